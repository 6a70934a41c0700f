import sys, os
os.environ['GPEMU_TRACE']='1'
sys.path.insert(0,'.')
from madaiemulator_amd import abi
c=abi.Context(0)
for cfg in (2,3):
    for k in (512,2048):
        reps=4
        ms,fl=c.gemm_bench(m=15488,n=15360,k=k,ld=15360,cfg=cfg,tri=1,beta=1,reps=reps)
        c.trace_dump("gpurun_out/gphase.txt")
        v=[int(x) for x in open("gpurun_out/gphase.txt").read().split("|")[1].split()]
        s,e,wsum,wn,wclk,pro,epi=v
        ghz=wclk/(wsum)  # clocks per ns
        print("cfg",cfg,"k",k,"ms %.4f TF/s %.1f | per WG: life %.1f us, prologue %.1f us, epilogue %.1f us, loop %.1f us (%.2f GHz)"%(
            ms,fl/ms/1e9, wsum/wn/1e3, pro/wn/ghz/1e3, epi/wn/ghz/1e3, (wclk-pro-epi)/wn/ghz/1e3, ghz),flush=True)

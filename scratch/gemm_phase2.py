import sys, os
os.environ['GPEMU_TRACE']='1'
sys.path.insert(0,'.')
from madaiemulator_amd import abi
c=abi.Context(0)
def run(m,n,k,cfg,tri,ld):
    ms,fl=c.gemm_bench(m=m,n=n,k=k,ld=ld,cfg=cfg,tri=tri,beta=1,reps=4)
    c.trace_dump("gpurun_out/gphase.txt")
    v=[int(x) for x in open("gpurun_out/gphase.txt").read().split("|")[1].split()]
    s,e,wsum,wn,wclk,pro,epi=v
    ghz=wclk/(wsum)
    print("cfg",cfg,"m",m,"n",n,"k",k,"ld",ld,"ms %.4f TF/s %.1f | WGs/launch %d per WG: life %.1f us, prologue %.1f us, epilogue %.1f us, loop %.1f us"%(
        ms,fl/ms/1e9, wn/5, wsum/wn/1e3, pro/wn/ghz/1e3, epi/wn/ghz/1e3, (wclk-pro-epi)/wn/ghz/1e3),flush=True)
run(1024,1024,512,3,0,15360)
for cfg in (3,2):
    run(1024,1024,512,cfg,0,15360)
    run(1024,1024,512,cfg,0,1024)
    run(2048,2048,512,cfg,0,15360)
    run(4096,4096,512,cfg,0,15360)
    run(8192,8192,512,cfg,0,15360)

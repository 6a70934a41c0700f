"""fixed cost per tile of the 128x128 LDS-DMA kernel: K = 16 .. 256 on the 15488 x 15360 triangular update (7320 tiles)"""
import sys
sys.path.insert(0,'.')
from madaiemulator_amd import abi
c=abi.Context(0)
for rnd in range(2):
    for beta in (1,0):
        for k in (16,32,64,128,256,512):
            ms,fl=c.gemm_bench(m=15488,n=15360,k=k,ld=15360,cfg=8,tri=1,beta=beta,reps=3)
            ms,fl=c.gemm_bench(m=15488,n=15360,k=k,ld=15360,cfg=8,tri=1,beta=beta,reps=10)
            print("round",rnd,"beta",beta,"k",k,"ms %.4f  us per tile per CU %.2f  TF/s %.1f"%(ms, ms*1e3/(7320/256.0), fl/ms/1e9),flush=True)

import sys
sys.path.insert(0,'.')
from madaiemulator_amd import abi
c=abi.Context(0)
for (m,n,k,tri) in ((16384,16384,1024,0),(16384,16384,512,0),(16384,16384,1024,1)):
    ms,fl=c.gemm_bench(m=m,n=n,k=k,ld=16384,cfg=3,tri=tri,beta=1,reps=3)
    ms,fl=c.gemm_bench(m=m,n=n,k=k,ld=16384,cfg=3,tri=tri,beta=1,reps=5)
    print("m",m,"n",n,"k",k,"tri",tri,"ms %.3f TF/s %.1f"%(ms,fl/ms/1e9),flush=True)

import sys
sys.path.insert(0,'.')
from madaiemulator_amd import abi
c=abi.Context(0)
for (m,n,k) in ((16384,64,64),(16384,128,128),(16384,256,256),(16384,512,512),(7744,7680,512)):
    for ld in (8192,8192+16,8192+32,8192+64,8192+256,9000):
        ms,fl=c.gemm_bench(m=m,n=n,k=k,ld=ld,cfg=2,tri=0,beta=1,reps=20)
        ms,fl=c.gemm_bench(m=m,n=n,k=k,ld=ld,cfg=2,tri=0,beta=1,reps=20)
        by=(2*m*n+ (m+n)*k)*8
        print("m",m,"n",n,"k",k,"ld",ld,"us %.1f TF/s %.1f  GB/s %.0f"%(ms*1e3,fl/ms/1e9,by/ms/1e6),flush=True)

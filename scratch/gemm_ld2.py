import sys
sys.path.insert(0,'.')
from madaiemulator_amd import abi
c=abi.Context(0)
for (m,n,k) in ((262144,64,64),(131072,128,128),(131072,256,256)):
    for ld in (max(n,k),8192,8224):
        ms,fl=c.gemm_bench(m=m,n=n,k=k,ld=ld,cfg=2,tri=0,beta=1,reps=3)
        by=(2*m*n+ (m+n)*k)*8
        print("m",m,"n",n,"k",k,"ld",ld,"us %.1f TF/s %.1f  GB/s %.0f"%(ms*1e3,fl/ms/1e9,by/ms/1e6),flush=True)

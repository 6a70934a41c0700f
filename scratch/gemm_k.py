import sys
sys.path.insert(0,'.')
from madaiemulator_amd import abi
c=abi.Context(0)
cfgs=[int(x) for x in sys.argv[1:]] or [3,7]
for rnd in range(3):                      # interleaved rounds in one process (methodology rule 24)
    for (m,n,ld) in ((15488,15360,15360),(6208,6144,8192)):
        for k in (512,1024,2048):
            for cfg in cfgs:
                ms,fl=c.gemm_bench(m=m,n=n,k=k,ld=ld,cfg=cfg,tri=1,beta=1,reps=3)
                ms,fl=c.gemm_bench(m=m,n=n,k=k,ld=ld,cfg=cfg,tri=1,beta=1,reps=8)
                print("round",rnd,"m",m,"ld",ld,"cfg",cfg,"k",k,"ms %.4f TF/s %.1f"%(ms,fl/ms/1e9),flush=True)

import sys
sys.path.insert(0,'.')
from madaiemulator_amd import abi
c=abi.Context(0)
cfgs=[int(x) for x in sys.argv[1:]] or [3,4,5,6]
for rnd in range(2):                      # interleaved rounds in one process (methodology rule 24)
    for k in (512,1024,2048):
        for cfg in cfgs:
            ms,fl=c.gemm_bench(m=15488,n=15360,k=k,ld=15360,cfg=cfg,tri=1,beta=1,reps=3)
            ms,fl=c.gemm_bench(m=15488,n=15360,k=k,ld=15360,cfg=cfg,tri=1,beta=1,reps=8)
            print("round",rnd,"cfg",cfg,"k",k,"ms %.4f TF/s %.1f"%(ms,fl/ms/1e9),flush=True)

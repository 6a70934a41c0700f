import sys, os
sys.path.insert(0,'.')
from madaiemulator_amd import abi
c=abi.Context(0)
for rnd in range(2):
    for ld0 in (0, 1):
        if ld0: os.environ["GPEMU_BENCH_LD0"] = "1"
        else: os.environ.pop("GPEMU_BENCH_LD0", None)
        for beta in (1, 0):
            for k in (512, 2048):
                ms,fl=c.gemm_bench(m=15488,n=15360,k=k,ld=15360,cfg=3,tri=1,beta=beta,reps=3)
                ms,fl=c.gemm_bench(m=15488,n=15360,k=k,ld=15360,cfg=3,tri=1,beta=beta,reps=8)
                print("round",rnd,"operands_in_cache",ld0,"beta",beta,"k",k,"ms %.4f TF/s %.1f"%(ms,fl/ms/1e9),flush=True)

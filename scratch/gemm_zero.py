import sys, os
sys.path.insert(0,'.')
os.environ["GPEMU_TRACE"] = "1"
from madaiemulator_amd import abi
c=abi.Context(0)
for rnd in range(2):
    for zero in (0, 1):
        if zero: os.environ["GPEMU_BENCH_ZERO"] = "1"
        else: os.environ.pop("GPEMU_BENCH_ZERO", None)
        for k in (2048,):
            ms,fl=c.gemm_bench(m=15488,n=15360,k=k,ld=15360,cfg=3,tri=1,beta=1,reps=3)
            ms,fl=c.gemm_bench(m=15488,n=15360,k=k,ld=15360,cfg=3,tri=1,beta=1,reps=8)
            c.trace_dump("/tmp/tr.txt")
            line = open("/tmp/tr.txt").read().strip().splitlines()[-1]
            q = [int(x) for x in line.rpartition("|")[2].split()]
            # q: start_ns end_ns sum_wg_ns n_wg sum_wg_clocks ...
            ghz = q[4] / max(q[2], 1)
            print("round",rnd,"zero",zero,"k",k,"ms %.4f TF/s %.1f  shader clock %.3f GHz -> %.1f TF/s clock-adjusted peak, frac %.3f"%(ms,fl/ms/1e9, ghz, 78.6*ghz/2.4, fl/ms/1e9/(78.6*ghz/2.4)),flush=True)

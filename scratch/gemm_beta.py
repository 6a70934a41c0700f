"""does the C-tile read in the prologue cost anything?  same shapes with beta = 1 (read-modify-write) and beta = 0"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madaiemulator_amd import abi
c = abi.Context(0)
for k in (256, 512, 1024):
    for beta in (1, 0):
        ms, fl = c.gemm_bench(m=15488, n=15360, k=k, ld=15360, cfg=3, tri=1, beta=beta, reps=5)
        ms, fl = c.gemm_bench(m=15488, n=15360, k=k, ld=15360, cfg=3, tri=1, beta=beta, reps=10)
        print("k", k, "beta", beta, "ms %.4f TF/s %.1f" % (ms, fl / ms / 1e9), flush=True)

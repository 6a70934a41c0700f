"""value+gradient batch of 16 (pow-exp, N=8192, d=8): where the 159 ms go (HIP events per class, one context)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
N, d, B = 8192, 8, 16
X, y = synth.design(N, d, 5)
c = abi.Context(0)
c.set_model(1, 0, X, y)
ths = np.array([synth.perturbed_thetas(1, d, 7, i) for i in range(B)])
for i in range(3):
    r = c.loglik_grad_batch(ths)
t0 = time.perf_counter(); r = c.loglik_grad_batch(ths); c.sync(); wall = (time.perf_counter() - t0) * 1e3
print("wall %.1f ms per batch of %d = %.2f ms per evaluation" % (wall, B, wall / B))
for name, cls in (("GEMM all", abi.PROF_GEMM), ("GEMM K>=512", abi.PROF_GEMM_K512), ("GEMM big tiles", abi.PROF_GEMM_BIG), ("leaf", abi.PROF_LEAF), ("potrf whole", abi.PROF_POTRF), ("fill", abi.PROF_FILL)):
    c.prof_begin(cls); c.loglik_grad_batch(ths); p = c.prof_end()
    print("%-16s launches %4d  %8.2f ms  %s" % (name, p["n"], p["ms"], ("%.1f TFLOP/s" % (p["flops"] / p["ms"] / 1e9)) if p["flops"] else ""))

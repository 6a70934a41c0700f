"""in-situ GEMM rate of a lock-step batch of 16 as a function of the leading dimension: N = 8192 gives ld = 8192 doubles
(rows 64 KB apart: a power of two), N = 8256 / 8320 do not"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madaiemulator_amd import abi, synth
B = 16
for rnd in range(2):
    for N in (8192, 8256, 8320, 8192):
        ctx = abi.Context(0)
        X, y = synth.design(N, 8, 5)
        ctx.set_model(3, 1, X, y)
        ths = np.array([synth.perturbed_thetas(3, 8, 7, i) for i in range(B)])
        for i in range(3):
            ctx.loglik_batch(ths)
        ctx.prof_begin(abi.PROF_GEMM); ctx.loglik_batch_enqueue(ths); p = ctx.prof_end(); ctx.loglik_batch_collect()
        t0 = time.perf_counter()
        for i in range(3):
            ctx.loglik_batch_enqueue(ths)
        ctx.loglik_batch_collect()
        dt = (time.perf_counter() - t0) / 3
        print("N", N, "gemm TF/s %.1f  gemm ms/batch %.2f  launches %d | batch ms %.2f = %.3f ms/eval, whole %.1f TF/s" % (
            p["flops"] / p["ms"] / 1e9, p["ms"], p["n"], dt * 1e3, dt * 1e3 / B, B * N ** 3 / 3 / dt / 1e12), flush=True)
        ctx.close()

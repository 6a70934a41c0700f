"""one evaluation at a time (N=8192, Matern 5/2, order 1): alternative leaves on the final GEMM kernels"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
N, d = 8192, 8
X, y = synth.design(N, d, 5)
combos = [dict(), dict(GPEMU_LEAF128="1"), dict(GPEMU_LEAF128="1", GPEMU_NB_TOP="1024"), dict(GPEMU_LEAF128="1", GPEMU_NB_TOP="256"),
          dict(GPEMU_PANEL_TRSM="256"), dict(GPEMU_PANEL_TRSM="128"), dict(GPEMU_FACTOR_AHEAD="0"), dict(GPEMU_SOLVE_AHEAD="1")]
keys = sorted({k for c in combos for k in c})
for rnd in range(2):
    for c in combos:
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(c)
        ctx = abi.Context(0)
        ctx.set_model(3, 1, X, y)
        ths = [synth.perturbed_thetas(3, d, 7, i) for i in range(16)]
        for i in range(3):
            ctx.loglik(ths[i])
        t0 = time.perf_counter()
        vals = [ctx.loglik(ths[3 + i])["value"] for i in range(10)]
        dt = (time.perf_counter() - t0) / 10
        print("%-70s %.3f ms  (check %.6f)" % (str(c), dt * 1e3, vals[0]), flush=True)
        ctx.close()

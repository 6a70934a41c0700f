"""factor-ahead on/off: one evaluation at a time and lock-step batches of 16 (N=8192, Matern 5/2, order 1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
N, d, B = 8192, 8, 16
X, y = synth.design(N, d, 5)
ths = np.array([synth.perturbed_thetas(3, d, 7, i) for i in range(B)])
for rnd in range(2):
    for fa, sa, pt in (("1", "0", "512"), ("1", "1", "512"), ("1", "0", "256"), ("1", "0", "0"), ("0", "0", "0")):
        os.environ["GPEMU_FACTOR_AHEAD"] = fa
        os.environ["GPEMU_SOLVE_AHEAD"] = sa
        os.environ["GPEMU_PANEL_TRSM"] = pt
        ctx = abi.Context(0)
        ctx.set_model(3, 1, X, y)
        for i in range(3): ctx.loglik(ths[i])
        t0 = time.perf_counter()
        v = [ctx.loglik(ths[i % B])["value"] for i in range(10)]
        t1 = (time.perf_counter() - t0) / 10
        for i in range(3): ctx.loglik_batch(ths)
        t0 = time.perf_counter()
        for i in range(4): ctx.loglik_batch_enqueue(ths)
        r = ctx.loglik_batch_collect()
        tb = (time.perf_counter() - t0) / 4
        ctx.prof_begin(abi.PROF_LEAF); ctx.loglik_batch_enqueue(ths); p = ctx.prof_end(); ctx.loglik_batch_collect()
        print("factor_ahead", fa, "solve_ahead", sa, "panel_trsm", pt, "single %.3f ms | batch16 %.2f ms = %.3f ms/eval | leaf launches %d %.2f ms | v0 %.9f vb %.9f" % (
            t1 * 1e3, tb * 1e3, tb * 1e3 / B, p["n"], p["ms"], v[0], r["value"][0]), flush=True)
        ctx.close()

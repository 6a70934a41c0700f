"""waves wholly above the diagonal of a diagonal tile (GPEMU_IDLE_WAVES): in-process A/B, contexts created under each
setting, rounds interleaved: stand-alone triangular updates by K and the lock-step likelihood batch at N=8192.
usage: python scratch/r04_idle_waves_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
def ctx(on):
    os.environ["GPEMU_IDLE_WAVES"] = "1" if on else "0"
    c = abi.Context(0)
    del os.environ["GPEMU_IDLE_WAVES"]
    return c
a, b = ctx(False), ctx(True)
for K in (512, 2048):
    best = {False: 1e9, True: 1e9}
    for rnd in range(3):
        for p, c in ((False, a), (True, b)):
            ms, fl = c.gemm_bench(7744, 7680, K, ld=8192, cfg=8, tri=1, beta=1, reps=6)
            best[p] = min(best[p], ms)
    print("stand-alone triangular update 7744 x 7680 (60 diagonal tiles of 1860), K=%4d: all waves compute %.1f TFLOP/s, idle waves %.1f TFLOP/s" % (K, fl / best[False] / 1e9, fl / best[True] / 1e9), flush=True)
N, d, B = 8192, 8, 16
X, y = synth.design(N, d, 6)
for c in (a, b): c.set_model(3, 1, X, y)
th8 = lambda j: np.array([synth.perturbed_thetas(3, d, 9, j * B + i) for i in range(B)])
for c in (a, b):
    c.loglik_batch(th8(0)); c.loglik_batch(th8(1))
va, vb = a.loglik_batch(th8(2)), b.loglik_batch(th8(2))
print("likelihood batch N=8192: bits equal", np.array_equal(va["value"], vb["value"]), flush=True)
best = {False: 1e9, True: 1e9}
for rnd in range(4):
    for p, c in ((False, a), (True, b)):
        t0 = time.perf_counter()
        for j in range(4): c.loglik_batch_enqueue(th8(3 + j))
        c.loglik_batch_collect()
        best[p] = min(best[p], (time.perf_counter() - t0) / 4)
print("one context, batches of 16 at N=8192: all waves compute %.2f ms per batch (%.1f /s), idle waves %.2f ms (%.1f /s)" % (best[False] * 1e3, B / best[False], best[True] * 1e3, B / best[True]))

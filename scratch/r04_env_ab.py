"""in-process A/B of one schedule switch of the device library: one context per value, rounds interleaved: stand-alone
triangular updates (128x128 tiles) by K, the lock-step likelihood batch of 16 at N=8192 (bits compared), value+gradient.
usage: python scratch/r04_env_ab.py GPEMU_SOMETHING value0 value1 [...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
var, settings = sys.argv[1], sys.argv[2:]
def ctx(v):
    os.environ[var] = v
    c = abi.Context(0)
    del os.environ[var]
    return c
cs = {v: ctx(v) for v in settings}
for (m, n, K) in ((6208, 6144, 2048), (6208, 6144, 512), (7744, 7680, 512), (7744, 7680, 256)):
    best = {v: 1e9 for v in settings}
    for rnd in range(3):
        for v in settings:
            ms, fl = cs[v].gemm_bench(m, n, K, ld=8192, cfg=8, tri=1, beta=1, reps=6)
            best[v] = min(best[v], ms)
    print("stand-alone triangular update %d x %d, K=%4d (ONE matrix): " % (m, n, K) + ", ".join("%s=%s: %.1f" % (var, v, fl / best[v] / 1e9) for v in settings) + " TFLOP/s", flush=True)
N, d, B = 8192, 8, 16
X, y = synth.design(N, d, 6)
for c in cs.values(): c.set_model(3, 1, X, y)
th8 = lambda j: np.array([synth.perturbed_thetas(3, d, 9, j * B + i) for i in range(B)])
vals = {}
for v, c in cs.items():
    c.loglik_batch(th8(0)); c.loglik_batch(th8(1))
    vals[v] = c.loglik_batch(th8(2))
print("likelihood batch N=8192: bits equal to the first context (value, sigma2, beta):",
      {v: bool(np.array_equal(vals[v]["value"], vals[settings[0]]["value"]) and np.array_equal(vals[v]["sigma2"], vals[settings[0]]["sigma2"])
              and np.array_equal(vals[v]["beta"], vals[settings[0]]["beta"])) for v in settings}, flush=True)
best = {v: 1e9 for v in settings}
for rnd in range(4):
    for v in settings:
        c = cs[v]
        t0 = time.perf_counter()
        for j in range(4): c.loglik_batch_enqueue(th8(3 + j))
        c.loglik_batch_collect()
        best[v] = min(best[v], (time.perf_counter() - t0) / 4)
print("one context, batches of 16 at N=8192: " + ", ".join("%s=%s: %.2f ms (%.1f /s)" % (var, v, best[v] * 1e3, B / best[v]) for v in settings), flush=True)
t1 = {}
for v, c in cs.items():
    bb = 1e9
    th1 = synth.perturbed_thetas(3, d, 9, 999)
    c.loglik(th1)
    for rnd in range(5):
        t0 = time.perf_counter(); r = c.loglik(th1); bb = min(bb, time.perf_counter() - t0)
    t1[v] = (bb, r["value"])
print("one evaluation at a time: " + ", ".join("%s=%s: %.3f ms" % (var, v, t1[v][0] * 1e3) for v in settings) + "; same value: %s" % (len({t1[v][1] for v in settings}) == 1), flush=True)

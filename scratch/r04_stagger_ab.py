"""first-round offset between the two workgroups of a CU in the 128x128 GEMM (GPEMU_STAGGER_US): in-process A/B, one context
per setting, rounds interleaved: stand-alone triangular updates by K and size, and the lock-step likelihood batch at N=8192.
usage: python scratch/r04_stagger_ab.py [us ...]   (default 0 20 40 80)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
settings = [int(v) for v in sys.argv[1:]] or [0, 20, 40, 80]
def ctx(us):
    os.environ["GPEMU_STAGGER_US"] = str(us)
    c = abi.Context(0)
    del os.environ["GPEMU_STAGGER_US"]
    return c
cs = {us: ctx(us) for us in settings}
for (m, n, K, nbatch_note) in ((6208, 6144, 2048, ""), (6208, 6144, 512, ""), (4160, 4096, 2048, ""), (2112, 2048, 2048, ""), (7744, 7680, 512, "")):
    best = {us: 1e9 for us in settings}
    for rnd in range(3):
        for us in settings:
            ms, fl = cs[us].gemm_bench(m, n, K, ld=8192, cfg=8, tri=1, beta=1, reps=6)
            best[us] = min(best[us], ms)
    print("stand-alone triangular update %d x %d, K=%4d (ONE matrix): " % (m, n, K) + ", ".join("%d us: %.1f" % (us, fl / best[us] / 1e9) for us in settings) + " TFLOP/s", flush=True)
N, d, B = 8192, 8, 16
X, y = synth.design(N, d, 6)
for c in cs.values(): c.set_model(3, 1, X, y)
th8 = lambda j: np.array([synth.perturbed_thetas(3, d, 9, j * B + i) for i in range(B)])
vals = {}
for us, c in cs.items():
    c.loglik_batch(th8(0)); c.loglik_batch(th8(1))
    vals[us] = c.loglik_batch(th8(2))["value"]
print("likelihood batch N=8192: bits equal to the 0-us context:", {us: bool(np.array_equal(vals[us], vals[settings[0]])) for us in settings}, flush=True)
best = {us: 1e9 for us in settings}
for rnd in range(4):
    for us in settings:
        c = cs[us]
        t0 = time.perf_counter()
        for j in range(4): c.loglik_batch_enqueue(th8(3 + j))
        c.loglik_batch_collect()
        best[us] = min(best[us], (time.perf_counter() - t0) / 4)
print("one context, batches of 16 at N=8192: " + ", ".join("%d us: %.2f ms (%.1f /s)" % (us, best[us] * 1e3, B / best[us]) for us in settings))

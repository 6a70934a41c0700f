"""in-process A/B of one schedule switch at N = 4096 (BASELINE configs[1] / configs[3]): one context per value, rounds
interleaved: lock-step likelihood batches of 64 (d=16, pow-exp, order 0: a pca8 component) and value+gradient batches of 16
and 64; bits compared against the first setting.
usage: python scratch/r05_env_ab_n4096.py GPEMU_SOMETHING value0 value1 [...]"""
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
N, d = 4096, 16
X, y = synth.design(N, d, 6)
for c in cs.values(): c.set_model(1, 0, X, y)
ths = lambda B, j: np.array([synth.perturbed_thetas(1, d, 9, j * B + i) for i in range(B)])
for name, B, fn_enq, fn_col in (("likelihood batch of 64", 64, "loglik_batch_enqueue", "loglik_batch_collect"),
                                ("value+gradient batch of 16", 16, "loglik_grad_batch_enqueue", None),
                                ("value+gradient batch of 64", 64, "loglik_grad_batch_enqueue", None)):
    vals = {}
    for v, c in cs.items():
        for j in range(3):
            r = c.loglik_batch(ths(B, j)) if fn_col else c.loglik_grad_batch(ths(B, j))
        vals[v] = r
    same = {v: bool(np.array_equal(vals[v]["value"], vals[settings[0]]["value"]) and (fn_col is not None or np.array_equal(vals[v]["grad"], vals[settings[0]]["grad"]))) for v in settings}
    best = {v: 1e9 for v in settings}
    for rnd in range(4):
        for v in settings:
            c = cs[v]
            t0 = time.perf_counter()
            for j in range(3):
                getattr(c, fn_enq)(ths(B, 3 + j))
                if fn_col is None: c.loglik_grad_batch_collect_back(0, B)
            if fn_col: c.loglik_batch_collect()
            best[v] = min(best[v], (time.perf_counter() - t0) / 3)
    print("%s at N=4096 d=16, one context: " % name + ", ".join("%s=%s: %.2f ms (%.0f /s)%s" % (var, v, best[v] * 1e3, B / best[v], "" if same[v] else " BITS DIFFER") for v in settings), flush=True)

"""ms per evaluation over the lock-step batch size at N=8192 (two contexts, enqueue/collect pipeline)"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madaiemulator_amd import abi, synth
kind, N, order, d = 3, 8192, 1, 8
X, y = synth.design(N, d, 5)
ths = np.array([synth.perturbed_thetas(kind, d, 7, i) for i in range(64)])
nctx = 2
cs = [abi.Context(0) for _ in range(nctx)]
for c in cs: c.set_model(kind, order, X, y)
for B in (16, 24, 32, 48, 64):
    for c in cs: c.loglik_batch(ths[:B]); c.loglik_batch(ths[:B])
    K = max(2, 96 // B)
    t = time.perf_counter()
    for i in range(K):
        for c in cs: c.loglik_batch_enqueue(ths[:B])
    for c in cs: c.loglik_batch_collect()
    dt = (time.perf_counter() - t) / K / nctx
    print("B %d ms/eval %.3f" % (B, dt * 1e3 / B), flush=True)

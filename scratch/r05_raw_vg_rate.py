"""raw value+gradient rate through the C-ABI (gpemu_loglik_grad_batch_enqueue / collect_back): nctx contexts x lock-step batches
of B, every batch collected -- the figure a search through the host layer is measured against.
usage: python scratch/r05_raw_vg_rate.py N d B nctx [steps] [exact]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
N, d, B, nctx = (int(v) for v in sys.argv[1:5])
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 12
exact = len(sys.argv) > 6 and sys.argv[6] == "exact"
X, y = synth.design(N, d, 20261003 + 3)
ctxs = [abi.Context(0) for _ in range(nctx)]
for c in ctxs:
    c.set_model(1, 0, X, y)
    if exact: c.set_mode(abi.MODE_EXACT_GRAD)
ths = lambda j: np.array([synth.perturbed_thetas(1, d, 7, j * B + i) for i in range(B)])
for j in range(3):
    for c in ctxs: c.loglik_grad_batch(ths(j))
t0 = time.perf_counter()
pend = [0] * nctx
for j in range(steps):
    k = j % nctx
    if pend[k]:
        r = ctxs[k].loglik_grad_batch_collect_back(0, B); assert np.all(r["status"] == 0); pend[k] = 0
    ctxs[k].loglik_grad_batch_enqueue(ths(10 + j)); pend[k] = 1
for k in range(nctx):
    if pend[k]:
        r = ctxs[k].loglik_grad_batch_collect_back(0, B); assert np.all(r["status"] == 0)
t = time.perf_counter() - t0
print("N=%d d=%d B=%d contexts=%d %s: %.1f value+gradient evaluations/s (%.2f ms per batch per context)" % (N, d, B, nctx, "exact" if exact else "literal", steps * B / t, t / steps * nctx * 1e3))

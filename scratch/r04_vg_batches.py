"""a few value+gradient batches at N=8192, d=8 (pow-exp), literal or exact mode: run under rocprofv3 --kernel-trace --stats
to read the per-kernel times of the gradient path.  usage: python scratch/r04_vg_batches.py [literal|exact] [batches]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
mode = sys.argv[1] if len(sys.argv) > 1 else "literal"
nbat = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N, d, B = 8192, 8, 16
X, y = synth.design(N, d, 20261003 + 3)
c = abi.Context(0)
if mode == "exact":
    c.set_mode(abi.MODE_EXACT_GRAD)
c.set_model(1, 0, X, y)
th = lambda j: np.array([synth.perturbed_thetas(1, d, 7, j * B + i) for i in range(B)])
c.loglik_grad_batch(th(0)); c.loglik_grad_batch(th(1))
t0 = time.perf_counter()
for j in range(nbat):
    r = c.loglik_grad_batch(th(2 + j))
    assert np.all(r["status"] == 0) and np.all(np.isfinite(r["grad"]))
dt = time.perf_counter() - t0
print(f"{mode}: {nbat * B / dt:.1f} value+gradient evaluations/s on one blocking context ({dt / nbat * 1e3:.1f} ms per batch of {B})")

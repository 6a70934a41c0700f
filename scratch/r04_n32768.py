"""beyond BASELINE's largest size: N = 32768, d = 8, pow-exp, order 0 -- ONE likelihood evaluation (an 8.6 GB factorisation
workspace) against LAPACK on a numpy-built matrix (tests/gradref.py; ~1-2 minutes of the host's cores, 17 GB of host memory),
then timing of one evaluation at a time and of a lock-step batch of 8.   usage: python scratch/r04_n32768.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, scipy.linalg as sl
from madaiemulator_amd import abi, synth
import gradref
N, d = (int(sys.argv[1]) if len(sys.argv) > 1 else 32768), 8
X, y = synth.design(N, d, 20261003 + 9)
c = abi.Context(0)
c.set_model(1, 0, X, y)
th = synth.default_thetas(1, d)
t0 = time.perf_counter(); a = c.loglik(th); t_first = time.perf_counter() - t0
print("N =", N, "status", a["status"], "info", a["info"], "value %.12g logdet %.12g quad %.12g first call %.2f s" % (a["value"], a["logdet"], a["quad"], t_first), flush=True)
best = 1e9
for i in range(3):
    t0 = time.perf_counter(); b = c.loglik(synth.perturbed_thetas(1, d, 3, i)); best = min(best, time.perf_counter() - t0)
print("one evaluation at a time: %.1f ms = %.1f TFLOP/s of N^3/3" % (best * 1e3, N ** 3 / 3.0 / best / 1e12), flush=True)
B = 8 if N <= 32768 else 2
thb = np.array([synth.perturbed_thetas(1, d, 5, i) for i in range(B)])
r = c.loglik_batch(thb)
t0 = time.perf_counter(); r = c.loglik_batch(thb); tb = time.perf_counter() - t0
print("lock-step batch of %d: %.1f ms per evaluation = %.1f TFLOP/s; all finite: %s" % (B, tb / B * 1e3, B * N ** 3 / 3.0 / tb / 1e12, bool(np.all(np.isfinite(r["value"])))), flush=True)
import threading
def _beat():
    k = 0
    while True:
        time.sleep(60); k += 1
        print("  ... LAPACK reference running, %d min" % k, flush=True)
threading.Thread(target=_beat, daemon=True).start()
t0 = time.perf_counter()
Cm, _ = gradref.powexp_matrix(X, th)
del _
print("  reference matrix built in %.0f s" % (time.perf_counter() - t0), flush=True)
cf = sl.cho_factor(Cm, lower=True, overwrite_a=True, check_finite=False)
logdet = 2.0 * np.log(np.diag(cf[0])).sum()
H = np.ones((N, 1))
AyH = sl.cho_solve(cf, np.column_stack([y, H]), check_finite=False)
beta = (H.T @ AyH[:, 0]) / (H.T @ AyH[:, 1])
rr = y - H[:, 0] * beta[0]
quad = rr @ sl.cho_solve(cf, rr, check_finite=False)
print("LAPACK reference in %.0f s: logdet rel %.2e  beta rel %.2e  quad rel %.2e  value rel %.2e" % (
    time.perf_counter() - t0, abs(a["logdet"] - logdet) / abs(logdet), abs(a["beta"][0] - beta[0]) / abs(beta[0]), abs(a["quad"] - quad) / abs(quad),
    abs(a["value"] + (-0.5 * logdet - N / 2.0 * 1.83788 - 0.5 * quad)) / abs(a["value"])), flush=True)
# predictions (the inverse rows under the matrix: (2 N + 64) x N elements per workspace, > 2^31 at N = 32768) against the same factor
if N <= 32768:
    Xq = synth.queries(8, d, 5)
    c.predict_setup(th)
    pm, pv = c.predict(Xq)
    Ks = np.exp(th[0]) * np.exp(-0.5 * (((Xq[:, None, :] - X[None, :, :]) / np.exp(np.array(th[2:]))[None, None, :]) ** 2).sum(axis=2))
    Ks[Ks < 1e-10] = 0.0
    alpha = sl.cho_solve(cf, rr, check_finite=False)
    mean = beta[0] + Ks @ alpha
    V = sl.cho_solve(cf, Ks.T, check_finite=False)
    hq = 1.0 - (H[:, 0] @ V)                       # h(x*) - H^T C^-1 k*
    var = np.exp(th[0]) + np.exp(th[1]) - np.einsum("qn,nq->q", Ks, V) + hq * hq / (H[:, 0] @ AyH[:, 1])
    print("8 predictions: max |mean - LAPACK| %.2e (scale %.2e)  max |var - LAPACK| %.2e (scale %.2e)" % (
        np.max(np.abs(pm - mean)), np.max(np.abs(mean)), np.max(np.abs(pv - var)), np.max(np.abs(var))), flush=True)
    nq = 16384
    Xb = synth.queries(nq, d, 6)
    c.predict(Xb[:64])
    t0 = time.perf_counter(); c.predict(Xb); tp = time.perf_counter() - t0
    print("%d predictions through the host-buffer entry: %.0f /s" % (nq, nq / tp), flush=True)

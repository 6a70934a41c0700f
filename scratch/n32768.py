"""N = 32768 (twice the largest BASELINE config): likelihood, y-scaling property, interpolation at training points, timing"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
N, d, kind, order = 32768, 8, 1, 0
X, y = synth.design(N, d, 77)
th = synth.default_thetas(kind, d)
c = abi.Context(0)
c.set_model(kind, order, X, y)
a = c.loglik(th); a = c.loglik(th)
t = time.perf_counter(); a = c.loglik(th); dt = time.perf_counter() - t
print("loglik", a["value"], "status", a["status"], "ms %.1f  TF/s %.1f" % (dt * 1e3, N ** 3 / 3 / dt / 1e12), flush=True)
c.set_training(2.0 * y)
b = c.loglik(th)
print("sigma2 ratio", b["sigma2"] / a["sigma2"], "quad ratio", b["quad"] / a["quad"], "logdet equal", b["logdet"] == a["logdet"])
c.set_training(y)
c.predict_setup(th)
idx = np.arange(0, N, 97)
m, v = c.predict(X[idx])
print("interp max err", np.max(np.abs(m - y[idx])), "max var", np.max(np.abs(v)))
r = c.loglik_batch(np.array([th, th, th, th]))
t = time.perf_counter(); r = c.loglik_batch(np.array([th, th, th, th])); dt = time.perf_counter() - t
print("batch of 4:", r["value"], "ms per eval %.1f" % (dt * 1e3 / 4), "rel diff vs single %.2e" % abs(r["value"][0] / a["value"] - 1))

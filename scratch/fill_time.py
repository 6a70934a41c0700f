"""time of the staging launch (fill + R rows) at N=8192, one matrix and a batch of 16, by HIP events (gpemu_prof)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
N, d = 8192, 8
X, y = synth.design(N, d, 5)
for kind, order in ((3, 1), (1, 0)):
    c = abi.Context(0)
    c.set_model(kind, order, X, y)
    ths = np.array([synth.perturbed_thetas(kind, d, 7, i) for i in range(16)])
    c.loglik(ths[0]); c.loglik(ths[0]); c.loglik_batch(ths); c.loglik_batch(ths)
    for nb in (1, 16):
        c.prof_begin(abi.PROF_FILL)
        for i in range(4):
            if nb == 1: c.loglik_enqueue(ths[i])
            else: c.loglik_batch_enqueue(ths)
        p = c.prof_end()
        print("kind %d batch %2d: %.1f us per matrix, %.0f GB/s" % (kind, nb, p["ms"] * 1e3 / p["n"] / nb, p["bytes"] / (p["ms"] * 1e-3) / 1e9), flush=True)
    c.close()

"""staging launch of a lock-step batch of 16 (N=8192): time per matrix, Matern 5/2 and pow-exp, Gram form on / off"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madaiemulator_amd import abi, synth
N, d, B = 8192, 8, 16
X, y = synth.design(N, d, 5)
for kind in (3, 1):
    for gram in ("1", "0"):
        os.environ["GPEMU_FILL_GRAM"] = gram
        c = abi.Context(0)
        del os.environ["GPEMU_FILL_GRAM"]
        c.set_model(kind, 1, X, y)
        ths = np.array([synth.perturbed_thetas(kind, d, 7, i) for i in range(B)])
        t0 = time.time()
        while time.time() - t0 < 1.5: c.loglik_batch(ths)           # clocks up
        best = 1e9
        for rep in range(3):
            c.prof_begin(abi.PROF_FILL)
            for i in range(4): c.loglik_batch_enqueue(ths)
            p = c.prof_end(); c.loglik_batch_collect()
            best = min(best, p["ms"] * 1e3 / (p["n"] * B))
        us = best
        print("kind %d gram %s: %.1f us per matrix = %.2f TB/s of 4 N^2 bytes (%.3f of 8 TB/s)" % (kind, gram, us, 4.0 * N * N / us / 1e6, 4.0 * N * N / us / 1e6 / 8))
        c.close()

"""one-query latency against N around 8192 (is the power-of-two row stride of L^-1 what holds the skinny product at 2.9 TB/s?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
d = 8
for N in (7936, 8000, 8128, 8192, 8256, 8320, 12288, 12352):
    X, y = synth.design(N, d, 5)
    c = abi.Context(0)
    c.set_model(3, 1, X, y)
    c.predict_setup(synth.default_thetas(3, d))
    Q = synth.queries(512, d, 3)
    c.predict(Q[:1])
    t0 = time.perf_counter(); n = 300
    for i in range(n): c.predict(Q[i:i + 1])
    t = (time.perf_counter() - t0) / n
    print("N=%5d: one query %.1f us per call = %.2f TB/s of the N^2/2 * 8 bytes of L^-1" % (N, t * 1e6, 4.0 * N * N / t / 1e12), flush=True)
    c.close()

"""emulate_point latency: ONE query through gpemu_predict_batch (host buffers, host call to host result) at N=8192 and 4096,
and 1 / 4 / 16 / 17 / 64 queries per call.  usage: python scratch/r04_point_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
for kind, N, order in ((3, 8192, 1), (1, 4096, 0)):
    d = 8
    X, y = synth.design(N, d, 5)
    c = abi.Context(0)
    c.set_model(kind, order, X, y)
    c.predict_setup(synth.default_thetas(kind, d))
    Q = synth.queries(4096, d, 3)
    for M in (1, 4, 16, 17, 64):
        c.predict(Q[:M])
        t0 = time.perf_counter()
        n = 200
        for i in range(n): c.predict(Q[i * M % 2048:i * M % 2048 + M])
        t = (time.perf_counter() - t0) / n
        print("N=%d kind %d: %2d queries per call: %.1f us per call, %.1f us per query" % (N, kind, M, t * 1e6, t * 1e6 / M), flush=True)
    c.close()

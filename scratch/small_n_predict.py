"""prediction latency / throughput on small models: one query per call (MCMC use: callEmulateMC), 64 and 4096 per call"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madaiemulator_amd import abi, synth
for N in (34, 200, 1024, 4096):
    kind, order, d = 1, 1, 4
    X, y = synth.design(N, d, 5)
    c = abi.Context(0)
    c.set_model(kind, order, X, y)
    c.predict_setup(synth.default_thetas(kind, d))
    out = []
    for M in (1, 64, 4096):
        Q = synth.queries(M, d, 3)
        c.predict(Q); c.predict(Q)
        K = 200 if M < 4096 else 40
        t = time.perf_counter()
        for i in range(K): c.predict(Q)
        dt = (time.perf_counter() - t) / K
        out.append("M=%d: %.0f us/call (%.2f us/point)" % (M, dt * 1e6, dt * 1e6 / M))
    print("N %5d  " % N + "   ".join(out), flush=True)
    c.close()

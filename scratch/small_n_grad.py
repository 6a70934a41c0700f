"""value+gradient batches of 16 (the lock-step group of estimate_thetas_threaded) on small models: us per evaluation"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madaiemulator_amd import abi, synth
B = 16
for N in (34, 128, 200, 512, 1024, 2048):
    kind, order, d = 1, 1, 4
    X, y = synth.design(N, d, 5)
    ths = np.array([synth.perturbed_thetas(kind, d, 7, i) for i in range(B)])
    c = abi.Context(0)
    c.set_model(kind, order, X, y)
    c.loglik_grad_batch(ths); c.loglik_grad_batch(ths)
    K = 50
    t = time.perf_counter()
    for i in range(K): c.loglik_grad_batch(ths)
    dt = (time.perf_counter() - t) / K
    c.loglik_batch(ths); c.loglik_batch(ths)
    t = time.perf_counter()
    for i in range(K): c.loglik_batch(ths)
    dv = (time.perf_counter() - t) / K
    print("N %5d  value+grad batch of 16: %.0f us = %.1f us/eval    value only: %.0f us = %.1f us/eval" % (N, dt * 1e6, dt * 1e6 / B, dv * 1e6, dv * 1e6 / B), flush=True)
    c.close()

"""small models (the sizes of the reference's own examples): microseconds per evaluation in batches of 64, one context,
and the split between host and device time"""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from madaiemulator_amd import abi, synth
B = 64
for N in (34, 64, 128, 200, 256, 512, 1024):
    kind, order, d = 1, 1, 4
    X, y = synth.design(N, d, 5)
    ths = np.array([synth.perturbed_thetas(kind, d, 7, i) for i in range(B)])
    c = abi.Context(0)
    c.set_model(kind, order, X, y)
    c.loglik_batch(ths); c.loglik_batch(ths)
    K = 200
    t = time.perf_counter()
    for i in range(K): c.loglik_batch(ths)
    dt = (time.perf_counter() - t) / K
    t = time.perf_counter()
    for i in range(K): c.loglik_batch_enqueue(ths)
    c.loglik_batch_collect()
    dp = (time.perf_counter() - t) / K
    t = time.perf_counter()
    for i in range(K): c.loglik(ths[0])
    d1 = (time.perf_counter() - t) / K
    print("N %5d  batch of 64: %.1f us/batch = %.2f us/eval (pipelined %.2f us/eval)   one at a time %.1f us" % (N, dt * 1e6, dt * 1e6 / B, dp * 1e6 / B, d1 * 1e6), flush=True)
    c.close()

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
for N in (1100, 4096, 8192, 32768):
    d, kind, order = 8, 1, 0
    X, y = synth.design(N, d, 77)
    th = synth.default_thetas(kind, d)
    c = abi.Context(0)
    c.set_model(kind, order, X, y)
    a = c.loglik(th)
    r = c.loglik_batch(np.array([th, th]))
    print(N, repr(a["value"]), repr(float(r["value"][0])), repr(a["logdet"]), repr(float(r["logdet"][0])), repr(a["quad"]), repr(float(r["quad"][0])))
    c.close()

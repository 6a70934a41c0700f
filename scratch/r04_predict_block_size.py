"""prediction sweep at N=8192 (Matern 5/2, order 1): predictions/s against the number of queries per gpemu_predict_batch_dev
call (the library works in blocks of at most 16384).  usage: python scratch/r04_predict_block_size.py"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
N, d, tot = 8192, 8, 262144
X, y = synth.design(N, d, 20261005)
th = synth.default_thetas(3, d)
Xq = synth.queries(tot, d, 17)
c = abi.Context(0)
c.set_model(3, 1, X, y); c.predict_setup(th)
dq, dm, dv = c.dev_alloc(Xq.nbytes), c.dev_alloc(tot * 8), c.dev_alloc(tot * 8)
c.upload(dq, Xq)
c.predict_dev(16384, dq, dm, dv); c.sync()
for per in (2048, 4096, 8192, 12288, 16384, 65536):
    best = 1e9
    for rnd in range(3):
        t0 = time.perf_counter()
        for b in range(tot // per):
            c.predict_dev(per, C.c_void_p(dq.value + b * per * d * 8), C.c_void_p(dm.value + b * per * 8), C.c_void_p(dv.value + b * per * 8))
        c.sync()
        best = min(best, time.perf_counter() - t0)
    print("%6d queries per call: %.0f predictions/s" % (per, tot / best), flush=True)

"""prediction sweep at N=8192 (Matern 5/2, order 1): in-process A/B of the k-vector fill form (GPEMU_KVEC_GRAM), contexts
created under each setting, rounds interleaved.  (Round 4 also ran it with a sum-of-squares epilogue for the product,
GPEMU_PRED_SUMSQ, which measured -0.2 % and left the tree: profiles/r04_persistent_gemm_and_sumsq_epilogue_ab.txt.)
usage: python scratch/r04_predict_ab.py"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from madaiemulator_amd import abi, synth
N, d, per, nb = 8192, 8, 50000, 6
X, y = synth.design(N, d, 20261005)
th = synth.default_thetas(3, d)
Xq = synth.queries(per * nb, d, 17)
variants = {"default": {}, "kvec difference form (round 3)": {"GPEMU_KVEC_GRAM": "0"}}
ctx = {}
for name, env in variants.items():
    for k, v in env.items(): os.environ[k] = v
    c = abi.Context(0)
    for k in env: del os.environ[k]
    c.set_model(3, 1, X, y); c.predict_setup(th)
    dq, dm, dv = c.dev_alloc(Xq.nbytes), c.dev_alloc(per * nb * 8), c.dev_alloc(per * nb * 8)
    c.upload(dq, Xq)
    c.predict_dev(per, dq, dm, dv); c.sync()
    ctx[name] = (c, dq, dm, dv)
best = {k: 1e9 for k in variants}
for rnd in range(4):
    for name, (c, dq, dm, dv) in ctx.items():
        t0 = time.perf_counter()
        for b in range(nb):
            c.predict_dev(per, C.c_void_p(dq.value + b * per * d * 8), C.c_void_p(dm.value + b * per * 8), C.c_void_p(dv.value + b * per * 8))
        c.sync()
        best[name] = min(best[name], time.perf_counter() - t0)
ref = None
for name, (c, dq, dm, dv) in ctx.items():
    m = c.download(dm, (per * nb,)); v = c.download(dv, (per * nb,))
    if ref is None: ref = (m, v)
    print("%-28s %.0f predictions/s   max |mean - default| %.2e  max |var - default| %.2e" % (name, per * nb / best[name], np.max(np.abs(m - ref[0])), np.max(np.abs(v - ref[1]))))

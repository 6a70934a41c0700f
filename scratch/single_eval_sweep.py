"""one evaluation at a time (N=8192, Matern 5/2, order 1): host call to host result, for the schedule switches"""
import os, sys, time, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if "--torch" in sys.argv:
    import torch; torch.cuda.is_available()
import numpy as np
from madaiemulator_amd import abi, synth
N, d = 8192, 8
X, y = synth.design(N, d, 5)
combos = [dict()] + [dict(GPEMU_LOOKAHEAD="1", GPEMU_RESERVE_CUS=str(r), GPEMU_NB_TOP=str(nb), GPEMU_NO_GRAPH=g)
                     for r in (0, 16, 32, 64) for nb in (512, 1024) for g in ("0", "1")] + \
         [dict(GPEMU_NB_TOP=str(nb)) for nb in (256, 1024)] + [dict(GPEMU_NO_GRAPH="1")]
keys = ["GPEMU_LOOKAHEAD", "GPEMU_RESERVE_CUS", "GPEMU_NB_TOP", "GPEMU_NO_GRAPH"]
for c in combos:
    for k in keys:
        os.environ.pop(k, None)
    os.environ.update(c)
    ctx = abi.Context(0)
    ctx.set_model(3, 1, X, y)
    ths = [synth.perturbed_thetas(3, d, 7, i) for i in range(16)]
    for i in range(3):
        ctx.loglik(ths[i])
    t0 = time.perf_counter()
    vals = [ctx.loglik(ths[3 + i])["value"] for i in range(10)]
    dt = (time.perf_counter() - t0) / 10
    print("%-95s %.3f ms  (check %.6f)" % (str(c), dt * 1e3, vals[0]), flush=True)
    ctx.close()

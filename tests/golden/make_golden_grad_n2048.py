#!/usr/bin/env python3
"""Generates tests/golden/golden_grad_n2048.npz: ONE oracle gradFnMulti at N = 2048.

N = 2048, d = 8, pow-exp, regression order 1, the seeded design of madaiemulator_amd.synth (seed 20261003 + 11):
  * orc_gradFnMulti  (maxmultimin.c:416-550 restated: fill, unblocked Cholesky, explicit inverse, estimateSigma, then per
    hyper-parameter the literal derivative matrix of emulator.c:173-209 and getGradientCn's N^3 dgemm + trace, :571-608)
  * orc_evalFnMulti  (maxmultimin.c:288-394) at the same theta -- the value evalFnGradMulti (:615-618) pairs it with.
N = 2048 has 32 x 33 / 2 = 528 lower 64x64 tiles: the device's second-stage reduction (grad_reduce_kernel, thread j takes
tiles j, j + 256, ...) makes more than one pass, which no live oracle comparison (N <= 900) reaches.  The oracle pass is
nine naive N^3 products (55 s of one core at N = 2048, measured here) next to other tests' CPU work: run here once,
~20 numbers committed.

Independent cross-check before writing: tests/gradref.py (numpy/LAPACK, O(N^2 d) form of the same formulas) at 1e-9.

The same script with the argument 4096 writes golden_grad_n4096.npz: the oracle's gradFnMulti at BASELINE.json configs[1]'s
size (N = 4096, d = 8; 2 080 tiles, nine passes of the stride loop; 768 s of one core, measured here).

Run from the repo root:  python tests/golden/make_golden_grad_n2048.py [4096]
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))
from oracle import oracle as O  # noqa: E402
from madaiemulator_amd import synth  # noqa: E402
import gradref  # noqa: E402

KIND, ORDER, D, SEED = 1, 1, 8, 20261003 + 11
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048       # 4096: BASELINE configs[1]'s size (golden_grad_n4096.npz)


def thetas():
    th = synth.perturbed_thetas(KIND, D, 41, 0)
    th[0] = 0.0                                     # gradFnMulti's theta[0] (maxmultimin.c:441)
    th[2:] += 0.1 * np.arange(D) - 0.3              # distinct length scales per direction
    return th


def grad_worker(_):
    X, y = synth.design(N, D, SEED)
    t = time.perf_counter()
    g, st = O.grad_fn_multi(KIND, ORDER, X, y, thetas()[1:])
    return dict(grad=g, status=st, seconds=time.perf_counter() - t)


def eval_worker(_):
    X, y = synth.design(N, D, SEED)
    t = time.perf_counter()
    o = O.eval_fn_multi(KIND, ORDER, X, y, thetas()[1:])
    o["seconds"] = time.perf_counter() - t
    return o


def main():
    with mp.get_context("spawn").Pool(2) as pool:
        ra = pool.map_async(grad_worker, [0])
        rb = pool.map_async(eval_worker, [0])
        X, y = synth.design(N, D, SEED)
        ref = gradref.value_and_gradients(X, y, ORDER, thetas())
        g = ra.get()[0]
        o = rb.get()[0]
    if g["status"] != 0 or o["info"] != 0:
        raise SystemExit("oracle reported a failed factorisation")
    scale = np.max(np.abs(ref["literal"]))
    err = np.max(np.abs(g["grad"] - ref["literal"])) / scale
    print("gradient", g["grad"], "\nnumpy   ", ref["literal"], "\nrel err of the largest component %.2e" % err)
    if not err < 1e-9:
        raise SystemExit("independent check FAILED for the gradient")
    for name in ("value", "sigma2", "logdet", "quad"):
        e = abs(o[name] - ref[name]) / abs(ref[name])
        print(name, o[name], ref[name], "rel err %.2e" % e)
        if not e < 1e-9:
            raise SystemExit("independent check FAILED for " + name)
    out = dict(grad=g["grad"], value=o["value"], sigma2=o["sigma2"], beta=o["beta"], logdet=o["logdet"], quad=o["quad"],
               exact_numpy=ref["exact"], thetas=thetas(), meta=np.array([KIND, ORDER, N, D, SEED]),
               oracle_seconds=np.array([g["seconds"], o["seconds"]]))
    path = os.path.join(HERE, "golden_grad_n%d.npz" % N)
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; oracle seconds", g["seconds"], o["seconds"])


if __name__ == "__main__":
    main()

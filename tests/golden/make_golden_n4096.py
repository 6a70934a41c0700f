#!/usr/bin/env python3
"""Generates tests/golden/golden_n4096.npz: ONE full oracle evaluation at BASELINE.json configs[1]'s size.

N = 4096, d = 8, pow-exp, regression order 0, the seeded design of madaiemulator_amd.synth (seed 20261003 + 1):
  * orc_evalFnMulti  (maxmultimin.c:288-394 restated: fill, unblocked Cholesky, explicit inverse, estimateBeta twice)
  * orc_emulator_setup + 64 x orc_emulate_points (emulator_struct.c:13-37,124-143; emulator.c:672-785)
The two passes are ~N^3 each through the oracle's naive row-major loops: 250 s and 220 s of one core each when they
have the machine to themselves (measured here), several times that next to other work (32 KB row stride: they live
on the cache) -- too close to the GPU box's "no output for 7 minutes = hung" rule for a test, so the run is done here,
once, and its ~150 numbers are committed as a fixture (the inputs are regenerated from the seeds wherever the test
runs).
tests/test_gpu_parity.py::test_n4096_oracle_fixture compares the HIP path with them; the same test runs the oracle
live at N = 3072 (two host cores, ~1.5 minutes).

Independent cross-check before writing: LAPACK (scipy) evaluation of the same quantities at 1e-9.

Run from the repo root:  python tests/golden/make_golden_n4096.py
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np
import scipy.linalg as sl

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from madaiemulator_amd import synth  # noqa: E402

KIND, ORDER, N, D, SEED, QSEED = 1, 0, 4096, 8, 20261003 + 1, 321


def thetas():
    th = synth.default_thetas(KIND, D).copy()
    th[0] = 0.0                                     # evalFnMulti's theta[0] (maxmultimin.c:311)
    return th


def eval_worker(_):
    X, y = synth.design(N, D, SEED)
    t = time.perf_counter()
    o = O.eval_fn_multi(KIND, ORDER, X, y, thetas()[1:])
    o["seconds"] = time.perf_counter() - t
    return o


def emu_worker(_):
    X, y = synth.design(N, D, SEED)
    t = time.perf_counter()
    e = O.Emulator(KIND, ORDER, X, y, thetas())
    m, v, st = e.emulate(synth.queries(64, D, QSEED))
    return dict(mean=m, var=v, beta=e.beta, logdet=e.logdet, seconds=time.perf_counter() - t)


def main():
    with mp.get_context("spawn").Pool(2) as pool:
        ra = pool.map_async(eval_worker, [0])
        rb = pool.map_async(emu_worker, [0])
        # LAPACK cross-check meanwhile
        X, y = synth.design(N, D, SEED)
        th = thetas()
        r2 = np.exp(th[2:]) ** 2
        Dm = X[:, None, :] - X[None, :, :]
        Cm = np.exp((-0.5 * Dm * Dm / r2).sum(-1)) + np.exp(th[1]) * np.eye(N)
        del Dm
        cf = sl.cho_factor(Cm, lower=True)
        logdet = 2 * np.log(np.diag(cf[0])).sum()
        H = np.ones((N, 1))
        AyH = sl.cho_solve(cf, np.column_stack([y, H]))
        beta = np.linalg.solve(H.T @ AyH[:, 1:], H.T @ AyH[:, 0])
        r = y - H @ beta
        Ar = sl.cho_solve(cf, r)
        quad = r @ Ar
        val = -(-0.5 * logdet - N / 2.0 * 1.83788 - 0.5 * quad)
        o = ra.get()[0]
        e = rb.get()[0]
    for name, a, b in (("value", o["value"], val), ("logdet", o["logdet"], logdet), ("quad", o["quad"], quad),
                       ("beta", o["beta"][0], beta[0]), ("sigma2", o["sigma2"], (y @ Ar) / N)):
        err = abs(a - b) / abs(b)
        print(name, a, b, "rel err %.2e" % err)
        if not err < 1e-9:
            raise SystemExit("independent check FAILED for " + name)
    out = dict(value=o["value"], sigma2=o["sigma2"], beta=o["beta"], logdet=o["logdet"], quad=o["quad"], info=o["info"],
               mean=e["mean"], var=e["var"], emu_beta=e["beta"], emu_logdet=e["logdet"], thetas=thetas(),
               meta=np.array([KIND, ORDER, N, D, SEED, QSEED]), oracle_seconds=np.array([o["seconds"], e["seconds"]]))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_n4096.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; oracle seconds", o["seconds"], e["seconds"])


if __name__ == "__main__":
    main()

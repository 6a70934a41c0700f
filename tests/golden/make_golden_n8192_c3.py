#!/usr/bin/env python3
"""Generates tests/golden/golden_n8192_c3.npz: ONE full oracle evaluation at BASELINE.json configs[2] exactly.

N = 8192, d = 8, Matern 5/2, regression order 1, the bench's own design (madaiemulator_amd.synth, seed 20261003 + 2)
and its supplied hyper-parameters [1.0, 0.01, log 0.6]:
  * orc_emulator_setup (emulator_struct.c:13-37 restated: fill, unblocked Cholesky, explicit inverse, estimateBeta)
    and from its pieces the likelihood at the given FULL theta exactly as tests/test_gpu_parity.py::check_loglik forms
    it for the Matern kernels (estimator-fns.c:38-103: value, sigma^2, beta, log det = 2 sum log L_ii, quadratic form)
  * 64 x orc_emulate_points (emulator_struct.c:124-143; emulator.c:578-593, 672-785)
About 8 x the N=4096 pass through the oracle's naive row-major loops: 35-60 minutes of one core -- far beyond the GPU
box's "no output for 7 minutes = hung" rule, so it is run here, once, offline, and its ~150 numbers are committed as a
fixture (the inputs are regenerated from the seeds wherever the test runs).

Independent cross-check before writing: LAPACK (scipy) evaluation of the same quantities at 1e-9, built from numpy's
own Matern 5/2 formula (literal constants of emulator.c:452-470).

Run from the repo root:  python tests/golden/make_golden_n8192_c3.py
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

KIND, ORDER, N, D, SEED, QSEED, NQ = 3, 1, 8192, 8, 20261003 + 2, 321, 64


def oracle_worker(_):
    X, y = synth.design(N, D, SEED)
    th = synth.default_thetas(KIND, D)
    t = time.perf_counter()
    e = O.Emulator(KIND, ORDER, X, y, th)
    t_setup = time.perf_counter() - t
    r = y - e.H @ e.beta
    Ar = e.cinverse @ r
    quad = float(r @ Ar)
    sigma2 = float(y @ Ar) / N                                   # maxmultimin.c:259-263 (y, not r)
    value = -(-0.5 * e.logdet - N / 2.0 * 1.83788 - 0.5 * quad)  # estimator-fns.c:48 literal
    t = time.perf_counter()
    m, v, st = e.emulate(synth.queries(NQ, D, QSEED))
    return dict(value=value, sigma2=sigma2, beta=e.beta, logdet=e.logdet, quad=quad, status=e.status, mean=m, var=v,
                emu_status=st, seconds=np.array([t_setup, time.perf_counter() - t]))


def matern52_lapack():
    """the same quantities from numpy / LAPACK (blocked, pivot-free Cholesky of a different implementation)"""
    X, y = synth.design(N, D, SEED)
    amp, nug, rho = synth.default_thetas(KIND, D)[0], synth.default_thetas(KIND, D)[1], np.exp(np.log(0.6))
    Cm = np.empty((N, N))
    for i0 in range(0, N, 512):
        Dm = X[i0:i0 + 512, None, :] - X[None, :, :]
        r = np.sqrt((Dm * Dm).sum(-1))
        s = r / rho
        Cm[i0:i0 + 512] = amp * (1.0 + 2.236067978 * s + (5.0 / 3.0) * s * s) * np.exp(-2.236067978 * s)
    Cm[np.diag_indices(N)] = amp + nug
    cf = sl.cho_factor(Cm, lower=True, overwrite_a=True)
    logdet = 2 * np.log(np.diag(cf[0])).sum()
    H = np.column_stack([np.ones(N), X])
    AyH = sl.cho_solve(cf, np.column_stack([y, H]))
    beta = np.linalg.solve(H.T @ AyH[:, 1:], H.T @ AyH[:, 0])
    r = y - H @ beta
    Ar = sl.cho_solve(cf, r)
    quad = r @ Ar
    val = -(-0.5 * logdet - N / 2.0 * 1.83788 - 0.5 * quad)
    # posterior mean / variance at the fixture's queries (emulator.c:672-785 in LAPACK terms)
    Xq = synth.queries(NQ, D, QSEED)
    Dq = Xq[:, None, :] - X[None, :, :]
    s = np.sqrt((Dq * Dq).sum(-1)) / rho
    K = amp * (1.0 + 2.236067978 * s + (5.0 / 3.0) * s * s) * np.exp(-2.236067978 * s)
    K[K < 1e-10] = 0.0
    Hq = np.column_stack([np.ones(NQ), Xq])
    AK = sl.cho_solve(cf, K.T)                                       # N x NQ
    mean = Hq @ beta + K @ Ar
    W = AyH[:, 1:]                                                   # C^-1 H
    q = Hq - K @ W
    Q = np.linalg.inv(H.T @ W)
    var = (amp + nug) - np.einsum("ij,ji->i", K, AK) + np.einsum("ij,jk,ik->i", q, Q, q)
    return dict(value=val, logdet=logdet, quad=quad, beta=beta, sigma2=(y @ Ar) / N, mean=mean, var=var)


def main():
    t0 = time.perf_counter()
    with mp.get_context("spawn").Pool(1) as pool:
        ra = pool.map_async(oracle_worker, [0])
        ref = matern52_lapack()
        print("LAPACK side done after %.0f s" % (time.perf_counter() - t0), flush=True)
        o = ra.get()[0]
    print("oracle side done after %.0f s (setup %.0f s, %d predictions %.0f s)" %
          (time.perf_counter() - t0, o["seconds"][0], NQ, o["seconds"][1]), flush=True)
    assert o["status"] == 0 and o["emu_status"] == 0
    for name in ("value", "logdet", "quad", "sigma2"):
        err = abs(o[name] - ref[name]) / abs(ref[name])
        print(name, o[name], ref[name], "rel err %.2e" % err)
        if not err < 1e-9:
            raise SystemExit("independent check FAILED for " + name)
    for name, scale in (("beta", np.max(np.abs(ref["beta"]))), ("mean", max(1.0, np.max(np.abs(ref["mean"])))), ("var", 1.01)):
        err = np.max(np.abs(o[name] - ref[name])) / scale
        print(name, "max err %.2e" % err)
        if not err < 1e-9:
            raise SystemExit("independent check FAILED for " + name)
    out = dict(value=o["value"], sigma2=o["sigma2"], beta=o["beta"], logdet=o["logdet"], quad=o["quad"],
               mean=o["mean"], var=o["var"], thetas=synth.default_thetas(KIND, D),
               meta=np.array([KIND, ORDER, N, D, SEED, QSEED, NQ]), oracle_seconds=o["seconds"])
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_n8192_c3.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Generates tests/golden/golden_v3.npz (round 3): an independent pin for the CORRECTED gradient forms of gpemu.h
(GPEMU_MODE_EXACT_GRAD, and GPEMU_MODE_MATERN_LOG for the Matern kernels; SURVEY App. C2-C4 policy).

The reference has no usable gradient for these (its Matern training path exits, maxmultimin.c:495; its pow-exp formula
keeps one coordinate's factor, emulator.c:189,203), so there is nothing of the reference's to restate: the vector is the
derivative of the VALUE -- the evalFnMulti value at theta[0] = 0 (maxmultimin.c:288-394 with the literal constants
1.83788, 1.732050808, 2.236067978; amplitude and nugget on the log scale for the Matern kernels, log det = 2 sum log
L_ii) -- evaluated with mpmath at 50 digits and differentiated NUMERICALLY there (mp.diff: central differences with
50-digit arithmetic, error ~1e-25).  The device computes the same derivative ANALYTICALLY (grad_exact_kernel), so
agreement checks the analytic forms (dC/dlog rho of the Matern kernels, the nugget direction, the beta-independence of
the GLS residual) and the value they belong to at once.

  X, y : the first 34 rows of test/multi-simple (d = 3), output column 0   -- N = 34 as G4
  cases: Matern 5/2 order 1, Matern 5/2 order 0, Matern 3/2 order 1, pow-exp order 1
Cross-check before writing: the same numbers from float64 central differences of the mpmath value (agreement 1e-6).

Run from the repo root:  python tests/golden/make_golden_v3.py     (about two minutes)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from madaiemulator_amd import synth  # noqa: E402
import make_golden as G1  # noqa: E402

import mpmath as mp  # noqa: E402
mp.mp.dps = 50
f = mp.mpf

HERE = os.path.dirname(os.path.abspath(__file__))
INP = os.path.join(HERE, "ref_inputs")


def neg_loglik(kind, order, X, y, th_less):
    """-logL at theta = [0, th_less...] as gpemu_loglik defines it in the corrected modes; th_less may hold mpf values"""
    N, d = X.shape
    Cm = mp.matrix(N, N)
    eps = 1e-10 if kind == 1 else 1e-16
    nug = mp.exp(th_less[0])
    for i in range(N):
        for j in range(N):
            same = all(abs(X[i, k] - X[j, k]) < eps for k in range(d))
            if kind == 1:
                e = f(0)
                for k in range(d):
                    r = mp.exp(th_less[1 + k])
                    dd = f(X[i, k]) - f(X[j, k])
                    e += f(-0.5) * dd * dd / (r * r)
                c = mp.exp(e)                                        # amplitude e^0
            else:
                r = mp.sqrt(sum((f(X[i, k]) - f(X[j, k])) ** 2 for k in range(d)))
                s = r / mp.exp(th_less[1])
                if kind == 2:
                    c = (1 + f("1.732050808") * s) * mp.exp(-f("1.732050808") * s)
                else:
                    c = (1 + f("2.236067978") * s + (f(5) / 3) * s * s) * mp.exp(-f("2.236067978") * s)
            Cm[i, j] = c + (nug if same else 0)
    L = mp.cholesky(Cm)
    logdet = 2 * sum(mp.log(L[i, i]) for i in range(N))
    H = mp.matrix(G1.hmat(order, X).tolist())
    yv = mp.matrix(y.tolist())
    # A [y|H] by two triangular solves per column
    AyH = mp.cholesky_solve(Cm, yv), [mp.cholesky_solve(Cm, H[:, a]) for a in range(H.cols)]
    HAH = mp.matrix(H.cols, H.cols)
    HAy = mp.matrix(H.cols, 1)
    for a in range(H.cols):
        HAy[a] = (H[:, a].T * AyH[0])[0]
        for b in range(H.cols):
            HAH[a, b] = (H[:, a].T * AyH[1][b])[0]
    beta = mp.lu_solve(HAH, HAy)
    r = yv - H * beta
    quad = (r.T * mp.cholesky_solve(Cm, r))[0]
    ll = -f(0.5) * logdet - (f(N) / 2) * f("1.83788") - f(0.5) * quad
    return -ll


def gradient(kind, order, X, y, th_less):
    g = []
    for k in range(len(th_less)):
        def fk(t, k=k):
            th = [f(v) for v in th_less]
            th[k] = t
            return neg_loglik(kind, order, X, y, th)
        g.append(mp.diff(fk, f(th_less[k])))
    return np.array([float(v) for v in g])


def main():
    X, Y = synth.read_input_model_file(os.path.join(INP, "multi-simple.input_model_file.dat"))
    X, y = X[:34].copy(), Y[:34, 0].copy()
    d = X.shape[1]
    cases = [(3, 1, np.array([-3.0, np.log(0.7)])), (3, 0, np.array([-2.0, np.log(0.4)])),
             (2, 1, np.array([-3.5, np.log(0.9)])), (1, 1, np.array([-4.0, np.log(0.6), np.log(0.8), np.log(0.5)]))]
    out = dict(X=X, y=y, ncases=np.array(len(cases)))
    for c, (kind, order, th) in enumerate(cases):
        val = float(neg_loglik(kind, order, X, y, [f(v) for v in th]))
        g = gradient(kind, order, X, y, th)
        # float64 central differences of the 50-digit value: a different differentiation of the same function
        h = 1e-4
        g2 = np.empty_like(g)
        for k in range(len(th)):
            tp, tm = th.copy(), th.copy()
            tp[k] += h
            tm[k] -= h
            g2[k] = float((neg_loglik(kind, order, X, y, [f(v) for v in tp]) - neg_loglik(kind, order, X, y, [f(v) for v in tm])) / (2 * h))
        err = np.max(np.abs(g - g2)) / np.max(np.abs(g))
        print("case", c, "kind", kind, "order", order, "value", val, "grad", g, "fd check %.1e" % err)
        if not err < 1e-6:
            raise SystemExit("finite-difference cross-check FAILED")
        out[f"kind{c}"] = np.array(kind)
        out[f"order{c}"] = np.array(order)
        out[f"th{c}"] = np.concatenate([[0.0], th])
        out[f"value{c}"] = np.array(val)
        out[f"grad{c}"] = g
    path = os.path.join(HERE, "golden_v3.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

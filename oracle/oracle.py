"""ctypes front end of the CPU oracle (oracle/gp_oracle.c).

TEST INFRASTRUCTURE ONLY -- importable from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg, never from the product package.
PARITY UNPINNED: see the header of gp_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")

POWEREXP, MATERN32, MATERN52 = 1, 2, 3  # optstruct.h:12-14

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


def build(force=False):
    src = os.path.join(_HERE, "gp_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.orc_cov.restype = C.c_double
        L.orc_cov.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int]
        L.orc_make_cov_matrix.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int, C.c_int]
        L.orc_make_kvector.argtypes = [C.c_int, _dp, _dp, _dp, _dp, C.c_int, C.c_int]
        L.orc_make_hmatrix.argtypes = [C.c_int, _dp, _dp, C.c_int, C.c_int]
        L.orc_cholesky_decomp.restype = C.c_int
        L.orc_cholesky_decomp.argtypes = [_dp, C.c_int]
        L.orc_cholesky_invert.argtypes = [_dp, C.c_int]
        L.orc_evalFnMulti.restype = C.c_double
        L.orc_evalFnMulti.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int,
                                      _dp, _dp, _dp, _dp, _ip]
        L.orc_gradFnMulti.restype = C.c_int
        L.orc_gradFnMulti.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, _dp]
        L.orc_emulator_setup.restype = C.c_int
        L.orc_emulator_setup.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, _dp, _dp, _dp, _dp]
        L.orc_emulate_points.restype = C.c_int
        L.orc_emulate_points.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_int,
                                         _dp, C.c_int, _dp, _dp]
        for name in ("orc_derivative_l_gauss", "orc_derivative_l_matern_three", "orc_derivative_l_matern_five"):
            getattr(L, name).argtypes = [_dp, _dp, C.c_double, C.c_int, C.c_int, C.c_int]
        L.orc_pca_backproject.argtypes = [C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _dp]
        _lib = L
    return _lib


def _a(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def nthetas_for(kind, d):
    """modelstruct.c:300-308: pow-exp has d+2 thetas, the Matern kernels 3."""
    return d + 2 if kind == POWEREXP else 3


def nreg_for(order, d):
    return 1 + order * d


def cov(kind, xm, xn, thetas):
    xm, xn, thetas = _a(xm), _a(xn), _a(thetas)
    return lib().orc_cov(kind, _p(xm), _p(xn), _p(thetas), xm.size)


def cov_matrix(kind, X, thetas):
    X, thetas = _a(X), _a(thetas)
    N, d = X.shape
    Cm = np.empty((N, N))
    lib().orc_make_cov_matrix(kind, _p(Cm), _p(X), _p(thetas), N, d)
    return Cm


def kvector(kind, X, xnew, thetas):
    X, xnew, thetas = _a(X), _a(xnew), _a(thetas)
    N, d = X.shape
    k = np.empty(N)
    lib().orc_make_kvector(kind, _p(k), _p(X), _p(xnew), _p(thetas), N, d)
    return k


def hmatrix(order, X):
    X = _a(X)
    N, d = X.shape
    H = np.empty((N, nreg_for(order, d)))
    lib().orc_make_hmatrix(order, _p(H), _p(X), N, d)
    return H


def cholesky_decomp(A):
    A = _a(A).copy()
    st = lib().orc_cholesky_decomp(_p(A), A.shape[0])
    return A, st


def cholesky_invert(LLT):
    A = _a(LLT).copy()
    lib().orc_cholesky_invert(_p(A), A.shape[0])
    return A


def eval_fn_multi(kind, order, X, y, theta_less_amp, det_mode=1):
    """-> dict(value=-logL, sigma2, beta, logdet, quad, info) -- maxmultimin.c:288-394."""
    X, y, th = _a(X), _a(y), _a(theta_less_amp)
    N, d = X.shape
    nthetas = th.size + 1
    beta = np.full(nreg_for(order, d), np.nan)
    s2, ld, qd = C.c_double(np.nan), C.c_double(np.nan), C.c_double(np.nan)
    info = C.c_int(0)
    v = lib().orc_evalFnMulti(kind, order, _p(X), _p(y), _p(th), N, d, nthetas, det_mode,
                              C.byref(s2), _p(beta), C.byref(ld), C.byref(qd), C.byref(info))
    return dict(value=v, sigma2=s2.value, beta=beta, logdet=ld.value, quad=qd.value, info=info.value)


def grad_fn_multi(kind, order, X, y, theta_less_amp):
    X, y, th = _a(X), _a(y), _a(theta_less_amp)
    N, d = X.shape
    g = np.full(th.size, np.nan)
    st = lib().orc_gradFnMulti(kind, order, _p(X), _p(y), _p(th), N, d, th.size + 1, _p(g))
    return g, st


def derivative_l(kind, X, theta_length, index):
    X = _a(X)
    N, d = X.shape
    dC = np.empty((N, N))
    fn = {POWEREXP: lib().orc_derivative_l_gauss, MATERN32: lib().orc_derivative_l_matern_three,
          MATERN52: lib().orc_derivative_l_matern_five}[kind]
    fn(_p(dC), _p(X), float(theta_length), index, N, d)
    return dC


class Emulator:
    """alloc_emulator_struct + emulate_point (emulator_struct.c:13-37,124-143)."""

    def __init__(self, kind, order, X, y, thetas):
        self.kind, self.order = kind, order
        self.X, self.y, self.thetas = _a(X), _a(y), _a(thetas)
        N, d = self.X.shape
        self.N, self.d = N, d
        self.cinverse = np.empty((N, N))
        self.beta = np.empty(nreg_for(order, d))
        self.H = np.empty((N, nreg_for(order, d)))
        ld = C.c_double(np.nan)
        self.status = lib().orc_emulator_setup(kind, order, _p(self.X), _p(self.y), _p(self.thetas), N, d,
                                               _p(self.cinverse), _p(self.beta), _p(self.H), C.byref(ld))
        self.logdet = ld.value

    def emulate(self, Xq):
        Xq = _a(Xq).reshape(-1, self.d)
        M = Xq.shape[0]
        mean, var = np.empty(M), np.empty(M)
        st = lib().orc_emulate_points(self.kind, self.order, _p(self.X), _p(self.y), _p(self.thetas),
                                      _p(self.cinverse), _p(self.beta), _p(self.H), self.N, self.d,
                                      _p(Xq), M, _p(mean), _p(var))
        return mean, var, st


def pca_backproject(ybar, evals, evecs, m, v):
    ybar, evals, evecs, m, v = map(_a, (ybar, evals, evecs, m, v))
    nt, nr = evecs.shape
    mo, vo = np.empty(nt), np.empty(nt)
    lib().orc_pca_backproject(nt, nr, _p(ybar), _p(evals), _p(evecs), _p(m), _p(v), _p(mo), _p(vo))
    return mo, vo

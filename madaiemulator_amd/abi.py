"""ctypes binding of the C-ABI in include/gpemu.h (lib/libgpemu_hip.so).

This is the same binding a reference maintainer would write for any FFI
(INTEGRATION.md); the tests and bench.py drive the device library through
it.  There is no CPU fallback: if the library is missing it is an error.
"""
import ctypes as C
import os

import numpy as np

from . import build as _build

POWEREXP, MATERN32, MATERN52 = 1, 2, 3

OK, ERR_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_NOT_PD, ERR_REGRESSION, ERR_STATE = range(7)
PROF_NONE, PROF_GEMM, PROF_FILL, PROF_LEAF, PROF_POTRF, PROF_GEMM_BIG, PROF_GEMM_K512 = range(7)
MODE_EXACT_GRAD, MODE_MATERN_LOG = 1, 2
RESULT_RING = 4

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

# every symbol include/gpemu.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "gpemu_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "gpemu_ctx_destroy": (None, [C.c_void_p]),
    "gpemu_last_error": (C.c_char_p, [C.c_void_p]),
    "gpemu_version": (C.c_char_p, []),
    "gpemu_device_count": (C.c_int, []),
    "gpemu_device_memory": (C.c_int, [C.c_int, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "gpemu_rccl_allgather": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_char_p, _dp, C.c_int, _dp, C.c_char_p, C.c_size_t]),
    "gpemu_rccl_unique_id": (C.c_int, [C.c_void_p, C.c_char_p, C.c_size_t]),
    "gpemu_rccl_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]),
    "gpemu_rccl_comm_allgather": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, C.c_char_p, C.c_size_t]),
    "gpemu_rccl_comm_destroy": (None, [C.c_void_p]),
    "gpemu_set_model": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, _dp, _dp]),
    "gpemu_set_training": (C.c_int, [C.c_void_p, _dp]),
    "gpemu_cov_matrix": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp]),
    "gpemu_kvectors": (C.c_int, [C.c_void_p, _dp, C.c_int, C.c_int, _dp, _dp]),
    "gpemu_loglik": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpemu_loglik_enqueue": (C.c_int, [C.c_void_p, _dp, C.c_int]),
    "gpemu_loglik_collect": (C.c_int, [C.c_void_p, _dp, _dp, _dp, _dp, _dp, _ip]),
    "gpemu_loglik_batch": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _ip]),
    "gpemu_loglik_batch_enqueue": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int]),
    "gpemu_loglik_batch_collect": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _ip]),
    "gpemu_loglik_batch_collect_back": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _dp, _ip, _ip]),
    "gpemu_grad": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, _ip]),
    "gpemu_loglik_grad": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, _dp, _dp, _dp, _ip]),
    "gpemu_loglik_grad_batch": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int, _dp, _dp, _dp, _dp, _ip, _ip]),
    "gpemu_loglik_grad_batch_enqueue": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int]),
    "gpemu_loglik_grad_batch_collect": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp, _dp, _ip, _ip]),
    "gpemu_loglik_grad_batch_collect_back": (C.c_int, [C.c_void_p, C.c_int, C.c_int, _dp, _dp, _dp, _dp, _ip, _ip]),
    "gpemu_predict_setup": (C.c_int, [C.c_void_p, _dp, C.c_int, _dp, _ip]),
    "gpemu_predict_setup_batch": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, _dp, C.c_int, _dp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "gpemu_warm_start": (C.c_int, [C.c_int]),
    "gpemu_get_cinverse": (C.c_int, [C.c_void_p, _dp]),
    "gpemu_predict_batch": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp, _dp]),
    "gpemu_predict_batch_enqueue": (C.c_int, [C.c_void_p, C.c_int, _dp]),
    "gpemu_predict_batch_collect": (C.c_int, [C.c_void_p, C.c_int, _dp, _dp]),
    "gpemu_predict_batch_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "gpemu_chol_inverse": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int, _dp, _ip]),
    "gpemu_symm_apply": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int, C.c_int, _dp, _dp]),
    "gpemu_symm_invalidate": (C.c_int, [C.c_void_p]),
    "gpemu_symm_pin": (C.c_int, [C.c_void_p, C.c_int]),
    "gpemu_set_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "gpemu_get_mode": (C.c_int, [C.c_void_p]),
    "gpemu_derivative_gauss": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_double, _dp, C.c_int]),
    "gpemu_trace_product": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int, _dp, C.c_int, _dp]),
    "gpemu_dev_alloc": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]),
    "gpemu_dev_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "gpemu_dev_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "gpemu_dev_download": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "gpemu_sync": (C.c_int, [C.c_void_p]),
    "gpemu_prof_begin": (C.c_int, [C.c_void_p, C.c_int]),
    "gpemu_prof_end": (C.c_int, [C.c_void_p, _ip, _dp, _dp, _dp]),
    "gpemu_trace_dump": (C.c_int, [C.c_void_p, C.c_char_p]),
    "gpemu_test_gemm_nt": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, _dp, _dp, _dp]),
    "gpemu_test_gemm_bench": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_int, _dp, _dp]),
    "gpemu_test_potrf": (C.c_int, [C.c_void_p, C.c_int, _dp, _ip]),
    "gpemu_test_staged_matrix": (C.c_int, [C.c_void_p, C.c_int, _dp, C.c_int, C.c_int, _dp]),
    "gpemu_test_tile_table": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _ip, C.c_int]),
    "gpemu_test_row_table": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _ip, C.c_int]),
}

_lib = None


def lib_path():
    return _build.HIP_LIB


def load():
    """dlopen the device library and bind every declared symbol (fails loudly if absent)."""
    global _lib
    if _lib is None:
        path = lib_path()
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `python -m madaiemulator_amd.build` (no CPU fallback exists)")
        L = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


class GpemuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"gpemu error {code}: {msg}")
        self.code = code


def _a(x):
    return np.ascontiguousarray(x, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp)


def nthetas_for(kind, d):
    return d + 2 if kind == POWEREXP else 3


def predict_setup_batch(ctxs, thetas):
    """gpemu_predict_setup_batch: the contexts of the components of a multi-output model (same design, own training vector)
    set up as ONE lock-step batch.  Returns (beta [n x nreg], info [n], status [n], rc)."""
    th = _a(thetas).reshape(len(ctxs), -1)
    n = len(ctxs)
    hs = (C.c_void_p * n)(*[c.h for c in ctxs])
    beta = np.full((n, ctxs[0].nreg), np.nan)
    info = np.zeros(n, dtype=np.int32)
    status = np.zeros(n, dtype=np.int32)
    rc = ctxs[0].L.gpemu_predict_setup_batch(hs, n, _p(th), th.shape[1], _p(beta), info.ctypes.data_as(C.POINTER(C.c_int)),
                                            status.ctypes.data_as(C.POINTER(C.c_int)))
    if rc not in (OK, ERR_NOT_PD, ERR_REGRESSION):
        raise GpemuError(rc, ctxs[0].L.gpemu_last_error(ctxs[0].h).decode())
    return beta, info, status, rc


class Context:
    """One gpemu_ctx: one HIP stream + its HBM workspace.  One per host thread."""

    def __init__(self, device=0):
        self.L = load()
        h = C.c_void_p()
        rc = self.L.gpemu_ctx_create(C.byref(h), device)
        if rc != OK:
            raise GpemuError(rc, "gpemu_ctx_create failed (no usable HIP device?)")
        self.h = h
        self.N = self.d = self.nreg = 0
        self.kind = self.order = 0

    def close(self):
        if getattr(self, "h", None):
            self.L.gpemu_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, allow=()):
        if rc != OK and rc not in allow:
            raise GpemuError(rc, self.L.gpemu_last_error(self.h).decode())
        return rc

    # -- model ---------------------------------------------------------
    def set_model(self, kind, order, X, y):
        X, y = _a(X), _a(y)
        N, d = X.shape
        self._chk(self.L.gpemu_set_model(self.h, kind, order, N, d, _p(X), _p(y)))
        self.kind, self.order, self.N, self.d, self.nreg = kind, order, N, d, 1 + order * d

    def set_training(self, y):
        y = _a(y)
        self._chk(self.L.gpemu_set_training(self.h, _p(y)))

    # -- a4 / a16 --------------------------------------------------------
    def cov_matrix(self, thetas):
        th = _a(thetas)
        out = np.empty((self.N, self.N))
        self._chk(self.L.gpemu_cov_matrix(self.h, _p(th), th.size, _p(out)))
        return out

    def kvectors(self, thetas, Xq):
        th, Xq = _a(thetas), _a(Xq).reshape(-1, self.d)
        out = np.empty((Xq.shape[0], self.N))
        self._chk(self.L.gpemu_kvectors(self.h, _p(th), th.size, Xq.shape[0], _p(Xq), _p(out)))
        return out

    # -- a11 ---------------------------------------------------------------
    def loglik(self, thetas):
        """-> dict(value=-logL, sigma2, beta, logdet, quad, info, status)"""
        th = _a(thetas)
        v, s2, ld, qd = (C.c_double(np.nan) for _ in range(4))
        info = C.c_int(0)
        beta = np.full(self.nreg, np.nan)
        rc = self.L.gpemu_loglik(self.h, _p(th), th.size, C.byref(v), C.byref(s2), _p(beta), C.byref(ld), C.byref(qd),
                                 C.byref(info))
        self._chk(rc, allow=(ERR_NOT_PD, ERR_REGRESSION))
        return dict(value=v.value, sigma2=s2.value, beta=beta, logdet=ld.value, quad=qd.value, info=info.value,
                    status=rc)

    def loglik_enqueue(self, thetas):
        th = _a(thetas)
        self._chk(self.L.gpemu_loglik_enqueue(self.h, _p(th), th.size))

    def loglik_collect(self):
        v, s2, ld, qd = (C.c_double(np.nan) for _ in range(4))
        info = C.c_int(0)
        beta = np.full(self.nreg, np.nan)
        rc = self.L.gpemu_loglik_collect(self.h, C.byref(v), C.byref(s2), _p(beta), C.byref(ld), C.byref(qd),
                                         C.byref(info))
        self._chk(rc, allow=(ERR_NOT_PD, ERR_REGRESSION))
        return dict(value=v.value, sigma2=s2.value, beta=beta, logdet=ld.value, quad=qd.value, info=info.value,
                    status=rc)

    def loglik_batch_enqueue(self, thetas):
        th = _a(thetas).reshape(-1, np.shape(thetas)[-1])
        self._nb = th.shape[0]
        self._chk(self.L.gpemu_loglik_batch_enqueue(self.h, th.shape[0], _p(th), th.shape[1]))

    def loglik_batch_collect(self):
        nb = self._nb
        v, s2, ld, qd = (np.full(nb, np.nan) for _ in range(4))
        beta = np.full((nb, self.nreg), np.nan)
        info = np.zeros(nb, dtype=np.int32)
        status = np.zeros(nb, dtype=np.int32)
        self._chk(self.L.gpemu_loglik_batch_collect(self.h, nb, _p(v), _p(s2), _p(beta), _p(ld), _p(qd),
                                                    info.ctypes.data_as(_ip), status.ctypes.data_as(_ip)))
        return dict(value=v, sigma2=s2, beta=beta, logdet=ld, quad=qd, info=info, status=status)

    def loglik_batch_collect_back(self, back, nb):
        """results of the batch enqueued `back` batches before the newest (ring of RESULT_RING); waits for it only"""
        v, s2, ld, qd = (np.full(nb, np.nan) for _ in range(4))
        beta = np.full((nb, self.nreg), np.nan)
        info = np.zeros(nb, dtype=np.int32)
        status = np.zeros(nb, dtype=np.int32)
        self._chk(self.L.gpemu_loglik_batch_collect_back(self.h, back, nb, _p(v), _p(s2), _p(beta), _p(ld), _p(qd),
                                                         info.ctypes.data_as(_ip), status.ctypes.data_as(_ip)))
        return dict(value=v, sigma2=s2, beta=beta, logdet=ld, quad=qd, info=info, status=status)

    def loglik_batch(self, thetas):
        """nb evaluations in lock-step (thetas: nb x nthetas) -> dict of arrays, as loglik() per element"""
        self.loglik_batch_enqueue(thetas)
        return self.loglik_batch_collect()

    def loglik_grad_batch(self, thetas):
        """value + gradient at nb thetas in lock-step -> dict(value, sigma2, beta, grad (nb x nthetas-1), info, status)"""
        th = _a(thetas).reshape(-1, np.shape(thetas)[-1])
        nb, nt = th.shape
        v, s2 = np.full(nb, np.nan), np.full(nb, np.nan)
        beta = np.full((nb, self.nreg), np.nan)
        g = np.full((nb, nt - 1), np.nan)
        info = np.zeros(nb, dtype=np.int32)
        status = np.zeros(nb, dtype=np.int32)
        self._chk(self.L.gpemu_loglik_grad_batch(self.h, nb, _p(th), nt, _p(v), _p(s2), _p(beta), _p(g),
                                                 info.ctypes.data_as(_ip), status.ctypes.data_as(_ip)))
        return dict(value=v, sigma2=s2, beta=beta, grad=g, info=info, status=status)

    def loglik_grad_batch_enqueue(self, thetas):
        """puts a value+gradient batch on the context's stream and returns at once"""
        th = _a(thetas).reshape(-1, np.shape(thetas)[-1])
        self._gnb, self._gnt = th.shape
        self._chk(self.L.gpemu_loglik_grad_batch_enqueue(self.h, th.shape[0], _p(th), th.shape[1]))

    def loglik_grad_batch_collect_back(self, back, nb=None, nthetas=None):
        """results of the value+gradient batch enqueued `back` batches before the newest; waits for it only"""
        nb = self._gnb if nb is None else nb
        nt = self._gnt if nthetas is None else nthetas
        v, s2 = np.full(nb, np.nan), np.full(nb, np.nan)
        beta = np.full((nb, self.nreg), np.nan)
        g = np.full((nb, nt - 1), np.nan)
        info = np.zeros(nb, dtype=np.int32)
        status = np.zeros(nb, dtype=np.int32)
        self._chk(self.L.gpemu_loglik_grad_batch_collect_back(self.h, back, nb, _p(v), _p(s2), _p(beta), _p(g),
                                                              info.ctypes.data_as(_ip), status.ctypes.data_as(_ip)))
        return dict(value=v, sigma2=s2, beta=beta, grad=g, info=info, status=status)

    def loglik_grad_batch_collect(self):
        return self.loglik_grad_batch_collect_back(0)

    # -- a12 ---------------------------------------------------------------
    def grad(self, thetas):
        th = _a(thetas)
        g = np.full(th.size - 1, np.nan)
        info = C.c_int(0)
        rc = self._chk(self.L.gpemu_grad(self.h, _p(th), th.size, _p(g), C.byref(info)), allow=(ERR_NOT_PD,))
        return g, rc

    def loglik_grad(self, thetas):
        th = _a(thetas)
        g = np.full(th.size - 1, np.nan)
        beta = np.full(self.nreg, np.nan)
        v, s2 = C.c_double(np.nan), C.c_double(np.nan)
        info = C.c_int(0)
        rc = self._chk(self.L.gpemu_loglik_grad(self.h, _p(th), th.size, C.byref(v), C.byref(s2), _p(beta), _p(g),
                                                C.byref(info)), allow=(ERR_NOT_PD,))
        return dict(value=v.value, sigma2=s2.value, beta=beta, grad=g, info=info.value, status=rc)

    # -- a15 / a19 ---------------------------------------------------------
    def predict_setup(self, thetas):
        th = _a(thetas)
        beta = np.full(self.nreg, np.nan)
        info = C.c_int(0)
        rc = self._chk(self.L.gpemu_predict_setup(self.h, _p(th), th.size, _p(beta), C.byref(info)),
                       allow=(ERR_NOT_PD, ERR_REGRESSION))
        return beta, rc

    def predict_enqueue(self, Xq):
        Xq = _a(Xq).reshape(-1, self.d)
        self._npred = Xq.shape[0]
        self._chk(self.L.gpemu_predict_batch_enqueue(self.h, Xq.shape[0], _p(Xq)))

    def predict_collect(self):
        m, v = np.empty(self._npred), np.empty(self._npred)
        self._chk(self.L.gpemu_predict_batch_collect(self.h, self._npred, _p(m), _p(v)))
        return m, v

    def chol_inverse(self, A):
        A = _a(A).copy()
        ld, info = C.c_double(np.nan), C.c_int(0)
        rc = self._chk(self.L.gpemu_chol_inverse(self.h, A.shape[0], _p(A), A.shape[1], C.byref(ld), C.byref(info)),
                       allow=(ERR_NOT_PD,))
        return A, ld.value, info.value, rc

    def symm_apply(self, A, V):
        A, V = _a(A), _a(V).reshape(-1, np.shape(A)[0])
        out = np.empty_like(V)
        self._chk(self.L.gpemu_symm_apply(self.h, A.shape[0], _p(A), A.shape[1], V.shape[0], _p(V), _p(out)))
        return out

    def set_mode(self, flags):
        self._chk(self.L.gpemu_set_mode(self.h, int(flags)))

    def get_mode(self):
        return self.L.gpemu_get_mode(self.h)

    def symm_invalidate(self):
        self._chk(self.L.gpemu_symm_invalidate(self.h))

    def symm_pin(self, pinned=True):
        self._chk(self.L.gpemu_symm_pin(self.h, 1 if pinned else 0))

    def derivative_gauss(self, xcol, theta_len):
        xcol = _a(xcol)
        out = np.empty((xcol.size, xcol.size))
        self._chk(self.L.gpemu_derivative_gauss(self.h, xcol.size, _p(xcol), float(theta_len), _p(out), xcol.size))
        return out

    def trace_product(self, A, B):
        A, B = _a(A), _a(B)
        t = C.c_double(np.nan)
        self._chk(self.L.gpemu_trace_product(self.h, A.shape[0], _p(A), A.shape[1], _p(B), B.shape[1], C.byref(t)))
        return t.value

    def cinverse(self):
        out = np.empty((self.N, self.N))
        self._chk(self.L.gpemu_get_cinverse(self.h, _p(out)))
        return out

    def predict(self, Xq):
        Xq = _a(Xq).reshape(-1, self.d)
        M = Xq.shape[0]
        mean, var = np.empty(M), np.empty(M)
        self._chk(self.L.gpemu_predict_batch(self.h, M, _p(Xq), _p(mean), _p(var)))
        return mean, var

    def predict_dev(self, M, xq_dev, mean_dev, var_dev):
        self._chk(self.L.gpemu_predict_batch_dev(self.h, M, xq_dev, mean_dev, var_dev))

    # -- memory / sync / profiling ------------------------------------------
    def dev_alloc(self, nbytes):
        p = C.c_void_p()
        self._chk(self.L.gpemu_dev_alloc(self.h, nbytes, C.byref(p)))
        return p

    def dev_free(self, p):
        self._chk(self.L.gpemu_dev_free(self.h, p))

    def upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.L.gpemu_dev_upload(self.h, dptr, arr.ctypes.data_as(C.c_void_p), arr.nbytes))

    def download(self, dptr, shape, dtype=np.float64):
        out = np.empty(shape, dtype=dtype)
        self._chk(self.L.gpemu_dev_download(self.h, out.ctypes.data_as(C.c_void_p), dptr, out.nbytes))
        return out

    def sync(self):
        self._chk(self.L.gpemu_sync(self.h))

    def prof_begin(self, cls):
        self._chk(self.L.gpemu_prof_begin(self.h, cls))

    def prof_end(self):
        n = C.c_int(0)
        ms, fl, by = C.c_double(0), C.c_double(0), C.c_double(0)
        self._chk(self.L.gpemu_prof_end(self.h, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
        return dict(n=n.value, ms=ms.value, flops=fl.value, bytes=by.value)

    # -- building blocks ------------------------------------------------------
    def test_gemm_nt(self, A, B, Cm, alpha=1.0, beta=0):
        A, B, Cm = _a(A), _a(B), _a(Cm).copy()
        m, k = A.shape
        n = B.shape[0]
        self._chk(self.L.gpemu_test_gemm_nt(self.h, m, n, k, alpha, beta, _p(A), _p(B), _p(Cm)))
        return Cm

    def trace_dump(self, path):
        self._chk(self.L.gpemu_trace_dump(self.h, str(path).encode()))

    def gemm_bench(self, m, n, k, ld=0, cfg=0, tri=0, beta=1, reps=5):
        ms, fl = C.c_double(0), C.c_double(0)
        self._chk(self.L.gpemu_test_gemm_bench(self.h, m, n, k, ld, cfg, tri, beta, reps, C.byref(ms), C.byref(fl)))
        return ms.value, fl.value

    def staged_matrix(self, thetas, b=0):
        """N x N block of matrix b as the batch staging launch (FILL_LOWER) leaves it; upper tiles are not written"""
        th = _a(thetas).reshape(-1, np.shape(thetas)[-1])
        out = np.zeros((self.N, self.N))
        self._chk(self.L.gpemu_test_staged_matrix(self.h, th.shape[0], _p(th), th.shape[1], int(b), _p(out)))
        return out

    def test_potrf(self, A):
        A = _a(A).copy()
        info = C.c_int(0)
        self._chk(self.L.gpemu_test_potrf(self.h, A.shape[0], _p(A), C.byref(info)))
        return A, info.value

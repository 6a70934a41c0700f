"""Independent numpy/LAPACK evaluation of the value+gradient path at sizes the oracle's N^3 loops cannot reach in a test.

TEST INFRASTRUCTURE (imported by tests/ and tests/golden/make_golden_grad_n2048.py only).  Nothing here comes from the
device library or from oracle/gp_oracle.c: the covariance matrix is a vectorised restatement of emulator.c:101-152, the
inverse is LAPACK's (scipy cho_factor / cho_solve), and the two gradient forms are written in their O(N^2 d) shape

  literal (gradFnMulti, maxmultimin.c:416-550 + getGradientCn :571-608 + derivative_l_gauss emulator.c:173-209):
      A = C(theta0 = 0)^-1, alpha = A y, sigma^2 = y.A.(y - H beta)/N, nug = e^theta1,
      G(dC) = -1/2 sum_ab A_ab dC_ba + 1/2 alpha^T dC alpha,
      g[0] = -G(nug I),   g[k+1] = -G(sigma^2 * dC_k),  dC_k,ab = exp(-1/2 e^{-2 t_k} D^2 - 2 t_k) D^2, D = x_ak - x_bk
  exact (include/gpemu.h GPEMU_MODE_EXACT_GRAD): d(-logL)/dtheta = 1/2 sum_ab (A_ab - a_a a_b) dC_ab, a = A (y - H beta),
      dC/dtheta_{k+2} = C0_ab D_k^2 e^{-2 t_k} (C0 = the matrix without its nugget), dC/dtheta_1 = nug [same point]
"""
import numpy as np
import scipy.linalg as sl


def hmatrix(order, X):
    """regression.c:9-67,100-112: h(x) = [1, x_1..x_d, x_1^2..x_d^2, x_1^3..x_d^3] up to `order`"""
    cols = [np.ones(X.shape[0])]
    for q in range(1, order + 1):
        cols += [X[:, k] ** q for k in range(X.shape[1])]
    return np.column_stack(cols)


def powexp_matrix(X, th, with_nugget=True):
    """emulator.c:101-152 vectorised: e^t0 exp(-1/2 sum_k D_k^2 / e^{2 t_{k+2}}) + e^t1 [all |D_k| < 1e-10]; returns (C, same)"""
    N, d = X.shape
    E = np.zeros((N, N))
    same = np.ones((N, N), dtype=bool)
    for k in range(d):
        D = X[:, k][:, None] - X[:, k][None, :]
        same &= np.abs(D) < 1e-10
        r = np.exp(th[k + 2])
        D *= D
        D *= -0.5 / (r * r)
        E += D
    np.exp(E, out=E)
    E *= np.exp(th[0])
    if with_nugget:
        E[same] += np.exp(th[1])
    return E, same


def _inverse(Cm):
    cf = sl.cho_factor(Cm, lower=True, overwrite_a=False, check_finite=False)
    A = sl.cho_solve(cf, np.eye(Cm.shape[0]), check_finite=False)
    return 0.5 * (A + A.T), 2.0 * np.log(np.diag(cf[0])).sum()


def value_and_gradients(X, y, order, th):
    """-> dict(value, sigma2, beta, logdet, literal (d+1), exact (d+1)) for the pow-exp kernel at theta (theta[0] taken as 0)"""
    N, d = X.shape
    th = np.array(th, float)
    th[0] = 0.0                                              # maxmultimin.c:311,441
    C0, same = powexp_matrix(X, th, with_nugget=False)
    nug = np.exp(th[1])
    Cm = C0.copy()
    Cm[same] += nug
    A, logdet = _inverse(Cm)
    del Cm
    H = hmatrix(order, X)
    AH, Ay = A @ H, A @ y
    beta = np.linalg.solve(H.T @ AH, H.T @ Ay)
    r = y - H @ beta
    Ar = A @ r
    sigma2 = (y @ Ar) / N                                    # maxmultimin.c:259-263 (y, not r, on the left)
    quad = r @ Ar
    value = -(-0.5 * logdet - (N / 2.0) * 1.83788 - 0.5 * quad)
    lit = np.empty(d + 1)
    exa = np.empty(d + 1)
    lit[0] = -(-0.5 * nug * np.trace(A) + 0.5 * nug * (Ay @ Ay))
    W = A - np.outer(Ar, Ar)                                 # exact form's weight
    exa[0] = 0.5 * nug * W[same].sum()
    for k in range(d):
        D2 = X[:, k][:, None] - X[:, k][None, :]
        D2 *= D2
        t = th[k + 2]
        dC = np.exp(-0.5 * np.exp(-2.0 * t) * D2 - 2.0 * t) * D2          # emulator.c:203
        G = -0.5 * np.sum(A * dC) + 0.5 * (Ay @ dC @ Ay)
        lit[k + 1] = -(sigma2 * G)
        del dC
        D2 *= np.exp(-2.0 * t)
        D2 *= C0
        exa[k + 1] = 0.5 * np.sum(W * D2)
    return dict(value=value, sigma2=sigma2, beta=beta, logdet=logdet, quad=quad, literal=lit, exact=exa)

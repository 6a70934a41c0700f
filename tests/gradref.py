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

and, since round 5, the posterior mean / variance of emulate_point (emulator_struct.c:124-143; emulator.c:578-593, 672-785:
SURVEY App. A.4) and the back-projection of emulate_point_multi (multivar_support.c:126-151) in the same style: `predict`.
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


def kvectors(X, th, Xq):
    """emulator.c:578-593 for a block of query rows: K[q, i] = cov(x_i, x*_q) of emulator.c:101-152 (amplitude e^t0 included,
    nugget where every |coordinate difference| < 1e-10), then the clamp `if (cov < 1e-10) cov = 0`"""
    M, d = Xq.shape
    E = np.zeros((M, X.shape[0]))
    same = np.ones((M, X.shape[0]), dtype=bool)
    for k in range(d):
        D = Xq[:, k][:, None] - X[:, k][None, :]
        same &= np.abs(D) < 1e-10
        r = np.exp(th[k + 2])
        E += -0.5 * D * D / (r * r)
    K = np.exp(th[0]) * np.exp(E)
    K[same] += np.exp(th[1])
    K[K < 1e-10] = 0.0
    return K


def predict(X, y, order, th, Xq):
    """emulate_point at the rows of Xq with the pow-exp kernel at the STORED thetas (amplitude e^theta0 included):
    mean = h.beta + k.(A y) - k.(A H beta); var = kappa - k.A.k + q.(H^T A H)^-1.q, q = h - (A H)^T k, kappa = e^t0 + e^t1
    (x* against itself: the nugget is in).  LAPACK factor + solves; returns (mean, var)."""
    th = np.asarray(th, float)
    Cm, _ = powexp_matrix(X, th)
    cf = sl.cho_factor(Cm, lower=True, overwrite_a=True, check_finite=False)
    H = hmatrix(order, X)
    AyH = sl.cho_solve(cf, np.column_stack([y, H]), check_finite=False)
    Ay, AH = AyH[:, 0], AyH[:, 1:]
    HAH = H.T @ AH
    beta = np.linalg.solve(HAH, H.T @ Ay)
    K = kvectors(X, th, Xq)
    hq = hmatrix(order, Xq)
    mean = hq @ beta + K @ Ay - K @ (AH @ beta)
    AK = sl.cho_solve(cf, K.T, check_finite=False)                    # N x M
    q = hq - K @ AH                                                   # M x nreg
    kappa = np.exp(th[0]) + np.exp(th[1])
    var = kappa - np.einsum("qi,iq->q", K, AK) + np.einsum("qa,qa->q", q, np.linalg.solve(HAH, q.T).T)
    return mean, var


def backproject(mean_pca, var_pca, evals, evecs, ybar):
    """multivar_support.c:126-151: mean_t = ybar_t + sum_j U_tj sqrt(lambda_j) m_j; var_t = sum_j U_tj^2 lambda_j v_j
    (mean_pca / var_pca: npoints x nr; evecs: nt x nr)"""
    return ybar[None, :] + (mean_pca * np.sqrt(evals)[None, :]) @ evecs.T, (var_pca * evals[None, :]) @ (evecs ** 2).T

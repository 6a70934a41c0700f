"""Seeded synthetic designs for tests and bench.py.

Self-contained splitmix64 generator (no libc rand, no numpy Generator state), so the
same seed gives the same bytes everywhere.
"""
import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(seed, n):
    """n uint64 values of the splitmix64 sequence started at `seed`."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = (np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        return z ^ (z >> np.uint64(31))


def uniform(seed, shape):
    """iid U[0,1) doubles: top 53 bits of splitmix64."""
    n = int(np.prod(shape))
    u = (splitmix64(seed, n) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def normal(seed, n):
    """iid N(0,1) by Box-Muller on two uniform streams."""
    u1 = uniform(seed, (n,))
    u2 = uniform(seed ^ 0x5DEECE66D, (n,))
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


def design(N, d, seed):
    """X: N x d iid U[0,1); y = sum_k sin(2 pi x_k (k+1)/d) + 0.01 N(0,1), standardised."""
    X = uniform(seed, (N, d))
    k = np.arange(1, d + 1, dtype=np.float64)
    y = np.sin(2.0 * np.pi * X * k / d).sum(axis=1) + 0.01 * normal(seed + 7919, N)
    y = (y - y.mean()) / y.std()
    return X, y


def multi_outputs(X, y, nt):
    """t outputs y^(j) = y cos(j) + sin(j pi x_{j mod d}) (config 4: PCA keeps nt-1 components)."""
    N, d = X.shape
    Y = np.empty((N, nt))
    for j in range(nt):
        Y[:, j] = y * np.cos(j) + np.sin(j * np.pi * X[:, j % d])
    return Y


def pca_zmatrix(Y, varfrac=1.0):
    """the PCA decomposition of a multi-output training matrix as the reference does it (multi_modelstruct.c:172-338,
    SURVEY App. A.6): column means, S = Yc^T Yc / N, eigenpairs sorted descending, nr = smallest i >= 1 whose leading
    eigenvalues reach varfrac -- the loop stops at nt - 1 --, Z = Yc U_r diag(lambda_r^-1/2).  Returns (Z, evals_r,
    evecs_r, ybar); eigenvector signs are LAPACK's (sign-ambiguous like GSL's)."""
    N, nt = Y.shape
    ybar = Y.mean(axis=0)
    Yc = Y - ybar
    lam, U = np.linalg.eigh(Yc.T @ Yc / N)
    lam, U = lam[::-1], U[:, ::-1]
    nr = 1
    if nt > 1:
        tot, frac = lam.sum(), 0.0
        nr = nt - 1
        for i in range(1, nt):
            frac = lam[:i].sum() / tot
            if frac >= varfrac:
                nr = i
                break
    Z = Yc @ U[:, :nr] / np.sqrt(lam[:nr])
    return Z, lam[:nr].copy(), U[:, :nr].copy(), ybar


def queries(M, d, seed):
    return uniform(seed, (M, d))


def default_thetas(kind, d):
    """supplied hyper-parameters of SURVEY 8(d): pow-exp [0,-4,log .6 x d]; Matern [1, .01, log .6]."""
    if kind == 1:
        return np.array([0.0, -4.0] + [np.log(0.6)] * d)
    return np.array([1.0, 0.01, np.log(0.6)])


def perturbed_thetas(kind, d, seed, i):
    """a fresh theta per evaluation: length scales += 0.01 U(-1,1), so nothing can be cached."""
    th = default_thetas(kind, d).copy()
    u = uniform(seed + 104729 * (i + 1), (th.size - 2,))
    th[2:] += 0.01 * (2.0 * u - 1.0)
    return th


def read_input_model_file(path):
    """INPUT_MODEL_FILE (interactive_emulator.c:222-238): nt d N, N*d design values, N*nt outputs."""
    vals = np.array(open(path).read().split(), dtype=np.float64)
    nt, d, N = int(vals[0]), int(vals[1]), int(vals[2])
    X = vals[3:3 + N * d].reshape(N, d)
    Y = vals[3 + N * d:3 + N * d + N * nt].reshape(N, nt)
    return X, Y


def snapshot_text(X, Y, evals, evecs, Z, cov, order, thetas_list, ranges_list=None, scales=None):
    """MODEL_SNAPSHOT_FILE text (multi_modelstruct.c:346-401, modelstruct.c:375-409) with the
    reference's printf formats ("%.17lf " reals, "%d\\n" ints) for a model whose thetas are SUPPLIED rather than
    trained: what bench.py's interactive_mode region and the tests feed `interactive_emulator interactive_mode`.
    evecs: nt x nr, Z: N x nr, thetas_list: nr vectors; grad_ranges / sample_scales are part of the file, not of the
    prediction (defaults: the reference's start ranges and unit scales)."""
    N, d = X.shape
    nt, nr = evecs.shape
    Y = np.asarray(Y, float).reshape(N, nt)
    row = lambda vals: "".join("%.17f " % v for v in vals) + "\n"
    xrows = "".join(row(X[i]) for i in range(N))
    parts = ["%d\n%d\n%d\n%d\n%d\n%d\n" % (nt, nr, d, N, cov, order), xrows, "".join(row(Y[i]) for i in range(N)),
             row(evals), "".join(row(evecs[t]) for t in range(nt)), "".join(row(Z[i]) for i in range(N))]
    nreg = 1 + order * d
    if scales is None:
        scales = np.ones(d)
    for c in range(nr):
        th = thetas_list[c]
        parts.append("%d\n%d\n%d\n%d\n%d\n%d\n%d\n%.17f\n%d\n%d\n" % (len(th), d, N, 0, order, nreg, 0, 0.0, cov, 1))
        rng_c = ranges_list[c] if ranges_list is not None else [(0.0001, 5.0), (-5.0, -2.0)] + [(-2.0, 1.2)] * (len(th) - 2)
        parts.append("".join("%.17f %.17f\n" % (lo, hi) for lo, hi in rng_c))
        parts += [xrows, row(Z[:, c]), row(th), row(scales)]
    return "".join(parts)


def single_output_snapshot(X, y, cov, order, thetas):
    """snapshot of a scalar-output model (nt = nr = 1) at supplied thetas: the PCA of one column is its centring and
    scaling (multi_modelstruct.c:172-338: eigenvalue = the column's variance, eigenvector 1)"""
    ybar = y.mean()
    lam = float(((y - ybar) ** 2).mean())
    Z = ((y - ybar) / np.sqrt(lam)).reshape(-1, 1)
    return snapshot_text(X, y.reshape(-1, 1), np.array([lam]), np.array([[1.0]]), Z, cov, order, [np.asarray(thetas, float)])

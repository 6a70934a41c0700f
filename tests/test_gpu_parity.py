"""Parity of the HIP path (through the C-ABI, include/gpemu.h) against the CPU oracle, the committed golden
vectors and size-independent properties.  Bar (BASELINE.json north_star): log-likelihood, sigma^2, beta and
posterior mean within 1e-8 relative, posterior variance within 1e-8*kappa absolute; covariance elements are
compared at 1e-13 relative (device exp/fma vs glibc differ in the last ulp)."""
import os
import numpy as np
import pytest
import scipy.linalg as sl

from madaiemulator_amd import abi, synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

RTOL = 1e-8          # the north_star parity bar
ELEM_RTOL = 1e-13    # covariance elements


def relerr(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def thetas_for(kind, d, golden=None):
    return synth.default_thetas(kind, d)


# ------------------------------------------------------------------ building blocks
# the last two shapes have >= 512 tiles: their tile order comes from the XCD-blocked table (kernels_linalg.hip, gemm_tile_table)
@pytest.mark.parametrize("m,n,k", [(16, 16, 16), (128, 128, 16), (200, 72, 64), (129, 257, 48), (384, 256, 256),
                                   (2048, 1600, 32), (2050, 1601, 16)])
def test_gemm_nt_asymmetric(gpu_ctx, m, n, k):
    rng = np.random.default_rng(m * 7 + n)
    A, B, C0 = rng.standard_normal((m, k)), rng.standard_normal((n, k)), rng.standard_normal((m, n))
    got = gpu_ctx.test_gemm_nt(A, B, C0, alpha=-1.0, beta=1)
    assert relerr(got, C0 - A @ B.T) < 1e-13
    got = gpu_ctx.test_gemm_nt(A, B, C0, alpha=1.0, beta=0)
    assert relerr(got, A @ B.T) < 1e-13


def _ctx_with_env(monkeypatch, env, device=0):
    """a context whose schedule switches come from `env`: they are copied into the context when it is created (gpemu::Sched)
    and the environment is restored at once -- contexts with different switches then live side by side in one process"""
    for k_, v in env.items():
        monkeypatch.setenv(k_, v)
    c = abi.Context(device)
    for k_ in env:
        monkeypatch.delenv(k_)
    return c


def test_gemm_tile_shapes_bit_identical(monkeypatch):
    """both tile shapes (128x128 with 8 waves, 64x64 with 4; operands by LDS-DMA into an XOR-swizzled image, two fragment
    sets, the barrier between the two MFMA blocks of a k-step) issue the same k-ordered MFMA chain per accumulator: same
    bits, ragged edges included.  Two contexts with different switches are alive at the same time and are called
    alternately: the switches are per context (nothing process-wide)."""
    shapes = [(128, 128, 16), (128, 128, 32), (200, 72, 64), (129, 257, 48), (384, 256, 256), (2050, 1601, 112)]
    big = _ctx_with_env(monkeypatch, {"GPEMU_GEMM_BIG_TILES": "1"})            # every launch on 128x128 tiles
    small = _ctx_with_env(monkeypatch, {"GPEMU_GEMM_BIG_TILES": "1000000"})     # every launch on 64x64 tiles
    for m, n, k in shapes:
        rng = np.random.default_rng(m * 11 + n + k)
        A, B, C0 = rng.standard_normal((m, k)), rng.standard_normal((n, k)), rng.standard_normal((m, n))
        outs = [(c.test_gemm_nt(A, B, C0, alpha=-1.0, beta=1), c.test_gemm_nt(A, B, C0, alpha=1.0, beta=0)) for c in (big, small, big)]
        assert relerr(outs[0][0], C0 - A @ B.T) < 1e-13 and relerr(outs[0][1], A @ B.T) < 1e-13, (m, n, k)
        for o in outs[1:]:
            assert np.array_equal(outs[0][0], o[0]) and np.array_equal(outs[0][1], o[1]), (m, n, k)
    big.close()
    small.close()


def test_gemm_identity_asymmetric_exact(gpu_ctx):
    # A = I against an asymmetric integer B: any row/col or k-slot mix-up of the MFMA maps shows up exactly
    n = 128
    A = np.eye(n)
    B = np.arange(n * n, dtype=np.float64).reshape(n, n)
    got = gpu_ctx.test_gemm_nt(A, B, np.zeros((n, n)))
    assert np.array_equal(got, B.T)


# n = 4500: ragged, outer panels, triangular trailing updates of > 512 tiles (table order)
@pytest.mark.parametrize("n", [1, 5, 64, 65, 100, 128, 200, 512, 1000, 4500])
def test_potrf_matches_lapack(gpu_ctx, n):
    rng = np.random.default_rng(n)
    M = rng.standard_normal((n, n))
    S = M @ M.T + n * np.eye(n)
    L, info = gpu_ctx.test_potrf(S)
    assert info == 0
    assert relerr(L, np.linalg.cholesky(S)) < 1e-13
    assert relerr(L @ L.T, S) < 1e-14


def test_potrf_reports_first_bad_pivot(gpu_ctx):
    S = np.eye(200)
    S[57, 57] = -1.0
    S[150, 150] = -2.0
    _, info = gpu_ctx.test_potrf(S)
    assert info == 58                       # 1-based index of the FIRST non-positive pivot


# ------------------------------------------------------------------ a1-a4, a16: covariance fill
@pytest.mark.parametrize("kind", [1, 2, 3])
@pytest.mark.parametrize("N,d", [(1, 1), (34, 1), (63, 2), (64, 3), (65, 8), (200, 16), (130, 20)])
def test_cov_matrix_vs_oracle(gpu_ctx, kind, N, d):
    X, y = synth.design(N, d, 100 + N + d) if N > 1 else (np.array([[0.25] * d]), np.array([1.0]))
    th = thetas_for(kind, d)
    gpu_ctx.set_model(kind, 0, X, y)
    got = gpu_ctx.cov_matrix(th)
    ref = O.cov_matrix(kind, X, th)
    assert got.shape == (N, N)
    assert relerr(got, ref) < ELEM_RTOL
    assert np.array_equal(got, got.T)


@pytest.mark.parametrize("kind", [1, 2, 3])
@pytest.mark.parametrize("loglen", [-6.0, -2.0, 3.0, 7.0])
def test_cov_matrix_extreme_hyperparameters(gpu_ctx, kind, loglen):
    """very short and very long length scales, large and tiny amplitudes / nuggets: the table-based exp and the Newton
    sqrt of the fill kernel against glibc through the oracle (entries that underflow must underflow alike)"""
    d, N = 5, 130
    X, y = synth.design(N, d, 77)
    X[7] = X[3]                                            # a duplicated design point: off-diagonal nugget
    gpu_ctx.set_model(kind, 0, X, y)
    if kind == 1:
        th = np.concatenate([[8.0, -18.0], np.full(d, loglen)])
    else:
        th = np.array([3.0e3, 1.0e-9, loglen])
    got = gpu_ctx.cov_matrix(th)
    ref = O.cov_matrix(kind, X, th)
    assert np.all(np.isfinite(got))
    big = np.abs(ref) > 1e-290
    # exp(-x) carries the rounding of its argument, x * 2^-53 relative: the tolerance grows with x = -log(c / amp)
    amp = np.exp(th[0]) if kind == 1 else th[0]
    tol = 1e-13 + 1e-15 * np.abs(np.log(np.abs(ref[big]) / amp))
    assert np.all(np.abs(got[big] - ref[big]) / np.abs(ref[big]) < tol)
    assert np.all(np.abs(got[~big]) < 1e-280)
    assert got[7, 3] == got[3, 3] and got[3, 7] == got[3, 3]     # duplicates carry the nugget off the diagonal too


def test_cov_golden_special_pairs(gpu_ctx, golden):
    # identical points, |delta| = 5e-11 / 2e-10 (pow-exp nugget threshold), 5e-17 (Matern), far points
    for x, y, kind, d, val, th in zip(golden["g1_x"], golden["g1_y"], golden["g1_kind"], golden["g1_d"],
                                      golden["g1_val"], golden["g1_th"]):
        kind, d = int(kind), int(d)
        X = np.vstack([x[:d], y[:d]])
        gpu_ctx.set_model(kind, 0, X, np.zeros(2))
        got = gpu_ctx.cov_matrix(th[:O.nthetas_for(kind, d)])
        assert got[0, 1] == pytest.approx(val, rel=ELEM_RTOL, abs=1e-300)
        assert got[1, 0] == got[0, 1]


def test_duplicate_design_points_get_offdiagonal_nugget(gpu_ctx):
    # the nugget is added wherever two rows coincide, not only on i == j
    X = np.array([[0.1, 0.2], [0.5, 0.5], [0.1, 0.2]])
    th = np.array([0.0, -1.0, 0.0, 0.0])
    gpu_ctx.set_model(1, 0, X, np.zeros(3))
    got = gpu_ctx.cov_matrix(th)
    assert relerr(got, O.cov_matrix(1, X, th)) < ELEM_RTOL
    assert got[0, 2] == pytest.approx(1.0 + np.exp(-1.0), rel=1e-14)


@pytest.mark.parametrize("kind", [1, 2, 3])
def test_kvectors_with_clamp(gpu_ctx, kind):
    X, y = synth.design(150, 3, 9)
    th = thetas_for(kind, 3)
    th = th.copy()
    th[2:] = np.log(0.05)                   # short length scale: many covariances fall below 1e-10
    Xq = np.vstack([synth.queries(70, 3, 5), X[:2]])
    gpu_ctx.set_model(kind, 0, X, y)
    got = gpu_ctx.kvectors(th, Xq)
    ref = np.vstack([O.kvector(kind, X, q, th) for q in Xq])
    assert (ref == 0.0).sum() > 100         # the clamp is exercised
    assert np.array_equal(got == 0.0, ref == 0.0)
    assert relerr(got, ref) < ELEM_RTOL


@pytest.mark.parametrize("kind", [1, 2, 3])
@pytest.mark.parametrize("N,d,M", [(300, 8, 200), (130, 3, 77), (64, 1, 64), (200, 16, 70)])
def test_kvectors_gram_form(gpu_ctx, kind, N, d, M):
    """a16 in the MFMA Gram form (cov_kvec_gram_kernel, the prediction sweep's k-vector fill since round 4) at length
    scales that admit it: queries inside the design's box, queries EQUAL to training points (the nugget rule of
    emulator.c:136-150 / :368-384 applies to k-vectors too), queries within 5e-11 of one (still "the same point" for
    pow-exp), and queries far outside the box -- a wave that holds such a row leaves the Gram form and takes exact
    differences -- against the oracle's makeKVector_fnptr (emulator.c:578-593, clamp included) at 1e-13, zero pattern
    equal; ragged M and N (padding rows and columns stay out of the result)."""
    X, y = synth.design(N, d, 31 + N)
    th = thetas_for(kind, d)
    Xq = synth.queries(M, d, 6)
    Xq[3] = X[5]
    Xq[4] = X[N - 1]
    Xq[5] = X[7] + 5e-11
    Xq[6] = X[9] + 3.0                       # outside: |x'|^2 > 16 for the default length scales
    Xq[7] = -20.0
    Xq[8] = 30.0                             # (round 5: at d = 16 this squared scaled distance, 2.3e4, came back as NaN)
    Xq[9] = 1.0e4
    Xq[M - 1] = X[0]
    Xq[40:44] = 1.0 + 0.5 * synth.queries(4, d, 8)     # a whole 16-row group of one wave mildly outside
    gpu_ctx.set_model(kind, 0, X, y)
    got = gpu_ctx.kvectors(th, Xq)
    ref = np.vstack([O.kvector(kind, X, q, th) for q in Xq])
    nug = np.exp(th[1]) if kind == 1 else th[1]
    amp = np.exp(th[0]) if kind == 1 else th[0]
    assert ref[3, 5] == pytest.approx(amp + nug, rel=1e-15) and ref[M - 1, 0] == pytest.approx(amp + nug, rel=1e-15)
    assert np.array_equal(got == 0.0, ref == 0.0)
    nz = ref != 0.0
    assert np.max(np.abs(got[nz] - ref[nz]) / ref[nz]) < ELEM_RTOL
    assert np.all(got[7] == 0.0) or kind != 1           # twenty length scales away: below the clamp (pow-exp)
    assert np.all(np.isfinite(got)) and np.all(got[8] == 0.0) and np.all(got[9] == 0.0)


# ------------------------------------------------------------------ a7-a11: likelihood
def check_loglik(gpu_ctx, kind, order, X, y, th_full):
    gpu_ctx.set_model(kind, order, X, y)
    got = gpu_ctx.loglik(th_full)
    if kind == 1 and th_full[0] == 0.0:
        ref = O.eval_fn_multi(kind, order, X, y, th_full[1:])
    else:
        # likelihood at a given FULL theta (Matern: amp is not exponentiated): same pieces, stored thetas
        e = O.Emulator(kind, order, X, y, th_full)
        H = e.H
        r = y - H @ e.beta
        quad = r @ e.cinverse @ r
        ref = dict(value=-(-0.5 * e.logdet - len(y) / 2.0 * 1.83788 - 0.5 * quad), sigma2=y @ e.cinverse @ r / len(y),
                   beta=e.beta, logdet=e.logdet, quad=quad, info=0)
    assert got["status"] == 0 and got["info"] == 0
    assert got["value"] == pytest.approx(ref["value"], rel=RTOL)
    assert got["sigma2"] == pytest.approx(ref["sigma2"], rel=RTOL)
    assert got["logdet"] == pytest.approx(ref["logdet"], rel=RTOL)
    assert got["quad"] == pytest.approx(ref["quad"], rel=RTOL)
    assert relerr(got["beta"], ref["beta"]) < RTOL
    return got, ref


@pytest.mark.parametrize("order", [0, 1, 2, 3])
def test_loglik_uni_simple(gpu_ctx, ref_inputs, golden, order):
    X, y = ref_inputs["uni"]
    th = np.concatenate([[0.0], golden["th_pe1"][1:]])
    got, _ = check_loglik(gpu_ctx, 1, order, X, y, th)
    row = golden["g3_uni"][order]
    assert got["value"] == pytest.approx(row[0], rel=RTOL)
    assert got["sigma2"] == pytest.approx(row[1], rel=RTOL)


@pytest.mark.parametrize("order", [0, 1])
def test_loglik_2d_param(gpu_ctx, ref_inputs, golden, order):
    X, y = ref_inputs["twod"]
    th = np.concatenate([[0.0], golden["th_pe2"][1:]])
    got, _ = check_loglik(gpu_ctx, 1, order, X, y, th)
    assert got["value"] == pytest.approx(golden["g3_2d"][order][0], rel=RTOL)


@pytest.mark.parametrize("kind,N,d,order", [(1, 512, 8, 0), (1, 512, 8, 1), (1, 1024, 8, 1), (3, 512, 8, 1),
                                            (2, 300, 4, 2), (1, 257, 16, 1)])
def test_loglik_seeded_designs(gpu_ctx, kind, N, d, order):
    X, y = synth.design(N, d, 20261003 + N)
    check_loglik(gpu_ctx, kind, order, X, y, thetas_for(kind, d))


def test_loglik_not_positive_definite_returns_nan(gpu_ctx, ref_inputs):
    # the Matern "training" failure mode (SURVEY C2 / golden G8): amp = 0 -> C = theta_1 * I, theta_1 < 0
    X, y = ref_inputs["uni"]
    gpu_ctx.set_model(2, 0, X, y)
    got = gpu_ctx.loglik(np.array([0.0, -3.0, 0.0]))
    assert got["status"] == abi.ERR_NOT_PD and got["info"] == 1 and np.isnan(got["value"])
    ref = O.eval_fn_multi(2, 0, X, y, np.array([-3.0, 0.0]))
    assert np.isnan(ref["value"]) and ref["info"] == 1


def test_loglik_is_deterministic_and_theta_sensitive(gpu_ctx):
    X, y = synth.design(700, 8, 4)
    gpu_ctx.set_model(1, 1, X, y)
    a = gpu_ctx.loglik(synth.perturbed_thetas(1, 8, 1, 0))
    b = gpu_ctx.loglik(synth.perturbed_thetas(1, 8, 1, 0))
    c = gpu_ctx.loglik(synth.perturbed_thetas(1, 8, 1, 1))
    assert a["value"] == b["value"] and np.array_equal(a["beta"], b["beta"])   # bit-identical re-run
    assert a["value"] != c["value"]


def test_loglik_enqueue_collect_pipeline(gpu_ctx):
    X, y = synth.design(400, 8, 5)
    gpu_ctx.set_model(1, 0, X, y)
    ths = [synth.perturbed_thetas(1, 8, 2, i) for i in range(4)]
    single = [gpu_ctx.loglik(t)["value"] for t in ths]
    for t in ths:
        gpu_ctx.loglik_enqueue(t)
    last = gpu_ctx.loglik_collect()
    assert last["value"] == single[-1]



# ------------------------------------------------------------------ a11 over a theta list (lock-step batch)
@pytest.mark.parametrize("kind,N,d,order,nb", [(1, 512, 8, 1, 5), (3, 1100, 8, 1, 3), (1, 2200, 4, 0, 4), (2, 300, 4, 2, 7)])
def test_loglik_batch_vs_oracle_and_single(gpu_ctx, kind, N, d, order, nb):
    """gpemu_loglik_batch: nb evaluations of one model factored in lock-step -- every element equals the oracle
    to 1e-8 and equals the single-evaluation entry bit for bit: the outer panel width of a batch differs above
    N = 512, but a trailing update continues the k-ordered accumulation from the stored C value (the accumulators
    start from the C tile), so the sums do not depend on where the panels are cut."""
    X, y = synth.design(N, d, 20261003 + N + nb)
    gpu_ctx.set_model(kind, order, X, y)
    ths = np.array([synth.perturbed_thetas(kind, d, 31, i) for i in range(nb)])
    got = gpu_ctx.loglik_batch(ths)
    assert np.all(got["status"] == 0) and np.all(got["info"] == 0)
    for b in range(nb):
        one = gpu_ctx.loglik(ths[b])
        for key in ("value", "sigma2", "logdet", "quad"):
            assert got[key][b] == one[key], (key, b)
        assert np.array_equal(got["beta"][b], one["beta"])
    for b in (0, nb - 1):
        e = O.Emulator(kind, order, X, y, ths[b])
        r = y - e.H @ e.beta
        quad = r @ e.cinverse @ r
        ref = -(-0.5 * e.logdet - N / 2.0 * 1.83788 - 0.5 * quad)
        assert got["value"][b] == pytest.approx(ref, rel=RTOL)
        assert got["sigma2"][b] == pytest.approx(y @ e.cinverse @ r / N, rel=RTOL)
        assert relerr(got["beta"][b], e.beta) < RTOL


def test_loglik_batch_reports_not_pd_per_element(gpu_ctx, ref_inputs):
    """one theta of the list gives a non-PD matrix (Matern, amp 0, nugget < 0): only that element fails"""
    X, y = ref_inputs["uni"]
    gpu_ctx.set_model(2, 0, X, y)
    good = np.array([1.0, 0.01, -0.5])
    bad = np.array([0.0, -3.0, 0.0])
    got = gpu_ctx.loglik_batch(np.array([good, bad, good]))
    assert list(got["status"]) == [0, abi.ERR_NOT_PD, 0] and list(got["info"]) == [0, 1, 0]
    assert np.isnan(got["value"][1]) and np.all(np.isnan(got["beta"][1]))
    one = gpu_ctx.loglik(good)
    assert got["value"][0] == one["value"] and got["value"][2] == one["value"]


def test_loglik_batch_pipeline_and_size_change(gpu_ctx):
    """batches of different sizes back to back on one context (workspace regrows, graphs are per size)"""
    X, y = synth.design(640, 8, 9)
    gpu_ctx.set_model(1, 1, X, y)
    ths = np.array([synth.perturbed_thetas(1, 8, 5, i) for i in range(6)])
    single = np.array([gpu_ctx.loglik(t)["value"] for t in ths])
    for nb in (2, 6, 3, 1):
        gpu_ctx.loglik_batch_enqueue(ths[:nb])
        gpu_ctx.loglik_batch_enqueue(ths[6 - nb:])
        got = gpu_ctx.loglik_batch_collect()
        assert relerr(got["value"], single[6 - nb:]) < 1e-12     # N > 512: the batch uses a wider outer panel
    # a prediction set-up after a batch uses matrix 0 of the workspace again
    beta, rc = gpu_ctx.predict_setup(ths[0])
    assert rc == 0 and relerr(beta, gpu_ctx.loglik(ths[0])["beta"]) < 1e-12


def test_result_ring_keeps_every_batch_of_a_pipeline(gpu_ctx):
    """gpemu_loglik_batch_collect_back: the results of the last RESULT_RING enqueued batches stay readable, so a caller
    that keeps the stream busy still collects EVERY batch (bench.py does; a restart pool consuming results would)"""
    X, y = synth.design(700, 8, 21)
    gpu_ctx.set_model(3, 1, X, y)
    batches = [np.array([synth.perturbed_thetas(3, 8, 40 + j, i) for i in range(3)]) for j in range(7)]
    ref = [gpu_ctx.loglik_batch(b) for b in batches]
    R = abi.RESULT_RING
    got = {}
    for j, b in enumerate(batches):
        if j >= R - 1:                                    # the ring is about to wrap: take the oldest batch still in it
            got[j - (R - 1)] = gpu_ctx.loglik_batch_collect_back(R - 2, 3)
        gpu_ctx.loglik_batch_enqueue(b)
    for back in range(R - 2, -1, -1):
        got[len(batches) - 1 - back] = gpu_ctx.loglik_batch_collect_back(back, 3)
    assert sorted(got) == list(range(len(batches)))
    for j in range(len(batches)):
        assert np.array_equal(got[j]["value"], ref[j]["value"]) and np.array_equal(got[j]["beta"], ref[j]["beta"])
    with pytest.raises(abi.GpemuError) as e:
        gpu_ctx.loglik_batch_collect_back(R, 3)
    assert e.value.code == abi.ERR_STATE


def test_set_training_swaps_outputs(gpu_ctx):
    # multi_modelstruct: one design, nr training vectors
    X, y = synth.design(300, 3, 6)
    Y = synth.multi_outputs(X, y, 3)
    th = thetas_for(1, 3)
    gpu_ctx.set_model(1, 1, X, Y[:, 0])
    for j in range(3):
        gpu_ctx.set_training(Y[:, j])
        got = gpu_ctx.loglik(th)
        ref = O.eval_fn_multi(1, 1, X, Y[:, j], th[1:])
        assert got["value"] == pytest.approx(ref["value"], rel=RTOL)


# ------------------------------------------------------------------ a5, a12: gradient
@pytest.mark.parametrize("N,d,order", [(34, 1, 1), (100, 3, 0), (300, 8, 1), (130, 2, 2)])
def test_grad_vs_oracle(gpu_ctx, N, d, order):
    X, y = synth.design(N, d, 77 + N)
    th = thetas_for(1, d)
    gpu_ctx.set_model(1, order, X, y)
    g, rc = gpu_ctx.grad(th)
    ref, st = O.grad_fn_multi(1, order, X, y, th[1:])
    assert rc == 0 and st == 0
    assert np.allclose(g, ref, rtol=1e-7, atol=1e-7 * np.abs(ref).max())


def test_grad_and_matern_prediction_golden_v2(gpu_ctx, ref_inputs, golden2):
    """round-2 fixtures whose independent re-derivation is mpmath at 50 digits (tests/golden/make_golden_v2.py):
    G4mp literal gradient d=1 and d=3, G5mp Matern 5/2 prediction"""
    X1, y1 = ref_inputs["uni"]
    gpu_ctx.set_model(1, 1, X1, y1)
    g, rc = gpu_ctx.grad(np.array([0.3, -3.0, -0.4]))
    ref = golden2["g4mp_uni"]
    assert rc == 0 and np.allclose(g, ref, rtol=1e-8, atol=1e-8 * np.abs(ref).max())
    X3, _ = ref_inputs["multi"]
    gpu_ctx.set_model(1, 0, X3[:34], golden2["g4mp_y34"])
    g, rc = gpu_ctx.grad(golden2["g4mp_th3"])
    ref = golden2["g4mp_multi34"]
    assert rc == 0 and np.allclose(g, ref, rtol=1e-8, atol=1e-8 * np.abs(ref).max())
    gpu_ctx.set_model(3, 1, X1, y1)
    th = golden2["g5mp_th"]
    _, rc = gpu_ctx.predict_setup(th)
    m, v = gpu_ctx.predict(golden2["g5mp_q"])
    assert rc == 0
    assert np.max(np.abs(m - golden2["g5mp_mean"])) < RTOL * max(1.0, np.abs(golden2["g5mp_mean"]).max())
    assert np.max(np.abs(v - golden2["g5mp_var"])) < RTOL * (th[0] + th[1])


@pytest.mark.parametrize("N,d,index,tl", [(130, 3, 2, -0.4), (257, 3, 4, -1.3), (64, 1, 2, 0.7)])
def test_derivative_l_gauss_materialised(gpu_ctx, N, d, index, tl):
    """a5 written out (emulator.c:173-209): the literal one-coordinate derivative matrix against the oracle, and
    getGradientCn's trace through gpemu_trace_product on it"""
    X, y = synth.design(N, d, 300 + N)
    got = gpu_ctx.derivative_gauss(X[:, index - 2], tl)
    ref = O.derivative_l(1, X, tl, index)
    assert relerr(got, ref) < 1e-13
    th = thetas_for(1, d)
    Cinv = np.linalg.inv(O.cov_matrix(1, X, th))
    assert gpu_ctx.trace_product(Cinv, ref) == pytest.approx(np.sum(Cinv * ref.T), rel=1e-11)


def test_grad_golden(gpu_ctx, ref_inputs, golden):
    X, y = ref_inputs["uni"]
    gpu_ctx.set_model(1, 1, X, y)
    g, _ = gpu_ctx.grad(np.concatenate([[0.0], golden["th_pe1"][1:]]))
    assert np.allclose(g, golden["g4_uni"], rtol=1e-7, atol=1e-7 * np.abs(golden["g4_uni"]).max())
    X3, _ = ref_inputs["multi"]
    gpu_ctx.set_model(1, 0, X3, golden["g4_y3"])
    g, _ = gpu_ctx.grad(np.concatenate([[0.0], golden["th_pe3"][1:]]))
    assert np.allclose(g, golden["g4_multi"], rtol=1e-7, atol=1e-7 * np.abs(golden["g4_multi"]).max())


@pytest.mark.parametrize("N,d,order,nb", [(300, 3, 1, 4), (700, 8, 0, 3)])
def test_loglik_grad_batch_vs_oracle_and_single(gpu_ctx, N, d, order, nb):
    """gpemu_loglik_grad_batch: value + gradient of nb independent thetas, factorisations in lock-step"""
    X, y = synth.design(N, d, 4242 + N)
    gpu_ctx.set_model(1, order, X, y)
    ths = np.array([synth.perturbed_thetas(1, d, 17, i) for i in range(nb)])
    ths[:, 0] = 0.0
    got = gpu_ctx.loglik_grad_batch(ths)
    assert np.all(got["status"] == 0) and np.all(got["info"] == 0)
    for b in range(nb):
        one = gpu_ctx.loglik_grad(ths[b])
        assert got["value"][b] == pytest.approx(one["value"], rel=1e-12)
        assert np.allclose(got["grad"][b], one["grad"], rtol=1e-9, atol=1e-9 * np.abs(one["grad"]).max())
        go, _ = O.grad_fn_multi(1, order, X, y, ths[b][1:])
        assert np.allclose(got["grad"][b], go, rtol=1e-7, atol=1e-7 * np.abs(go).max())
        ref = O.eval_fn_multi(1, order, X, y, ths[b][1:])
        assert got["value"][b] == pytest.approx(ref["value"], rel=RTOL)


@pytest.mark.parametrize("mode,kind", [(0, 1), (abi.MODE_EXACT_GRAD, 1), (abi.MODE_EXACT_GRAD | abi.MODE_MATERN_LOG, 3)])
def test_loglik_grad_batch_enqueue_collect_pipeline(mode, kind):
    """the asynchronous halves of gpemu_loglik_grad_batch: three value+gradient batches and a likelihood batch are enqueued
    back to back on one context (no host synchronisation in between) and collected afterwards from the pinned result ring;
    every batch equals, bit for bit, the same batch run through the blocking entry, the ring refuses a collect of the wrong
    kind, and an element equals the one-at-a-time call (gpemu_loglik_grad).  Literal and exact (device-side beta) modes."""
    N, d, order, nb = 600, 5, 1, 5
    X, y = synth.design(N, d, 77)
    c = abi.Context(0)
    c.set_mode(mode)
    c.set_model(kind, order, X, y)
    def th(i):
        t = synth.perturbed_thetas(kind, d, 23, i)
        if kind != 1:
            t[0], t[1] = 0.0, -3.0                      # log scale (MODE_MATERN_LOG)
        return t
    batches = [np.array([th(10 * j + i) for i in range(nb - (j == 1))]) for j in range(3)]      # sizes 5, 4, 5
    ref = [c.loglik_grad_batch(b) for b in batches]
    lik = c.loglik_batch(batches[0])
    for b in batches[:2]:
        c.loglik_grad_batch_enqueue(b)
    c.loglik_batch_enqueue(batches[0])
    c.loglik_grad_batch_enqueue(batches[2])
    got = [c.loglik_grad_batch_collect_back(3, len(batches[0])), c.loglik_grad_batch_collect_back(2, len(batches[1])),
           c.loglik_grad_batch_collect_back(0, len(batches[2]))]
    with pytest.raises(abi.GpemuError) as e:
        c.loglik_grad_batch_collect_back(1, len(batches[0]))          # that entry is the likelihood batch
    assert e.value.code == abi.ERR_STATE
    with pytest.raises(abi.GpemuError) as e:
        c.loglik_batch_collect_back(0, len(batches[2]))               # and the newest one is a gradient batch
    assert e.value.code == abi.ERR_STATE
    lik2 = c.loglik_batch_collect_back(1, len(batches[0]))
    assert np.array_equal(lik2["value"], lik["value"]) and np.array_equal(lik2["beta"], lik["beta"])
    for r, g in zip(ref, got):
        assert np.all(g["status"] == 0) and np.all(np.isfinite(g["grad"]))
        for k in ("value", "sigma2", "beta", "grad"):
            assert np.array_equal(r[k], g[k]), k
    assert np.array_equal(got[0]["value"], lik["value"])              # value of a gradient batch = likelihood batch, bit for bit
    one = c.loglik_grad(batches[2][3])
    assert one["value"] == got[2]["value"][3] and np.array_equal(one["grad"], got[2]["grad"][3])
    c.close()


def test_loglik_grad_batch_matern_is_refused(gpu_ctx, ref_inputs):
    X, y = ref_inputs["uni"]
    gpu_ctx.set_model(3, 0, X, y)
    with pytest.raises(abi.GpemuError) as e:
        gpu_ctx.loglik_grad_batch(np.array([[1.0, 0.01, -0.5], [1.0, 0.02, -0.4]]))
    assert e.value.code == abi.ERR_ARG


def test_grad_matern_is_refused(gpu_ctx, ref_inputs):
    """literal mode (the default): the reference's Matern training path exits (maxmultimin.c:495) -- no gradient"""
    X, y = ref_inputs["uni"]
    gpu_ctx.set_model(3, 0, X, y)
    for mode in (0, abi.MODE_EXACT_GRAD, abi.MODE_MATERN_LOG):         # both flags are needed
        gpu_ctx.set_mode(mode)
        with pytest.raises(abi.GpemuError) as e:
            gpu_ctx.grad(np.array([1.0, 0.01, 0.0]))
        assert e.value.code == abi.ERR_ARG
    gpu_ctx.set_mode(0)


# ------------------------------------------------------------------ corrected forms behind flags
def _fd_gradient(ctx, th, h=2e-3):
    """central differences of the value gpemu_loglik returns (theta[0] = 0), Richardson-extrapolated (h, 2h)"""
    th = np.array(th, float)
    th[0] = 0.0
    g = np.zeros(th.size - 1)
    for i in range(1, th.size):
        def f(t):
            x = th.copy()
            x[i] += t
            return ctx.loglik(x)["value"]
        d1 = (f(h) - f(-h)) / (2 * h)
        d2 = (f(2 * h) - f(-2 * h)) / (4 * h)
        g[i - 1] = (4.0 * d1 - d2) / 3.0
    return g


@pytest.mark.parametrize("kind,d,order", [(1, 1, 1), (1, 3, 0), (1, 8, 1), (2, 3, 1), (3, 1, 0), (3, 8, 1)])
def test_exact_gradient_matches_finite_differences(kind, d, order):
    """GPEMU_MODE_EXACT_GRAD: the gradient is the derivative of the value gpemu_loglik returns -- checked against
    central finite differences at 1e-6 of the largest component (pow-exp d = 1, 3, 8; both Matern kernels on the log
    scale).  The literal gradient (emulator.c:189,203; maxmultimin.c:514,532,594) is not, for d > 1."""
    N = 400
    X, y = synth.design(N, d, 77 + d)
    y = y + 0.1 * synth.normal(5, N)
    ctx = abi.Context(0)
    ctx.set_mode(abi.MODE_EXACT_GRAD | (abi.MODE_MATERN_LOG if kind != 1 else 0))
    ctx.set_model(kind, order, X, y)
    if kind == 1:
        th = np.concatenate([[0.0, -3.0], np.log(0.5) + 0.1 * np.arange(d)])
    else:
        th = np.array([0.0, -3.5, np.log(0.7)])
    r = ctx.loglik_grad(th)
    assert r["status"] == 0
    assert r["value"] == ctx.loglik(th)["value"]                        # same factorisation, same value
    fd = _fd_gradient(ctx, th)
    scale = np.max(np.abs(fd))
    assert np.max(np.abs(r["grad"] - fd)) <= 1e-6 * scale, (r["grad"], fd)
    g2, rc = ctx.grad(th)
    assert rc == 0 and np.array_equal(g2, r["grad"])
    # lock-step batch: element for element the single evaluation
    ths = np.array([th, th + 0.05, th - 0.03])
    ths[:, 0] = 0.0
    b = ctx.loglik_grad_batch(ths)
    assert np.all(b["status"] == 0)
    for i in range(3):
        one = ctx.loglik_grad(ths[i])
        assert np.array_equal(b["grad"][i], one["grad"]) and b["value"][i] == one["value"]
    if kind == 1 and d > 1:
        # and the literal formula really is something else
        ctx.set_mode(0)
        lit, _ = ctx.grad(th)
        assert np.max(np.abs(lit - fd)) > 1e-3 * scale
    ctx.close()


def test_matern_log_scale_mode_changes_the_kernel_consistently():
    """GPEMU_MODE_MATERN_LOG: amp = e^theta0, nug = e^theta1 in fill, likelihood and prediction alike; the literal
    kernel at (e^theta0, e^theta1, theta2) is the same matrix"""
    N, d = 300, 4
    X, y = synth.design(N, d, 3)
    th_log = np.array([0.3, -3.0, np.log(0.8)])
    th_raw = np.array([np.exp(0.3), np.exp(-3.0), np.log(0.8)])
    for kind in (2, 3):
        a, b = abi.Context(0), abi.Context(0)
        a.set_mode(abi.MODE_MATERN_LOG)
        a.set_model(kind, 1, X, y)
        b.set_model(kind, 1, X, y)
        assert np.array_equal(a.cov_matrix(th_log), b.cov_matrix(th_raw))
        la, lb = a.loglik(th_log), b.loglik(th_raw)
        assert la["value"] == lb["value"] and np.array_equal(la["beta"], lb["beta"])
        a.predict_setup(th_log)
        b.predict_setup(th_raw)
        Xq = synth.queries(50, d, 9)
        ma, va = a.predict(Xq)
        mb, vb = b.predict(Xq)
        assert np.array_equal(ma, mb) and np.array_equal(va, vb)
        # and against the oracle at the raw thetas
        e = O.Emulator(kind, 1, X, y, th_raw)
        mo, vo, _ = e.emulate(Xq)
        assert np.max(np.abs(ma - mo)) < RTOL * max(1.0, np.abs(mo).max())
        assert np.max(np.abs(va - vo)) < RTOL * (th_raw[0] + th_raw[1])
        a.close()
        b.close()


# ------------------------------------------------------------------ a14-a19: prediction
def check_predict(gpu_ctx, kind, order, X, y, th, Xq):
    gpu_ctx.set_model(kind, order, X, y)
    beta, rc = gpu_ctx.predict_setup(th)
    assert rc == 0
    e = O.Emulator(kind, order, X, y, th)
    m, v = gpu_ctx.predict(Xq)
    mo, vo, st = e.emulate(Xq)
    assert st == 0
    kappa = O.cov(kind, Xq[0], Xq[0], th)
    assert relerr(beta, e.beta) < RTOL
    assert np.max(np.abs(m - mo)) <= RTOL * max(1.0, np.max(np.abs(mo)))
    assert np.max(np.abs(v - vo)) <= RTOL * kappa
    return m, v, e


@pytest.mark.parametrize("kind", [1, 2, 3])
@pytest.mark.parametrize("order", [0, 1])
def test_predict_golden_inputs(gpu_ctx, ref_inputs, golden, kind, order):
    for tag, key, th in (("uni", "uni", golden["th_pe1"]), ("2d", "twod", golden["th_pe2"])):
        X, y = ref_inputs[key]
        thk = th if kind == 1 else golden["th_mat"]
        Q = golden["g5_q_" + tag]
        m, v, _ = check_predict(gpu_ctx, kind, order, X, y, thk, Q)
        ref = golden[f"g5_{tag}_k{kind}_o{order}"]
        kappa = O.cov(kind, Q[0], Q[0], thk)
        assert np.max(np.abs(m - ref[0])) <= RTOL * max(1.0, np.max(np.abs(ref[0])))
        assert np.max(np.abs(v - ref[1])) <= RTOL * kappa


@pytest.mark.parametrize("kind,N,d,order,M", [(1, 512, 8, 1, 300), (3, 512, 8, 1, 300), (2, 200, 5, 3, 65),
                                               (1, 100, 16, 0, 1), (3, 1000, 8, 1, 129)])
def test_predict_seeded(gpu_ctx, kind, N, d, order, M):
    X, y = synth.design(N, d, 31 + N)
    Xq = np.vstack([synth.queries(M, d, 17), X[:1]])
    check_predict(gpu_ctx, kind, order, X, y, thetas_for(kind, d), Xq)


def test_predict_at_training_points_interpolates(gpu_ctx):
    # k* at a training point carries the nugget -> C^-1 k* = e_i: mean = y_i, variance = 0
    X, y = synth.design(2048, 8, 8)
    th = thetas_for(3, 8)
    gpu_ctx.set_model(3, 1, X, y)
    gpu_ctx.predict_setup(th)
    m, v = gpu_ctx.predict(X[:500])
    assert np.max(np.abs(m - y[:500])) < 1e-8
    assert np.max(np.abs(v)) < 1e-8 * (th[0] + th[1])


def test_cinverse_matches_oracle_and_identity(gpu_ctx):
    X, y = synth.design(300, 4, 12)
    th = thetas_for(1, 4)
    gpu_ctx.set_model(1, 1, X, y)
    gpu_ctx.predict_setup(th)
    Ai = gpu_ctx.cinverse()
    e = O.Emulator(1, 1, X, y, th)
    assert relerr(Ai, e.cinverse) < RTOL
    assert np.array_equal(Ai, Ai.T)
    assert relerr(Ai @ O.cov_matrix(1, X, th), np.eye(300)) < 1e-9


def test_predict_batching_is_consistent(gpu_ctx):
    # results do not depend on how the queries are split into calls
    X, y = synth.design(640, 8, 13)
    gpu_ctx.set_model(1, 1, X, y)
    gpu_ctx.predict_setup(thetas_for(1, 8))
    Xq = synth.queries(1000, 8, 3)
    m_all, v_all = gpu_ctx.predict(Xq)
    m_a, v_a = gpu_ctx.predict(Xq[:333])
    m_b, v_b = gpu_ctx.predict(Xq[333:])
    assert np.array_equal(m_all, np.concatenate([m_a, m_b])) and np.array_equal(v_all, np.concatenate([v_a, v_b]))


# ------------------------------------------------------------------ BASELINE sizes: size-independent properties
@pytest.mark.parametrize("kind,N", [(1, 4096), (3, 8192)])
def test_full_size_properties(gpu_ctx, kind, N):
    d, order = 8, (0 if kind == 1 else 1)
    X, y = synth.design(N, d, 20261003 + kind)
    th = thetas_for(kind, d)
    gpu_ctx.set_model(kind, order, X, y)
    a = gpu_ctx.loglik(th)
    assert a["status"] == 0 and np.isfinite(a["value"])
    # (1) independent LAPACK evaluation of the same theta (not the oracle: scipy on the host cores)
    Cm = gpu_ctx.cov_matrix(th)
    cf = sl.cho_factor(Cm, lower=True, overwrite_a=True, check_finite=False)
    logdet = 2.0 * np.log(np.diag(cf[0])).sum()
    H = O.hmatrix(order, X)
    AyH = sl.cho_solve(cf, np.column_stack([y, H]), check_finite=False)
    beta = np.linalg.solve(H.T @ AyH[:, 1:], H.T @ AyH[:, 0])
    r = y - H @ beta
    quad = r @ sl.cho_solve(cf, r, check_finite=False)
    assert a["logdet"] == pytest.approx(logdet, rel=RTOL)
    assert relerr(a["beta"], beta) < RTOL
    assert a["quad"] == pytest.approx(quad, rel=RTOL)
    assert a["value"] == pytest.approx(-(-0.5 * logdet - N / 2.0 * 1.83788 - 0.5 * quad), rel=RTOL)
    del Cm, cf
    # (2) scaling y by 2 scales sigma^2 and the quadratic form by 4, beta by 2, leaves log det alone
    gpu_ctx.set_training(2.0 * y)
    b = gpu_ctx.loglik(th)
    assert b["sigma2"] == pytest.approx(4.0 * a["sigma2"], rel=1e-12)
    assert b["quad"] == pytest.approx(4.0 * a["quad"], rel=1e-12)
    assert relerr(b["beta"], 2.0 * a["beta"]) < 1e-10
    assert b["logdet"] == a["logdet"]
    gpu_ctx.set_training(y)
    # (3) fit -> predict at the training points returns the training values with zero variance
    gpu_ctx.predict_setup(th)
    idx = np.arange(0, N, 7)
    m, v = gpu_ctx.predict(X[idx])
    kappa = O.cov(kind, X[0], X[0], th)
    assert np.max(np.abs(m - y[idx])) < 1e-8 * max(1.0, np.abs(y).max())
    assert np.max(np.abs(v)) < 1e-8 * kappa
    # (4) away from the design the variance lies in (0, kappa + regression term] and the mean is finite
    m, v = gpu_ctx.predict(synth.queries(4096, d, 99))
    assert np.all(np.isfinite(m)) and np.all(v > 0) and np.all(v < 2.0 * kappa)


# ------------------------------------------------------------------ BASELINE sizes against the ORACLE itself
def _sample_pairs(N, rng, n_random=10000):
    """(i, j <= i) index pairs: random, diagonal, last row, last tile column, first column"""
    i = rng.integers(0, N, n_random)
    j = (rng.random(n_random) * (i + 1)).astype(np.int64)
    diag = rng.integers(0, N, 400)
    last_row_j = rng.integers(0, N, 300)
    ltc_j = rng.integers(N - 64, N, 300)                          # last tile column: j in the last 64, i >= j
    ltc_i = ltc_j + (rng.random(300) * (N - ltc_j)).astype(np.int64)
    fc_i = rng.integers(0, N, 200)
    I = np.concatenate([i, diag, np.full(300, N - 1), ltc_i, fc_i, [0, N - 1, N - 1, 63, 64, 64]])
    J = np.concatenate([j, diag, last_row_j, ltc_j, np.zeros(200, np.int64), [0, N - 1, 0, 63, 63, 64]])
    return I.astype(np.int64), J.astype(np.int64)


@pytest.mark.parametrize("kind,N,d", [(3, 8192, 8), (1, 16384, 8), (1, 4096, 16)])
def test_full_size_fill_elements_against_the_oracle(gpu_ctx, kind, N, d):
    """BASELINE.json configs[2], configs[4] and configs[3] (d = 16) sizes: > 10^4 sampled elements of the device fill -- random lower pairs,
    diagonal, last row, last tile column, duplicated design points (off-diagonal nugget, emulator.c:136-150) -- against
    the oracle's covariance function (emulator.c:101-152, 438-480 restated) at 1e-13, for BOTH fill paths: the 2-D
    cov_fill_kernel (gpemu_cov_matrix) and the one-launch lower-tile staging of a lock-step batch (what
    gpemu_loglik_batch factors), element 0 and the last element of a batch of 3."""
    X, y = synth.design(N, d, 20261003 + kind + d)
    X = X.copy()
    dups = [(N - 1, 5), (N // 2 + 1, N // 2), (N // 2 + 63, 64), (1000, 999)]      # row i repeats row j
    for i, j in dups:
        X[i] = X[j]
    rng = np.random.default_rng(kind * 1000 + N + d)
    I, J = _sample_pairs(N, rng)
    I = np.concatenate([I, [p[0] for p in dups]])
    J = np.concatenate([J, [p[1] for p in dups]])
    assert len(I) > 10000 and np.all(J <= I)
    ths = np.array([synth.perturbed_thetas(kind, d, 5, i) for i in range(3)])
    gpu_ctx.set_model(kind, 0, X, y)

    def ref_for(th):
        return np.array([O.cov(kind, X[a], X[b], th) for a, b in zip(I, J)])

    refs = [ref_for(th) for th in ths]
    # the duplicated pairs carry the nugget off the diagonal
    th0 = ths[0]
    nug = th0[1] if kind != 1 else np.exp(th0[1])
    amp = th0[0] if kind != 1 else np.exp(th0[0])
    assert np.allclose(refs[0][-len(dups):], amp + nug, rtol=1e-15)
    # (a) 2-D fill, full matrix
    Cm = gpu_ctx.cov_matrix(th0)
    got = Cm[I, J]
    assert np.max(np.abs(got - refs[0]) / np.abs(refs[0])) < ELEM_RTOL
    assert np.array_equal(Cm[J, I], got)                         # both triangles computed, symmetric
    del Cm
    # (b) the staging launch of a lock-step batch (lower tiles only), first and last matrix of the batch
    for b in (0, 2):
        Sm = gpu_ctx.staged_matrix(ths, b)
        got = Sm[I, J]
        assert np.max(np.abs(got - refs[b]) / np.abs(refs[b])) < ELEM_RTOL, b
        del Sm


def _oracle_eval_worker(args):
    kind, order, N, d, seed, th = args
    X, y = synth.design(N, d, seed)
    return O.eval_fn_multi(kind, order, X, y, np.asarray(th)[1:])


def _oracle_emulate_worker(args):
    kind, order, N, d, seed, th, qseed = args
    X, y = synth.design(N, d, seed)
    e = O.Emulator(kind, order, X, y, np.asarray(th))
    m, v, st = e.emulate(synth.queries(64, d, qseed))
    return m, v, st, e.beta, e.logdet


def _check_full_evaluation(gpu_ctx, kind, order, N, d, seed, th, qseed, o, mo, vo, obeta, ologdet):
    X, y = synth.design(N, d, seed)
    gpu_ctx.set_model(kind, order, X, y)
    got = gpu_ctx.loglik(th)
    gotb = gpu_ctx.loglik_batch(np.array([th, synth.perturbed_thetas(kind, d, 9, 1), th]))
    gpu_ctx.predict_setup(th)
    m, v = gpu_ctx.predict(synth.queries(64, d, qseed))
    assert got["status"] == 0 and o["info"] == 0
    assert got["value"] == pytest.approx(o["value"], rel=RTOL)
    assert got["sigma2"] == pytest.approx(o["sigma2"], rel=RTOL)
    assert got["logdet"] == pytest.approx(o["logdet"], rel=RTOL)
    assert got["quad"] == pytest.approx(o["quad"], rel=RTOL)
    assert relerr(got["beta"], o["beta"]) < RTOL
    assert gotb["value"][0] == got["value"] and gotb["value"][2] == got["value"]     # batch element = single evaluation
    assert relerr(got["beta"], obeta) < RTOL and got["logdet"] == pytest.approx(ologdet, rel=RTOL)
    kappa = np.exp(th[0]) + np.exp(th[1])
    assert np.max(np.abs(m - mo)) < RTOL * max(1.0, np.abs(mo).max())
    assert np.max(np.abs(v - vo)) < RTOL * kappa


def test_n3072_full_oracle_evaluation_and_predictions_live(gpu_ctx):
    """A full oracle evalFnMulti (maxmultimin.c:288-394 restated: fill, unblocked Cholesky, explicit inverse,
    estimateBeta twice) and 64 oracle emulate_point calls (emulator_struct.c:124-143, emulator.c:672-785) run LIVE next
    to the device at N = 3072, d = 8, pow-exp -- the largest size the oracle's naive N^3 loops finish in about a
    minute and a half (two passes side by side on two host cores); N = 4096 itself is the fixture test below."""
    import multiprocessing as mp
    kind, order, N, d, seed = 1, 0, 3072, 8, 20261003 + 1
    th = synth.default_thetas(kind, d).copy()
    th[0] = 0.0                                               # evalFnMulti's theta[0] (maxmultimin.c:311)
    with mp.get_context("spawn").Pool(2) as pool:
        ra = pool.map_async(_oracle_eval_worker, [(kind, order, N, d, seed, th.tolist())])
        rb = pool.map_async(_oracle_emulate_worker, [(kind, order, N, d, seed, th.tolist(), 321)])
        o = ra.get(timeout=600)[0]
        mo, vo, st, obeta, ologdet = rb.get(timeout=600)[0]
    _check_full_evaluation(gpu_ctx, kind, order, N, d, seed, th, 321, o, mo, vo, obeta, ologdet)


def test_n4096_oracle_fixture(gpu_ctx):
    """BASELINE.json configs[1] (N=4096, d=8, pow-exp): the device against ONE full oracle evaluation + 64 oracle
    predictions computed offline by tests/golden/make_golden_n4096.py (two passes of 4 minutes of one core each when
    alone on the machine, much longer under load) and committed as ~150 numbers; the design is regenerated here from
    the same seeds."""
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_n4096.npz"))
    kind, order, N, d, seed, qseed = (int(v) for v in f["meta"])
    o = dict(value=float(f["value"]), sigma2=float(f["sigma2"]), beta=f["beta"], logdet=float(f["logdet"]),
             quad=float(f["quad"]), info=int(f["info"]))
    _check_full_evaluation(gpu_ctx, kind, order, N, d, seed, f["thetas"], qseed, o, f["mean"], f["var"], f["emu_beta"],
                           float(f["emu_logdet"]))


def test_n8192_c3_oracle_fixture_and_a_50000_query_call(gpu_ctx):
    """BASELINE.json configs[2] EXACTLY -- N=8192, d=8, Matern 5/2, regression order 1, the bench's own design and
    hyper-parameters: the device against ONE full oracle evaluation (emulator_struct.c:13-37 + estimator-fns.c:38-103
    restated: fill, unblocked Cholesky, explicit inverse, estimateBeta) and 64 oracle emulate_point calls
    (emulator_struct.c:124-143), computed offline by tests/golden/make_golden_n8192_c3.py (most of an hour of one core,
    LAPACK cross-checked) and committed as ~150 numbers -- one oracle number at this size, no chain through the device's
    own fill.  Then the query-block size the bench pushes: ONE 50 000-query gpemu_predict_batch call whose first and
    last 64 queries are the fixture's; their results match the fixture at the parity bar, equal to rounding the 64-query
    call (which takes the split-K path: another summation order) and equal bit for bit the same queries inside a
    1024-query call (unsplit products: one k-ordered chain per element wherever the tile sits)."""
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_n8192_c3.npz"))
    kind, order, N, d, seed, qseed, nq = (int(v) for v in f["meta"])
    assert (kind, order, N, d, seed) == (3, 1, 8192, 8, 20261003 + 2)          # bench.py WORKLOADS["c3"] and its seed
    th = f["thetas"]
    assert np.array_equal(th, synth.default_thetas(kind, d))
    X, y = synth.design(N, d, seed)
    gpu_ctx.set_model(kind, order, X, y)
    got = gpu_ctx.loglik(th)
    gotb = gpu_ctx.loglik_batch(np.array([th, synth.perturbed_thetas(kind, d, 9, 1), th]))
    assert got["status"] == 0
    for k in ("value", "sigma2", "logdet", "quad"):
        assert got[k] == pytest.approx(float(f[k]), rel=RTOL), k
    assert relerr(got["beta"], f["beta"]) < RTOL
    assert gotb["value"][0] == got["value"] and gotb["value"][2] == got["value"]     # batch element = single evaluation
    assert np.array_equal(gotb["beta"][0], got["beta"])
    gpu_ctx.predict_setup(th)
    Xq = synth.queries(nq, d, qseed)
    m64, v64 = gpu_ctx.predict(Xq)
    kappa = th[0] + th[1]                                                        # raw Matern amplitude + nugget
    assert np.max(np.abs(m64 - f["mean"])) < RTOL * max(1.0, np.abs(f["mean"]).max())
    assert np.max(np.abs(v64 - f["var"])) < RTOL * kappa
    big = synth.queries(50000, d, qseed + 1)
    big[:nq] = Xq
    big[-nq:] = Xq
    mb, vb = gpu_ctx.predict(big)                                                # one call, 50 000 queries
    assert np.all(np.isfinite(mb)) and np.all(np.isfinite(vb)) and np.all(vb > -RTOL * kappa)
    for sl in (slice(0, nq), slice(50000 - nq, 50000)):
        assert np.max(np.abs(mb[sl] - f["mean"])) < RTOL * max(1.0, np.abs(f["mean"]).max())
        assert np.max(np.abs(vb[sl] - f["var"])) < RTOL * kappa
        assert np.max(np.abs(mb[sl] - m64)) < 1e-11 * max(1.0, np.abs(m64).max()) and np.max(np.abs(vb[sl] - v64)) < 1e-11 * kappa
    assert np.array_equal(mb[:nq], mb[-nq:]) and np.array_equal(vb[:nq], vb[-nq:])   # same query, different block: same bits
    mid = synth.queries(1024, d, qseed + 2)
    mid[:nq] = Xq
    mm, vm = gpu_ctx.predict(mid)
    assert np.array_equal(mm[:nq], mb[:nq]) and np.array_equal(vm[:nq], vb[:nq])
    # emulate_point's own path (round 4): ONE query per call (small k-vector kernel, matrix-vector stream over L^-1, epilogue
    # with the slice sums in front) and three per call (the skinny MFMA product) against the same oracle numbers
    for i in range(6):
        m1, v1 = gpu_ctx.predict(Xq[i:i + 1])
        assert abs(m1[0] - f["mean"][i]) < RTOL * max(1.0, np.abs(f["mean"]).max()) and abs(v1[0] - f["var"][i]) < RTOL * kappa
        assert abs(m1[0] - m64[i]) < 1e-11 * max(1.0, np.abs(m64).max()) and abs(v1[0] - v64[i]) < 1e-11 * kappa
    m3, v3 = gpu_ctx.predict(Xq[6:9])
    assert np.max(np.abs(m3 - f["mean"][6:9])) < RTOL * max(1.0, np.abs(f["mean"]).max()) and np.max(np.abs(v3 - f["var"][6:9])) < RTOL * kappa


@pytest.mark.parametrize("kind,d", [(1, 5), (1, 16), (3, 4), (2, 3)])
def test_exact_gradient_tile_distances_from_the_matrix_unit_or_from_differences(monkeypatch, kind, d):
    """grad_exact_gram_kernel (round 5: the tile's squared scaled distances from the fp64 MFMA on the centred design, at ANY
    length scale -- the weights enter plain sums) against grad_exact_kernel (coordinate differences, GPEMU_GRAD_GRAM=0): the
    same gradient to rounding, with length scales far below the Gram-form fill's bound (|x'|^2 of several hundred), a pair of
    design points 3e-9 apart (a candidate of the nugget rule that fails the exact "same point" test in both; pairs that pass
    it off the diagonal make the matrix singular by the reference's own rule) and a ragged N."""
    N = 1000
    X, y = synth.design(N, d, 404 + d)
    X[17] = X[3] + 3e-9                  # a near-coincident pair: a nugget-rule candidate that fails the exact test
    a, b = _ctx_with_env(monkeypatch, {"GPEMU_GRAD_GRAM": "1"}), _ctx_with_env(monkeypatch, {"GPEMU_GRAD_GRAM": "0"})
    modes = abi.MODE_EXACT_GRAD | (abi.MODE_MATERN_LOG if kind != 1 else 0)
    for c in (a, b):
        c.set_model(kind, 1, X, y)
        c.set_mode(modes)
    nth = abi.nthetas_for(kind, d)
    for scale in (0.6, 0.08, 0.02):
        th = np.zeros(nth)
        th[1] = -3.0
        th[2:] = np.log(scale) + 0.1 * np.arange(nth - 2)
        ths = np.array([th, th + 0.05])
        ga, gb = a.loglik_grad_batch(ths), b.loglik_grad_batch(ths)
        assert np.all(ga["status"] == 0) and np.array_equal(ga["value"], gb["value"])
        big = np.max(np.abs(gb["grad"]), axis=1, keepdims=True)
        assert np.all(np.isfinite(ga["grad"])) and np.max(np.abs(ga["grad"] - gb["grad"]) / big) < 1e-11, (scale, ga["grad"], gb["grad"])
    a.close(); b.close()


def test_exact_gradient_against_mpmath_golden_v3():
    """the corrected gradient forms (GPEMU_MODE_EXACT_GRAD, + GPEMU_MODE_MATERN_LOG for the Matern kernels) against an
    independent vector: the derivative of the 50-digit mpmath value, taken numerically there
    (tests/golden/make_golden_v3.py; N = 34, d = 3: Matern 5/2 orders 1 and 0, Matern 3/2, pow-exp).  Until round 3 the
    analytic Matern forms were only checked against finite differences of the device's own likelihood."""
    import os
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_v3.npz"))
    X, y = f["X"], f["y"]
    c = abi.Context(0)
    for i in range(int(f["ncases"])):
        kind, order, th = int(f[f"kind{i}"]), int(f[f"order{i}"]), f[f"th{i}"]
        c.set_mode(abi.MODE_EXACT_GRAD | (abi.MODE_MATERN_LOG if kind != 1 else 0))
        c.set_model(kind, order, X, y)
        got = c.loglik_grad(th)
        ref = f[f"grad{i}"]
        assert got["status"] == 0
        assert got["value"] == pytest.approx(float(f[f"value{i}"]), rel=RTOL)
        assert np.max(np.abs(got["grad"] - ref)) < RTOL * np.max(np.abs(ref)), (i, got["grad"], ref)
        gb = c.loglik_grad_batch(np.array([th, th]))
        assert np.array_equal(gb["grad"][0], got["grad"]) and np.array_equal(gb["grad"][1], got["grad"])
    c.close()


def _check_gradient_entries(c, ths, ref_grads, ref_values, tol=RTOL):
    """every way the library returns a value+gradient -- gpemu_loglik_grad, an element of a lock-step batch, the
    enqueue/collect halves -- against the reference vectors at `tol` of the largest component; the three entries agree
    bit for bit among themselves"""
    nb = len(ths)
    one = c.loglik_grad(ths[0])
    assert one["status"] == 0 and one["info"] == 0
    bat = c.loglik_grad_batch(ths)
    assert np.all(bat["status"] == 0)
    c.loglik_grad_batch_enqueue(ths[::-1].copy())
    c.loglik_grad_batch_enqueue(ths)
    late, early = c.loglik_grad_batch_collect_back(0, nb), c.loglik_grad_batch_collect_back(1, nb)
    assert np.array_equal(one["grad"], bat["grad"][0]) and one["value"] == bat["value"][0]
    assert np.array_equal(late["grad"], bat["grad"]) and np.array_equal(early["grad"], bat["grad"][::-1])
    assert np.array_equal(late["value"], bat["value"])
    errs = []
    for b in range(nb):
        scale = np.max(np.abs(ref_grads[b]))
        errs.append(float(np.max(np.abs(bat["grad"][b] - ref_grads[b])) / scale))
        assert errs[-1] < tol, (b, bat["grad"][b], ref_grads[b])
        assert bat["value"][b] == pytest.approx(ref_values[b], rel=RTOL)
    return errs


@pytest.mark.parametrize("N,d,order", [(4096, 8, 0), (4096, 16, 1), (8192, 8, 1)])
def test_gradient_at_baseline_sizes_against_numpy_lapack(N, d, order):
    """a12/a13 at BASELINE.json sizes (configs[1], configs[3]'s N and d, configs[2]'s N): gpemu_loglik_grad, a lock-step
    batch and the enqueue/collect entry against tests/gradref.py -- a numpy restatement of the pow-exp matrix
    (emulator.c:101-152, itself checked here on sampled elements against the oracle's covariance function), LAPACK's
    inverse, and the O(N^2 d) form of gradFnMulti / getGradientCn / derivative_l_gauss (maxmultimin.c:416-550, 571-608;
    emulator.c:173-209) -- nothing in that chain comes from the device.  Literal form (the default) and exact form, 1e-8
    of the largest component.  At N = 8192 grad_part_kernel runs 8 256 tiles and grad_reduce_kernel's stride loop makes
    33 passes; the live oracle comparisons stop at N = 900."""
    import gradref
    X, y = synth.design(N, d, 20261003 + 17 + d)
    ths = np.array([synth.perturbed_thetas(1, d, 29, i) for i in range(2)])
    ths[:, 0] = 0.0
    ths[1, 2:] += 0.05 * np.arange(d) - 0.2                              # distinct length scales in the second element
    rng = np.random.default_rng(N + d)
    I, J = _sample_pairs(N, rng, n_random=2000)
    Cn, _ = gradref.powexp_matrix(X, ths[0])
    refc = np.array([O.cov(1, X[a], X[b], ths[0]) for a, b in zip(I, J)])
    assert np.max(np.abs(Cn[I, J] - refc) / refc) < ELEM_RTOL             # the numpy matrix IS the oracle's matrix
    del Cn
    refs = [gradref.value_and_gradients(X, y, order, th) for th in ths]
    c = abi.Context(0)
    c.set_model(1, order, X, y)
    lit = _check_gradient_entries(c, ths, [r["literal"] for r in refs], [r["value"] for r in refs])
    one = c.loglik_grad(ths[1])
    assert one["sigma2"] == pytest.approx(refs[1]["sigma2"], rel=RTOL) and relerr(one["beta"], refs[1]["beta"]) < RTOL
    c.set_mode(abi.MODE_EXACT_GRAD)
    exa = _check_gradient_entries(c, ths, [r["exact"] for r in refs], [r["value"] for r in refs])
    print(f"gradient N={N} d={d}: literal {max(lit):.2e} exact {max(exa):.2e} of the largest component")
    c.close()


@pytest.mark.parametrize("size", [2048, 4096])
def test_oracle_gradient_fixtures(size):
    """ONE oracle gradFnMulti each (maxmultimin.c:416-550 restated: fill, unblocked Cholesky, explicit inverse, then nine
    naive N^3 products with the literal derivative matrices) at N = 2048 and at BASELINE.json configs[1]'s size N = 4096
    (d = 8, order 1), computed offline by tests/golden/make_golden_grad_n2048.py and cross-checked there against
    tests/gradref.py: 528 / 2 080 lower tiles, so the device's second-stage reduction (grad_reduce_kernel: thread j takes
    tiles j, j + 256, ...) makes three / nine passes against the oracle itself."""
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_grad_n%d.npz" % size))
    kind, order, N, d, seed = (int(v) for v in f["meta"])
    assert N == size
    X, y = synth.design(N, d, seed)
    th = f["thetas"]
    c = abi.Context(0)
    c.set_model(kind, order, X, y)
    got = c.loglik_grad(th)
    ref = f["grad"]
    assert got["status"] == 0
    assert np.max(np.abs(got["grad"] - ref)) < RTOL * np.max(np.abs(ref)), (got["grad"], ref)
    assert got["value"] == pytest.approx(float(f["value"]), rel=RTOL)
    assert got["sigma2"] == pytest.approx(float(f["sigma2"]), rel=RTOL) and relerr(got["beta"], f["beta"]) < RTOL
    g2, rc = c.grad(th)
    assert rc == 0 and np.array_equal(g2, got["grad"])
    bat = c.loglik_grad_batch(np.array([th, th + 0.01, th]))
    assert np.array_equal(bat["grad"][0], got["grad"]) and np.array_equal(bat["grad"][2], got["grad"])
    c.set_mode(abi.MODE_EXACT_GRAD)
    ex = c.loglik_grad(th)
    assert np.max(np.abs(ex["grad"] - f["exact_numpy"])) < RTOL * np.max(np.abs(f["exact_numpy"]))
    c.close()


def test_randomised_parity_sweep(gpu_ctx):
    """40 random configurations (covariance function, N from 2 to 900, d up to 16, regression order, batch size, thetas,
    query counts) through the whole path -- likelihood alone and in a lock-step batch, value+gradient batches through
    the asynchronous entry (every pow-exp case), predictions -- against the oracle at the parity bar; the bar is scaled
    only for cases whose conditioning the oracle itself cannot resolve (cond > 5e6: the oracle's own explicit inverse
    then carries more than 1e-9), and the test counts how many cases ran at the unscaled 1e-8 bar and how many were
    skipped as numerically singular (tests/tools/fuzz_parity.py is the long form: 700 cases in round 1, none above 2e-9)."""
    rng = np.random.default_rng(20261003)
    worst = 0.0
    n_unscaled = n_scaled = n_singular = n_grad = 0
    for it in range(40):
        kind = int(rng.integers(1, 4))
        N = int(rng.choice([rng.integers(2, 70), rng.integers(60, 140), rng.integers(120, 500), rng.integers(400, 900), rng.integers(64, 66)]))
        d = int(rng.choice([1, 2, 3, 5, 8, 13, 16]))
        order = int(rng.integers(0, 4))
        while 1 + order * d >= N:
            order -= 1
        nb = int(rng.choice([1, 2, 3, 5, 9, 16]))
        X, y = synth.design(N, d, int(rng.integers(1, 1 << 30)))
        nth = d + 2 if kind == 1 else 3

        def draw():
            th = synth.default_thetas(kind, d).copy()
            th[2:] += rng.uniform(-0.7, 0.7, size=nth - 2)
            if kind == 1:
                th[0], th[1] = rng.uniform(-1, 1), rng.uniform(-6, -2)
            else:
                th[0], th[1] = rng.uniform(0.3, 2.0), 10 ** rng.uniform(-4, -1)
            return th
        ths = np.array([draw() for _ in range(nb)])
        what = dict(it=it, kind=kind, N=N, d=d, order=order, nb=nb)
        gpu_ctx.set_model(kind, order, X, y)
        got = gpu_ctx.loglik_batch(ths)
        one = gpu_ctx.loglik(ths[0])
        e = O.Emulator(kind, order, X, y, ths[0])
        r = y - e.H @ e.beta
        ref = -(-0.5 * e.logdet - N / 2.0 * 1.83788 - 0.5 * (r @ e.cinverse @ r))
        cond = np.linalg.cond(O.cov_matrix(kind, X, ths[0]))
        tol = RTOL * max(1.0, cond * 2e-16 / 1e-9)
        if got["status"][0] != 0:
            assert e.status != 0 or cond > 1e12, what          # only a numerically singular matrix may fail
            n_singular += 1
            continue
        if tol > RTOL:
            n_scaled += 1
        else:
            n_unscaled += 1
        errs = [abs(got["value"][0] - ref) / abs(ref), abs(one["value"] - ref) / abs(ref),
                abs(one["sigma2"] - y @ e.cinverse @ r / N) / abs(one["sigma2"]),
                float(np.max(np.abs(one["beta"] - e.beta)) / max(np.max(np.abs(e.beta)), 1e-6 * np.max(np.abs(y))))]
        assert got["value"][0] == one["value"], what           # batch element = single evaluation, bit for bit
        if kind == 1:
            n_grad += 1
            thg = ths.copy()
            thg[:, 0] = 0.0
            gpu_ctx.loglik_grad_batch_enqueue(thg)
            gpu_ctx.loglik_grad_batch_enqueue(thg[::-1].copy())
            ga, gb = gpu_ctx.loglik_grad_batch_collect_back(1, nb), gpu_ctx.loglik_grad_batch_collect_back(0, nb)
            assert np.array_equal(ga["grad"], gb["grad"][::-1]) and np.array_equal(ga["value"], gb["value"][::-1]), what
            gref, st = O.grad_fn_multi(kind, order, X, y, thg[0][1:])
            if st == 0 and ga["status"][0] == 0:
                errs.append(relerr(ga["grad"][0], gref))
        M = int(rng.choice([1, 7, 16, 17, 100, 300]))
        Xq = synth.queries(M, d, int(rng.integers(1, 1 << 30)))
        if M > 3:
            Xq[0] = X[0]
        gpu_ctx.predict_setup(ths[0])
        m, v = gpu_ctx.predict(Xq)
        mo, vo, _ = e.emulate(Xq)
        kappa = abs(ths[0][0]) + 1 if kind != 1 else np.exp(ths[0][0]) + np.exp(ths[0][1])
        errs += [float(np.max(np.abs(m - mo)) / max(1.0, np.max(np.abs(mo)))), float(np.max(np.abs(v - vo)) / kappa)]
        assert np.all(np.isfinite(errs)) and max(errs) < tol, (what, errs, cond)
        worst = max(worst, max(errs) / max(1.0, cond * 2e-16 / 1e-9))
    assert worst < RTOL
    print(f"sweep: {n_unscaled} cases at the unscaled 1e-8 bar, {n_scaled} at a conditioning-scaled bar, {n_singular} singular, "
          f"{n_grad} with the gradient")
    # fixed seed: the counts are facts of this seed's 40 cases (cond(C) up to 4.4e5) -- all of them at the plain bar, none singular, 13 pow-exp ones with the gradient
    assert (n_unscaled, n_scaled, n_singular, n_grad) == (40, 0, 0, 13)


# ------------------------------------------------------------------ ragged and extreme shapes
@pytest.mark.parametrize("N", [2, 3, 63, 64, 65, 127, 129, 513])
@pytest.mark.parametrize("kind", [1, 3])
def test_ragged_sizes_full_path(gpu_ctx, kind, N):
    """N around the 64-column padding granule and the 512-column outer panel: likelihood, batch, gradient
    (pow-exp) and predictions against the oracle"""
    d, order = 3, (1 if N > 8 else 0)
    X, y = synth.design(N, d, 900 + N)
    y = y + 1.0                                  # the synthetic outputs are standardised: keep beta_0 away from 0
    th = thetas_for(kind, d)
    got, _ = check_loglik(gpu_ctx, kind, order, X, y, th)
    gb = gpu_ctx.loglik_batch(np.array([th, th]))
    assert gb["value"][0] == got["value"] and gb["value"][1] == got["value"]
    e = O.Emulator(kind, order, X, y, th)
    gpu_ctx.predict_setup(th)
    Xq = np.vstack([synth.queries(5, d, N), X[:1]])
    m, v = gpu_ctx.predict(Xq)
    mo, vo, _ = e.emulate(Xq)
    kappa = O.cov(kind, X[0], X[0], th)
    assert np.max(np.abs(m - mo)) < RTOL * max(1.0, np.abs(mo).max())
    assert np.max(np.abs(v - vo)) < RTOL * kappa
    if kind == 1 and N > 8:
        th0 = np.concatenate([[0.0], th[1:]])
        g, rc = gpu_ctx.grad(th0)
        go, _ = O.grad_fn_multi(kind, order, X, y, th0[1:])
        assert rc == 0 and np.allclose(g, go, rtol=1e-7, atol=1e-7 * np.abs(go).max())


def test_one_point_model(gpu_ctx):
    """N = 1: C is the scalar amp + nugget"""
    X, y = np.array([[0.25, 0.5]]), np.array([1.5])
    th = thetas_for(1, 2)
    gpu_ctx.set_model(1, 0, X, y)
    got = gpu_ctx.loglik(th)
    c = np.exp(th[0]) + np.exp(th[1])
    assert got["status"] == 0 and got["logdet"] == pytest.approx(np.log(c), rel=1e-14)
    assert got["beta"][0] == pytest.approx(1.5, rel=1e-14) and abs(got["quad"]) < 1e-20


def test_maximum_dimensions(gpu_ctx):
    """d = GPEMU_MAX_PARAMS = 64 coordinates, and the largest regression basis the augmented rows hold
    (1 + nreg = 64: d = 31, quadratic)"""
    X, y = synth.design(200, 64, 77)
    th = thetas_for(1, 64)
    th[2:] = np.log(3.0)                     # long length scales: 64-dimensional distances are large
    check_loglik(gpu_ctx, 1, 0, X, y, th)
    ref = O.cov_matrix(1, X, th)
    assert relerr(gpu_ctx.cov_matrix(th), ref) < ELEM_RTOL
    X, y = synth.design(400, 31, 78)
    th = thetas_for(1, 31)
    th[2:] = np.log(2.0)
    check_loglik(gpu_ctx, 1, 2, X, y, th)                 # nreg = 1 + 2*31 = 63: y and H fill all 64 augmented rows
    # the gradient kernels at d = 64: 130 sums per tile (nine block sums of 8 directions), 111 KB of dynamic LDS (beyond the
    # 64 KB a kernel gets without asking), literal and exact form against the oracle / tests/gradref.py; Gram-form k-vectors
    import gradref
    X, y = synth.design(200, 64, 77)
    th = thetas_for(1, 64)
    th[2:] = np.log(3.0) + 0.01 * np.arange(64)
    c = abi.Context(0)
    c.set_model(1, 0, X, y)
    g, rc = c.grad(th)
    ref, st = O.grad_fn_multi(1, 0, X, y, th[1:])
    assert rc == 0 and st == 0 and np.max(np.abs(g - ref)) < RTOL * np.max(np.abs(ref))
    c.set_mode(abi.MODE_EXACT_GRAD)
    ge, rc = c.grad(th)
    assert rc == 0 and np.max(np.abs(ge - gradref.value_and_gradients(X, y, 0, th)["exact"])) < RTOL * np.max(np.abs(ge))
    Xq = synth.queries(70, 64, 3)
    kv = c.kvectors(th, Xq)
    kref = np.vstack([O.kvector(1, X, q, th) for q in Xq])
    assert np.array_equal(kv == 0.0, kref == 0.0) and relerr(kv, kref) < ELEM_RTOL
    c.close()

def test_predict_enqueue_collect_on_several_contexts(gpu_ctx):
    """the asynchronous halves of gpemu_predict_batch: three contexts (PCA components of one design) are all started
    before the first is collected; results equal the blocking call; misuse is reported, not executed"""
    X, y = synth.design(500, 4, 21)
    Y = synth.multi_outputs(X, y, 3)
    th = thetas_for(1, 4)
    ctxs = []
    for c in range(3):
        k = abi.Context(0)
        k.set_model(1, 1, X, Y[:, c])
        k.predict_setup(th)
        ctxs.append(k)
    for M in (1, 9, 40, 300):
        Xq = synth.queries(M, 4, 5 + M)
        ref = [k.predict(Xq) for k in ctxs]
        for k in ctxs:
            k.predict_enqueue(Xq)
        got = [k.predict_collect() for k in ctxs]
        for (m0, v0), (m1, v1) in zip(ref, got):
            assert np.array_equal(m0, m1) and np.array_equal(v0, v1)
    ctxs[0].predict_enqueue(Xq)
    with pytest.raises(abi.GpemuError) as e:
        ctxs[0].predict_enqueue(Xq)                      # a batch is already pending on this context
    assert e.value.code == abi.ERR_STATE
    ctxs[0].predict_collect()
    with pytest.raises(abi.GpemuError) as e:
        ctxs[0].predict_collect()                        # nothing pending
    assert e.value.code == abi.ERR_STATE
    for k in ctxs:
        k.close()


@pytest.mark.parametrize("n", [5, 64, 300, 1100])
def test_chol_inverse_and_symm_apply_on_host_matrices(gpu_ctx, n):
    """the host-matrix entries behind chol_inverse_cov_matrix / estimateBeta / makeEmulatedMean: inverse and log det of
    a host matrix, and C^-1 applied to a few vectors, against LAPACK"""
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n))
    A = B @ B.T / n + np.eye(n) * 0.5
    inv, logdet, info, rc = gpu_ctx.chol_inverse(A)
    assert rc == 0 and info == 0
    ref = np.linalg.inv(A)
    assert relerr(inv, ref) < 1e-9 and np.array_equal(inv, inv.T)
    assert logdet == pytest.approx(np.linalg.slogdet(A)[1], rel=1e-11)
    V = rng.standard_normal((7, n))
    out = gpu_ctx.symm_apply(ref, V)
    assert relerr(out, V @ ref) < 1e-12
    out2 = gpu_ctx.symm_apply(ref, V[:2])                # cached matrix, fewer vectors
    assert np.array_equal(out2, out[:2])
    # callers reuse ONE cinverse buffer and rewrite it in place (regression.c:120-176 callers, rbind.c loops): a change
    # of a single interior element -- none of the corners or of the middle sample an abbreviated fingerprint would
    # look at -- must reach the device copy
    i, j = (n * 3) // 7, (n * 2) // 7
    ref[i, j] += 0.25
    out3 = gpu_ctx.symm_apply(ref, V)
    assert relerr(out3, V @ ref.T) < 1e-12 and not np.array_equal(out3, out)
    assert abs((out3 - out)[:, i] - 0.25 * V[:, j]).max() < 1e-12 * max(1.0, np.abs(V).max())
    ref[i, j] -= 0.25
    ref[n - 1, 0] = ref[0, n - 1] = ref[n - 1, 0] + 1.0   # and back, plus a corner
    assert relerr(gpu_ctx.symm_apply(ref, V), V @ ref) < 1e-12
    gpu_ctx.symm_invalidate()
    assert relerr(gpu_ctx.symm_apply(ref, V), V @ ref) < 1e-12
    # gpemu_symm_pin: the caller's promise that the buffer stays as it is -- the per-call checksum pass is skipped, so an
    # in-place change made under the promise is (by contract) NOT seen; unpinning restores the check, a new pointer or
    # gpemu_symm_invalidate always uploads
    gpu_ctx.symm_pin(True)
    before = gpu_ctx.symm_apply(ref, V)
    ref[i, j] += 0.5
    assert np.array_equal(gpu_ctx.symm_apply(ref, V), before)             # pinned: the device copy is trusted
    gpu_ctx.symm_pin(False)
    after = gpu_ctx.symm_apply(ref, V)                                     # checked again: the change is seen
    assert relerr(after, V @ ref.T) < 1e-12 and not np.array_equal(after, before)
    gpu_ctx.symm_pin(True)
    other = ref.copy()
    other[0, 0] += 1.0
    assert relerr(gpu_ctx.symm_apply(other, V), V @ other.T) < 1e-12      # another buffer: uploaded whatever the pin says
    gpu_ctx.symm_invalidate()                                              # (also drops the pin)
    ref[i, j] -= 0.5
    assert relerr(gpu_ctx.symm_apply(ref, V), V @ ref.T) < 1e-12
    bad = A.copy()
    bad[n // 2, n // 2] = -1.0
    _, _, info, rc = gpu_ctx.chol_inverse(bad)
    assert rc == abi.ERR_NOT_PD and info == n // 2 + 1


_schedule_cache = {}


def _schedule_run(monkeypatch, env):
    """N=1500 likelihood / batch / gradient / 700 predictions on a context of its own created under the given schedule
    switches.  The switches live in the context (gpemu::Sched, copied from the environment at creation), so every variant
    runs in this process next to the session context.

    Round 2 ran every variant in a child process because two GPU sessions of that round had gone silent and the cause was
    put down, without evidence, to the look-ahead context.  Read again in round 3 against the test file as it stood at
    those commits (DESIGN.md section 8 has the details): the first (gpurun_out/r02_t2.txt, four dots) was the then-live
    N=4096 CPU oracle pass -- two ~N^3 passes that print nothing for longer than the GPU box's seven-minute silence limit,
    moved into a fixture twenty minutes later; the second (r02_t8.txt) was the one session ever run with the
    inter-workgroup spin wait of solve-ahead AND the diag-first panels both on by default in the session context, and no
    look-ahead context precedes the test it stopped in.  Every construct involved has left the library: there is no second
    stream, no CU mask, no cross-stream graph capture and no wait between workgroups anywhere."""
    key = tuple(sorted(env.items()))
    if key not in _schedule_cache:
        N, d = 1500, 4
        X, y = synth.design(N, d, 8)
        th = synth.default_thetas(1, d)
        ths = np.array([synth.perturbed_thetas(1, d, 3, i) for i in range(3)])
        c = _ctx_with_env(monkeypatch, env)
        c.set_model(1, 1, X, y)
        out = [c.loglik(th), c.loglik(th), c.loglik_batch(ths), c.loglik_batch(ths), c.loglik_grad(th)]
        c.predict_setup(th)
        pm, pv = c.predict(synth.design(700, d, 99)[0])
        p1 = c.predict(synth.design(700, d, 99)[0][:1])              # one query: emulate_point's path
        c.close()
        _schedule_cache[key] = {"v0": out[0]["value"], "v1": out[1]["value"], "s2": out[0]["sigma2"], "b2": out[2]["value"],
                                "b3": out[3]["value"], "beta2": out[2]["beta"], "grad": out[4]["grad"], "pm": pm, "pv": pv,
                                "p1m": p1[0][0], "p1v": p1[1][0]}
    return _schedule_cache[key]


@pytest.mark.parametrize("env", [{"GPEMU_NO_GRAPH": "1"}, {"GPEMU_FACTOR_AHEAD": "0"}, {"GPEMU_FILL_GRAM": "0"},
                                 {"GPEMU_NB_TOP": "256"}, {"GPEMU_NB_TOP": "2048"}, {"GPEMU_GEMM_BIG_TILES": "1"},
                                 {"GPEMU_GEMM_BIG_TILES": "1000000"}, {"GPEMU_GEMM_TABLE": "0"}, {"GPEMU_GEMM_TABLE": "5"},
                                 {"GPEMU_FACTOR_AHEAD": "0", "GPEMU_NO_GRAPH": "1", "GPEMU_GEMM_BIG_TILES": "1"},
                                 {"GPEMU_SPLIT_RHS_ROWS": "0", "GPEMU_GEMM_BIG_TILES": "1"}, {"GPEMU_SPLIT_RHS_ROWS": "0"},
                                 {"GPEMU_KVEC_GRAM": "0"}, {"GPEMU_IDLE_WAVES": "0"}, {"GPEMU_IDLE_WAVES": "0", "GPEMU_GEMM_BIG_TILES": "1"},
                                 {"GPEMU_GEMV_POINT": "0"}, {"GPEMU_STAGGER_US": "0", "GPEMU_GEMM_BIG_TILES": "1"},
                                 {"GPEMU_STAGGER_US": "200", "GPEMU_GEMM_BIG_TILES": "1"}, {"GPEMU_NEG_MODIFIER": "0"},
                                 {"GPEMU_NEG_MODIFIER": "0", "GPEMU_GEMM_BIG_TILES": "1"},
                                 {"GPEMU_LEAF_STAGED": "1"}, {"GPEMU_LEAF_STAGED": "0"}, {"GPEMU_DIAG_INV_AHEAD": "0"},
                                 {"GPEMU_DIAG_INV_AHEAD": "0", "GPEMU_LEAF_STAGED": "1"},
                                 {"GPEMU_LEAF_STAGED": "1", "GPEMU_FACTOR_AHEAD": "0"}, {"GPEMU_LEAF_PAIR": "0"},
                                 {"GPEMU_LEAF_PAIR": "0", "GPEMU_LEAF_STAGED": "1"}, {"GPEMU_CORNER_ROW_TABLE": "0"}])
def test_schedule_switches_keep_parity(monkeypatch, env):
    """the measurement switches of INTEGRATION.md (factor-ahead, panel widths, tile shapes, tile order, no graph, the form
    of the prediction sweep's k-vector fill) change the
    schedule, not the result: every switch that only moves work between launches or workgroups leaves every bit alone
    (a trailing update continues the k-ordered chain of MFMA accumulations from the stored value, so a sum does not depend
    on where the panels are cut or which tile shape ran it); the difference form of the fill agrees to rounding"""
    N, d = 1500, 4
    X, y = synth.design(N, d, 8)
    th = synth.default_thetas(1, d)
    base = _schedule_run(monkeypatch, {})
    got = _schedule_run(monkeypatch, env)
    assert got["v0"] == got["v1"] and np.array_equal(got["b2"], got["b3"])
    assert got["v0"] == pytest.approx(base["v0"], rel=1e-11)
    assert got["s2"] == pytest.approx(base["s2"], rel=1e-10)
    assert np.allclose(got["b2"], base["b2"], rtol=1e-11, atol=0)
    assert np.allclose(got["grad"], base["grad"], rtol=1e-8, atol=1e-9 * np.max(np.abs(base["grad"])))
    # 700 predictions (triangular-operand products, k-ranges per tile) through the same switches
    assert np.max(np.abs(got["pm"] - base["pm"])) < 1e-9 * max(1.0, np.max(np.abs(base["pm"])))
    assert np.max(np.abs(got["pv"] - base["pv"])) < 1e-9 * max(1e-3, np.max(np.abs(base["pv"])))
    # one query per call: the same numbers as inside the 700-query call to rounding, whichever product kernel ran
    assert abs(got["p1m"] - base["pm"][0]) < 1e-9 * max(1.0, np.max(np.abs(base["pm"]))) and abs(got["p1v"] - base["pv"][0]) < 1e-9 * max(1e-3, np.max(np.abs(base["pv"])))
    if "GPEMU_FILL_GRAM" not in env and "GPEMU_GEMV_POINT" not in env:
        assert got["p1m"] == base["p1m"] and got["p1v"] == base["p1v"]
    if "GPEMU_FILL_GRAM" not in env:
        assert got["v0"] == base["v0"] and np.array_equal(got["b2"], base["b2"])
        assert np.array_equal(got["beta2"], base["beta2"]) and np.array_equal(got["grad"], base["grad"])
        # (the difference form of the k-vectors agrees with the Gram form to rounding; every other switch to the bit)
        if "GPEMU_KVEC_GRAM" not in env:
            assert np.array_equal(got["pm"], base["pm"]) and np.array_equal(got["pv"], base["pv"])
    e = O.Emulator(1, 1, X, y, th)
    r = y - e.H @ e.beta
    ref = -(-0.5 * e.logdet - N / 2.0 * 1.83788 - 0.5 * (r @ e.cinverse @ r))
    assert got["v0"] == pytest.approx(ref, rel=RTOL)


def test_tile_order_does_not_change_the_bits(monkeypatch):
    """GPEMU_GEMM_TABLE only permutes which workgroup computes which tile: same sums in the same order per tile"""
    import os
    N, d = 4096, 8
    X, y = synth.design(N, d, 3)
    th = synth.default_thetas(3, d)
    vals = []
    for table in ("0", "8", "5"):
        monkeypatch.setenv("GPEMU_GEMM_TABLE", table)      # copied into the context when it is created
        c = abi.Context(0)
        c.set_model(3, 1, X, y)
        r1 = c.loglik(th)                                   # plain launches
        r2 = c.loglik(th)                                   # recorded graph
        assert r1["value"] == r2["value"]
        rb = c.loglik_batch(np.array([th, th]))
        assert rb["value"][0] == rb["value"][1]
        vals.append((r1["value"], r1["sigma2"], rb["value"][0]))
        c.close()
    monkeypatch.delenv("GPEMU_GEMM_TABLE")
    assert vals[0] == vals[1] == vals[2]


def test_round5_chain_kernels_do_not_change_the_bits_at_a_baseline_size(monkeypatch):
    """the round-5 kernels of the panel chain at the size where they are chosen automatically (N = 4096, lock-step batches of
    32: the staged leaf solve, the diagonal inverses left by the factoring workgroup, leaf_pair_kernel, the row table of
    C^-1 = U U^T): every one switched off gives the same value, sigma^2, beta and gradient to the last bit, and a batch element
    equals the single evaluation"""
    N, d, B = 4096, 8, 32
    X, y = synth.design(N, d, 11)
    ths = np.array([synth.perturbed_thetas(1, d, 4, i) for i in range(B)])
    got = []
    for env in ({}, {"GPEMU_LEAF_STAGED": "0"}, {"GPEMU_DIAG_INV_AHEAD": "0"}, {"GPEMU_LEAF_PAIR": "0"},
                {"GPEMU_LEAF_STAGED": "0", "GPEMU_DIAG_INV_AHEAD": "0", "GPEMU_LEAF_PAIR": "0", "GPEMU_CORNER_ROW_TABLE": "0"}):
        c = _ctx_with_env(monkeypatch, env)
        c.set_model(1, 0, X, y)
        rb = c.loglik_batch(ths)
        r1 = c.loglik(ths[5])
        assert r1["value"] == rb["value"][5] and r1["sigma2"] == rb["sigma2"][5]
        rg = c.loglik_grad_batch(ths[:8])
        assert np.array_equal(rg["value"], rb["value"][:8])
        got.append((rb["value"].copy(), rb["sigma2"].copy(), rb["beta"].copy(), rg["grad"].copy()))
        c.close()
    for g in got[1:]:
        for a, b in zip(got[0], g):
            assert np.array_equal(a, b)
    assert np.all(np.isfinite(got[0][0])) and np.all(np.isfinite(got[0][3]))


def test_host_threads_with_their_own_contexts():
    """the boundary's threading contract (re-entrant across threads with distinct params): three host
    threads drive their own contexts at the same time (ctypes releases the GIL); every result equals, bit for
    bit, the one the same context produced alone.  scratch/threads_soak.py is the long form (90 s, 23 000 rounds)."""
    import threading
    import time
    work = []
    for kind, order, N, d, nb in [(1, 1, 1500, 4, 5), (3, 1, 2300, 8, 3), (2, 2, 777, 3, 7)]:
        X, y = synth.design(N, d, N)
        ths = np.array([synth.perturbed_thetas(kind, d, 5, i) for i in range(nb)])
        c = abi.Context(0)
        c.set_model(kind, order, X, y)
        ref_b = c.loglik_batch(ths)
        c.predict_setup(ths[0])
        Q = synth.queries(65, d, 3)
        work.append((c, ths, Q, ref_b, c.predict(Q)))
    errors, rounds = [], [0] * len(work)
    stop = time.time() + 3.0

    def run(i):
        c, ths, Q, ref_b, ref_p = work[i]
        while time.time() < stop and not errors:
            b = c.loglik_batch(ths)
            c.predict_setup(ths[0])
            m, v = c.predict(Q)
            if not (np.array_equal(b["value"], ref_b["value"]) and np.array_equal(m, ref_p[0]) and np.array_equal(v, ref_p[1])):
                errors.append(i)
            rounds[i] += 1
    ts = [threading.Thread(target=run, args=(i,)) for i in range(len(work))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for w in work:
        w[0].close()
    assert not errors and min(rounds) >= 2, (errors, rounds)


def test_model_switching_soak_is_deterministic():
    """two contexts, a random sequence of model changes (size, dimension, kernel, order) with evaluations, batches,
    gradients and predictions in between: workspaces and launch graphs are re-used / rebuilt correctly, and the
    same inputs always give bit-identical outputs"""
    rng = np.random.default_rng(1)
    ctxs = [abi.Context(0), abi.Context(0)]
    seen = {}
    for it in range(60):
        c = ctxs[it % 2]
        N = int(rng.choice([100, 257, 640, 1500])); d = int(rng.choice([1, 3, 8]))
        kind = int(rng.choice([1, 1, 3])); order = int(rng.choice([0, 1]))
        X, y = synth.design(N, d, N + d)
        c.set_model(kind, order, X, y)
        th = thetas_for(kind, d)
        v = c.loglik(th)["value"]
        ths = np.array([synth.perturbed_thetas(kind, d, 3, i) for i in range(int(rng.integers(1, 7)))])
        ths[0] = th
        vb = c.loglik_batch(ths)["value"]
        assert vb[0] == pytest.approx(v, rel=1e-11)
        g0 = None
        if kind == 1:
            th0 = np.concatenate([[0.0], th[1:]])
            g0 = c.loglik_grad(th0)["grad"]
            gb = c.loglik_grad_batch(np.array([th0, th0]))["grad"]
            assert np.allclose(gb[0], g0, rtol=1e-9, atol=1e-12) and np.array_equal(gb[0], gb[1])
        c.predict_setup(th)
        nq = int(rng.choice([1, 5, 64, 200]))
        m, var = c.predict(synth.queries(nq, d, 7))
        sig = (v, float(m[0]), float(var[0]), None if g0 is None else tuple(g0))
        key = (N, d, kind, order, nq)           # small query batches take the split-K route: same nq, same bits
        assert seen.setdefault(key, sig) == sig, key
    for c in ctxs:
        c.close()


# ------------------------------------------------------------------ error behaviour at the boundary
def test_error_codes(gpu_ctx):
    X, y = synth.design(50, 2, 1)
    ctx = abi.Context(0)
    with pytest.raises(abi.GpemuError) as e:
        ctx.loglik(np.zeros(4))
    assert e.value.code == abi.ERR_STATE                                   # model not set
    ctx.set_model(1, 0, X, y)
    with pytest.raises(abi.GpemuError) as e:
        ctx.predict(X[:2])
    assert e.value.code == abi.ERR_STATE                                   # predict before predict_setup
    with pytest.raises(abi.GpemuError) as e:
        ctx.loglik(np.zeros(2))
    assert e.value.code == abi.ERR_ARG                                     # nthetas too small for d+2
    with pytest.raises(abi.GpemuError) as e:
        ctx.set_model(1, 0, np.zeros((4, 65)), np.zeros(4))
    assert e.value.code == abi.ERR_ARG                                     # d > GPEMU_MAX_PARAMS
    with pytest.raises(abi.GpemuError) as e:
        ctx.set_model(1, 3, np.zeros((40, 30)), np.zeros(40))
    assert e.value.code == abi.ERR_ARG                                     # 1 + nregression_fns > 64
    ctx.close()


# ------------------------------------------------------------------ BASELINE.json configs[3] and configs[4]
def test_config4_eight_pca_components_n4096_d16(gpu_ctx):
    """N=4096, d=16, multi-output model whose PCA keeps 8 components: the 8 scalar GPs share the design (one
    upload) and differ in the training vector; each likelihood is checked against one LAPACK factorisation of a
    numpy-built matrix (tests/gradref.py; the device's own fill is not part of the reference chain)."""
    from madaiemulator_amd import shard
    N, d, nt = 4096, 16, 9
    X, y = synth.design(N, d, 20261003 + 3)
    Y = synth.multi_outputs(X, y, nt)
    Yc = Y - Y.mean(axis=0)
    w, U = np.linalg.eigh(Yc.T @ Yc / N)
    order = np.argsort(w)[::-1][:8]
    Z = Yc @ U[:, order] / np.sqrt(w[order])                 # multi_modelstruct.c:295-316
    th = synth.default_thetas(1, d)
    gpu_ctx.set_model(1, 1, X, Z[:, 0])
    import gradref
    Cm, _ = gradref.powexp_matrix(X, th)                     # numpy restatement of emulator.c:101-152: no device link in the chain
    cf = sl.cho_factor(Cm, lower=True, overwrite_a=True, check_finite=False)
    logdet = 2.0 * np.log(np.diag(cf[0])).sum()
    H = O.hmatrix(1, X)
    AH = sl.cho_solve(cf, H, check_finite=False)

    def component(c):
        gpu_ctx.set_training(Z[:, c])
        r = gpu_ctx.loglik(th)
        assert r["status"] == 0
        return [r["value"], r["sigma2"]]

    got = shard.farm_components(component, 8, 2, rank=0, world_size=1)
    for c in range(8):
        z = Z[:, c]
        Az = sl.cho_solve(cf, z, check_finite=False)
        beta = np.linalg.solve(H.T @ AH, H.T @ Az)
        r = z - H @ beta
        quad = r @ sl.cho_solve(cf, r, check_finite=False)
        ref = -(-0.5 * logdet - N / 2.0 * 1.83788 - 0.5 * quad)
        assert got[c, 0] == pytest.approx(ref, rel=RTOL)
        assert got[c, 1] == pytest.approx(z @ sl.cho_solve(cf, r, check_finite=False) / N, rel=RTOL)
    del Cm, cf


@pytest.mark.parametrize("kind,N,d,order,nr", [(1, 700, 3, 1, 5), (3, 1100, 2, 2, 3), (1, 4096, 16, 0, 8)])
def test_predict_setup_batch_is_the_per_component_call_bit_for_bit(kind, N, d, order, nr):
    """gpemu_predict_setup_batch (alloc_multi_emulator, multivar_support.c:30-52, as ONE lock-step factorisation with inverse
    rows: blockIdx.y = component, every component under its own right-hand-side rows): each context ends with the prediction
    state gpemu_predict_setup gives it alone -- beta and every predicted mean / variance bit for bit, the last case at
    BASELINE configs[3]'s size (8 components, N = 4096, d = 16).  A component whose matrix does not factor is reported in its
    status slot and leaves the others usable."""
    X, y = synth.design(N, d, 77 + N)
    Ys = [np.cos(0.7 * c) * y + np.sin(1.3 * (c + 1) * X[:, c % d]) for c in range(nr)]
    ths = np.array([synth.perturbed_thetas(kind, d, 31, c) for c in range(nr)])
    Xq = np.vstack([synth.queries(130, d, 5), X[:2]])
    alone = []
    c0 = abi.Context(0)
    for c in range(nr):
        c0.set_model(kind, order, X, Ys[c])
        beta, rc = c0.predict_setup(ths[c])
        assert rc == 0
        alone.append((beta.copy(),) + tuple(a.copy() for a in c0.predict(Xq)) + tuple(a.copy() for a in c0.predict(Xq[:1])))
    c0.close()
    ctxs = [abi.Context(0) for _ in range(nr)]
    for c in range(nr):
        ctxs[c].set_model(kind, order, X, Ys[c])
    import time
    t0 = time.perf_counter()
    beta, info, status, rc = abi.predict_setup_batch(ctxs, ths)
    t_first = time.perf_counter() - t0
    assert rc == 0 and np.all(status == 0) and np.all(info == 0)
    for c in range(nr):
        m, v = ctxs[c].predict(Xq)
        m1, v1 = ctxs[c].predict(Xq[:1])
        assert np.array_equal(beta[c], alone[c][0]) and np.array_equal(m, alone[c][1]) and np.array_equal(v, alone[c][2]), c
        assert m1[0] == alone[c][3][0] and v1[0] == alone[c][4][0]
    # again (replayed launch graph), with the components' thetas rotated: every context gets the other state
    rot = np.roll(ths, 1, axis=0)
    for rep in range(2):
        t0 = time.perf_counter()
        beta2, _, status2, rc2 = abi.predict_setup_batch(ctxs, rot)
        t_again = time.perf_counter() - t0
    assert rc2 == 0 and not np.array_equal(beta2, beta)
    ref = abi.Context(0)
    ref.set_model(kind, order, X, Ys[1])
    ref.predict_setup(rot[1])
    assert np.array_equal(ctxs[1].predict(Xq)[0], ref.predict(Xq)[0])
    if N <= 1100:
        # the explicit inverse (emulator_struct.cinverse) of a context that was set up as a LATER component of a batch: its
        # factorisation ran in the first context's workspace (round 5: reading its own, never allocated one was a GPU memory
        # fault) -- the entry factors the component alone and gives the single call's matrix bit for bit; the prediction
        # state is the same afterwards
        assert np.array_equal(ctxs[1].cinverse(), ref.cinverse()) and np.array_equal(ctxs[0].cinverse().shape, (N, N))
        assert np.array_equal(ctxs[1].predict(Xq)[0], ref.predict(Xq)[0])
    ref.close()
    flops = nr * 2.0 * (64 * ((N + 63) // 64)) ** 3 / 3.0
    print(f"predict_setup_batch kind={kind} N={N} d={d} nr={nr}: first {t_first * 1e3:.2f} ms, replay {t_again * 1e3:.2f} ms = "
          f"{t_again / nr * 1e3:.2f} ms per component = {flops / t_again / 78.6e12:.3f} of the 2N^3/3 roofline")
    if kind == 3:
        bad = ths.copy()
        bad[1, 0] = -1.0                                    # raw Matern amplitude < 0: not positive definite
        beta3, info3, status3, rc3 = abi.predict_setup_batch(ctxs, bad)
        assert rc3 == abi.ERR_NOT_PD and status3[1] == abi.ERR_NOT_PD and info3[1] > 0 and status3[0] == 0 and status3[2] == 0
        assert np.array_equal(ctxs[0].predict(Xq)[0], alone[0][1])
        with pytest.raises(abi.GpemuError):
            ctxs[1].predict(Xq)                             # left without a set-up
    for cx in ctxs:
        cx.close()


def test_config4_predictions_per_component_and_backprojection_d16():
    """a16 - a19 at configs[3]'s size and dimension (N=4096, d=16, 8 PCA components of 9 outputs): 64 predictions per component
    -- each component at thetas of its own -- against LAPACK solves on numpy-built matrices (tests/gradref.py predict:
    emulator.c:578-593, 672-785 restated), and the 9 back-projected outputs of emulate_point_multi against numpy
    (multivar_support.c:126-151).  Regression order 1 (17 basis functions at d = 16): the W^T rows of the sweep are in."""
    import gradref
    N, d, nt = 4096, 16, 9
    X, y = synth.design(N, d, 20261003 + 3)
    Y = synth.multi_outputs(X, y, nt)
    Z, evals, evecs, ybar = synth.pca_zmatrix(Y)
    nr = Z.shape[1]
    assert nr == 8
    Xq = np.vstack([synth.queries(62, d, 17), X[9:10], np.full((1, d), 30.0)])
    mp, vp = np.empty((64, nr)), np.empty((64, nr))
    mo, vo = np.empty((64, nr)), np.empty((64, nr))
    c = abi.Context(0)
    for j in range(nr):
        th = synth.perturbed_thetas(1, d, 77, j)
        c.set_model(1, 1, X, Z[:, j].copy())
        c.predict_setup(th)
        mp[:, j], vp[:, j] = c.predict(Xq)
        mo[:, j], vo[:, j] = gradref.predict(X, Z[:, j], 1, th, Xq)
        kappa = np.exp(th[0]) + np.exp(th[1])
        assert np.max(np.abs(mp[:, j] - mo[:, j])) <= RTOL * max(1.0, np.max(np.abs(mo[:, j]))), j
        assert np.max(np.abs(vp[:, j] - vo[:, j])) <= RTOL * kappa, j
    c.close()
    print("config4 predictions: mean", np.max(np.abs(mp - mo)), "variance", np.max(np.abs(vp - vo)))
    # observable space: the device's PCA-space numbers through numpy's back-projection against the all-numpy chain
    ym, yv = gradref.backproject(mp, vp, evals, evecs, ybar)
    ymo, yvo = gradref.backproject(mo, vo, evals, evecs, ybar)
    assert ym.shape == (64, nt) and np.max(np.abs(ym - ymo)) <= RTOL * np.max(np.abs(ymo)) and np.max(np.abs(yv - yvo)) <= RTOL * np.max(yvo)


def test_n32768_beyond_the_baseline_sizes(gpu_ctx):
    """N = 32768 (twice BASELINE's largest size; an 8.6 GB workspace per matrix, row offsets beyond 2^31 bytes): what the
    domain offers without a 50-second LAPACK pass in the suite -- a batch element equals the same evaluation alone bit for
    bit (two launch geometries over the same memory), scaling y by 3 scales the quadratic form by 9 and leaves log det
    alone, the trend coefficient follows y.  The LAPACK comparison itself (1e-15, also at N = 65536) is
    profiles/r04_large_n_32768_65536_against_lapack.txt (scratch/r04_n32768.py)."""
    N, d = 32768, 8
    X, y = synth.design(N, d, 20261003 + 9)
    gpu_ctx.set_model(1, 0, X, y)
    th = synth.default_thetas(1, d)
    a = gpu_ctx.loglik(th)
    assert a["status"] == 0 and a["info"] == 0 and np.isfinite(a["value"])
    # the committed LAPACK run used this seed and these thetas
    assert a["logdet"] == pytest.approx(-121957.945232, rel=1e-11) and a["quad"] == pytest.approx(2868.66861847, rel=1e-11)
    b = gpu_ctx.loglik_batch(np.array([synth.perturbed_thetas(1, d, 3, 0), th]))
    assert b["value"][1] == a["value"] and b["sigma2"][1] == a["sigma2"] and b["value"][0] != a["value"]
    gpu_ctx.set_training(3.0 * y)
    c = gpu_ctx.loglik(th)
    assert c["quad"] == pytest.approx(9.0 * a["quad"], rel=1e-12) and c["logdet"] == a["logdet"]
    assert c["beta"][0] == pytest.approx(3.0 * a["beta"][0], rel=1e-12)


def test_config5_n16384_powexp(gpu_ctx):
    """N=16384, d=8, pow-exp (one rank's share of the hyper-parameter search): an evaluation against LAPACK on a
    numpy-built matrix (tests/gradref.py), theta sensitivity, and the y-scaling property."""
    N, d = 16384, 8
    X, y = synth.design(N, d, 20261003 + 4)
    gpu_ctx.set_model(1, 0, X, y)
    th = synth.default_thetas(1, d)
    a = gpu_ctx.loglik(th)
    assert a["status"] == 0 and a["info"] == 0 and np.isfinite(a["value"])
    import gradref
    Cm, _ = gradref.powexp_matrix(X, th)                     # numpy restatement of emulator.c:101-152, not the device's fill
    cf = sl.cho_factor(Cm, lower=True, overwrite_a=True, check_finite=False)
    logdet = 2.0 * np.log(np.diag(cf[0])).sum()
    H = np.ones((N, 1))
    AyH = sl.cho_solve(cf, np.column_stack([y, H]), check_finite=False)
    beta = (H.T @ AyH[:, 0]) / (H.T @ AyH[:, 1])
    r = y - H[:, 0] * beta[0]
    quad = r @ sl.cho_solve(cf, r, check_finite=False)
    del Cm, cf
    assert a["logdet"] == pytest.approx(logdet, rel=RTOL)
    print("config5 beta", a["beta"][0], beta[0], abs(a["beta"][0] - beta[0]) / abs(beta[0]))
    assert a["beta"][0] == pytest.approx(beta[0], rel=RTOL)
    assert a["quad"] == pytest.approx(quad, rel=RTOL)
    assert a["value"] == pytest.approx(-(-0.5 * logdet - N / 2.0 * 1.83788 - 0.5 * quad), rel=RTOL)
    b = gpu_ctx.loglik(synth.perturbed_thetas(1, d, 3, 0))
    assert b["value"] != a["value"] and np.isfinite(b["value"])
    gpu_ctx.set_training(3.0 * y)
    c = gpu_ctx.loglik(th)
    assert c["quad"] == pytest.approx(9.0 * a["quad"], rel=1e-12) and c["logdet"] == a["logdet"]
    gpu_ctx.set_training(y)
    # a12 / a13 and a16 - a19 at this size (the config IS a hyper-parameter search, maxmultimin.c:416-550): value, literal and
    # exact gradient and 64 predictions against tests/golden/golden_n16384_c5.npz -- the numpy / LAPACK chain of
    # tests/gradref.py run offline (make_golden_n16384_c5.py: one explicit N = 16384 inverse, about 4 minutes of 8 host cores)
    f = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_n16384_c5.npz"))
    assert [int(v) for v in f["meta"]] == [1, 0, N, d, 20261003 + 4] and np.array_equal(f["thetas"], th)
    assert a["value"] == pytest.approx(float(f["value"]), rel=RTOL) and a["sigma2"] == pytest.approx(float(f["sigma2"]), rel=RTOL)
    assert logdet == pytest.approx(float(f["logdet"]), rel=1e-10) and quad == pytest.approx(float(f["quad"]), rel=1e-9)   # fixture == the live LAPACK pass above
    ths = np.array([th])
    ths[:, 0] = 0.0
    lit = _check_gradient_entries(gpu_ctx, ths, [f["literal"]], [float(f["value"])])
    gpu_ctx.set_mode(abi.MODE_EXACT_GRAD)
    exa = _check_gradient_entries(gpu_ctx, ths, [f["exact"]], [float(f["value"])])
    gpu_ctx.set_mode(0)
    print(f"config5 gradient N={N}: literal {max(lit):.2e} exact {max(exa):.2e} of the largest component")
    # 62 random queries, a training point, a far point whose k-vector is clamped to zero (its mean is the trend alone)
    Xq, mo, vo = f["Xq"], f["mean"], f["var"]
    gpu_ctx.predict_setup(th)
    m, v = gpu_ctx.predict(Xq)
    kappa = np.exp(th[0]) + np.exp(th[1])
    print("config5 predictions", np.max(np.abs(m - mo)), np.max(np.abs(v - vo)) / kappa)
    assert np.max(np.abs(m - mo)) <= RTOL * max(1.0, np.max(np.abs(mo))) and np.max(np.abs(v - vo)) <= RTOL * kappa
    assert mo[-1] == pytest.approx(float(f["beta"][0]), rel=1e-9) and m[-1] == pytest.approx(a["beta"][0], rel=1e-12)
    m1, v1 = gpu_ctx.predict(Xq[:1])                         # the one-query path (gemv_tri_kernel) at this size
    assert abs(m1[0] - mo[0]) <= RTOL * max(1.0, abs(mo[0])) and abs(v1[0] - vo[0]) <= RTOL * kappa

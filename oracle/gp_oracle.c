/*
 * gp_oracle.c -- CPU restatement of the MADAIEmulator GP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / the timed CPU baseline.
 *
 * PARITY UNPINNED: the reference ships no golden vectors, known-answer tests
 * or expected outputs for this path, and the
 * reference itself cannot be built here (it needs GSL, which is absent, and
 * writing stand-in GSL headers/libraries is not allowed).  This file therefore
 * restates the reference's formulas line by line (citations below, all
 * relative to /root/reference/src) and restates the *published* algorithms of
 * the GSL 1.x routines it calls (gsl_linalg_cholesky_decomp/_invert, reference
 * cblas dgemm/dgemv/ddot; GSL version is not pinned by the reference's
 * FindGSL.cmake).  It is cross-checked against independent scipy/LAPACK and
 * mpmath computations in tests/test_oracle.py.
 *
 * Layout: every matrix is row-major, element (i,j) at a[i*ld+j] (the GSL
 * layout, gsl_matrix.tda == ld).
 *
 * Kernel index (optstruct.h:12-14): 1 = power-exponential ("gaussian"),
 * 2 = Matern 3/2, 3 = Matern 5/2.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_POWEREXP 1
#define ORC_MATERN32 2
#define ORC_MATERN52 3

#define ORC_SUCCESS 0
#define ORC_EDOM 1

/* ------------------------------------------------------------------ */
/* covariance functions                                               */
/* ------------------------------------------------------------------ */

/* libEmu/emulator.c:101-152  covariance_fn_gaussian */
static double cov_gaussian(const double *xm, const double *xn, const double *thetas, int nparams)
{
	int i, truecount = 0;
	double covariance, exponent = 0.0, r_temp, dist_temp;
	double amp = exp(thetas[0]);
	double nug = exp(thetas[1]);
	for (i = 0; i < nparams; i++) {
		r_temp = exp(thetas[i + 2]);
		r_temp = r_temp * r_temp;
		dist_temp = fabs(xm[i] - xn[i]);
		exponent += (-1.0 / 2.0) * dist_temp * dist_temp / (r_temp);
		if (dist_temp < 0.0000000001)
			truecount++;
	}
	covariance = exp(exponent) * amp;
	if (truecount == nparams)
		covariance += nug;
	return covariance;
}

/* libEmu/emulator.c:344-386  covariance_fn_matern_three (amp, nugget NOT exponentiated) */
static double cov_matern_three(const double *xm, const double *xn, const double *thetas, int nparams)
{
	double covariance, distance = 0.0, temp_dist;
	int i, truecount = 0;
	double amp = thetas[0];
	double nugget = thetas[1];
	double rho = exp(thetas[2]);
	double root3 = 1.732050808;
	for (i = 0; i < nparams; i++) {
		temp_dist = fabs(xm[i] - xn[i]);
		distance += temp_dist * temp_dist;
		if (temp_dist < 0.0000000000000001)
			truecount++;
	}
	distance = sqrt(distance);
	if (distance > 0.0)
		covariance = amp * (1 + root3 * (distance / rho)) * exp(-root3 * (distance / rho));
	else
		covariance = amp;
	if (truecount == nparams)
		covariance += nugget;
	return covariance;
}

/* libEmu/emulator.c:438-480  covariance_fn_matern_five */
static double cov_matern_five(const double *xm, const double *xn, const double *thetas, int nparams)
{
	double covariance = 0.0, distance = 0.0, d_over_r;
	int i, truecount = 0;
	double amp = thetas[0];
	double nugget = thetas[1];
	double rho = exp(thetas[2]);
	double root5 = 2.236067978;
	for (i = 0; i < nparams; i++) {
		distance += pow(fabs(xm[i] - xn[i]), 2.0);
		if (fabs(xm[i] - xn[i]) < 0.0000000000000001)
			truecount++;
	}
	distance = sqrt(distance);
	d_over_r = distance / rho;
	if (distance > 0.0)
		covariance = amp * (1 + root5 * (d_over_r) + (5.0 / 3.0) * (d_over_r) * (d_over_r)) * exp(-root5 * (d_over_r));
	else if (distance == 0)
		covariance = amp;
	if (truecount == nparams)
		covariance += nugget;
	return covariance;
}

double orc_cov(int kind, const double *xm, const double *xn, const double *thetas, int nparams)
{
	switch (kind) {
	case ORC_MATERN32: return cov_matern_three(xm, xn, thetas, nparams);
	case ORC_MATERN52: return cov_matern_five(xm, xn, thetas, nparams);
	default:           return cov_gaussian(xm, xn, thetas, nparams);
	}
}

/* libEmu/emulator.c:636-653  makeCovMatrix_fnptr: full N x N, both triangles */
void orc_make_cov_matrix(int kind, double *C, const double *X, const double *thetas, int N, int nparams)
{
	int i, j;
	for (i = 0; i < N; i++)
		for (j = 0; j < N; j++)
			C[(size_t)i * N + j] = orc_cov(kind, X + (size_t)i * nparams, X + (size_t)j * nparams, thetas, nparams);
}

/* libEmu/emulator.c:578-593  makeKVector_fnptr, including the 1e-10 clamp */
void orc_make_kvector(int kind, double *kvec, const double *X, const double *xnew, const double *thetas, int N, int nparams)
{
	int i;
	double cov;
	for (i = 0; i < N; i++) {
		cov = orc_cov(kind, X + (size_t)i * nparams, xnew, thetas, nparams);
		if (cov < 1E-10)
			cov = 0.0;
		kvec[i] = cov;
	}
}

/* ------------------------------------------------------------------ */
/* derivative matrices  dC/dtheta_length                              */
/* ------------------------------------------------------------------ */

/* libEmu/emulator.c:173-209  derivative_l_gauss: uses ONE coordinate (index-2) only */
void orc_derivative_l_gauss(double *dC, const double *X, double thetaLength, int index, int N, int nparams)
{
	int i, j, k = index - 2;
	double rtemp, expTheta = exp(-2.0 * thetaLength);
	for (i = 0; i < N; i++)
		for (j = 0; j < N; j++) {
			rtemp = X[(size_t)i * nparams + k] - X[(size_t)j * nparams + k];
			dC[(size_t)i * N + j] = exp(-0.5 * expTheta * rtemp * rtemp - 2 * thetaLength) * rtemp * rtemp;
		}
}

/* libEmu/emulator.c:401-433  derivative_l_matern_three.  rtemp is NOT reset
 * between (i,j) iterations (line 410 vs 423-425) and thetaLength is the raw
 * (log-scale) value: reproduced literally, so the result depends on the
 * sequential loop order. */
void orc_derivative_l_matern_three(double *dC, const double *X, double thetaLength, int index, int N, int nparams)
{
	int i, j, p;
	double root3 = 1.732050808, rtemp = 0.0, x_temp, y_temp;
	double thetaLCubed = thetaLength * thetaLength * thetaLength;
	(void)index;
	for (i = 0; i < N; i++)
		for (j = 0; j < N; j++) {
			for (p = 0; p < nparams; p++) {
				x_temp = X[(size_t)i * nparams + p];
				y_temp = X[(size_t)j * nparams + p];
				rtemp += (x_temp - y_temp) * (x_temp - y_temp);
			}
			rtemp = sqrt(rtemp);
			dC[(size_t)i * N + j] = 3.0 * exp(-root3 * rtemp / thetaLength) * (rtemp * rtemp / thetaLCubed);
		}
}

/* libEmu/emulator.c:497-532  derivative_l_matern_five (same carried rtemp) */
void orc_derivative_l_matern_five(double *dC, const double *X, double thetaLength, int index, int N, int nparams)
{
	int i, j, p;
	double root5 = 2.2360680, rtemp = 0.0, rsq, x_temp, y_temp;
	double thetaLCubed = thetaLength * thetaLength * thetaLength;
	(void)index;
	for (i = 0; i < N; i++)
		for (j = 0; j < N; j++) {
			for (p = 0; p < nparams; p++) {
				x_temp = X[(size_t)i * nparams + p];
				y_temp = X[(size_t)j * nparams + p];
				rtemp += (x_temp - y_temp) * (x_temp - y_temp);
			}
			rsq = rtemp;
			rtemp = sqrt(rtemp);
			dC[(size_t)i * N + j] = (rsq / (thetaLCubed)) * exp(-root5 * rtemp / thetaLength) *
			                        (3.72768 * rtemp + 1.66667 * thetaLength);
		}
}

static void orc_derivative(int kind, double *dC, const double *X, double thetaLength, int index, int N, int nparams)
{
	switch (kind) {
	case ORC_MATERN32: orc_derivative_l_matern_three(dC, X, thetaLength, index, N, nparams); break;
	case ORC_MATERN52: orc_derivative_l_matern_five(dC, X, thetaLength, index, N, nparams); break;
	default:           orc_derivative_l_gauss(dC, X, thetaLength, index, N, nparams); break;
	}
}

/* ------------------------------------------------------------------ */
/* regression basis (libEmu/regression.c:9-67, 100-112)                */
/* ------------------------------------------------------------------ */

int orc_nregression_fns(int order, int nparams) { return 1 + order * nparams; }

void orc_make_hvector(int order, double *h, const double *x, int nparams)
{
	int i;
	h[0] = 1;
	if (order >= 1)
		for (i = 0; i < nparams; i++) h[i + 1] = x[i];
	if (order >= 2)
		for (i = 0; i < nparams; i++) h[nparams + i + 1] = x[i] * x[i];
	if (order >= 3)
		for (i = 0; i < nparams; i++) h[2 * nparams + i + 1] = x[i] * x[i] * x[i];
}

void orc_make_hmatrix(int order, double *H, const double *X, int N, int nparams)
{
	int i, nreg = orc_nregression_fns(order, nparams);
	for (i = 0; i < N; i++)
		orc_make_hvector(order, H + (size_t)i * nreg, X + (size_t)i * nparams, nparams);
}

/* ------------------------------------------------------------------ */
/* restated GSL 1.x numerics (published algorithms)     */
/* ------------------------------------------------------------------ */

/* reference-cblas ddot: plain left-to-right accumulation */
static double ddot_s(int n, const double *x, int incx, const double *y, int incy)
{
	double r = 0.0;
	int i;
	for (i = 0; i < n; i++)
		r += x[(size_t)i * incx] * y[(size_t)i * incy];
	return r;
}

/* y = alpha*op(A) x + beta*y, row-major A (m x n), reference-cblas loop order */
static void dgemv_s(int trans, int m, int n, double alpha, const double *A, int lda,
                    const double *x, double beta, double *y)
{
	int i, j;
	int leny = trans ? n : m;
	if (beta == 0.0) for (i = 0; i < leny; i++) y[i] = 0.0;
	else if (beta != 1.0) for (i = 0; i < leny; i++) y[i] *= beta;
	if (alpha == 0.0) return;
	if (!trans) {
		for (i = 0; i < m; i++) {
			double temp = 0.0;
			for (j = 0; j < n; j++) temp += x[j] * A[(size_t)lda * i + j];
			y[i] += alpha * temp;
		}
	} else {
		for (j = 0; j < m; j++) {
			const double temp = alpha * x[j];
			if (temp != 0.0)
				for (i = 0; i < n; i++) y[i] += temp * A[(size_t)lda * j + i];
		}
	}
}

/* C = alpha*op(A) op(B) + beta*C, row-major, reference-cblas loop order
 * (i,k,j with temp = alpha*A_ik for NN; dots for the transposed forms) */
static void dgemm_s(int transA, int transB, int M, int N, int K, double alpha,
                    const double *A, int lda, const double *B, int ldb, double beta, double *C, int ldc)
{
	int i, j, k;
	if (beta == 0.0) { for (i = 0; i < M; i++) for (j = 0; j < N; j++) C[(size_t)ldc * i + j] = 0.0; }
	else if (beta != 1.0) { for (i = 0; i < M; i++) for (j = 0; j < N; j++) C[(size_t)ldc * i + j] *= beta; }
	if (alpha == 0.0) return;
	if (!transA && !transB) {
		for (k = 0; k < K; k++)
			for (i = 0; i < M; i++) {
				const double temp = alpha * A[(size_t)lda * i + k];
				if (temp != 0.0)
					for (j = 0; j < N; j++) C[(size_t)ldc * i + j] += temp * B[(size_t)ldb * k + j];
			}
	} else if (!transA && transB) {
		for (i = 0; i < M; i++)
			for (j = 0; j < N; j++) {
				double temp = 0.0;
				for (k = 0; k < K; k++) temp += A[(size_t)lda * i + k] * B[(size_t)ldb * j + k];
				C[(size_t)ldc * i + j] += alpha * temp;
			}
	} else if (transA && !transB) {
		for (k = 0; k < K; k++)
			for (i = 0; i < M; i++) {
				const double temp = alpha * A[(size_t)lda * k + i];
				if (temp != 0.0)
					for (j = 0; j < N; j++) C[(size_t)ldc * i + j] += temp * B[(size_t)ldb * k + j];
			}
	} else {
		for (i = 0; i < M; i++)
			for (j = 0; j < N; j++) {
				double temp = 0.0;
				for (k = 0; k < K; k++) temp += A[(size_t)lda * k + i] * B[(size_t)ldb * j + k];
				C[(size_t)ldc * i + j] += alpha * temp;
			}
	}
}

/* gsl_linalg_cholesky_decomp (GSL 1.x): row-oriented Cholesky-Banachiewicz,
 * L in the lower triangle, L^T mirrored into the upper triangle, EDOM if any
 * pivot <= 0 (the loop keeps going, sqrt of a negative gives NaN). */
int orc_cholesky_decomp(double *A, int n)
{
	int i, j, k, status = ORC_SUCCESS;
	for (k = 0; k < n; k++) {
		double *rk = A + (size_t)k * n;
		double diag;
		for (i = 0; i < k; i++) {
			const double *ri = A + (size_t)i * n;
			double sum = ddot_s(i, ri, 1, rk, 1);
			rk[i] = (rk[i] - sum) / ri[i];
		}
		diag = rk[k] - ddot_s(k, rk, 1, rk, 1);
		if (diag <= 0) status = ORC_EDOM;
		rk[k] = sqrt(diag);
	}
	for (i = 1; i < n; i++)
		for (j = 0; j < i; j++)
			A[(size_t)j * n + i] = A[(size_t)i * n + j];
	return status;
}

/* gsl_linalg_cholesky_invert (GSL >= 1.14): invert L in place, form
 * A^-1 = L^-T L^-1 from column dots of L^-1, fill both triangles. */
void orc_cholesky_invert(double *A, int n)
{
	int i, j, k;
	/* lower triangle <- L^-1, column by column from the last */
	for (j = n - 1; j >= 0; j--) {
		double ajj = 1.0 / A[(size_t)j * n + j];
		A[(size_t)j * n + j] = ajj;
		if (j < n - 1) {
			/* v = T[j+1:, j+1:] * L[j+1:, j]  (T already inverted, lower, non-unit),
			 * dtrmv lower/notrans walks rows bottom-up so it can work in place */
			for (i = n - 1; i > j; i--) {
				double temp = 0.0;
				for (k = j + 1; k <= i; k++)
					temp += A[(size_t)i * n + k] * A[(size_t)k * n + j];
				A[(size_t)i * n + j] = temp;
			}
			for (i = j + 1; i < n; i++)
				A[(size_t)i * n + j] *= -ajj;
		}
	}
	/* upper triangle (incl. diagonal) <- columns of L^-1 dotted together */
	for (i = 0; i < n; i++)
		for (j = i; j < n; j++) {
			double sum = 0.0;
			for (k = j; k < n; k++)
				sum += A[(size_t)k * n + i] * A[(size_t)k * n + j];
			A[(size_t)i * n + j] = sum;
		}
	for (i = 1; i < n; i++)
		for (j = 0; j < i; j++)
			A[(size_t)i * n + j] = A[(size_t)j * n + i];
}

/* ------------------------------------------------------------------ */
/* regression / likelihood pieces                                      */
/* ------------------------------------------------------------------ */

/* libEmu/regression.c:120-176  estimateBeta.  Returns ORC_EDOM where the
 * reference would print and exit(1). */
int orc_estimate_beta(double *beta, const double *H, const double *cinverse, const double *y, int N, int nreg)
{
	double *htc = malloc(sizeof(double) * (size_t)nreg * N);
	double *den = malloc(sizeof(double) * (size_t)nreg * nreg);
	double *num = malloc(sizeof(double) * (size_t)nreg);
	int st;
	dgemm_s(1, 0, nreg, N, N, 1.0, H, nreg, cinverse, N, 0.0, htc, N);
	dgemm_s(0, 0, nreg, nreg, N, 1.0, htc, N, H, nreg, 0.0, den, nreg);
	st = orc_cholesky_decomp(den, nreg);
	if (st == ORC_SUCCESS) {
		orc_cholesky_invert(den, nreg);
		dgemv_s(0, nreg, N, 1.0, htc, N, y, 0.0, num);
		dgemv_s(0, nreg, nreg, 1.0, den, nreg, num, 0.0, beta);
	}
	free(htc); free(den); free(num);
	return st;
}

/* mean_i = h(x_i) . beta   (maxmultimin.c:246-252, estimator-fns.c:64-71) */
static void estimated_mean(double *mean, const double *beta, const double *X, int order, int N, int nparams, int nreg)
{
	double *h = malloc(sizeof(double) * (size_t)nreg);
	int i;
	for (i = 0; i < N; i++) {
		orc_make_hvector(order, h, X + (size_t)i * nparams, nparams);
		mean[i] = ddot_s(nreg, beta, 1, h, 1);
	}
	free(h);
}

/* libEmu/maxmultimin.c:215-273  estimateSigma: sigma^2 = y . Cinv (y - H beta) / N */
double orc_estimate_sigma(const double *cinverse, const double *X, const double *y, const double *H,
                          int order, int N, int nparams, int *status)
{
	int nreg = orc_nregression_fns(order, nparams), i;
	double *beta = malloc(sizeof(double) * (size_t)nreg);
	double *mean = malloc(sizeof(double) * (size_t)N);
	double *tsm = malloc(sizeof(double) * (size_t)N);
	double *tmp = malloc(sizeof(double) * (size_t)N);
	double sigma = NAN;
	int st = orc_estimate_beta(beta, H, cinverse, y, N, nreg);
	if (status) *status = st;
	if (st == ORC_SUCCESS) {
		estimated_mean(mean, beta, X, order, N, nparams, nreg);
		for (i = 0; i < N; i++) tsm[i] = y[i] - mean[i];
		dgemv_s(0, N, N, 1.0, cinverse, N, tsm, 0.0, tmp);
		sigma = ddot_s(N, y, 1, tmp, 1) / (double)N;
	}
	free(beta); free(mean); free(tsm); free(tmp);
	return sigma;
}

/* libEmu/estimator-fns.c:38-103  getLogLikelyhood with log_det_c supplied
 * (the caller decides between log(prod^2) and 2*sum(log)) */
double orc_get_loglikelyhood(const double *cinverse, double log_det_c, const double *X, const double *y,
                             const double *H, int order, int N, int nparams, double *beta_out,
                             double *quad_out, int *status)
{
	int nreg = orc_nregression_fns(order, nparams), i;
	double log_2_pi = 1.83788;
	double *beta = malloc(sizeof(double) * (size_t)nreg);
	double *mean = malloc(sizeof(double) * (size_t)N);
	double *tsm = malloc(sizeof(double) * (size_t)N);
	double *tmp = malloc(sizeof(double) * (size_t)N);
	double the_likelyhood = NAN, vmv;
	int st = orc_estimate_beta(beta, H, cinverse, y, N, nreg);
	if (status) *status = st;
	if (st == ORC_SUCCESS) {
		estimated_mean(mean, beta, X, order, N, nparams, nreg);
		for (i = 0; i < N; i++) tsm[i] = y[i] - mean[i];
		the_likelyhood = -(1.0 / 2.0) * log_det_c - (N / 2.0) * log_2_pi;
		dgemv_s(0, N, N, 1.0, cinverse, N, tsm, 0.0, tmp);
		vmv = ddot_s(N, tsm, 1, tmp, 1);
		the_likelyhood += vmv * (-1.0 / 2.0);
		if (quad_out) *quad_out = vmv;
		if (beta_out) memcpy(beta_out, beta, sizeof(double) * (size_t)nreg);
	}
	free(beta); free(mean); free(tsm); free(tmp);
	return the_likelyhood;
}

/*
 * libEmu/maxmultimin.c:288-394  evalFnMulti.
 *   theta_less_amp : nthetas-1 values {nugget, length...}
 *   det_mode 0     : det = (prod L_ii)^2, log(det)   (bit-faithful; under/overflows)
 *   det_mode 1     : log det = 2*sum(log L_ii)       (the usable form parity is asserted against)
 * Returns -logL, NaN on a non-PD matrix.  Optional outputs: sigma2 (estimateSigma),
 * beta (nreg), logdet, quad = r.Cinv.r, info = ORC_EDOM / ORC_SUCCESS.
 */
double orc_evalFnMulti(int kind, int order, const double *X, const double *y, const double *theta_less_amp,
                       int N, int nparams, int nthetas, int det_mode,
                       double *sigma2_out, double *beta_out, double *logdet_out, double *quad_out, int *info)
{
	int nreg = orc_nregression_fns(order, nparams), i, st;
	size_t nn = (size_t)N * N;
	double *theta_local = malloc(sizeof(double) * (size_t)nthetas);
	double *temp = malloc(sizeof(double) * nn);
	double *H = malloc(sizeof(double) * (size_t)N * nreg);
	double det, logdet, sigma2, val;

	theta_local[0] = 0.0;
	for (i = 1; i < nthetas; i++) theta_local[i] = theta_less_amp[i - 1];
	orc_make_hmatrix(order, H, X, N, nparams);               /* maxmultimin.c:75-79 */
	orc_make_cov_matrix(kind, temp, X, theta_local, N, nparams);
	st = orc_cholesky_decomp(temp, N);
	if (info) *info = st;
	if (st == ORC_EDOM) {
		free(theta_local); free(temp); free(H);
		return NAN;
	}
	if (det_mode == 0) {
		det = 1.0;
		for (i = 0; i < N; i++) det *= temp[(size_t)i * N + i];
		det = det * det;
		logdet = log(det);
	} else {
		logdet = 0.0;
		for (i = 0; i < N; i++) logdet += log(temp[(size_t)i * N + i]);
		logdet *= 2.0;
	}
	orc_cholesky_invert(temp, N);
	sigma2 = orc_estimate_sigma(temp, X, y, H, order, N, nparams, &st);
	if (st == ORC_SUCCESS)
		val = orc_get_loglikelyhood(temp, logdet, X, y, H, order, N, nparams, beta_out, quad_out, &st);
	else
		val = NAN;
	if (info) *info = st;
	if (sigma2_out) *sigma2_out = sigma2;
	if (logdet_out) *logdet_out = logdet;
	free(theta_local); free(temp); free(H);
	return -1 * val;
}

/* libEmu/maxmultimin.c:571-608  getGradientCn */
static double get_gradient_cn(const double *dCdtheta, const double *cinverse, const double *y, int N,
                              double *temp, double *v, double *w)
{
	double grad, trace = 0.0;
	int i;
	dgemm_s(0, 0, N, N, N, 1.0, cinverse, N, dCdtheta, N, 0.0, temp, N);
	for (i = 0; i < N; i++) trace += temp[(size_t)i * N + i];
	trace *= -(0.5);
	dgemv_s(0, N, N, 0.5, cinverse, N, y, 0.0, v);
	dgemv_s(0, N, N, 1.0, temp, N, v, 0.0, w);
	grad = ddot_s(N, y, 1, w, 1);
	grad += trace;
	return grad;
}

/* libEmu/maxmultimin.c:416-550  gradFnMulti.  grad has nthetas-1 entries.
 * Returns ORC_EDOM where the reference would exit(EXIT_FAILURE). */
int orc_gradFnMulti(int kind, int order, const double *X, const double *y, const double *theta_less_amp,
                    int N, int nparams, int nthetas, double *grad)
{
	int nreg = orc_nregression_fns(order, nparams), i, st;
	size_t nn = (size_t)N * N, q;
	double *theta_local = malloc(sizeof(double) * (size_t)nthetas);
	double *cinverse = malloc(sizeof(double) * nn);
	double *temp_matrix = malloc(sizeof(double) * nn);
	double *gtemp = malloc(sizeof(double) * nn);
	double *H = malloc(sizeof(double) * (size_t)N * nreg);
	double *v = malloc(sizeof(double) * (size_t)N);
	double *w = malloc(sizeof(double) * (size_t)N);
	double amp, nug, sigma_est;

	theta_local[0] = 0.0;
	for (i = 1; i < nthetas; i++) theta_local[i] = theta_less_amp[i - 1];
	orc_make_hmatrix(order, H, X, N, nparams);
	orc_make_cov_matrix(kind, cinverse, X, theta_local, N, nparams);
	st = orc_cholesky_decomp(cinverse, N);
	if (st == ORC_SUCCESS) {
		orc_cholesky_invert(cinverse, N);
		sigma_est = log(orc_estimate_sigma(cinverse, X, y, H, order, N, nparams, &st));
	}
	if (st == ORC_SUCCESS) {
		theta_local[0] = sigma_est;
		amp = exp(theta_local[0]);
		nug = exp(theta_local[1]);
		for (q = 0; q < nn; q++) temp_matrix[q] = 0.0;
		for (i = 0; i < N; i++) temp_matrix[(size_t)i * N + i] = 1.0 * nug;
		grad[0] = -1.0 * get_gradient_cn(temp_matrix, cinverse, y, N, gtemp, v, w);
		for (i = 2; i < nthetas; i++) {
			orc_derivative(kind, temp_matrix, X, theta_local[i], i, N, nparams);
			for (q = 0; q < nn; q++) temp_matrix[q] *= amp;
			grad[i - 1] = -1.0 * get_gradient_cn(temp_matrix, cinverse, y, N, gtemp, v, w);
		}
	}
	free(theta_local); free(cinverse); free(temp_matrix); free(gtemp); free(H); free(v); free(w);
	return st;
}

/* ------------------------------------------------------------------ */
/* prediction (emulator_struct.c:13-37,124-143; emulator.c:672-785)    */
/* ------------------------------------------------------------------ */

/* alloc_emulator_struct: cinverse (N*N), beta (nreg), H (N*nreg) from the
 * STORED full thetas.  log_det_out receives 2*sum(log L_ii).  Returns EDOM
 * where chol_inverse_cov_matrix (emulate-fns.c:275-299) would exit(1). */
int orc_emulator_setup(int kind, int order, const double *X, const double *y, const double *thetas,
                       int N, int nparams, double *cinverse, double *beta, double *H, double *log_det_out)
{
	int nreg = orc_nregression_fns(order, nparams), i, st;
	double logdet = 0.0;
	orc_make_cov_matrix(kind, cinverse, X, thetas, N, nparams);
	st = orc_cholesky_decomp(cinverse, N);
	if (st != ORC_SUCCESS) return st;
	for (i = 0; i < N; i++) logdet += log(cinverse[(size_t)i * N + i]);
	if (log_det_out) *log_det_out = 2.0 * logdet;
	orc_cholesky_invert(cinverse, N);
	orc_make_hmatrix(order, H, X, N, nparams);
	return orc_estimate_beta(beta, H, cinverse, y, N, nreg);
}

/* emulator.c:672-704  makeEmulatedMean */
static double make_emulated_mean(const double *cinverse, const double *y, const double *kplus, const double *h,
                                 const double *H, const double *beta, int N, int nreg, double *r1, double *r2)
{
	double emulated_mean, regression_cpt, residual_cpt;
	dgemv_s(0, N, N, 1.0, cinverse, N, y, 0.0, r1);
	emulated_mean = ddot_s(N, kplus, 1, r1, 1);
	regression_cpt = ddot_s(nreg, h, 1, beta, 1);
	dgemv_s(0, N, nreg, 1.0, H, nreg, beta, 0.0, r1);
	dgemv_s(0, N, N, 1.0, cinverse, N, r1, 0.0, r2);
	residual_cpt = ddot_s(N, kplus, 1, r2, 1);
	return regression_cpt + emulated_mean - residual_cpt;
}

/* emulator.c:720-785  makeEmulatedVariance; *status = EDOM where it would exit(1) */
static double make_emulated_variance(const double *cinverse, const double *kplus, const double *h,
                                     const double *H, double kappa, int N, int nreg, int *status)
{
	double *rn = malloc(sizeof(double) * (size_t)nreg);
	double *rn2 = malloc(sizeof(double) * (size_t)nreg);
	double *holder = malloc(sizeof(double) * (size_t)N);
	double *minv_h = malloc(sizeof(double) * (size_t)N * nreg);
	double *hmh = malloc(sizeof(double) * (size_t)nreg * nreg);
	double emulated_variance = NAN, regression_cpt = NAN;
	int i, st;
	dgemm_s(0, 0, N, nreg, N, 1.0, cinverse, N, H, nreg, 0.0, minv_h, nreg);
	dgemv_s(1, N, nreg, -1.0, minv_h, nreg, kplus, 0.0, rn);
	for (i = 0; i < nreg; i++) rn[i] += h[i];
	dgemm_s(1, 0, nreg, nreg, N, 1.0, H, nreg, minv_h, nreg, 0.0, hmh, nreg);
	st = orc_cholesky_decomp(hmh, nreg);
	if (status) *status = st;
	if (st == ORC_SUCCESS) {
		orc_cholesky_invert(hmh, nreg);
		dgemv_s(0, nreg, nreg, 1.0, hmh, nreg, rn, 0.0, rn2);
		regression_cpt = ddot_s(nreg, rn, 1, rn2, 1);
		dgemv_s(0, N, N, 1.0, cinverse, N, kplus, 0.0, holder);
		emulated_variance = ddot_s(N, kplus, 1, holder, 1);
	}
	free(rn); free(rn2); free(holder); free(minv_h); free(hmh);
	return kappa - emulated_variance + regression_cpt;
}

/* emulator_struct.c:124-143  emulate_point for M query rows */
int orc_emulate_points(int kind, int order, const double *X, const double *y, const double *thetas,
                       const double *cinverse, const double *beta, const double *H,
                       int N, int nparams, const double *Xq, int M, double *mean, double *var)
{
	int nreg = orc_nregression_fns(order, nparams), q, st = ORC_SUCCESS, s;
	double *kplus = malloc(sizeof(double) * (size_t)N);
	double *h = malloc(sizeof(double) * (size_t)nreg);
	double *r1 = malloc(sizeof(double) * (size_t)N);
	double *r2 = malloc(sizeof(double) * (size_t)N);
	for (q = 0; q < M; q++) {
		const double *point = Xq + (size_t)q * nparams;
		double kappa;
		orc_make_kvector(kind, kplus, X, point, thetas, N, nparams);
		orc_make_hvector(order, h, point, nparams);
		mean[q] = make_emulated_mean(cinverse, y, kplus, h, H, beta, N, nreg, r1, r2);
		kappa = orc_cov(kind, point, point, thetas, nparams);
		var[q] = make_emulated_variance(cinverse, kplus, h, H, kappa, N, nreg, &s);
		if (s != ORC_SUCCESS) st = s;
	}
	free(kplus); free(h); free(r1); free(r2);
	return st;
}

/* multivar_support.c:126-151 back-projection of nr PCA-space (mean,var) pairs to nt outputs:
 * mean_t = ybar_t + sum_j U_tj sqrt(lambda_j) m_j ;  var_t = sum_j U_tj^2 lambda_j v_j */
void orc_pca_backproject(int nt, int nr, const double *ybar, const double *evals, const double *evecs,
                         const double *m, const double *v, double *mean_out, double *var_out)
{
	int i, j;
	for (i = 0; i < nt; i++) {
		double mu = 0.0, vv = 0.0;
		for (j = 0; j < nr; j++) {
			double e = evecs[(size_t)i * nr + j];
			mu += e * sqrt(evals[j]) * m[j];
			vv += e * e * evals[j] * v[j];
		}
		mean_out[i] = ybar[i] + mu;
		var_out[i] = vv;
	}
}

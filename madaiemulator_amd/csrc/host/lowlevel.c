/*
 * lowlevel.c -- the reference's host-matrix interface below evalFnMulti / emulate_point, the functions libRbind
 * links against: they take and return N x N matrices in host memory.
 *
 *   chol_inverse_cov_matrix   libEmu/emulate-fns.c:275-299
 *   estimateBeta              libEmu/regression.c:120-176
 *   estimateSigma             libEmu/maxmultimin.c:215-273
 *   getLogLikelyhood          libEmu/estimator-fns.c:38-103
 *   makeEmulatedMean          libEmu/emulator.c:672-704
 *   makeEmulatedVariance      libEmu/emulator.c:720-785
 *   getGradientCn             libEmu/maxmultimin.c:571-608
 *   make*_es, estimateBeta_es emulator_struct.c:63-118
 *
 * The O(N^3) factorisation and every C^-1-times-vector product run on the device (gpemu_chol_inverse,
 * gpemu_symm_apply: the matrix is uploaded once per pointer/content and cached in the thread's scratch context);
 * the host only does the nreg x nreg algebra and dot products of length N.  Callers that can should use evalFnMulti /
 * emulate_point(s) instead: those keep everything resident.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "libemu.h"
#include "gpemu.h"

extern gpemu_ctx *gpemu_host_scratch_ctx(const char *where);

static void die_ll(gpemu_ctx *ctx, int rc, const char *where)
{
	fprintf(stderr, "%s: gpemu error %d: %s\n", where, rc, gpemu_last_error(ctx));
	gpemu_host_exit(EXIT_FAILURE);
}

/* rows: y (optional), then the columns of H -> C^-1 applied to each; returns nvec x N (caller frees) */
static double *apply_cinv(gsl_matrix *cinverse, const gsl_vector *first, const gsl_matrix *h_matrix, int N, int nreg, const char *where)
{
	const int nvec = (first ? 1 : 0) + nreg;
	double *v = (double *)malloc(sizeof(double) * (size_t)nvec * N), *out = (double *)malloc(sizeof(double) * (size_t)nvec * N);
	int r = 0;
	if (first) { for (int i = 0; i < N; i++) v[i] = gsl_vector_get(first, i); r = 1; }
	for (int a = 0; a < nreg; a++)
		for (int i = 0; i < N; i++) v[(size_t)(r + a) * N + i] = gsl_matrix_get(h_matrix, i, a);
	gpemu_ctx *ctx = gpemu_host_scratch_ctx(where);
	int rc = gpemu_symm_apply(ctx, N, cinverse->data, (int)cinverse->tda, nvec, v, out);
	if (rc) die_ll(ctx, rc, where);
	free(v);
	return out;
}

/* in-place Cholesky inverse of a small SPD matrix (nreg x nreg, row-major); 0 if not positive definite */
static int small_spd_inverse(double *A, int n)
{
	double *L = (double *)calloc((size_t)n * n, sizeof(double)), *Li = (double *)calloc((size_t)n * n, sizeof(double));
	int ok = 1;
	for (int j = 0; j < n && ok; j++) {
		double s = A[j * n + j];
		for (int k = 0; k < j; k++) s -= L[j * n + k] * L[j * n + k];
		if (!(s > 0.0)) { ok = 0; break; }
		L[j * n + j] = sqrt(s);
		for (int i = j + 1; i < n; i++) {
			double t = A[i * n + j];
			for (int k = 0; k < j; k++) t -= L[i * n + k] * L[j * n + k];
			L[i * n + j] = t / L[j * n + j];
		}
	}
	if (ok) {
		for (int c = 0; c < n; c++)
			for (int i = c; i < n; i++) {
				double s = (i == c) ? 1.0 : 0.0;
				for (int k = c; k < i; k++) s -= L[i * n + k] * Li[k * n + c];
				Li[i * n + c] = s / L[i * n + i];
			}
		for (int i = 0; i < n; i++)
			for (int j = 0; j < n; j++) {
				double s = 0.0;
				for (int k = (i > j ? i : j); k < n; k++) s += Li[k * n + i] * Li[k * n + j];
				A[i * n + j] = s;
			}
	}
	free(L); free(Li);
	return ok;
}

/* emulate-fns.c:275-299 */
void chol_inverse_cov_matrix(optstruct *options, gsl_matrix *temp_matrix, gsl_matrix *result_matrix, double *final_determinant_c)
{
	const int N = options->nmodel_points;
	gpemu_ctx *ctx = gpemu_host_scratch_ctx("chol_inverse_cov_matrix");
	double logdet = 0.0;
	int info = 0;
	int rc = gpemu_chol_inverse(ctx, N, temp_matrix->data, (int)temp_matrix->tda, &logdet, &info);
	if (rc == GPEMU_ERR_NOT_PD) {
		fprintf(stderr, "trying to cholesky a non postive def matrix, in emulate-fns.c sorry...\n");
		gpemu_host_exit(1);
	}
	if (rc) die_ll(ctx, rc, "chol_inverse_cov_matrix");
	gsl_matrix_memcpy(result_matrix, temp_matrix);
	*final_determinant_c = exp(logdet);      /* (prod L_ii)^2: under/overflows for large N exactly as the reference's product */
}

/* regression.c:120-176: beta = (H^T C^-1 H)^-1 H^T C^-1 y */
void estimateBeta(gsl_vector *beta_vector, gsl_matrix *h_matrix, gsl_matrix *cinverse, gsl_vector *trainingvector,
                  int nmodel_points, int nregression_fns)
{
	const int N = nmodel_points, nreg = nregression_fns;
	double *W = apply_cinv(cinverse, trainingvector, h_matrix, N, nreg, "estimateBeta");    /* rows: C^-1 y, C^-1 H_a */
	double *A = (double *)calloc((size_t)nreg * nreg, sizeof(double)), *b = (double *)calloc((size_t)nreg, sizeof(double));
	for (int a = 0; a < nreg; a++) {
		for (int i = 0; i < N; i++) b[a] += gsl_matrix_get(h_matrix, i, a) * W[i];
		for (int c = 0; c < nreg; c++) {
			double s = 0.0;
			for (int i = 0; i < N; i++) s += gsl_matrix_get(h_matrix, i, a) * W[(size_t)(1 + c) * N + i];
			A[a * nreg + c] = s;
		}
	}
	if (!small_spd_inverse(A, nreg)) {
		fprintf(stderr, "# err: estimateBeta\n# trying to cholesky a non postive def matrix, sorry...\n");
		gpemu_host_exit(1);                                          /* regression.c:134-160 */
	}
	for (int a = 0; a < nreg; a++) {
		double s = 0.0;
		for (int c = 0; c < nreg; c++) s += A[a * nreg + c] * b[c];
		gsl_vector_set(beta_vector, a, s);
	}
	free(W); free(A); free(b);
}

/* maxmultimin.c:215-273: sigma^2 = y^T C^-1 (y - H beta) / N for the model behind params */
double estimateSigma(gsl_matrix *cinverse, void *params_in)
{
	struct estimate_thetas_params *params = (struct estimate_thetas_params *)params_in;
	const int N = params->options->nmodel_points, nreg = params->options->nregression_fns;
	gsl_vector *y = params->the_model->training_vector;
	gsl_matrix *H = params->h_matrix;
	gsl_vector *beta = gsl_vector_alloc(nreg);
	estimateBeta(beta, H, cinverse, y, N, nreg);
	double *W = apply_cinv(cinverse, y, H, N, 0, "estimateSigma");          /* C^-1 y */
	double s = 0.0;
	for (int i = 0; i < N; i++) {
		double r = gsl_vector_get(y, i);
		for (int a = 0; a < nreg; a++) r -= gsl_matrix_get(H, i, a) * gsl_vector_get(beta, a);
		s += W[i] * r;
	}
	free(W);
	gsl_vector_free(beta);
	return s / (double)N;
}

/* estimator-fns.c:38-103 */
double getLogLikelyhood(gsl_matrix *cinverse, double det_cmatrix, gsl_matrix *xmodel, gsl_vector *trainingvector,
                        gsl_vector *thetas, gsl_matrix *h_matrix, int nmodel_points, int nthetas, int nparams,
                        int nregression_fns, void (*makeHVector)(gsl_vector *, gsl_vector *, int))
{
	(void)thetas; (void)nthetas;
	const int N = nmodel_points, nreg = nregression_fns;
	const double log_2_pi = 1.83788;
	gsl_vector *beta = gsl_vector_alloc(nreg), *h = gsl_vector_alloc(nreg), *r = gsl_vector_alloc(N);
	estimateBeta(beta, h_matrix, cinverse, trainingvector, N, nreg);
	for (int i = 0; i < N; i++) {
		gsl_vector_view row = gsl_matrix_row(xmodel, i);
		makeHVector(h, &row.vector, nparams);
		double m = 0.0;
		for (int a = 0; a < nreg; a++) m += gsl_vector_get(beta, a) * gsl_vector_get(h, a);
		gsl_vector_set(r, i, gsl_vector_get(trainingvector, i) - m);
	}
	double *W = apply_cinv(cinverse, r, h_matrix, N, 0, "getLogLikelyhood");   /* C^-1 r */
	double quad = 0.0;
	for (int i = 0; i < N; i++) quad += gsl_vector_get(r, i) * W[i];
	free(W);
	gsl_vector_free(beta); gsl_vector_free(h); gsl_vector_free(r);
	return -(1.0 / 2.0) * log(det_cmatrix) - (N / 2.0) * log_2_pi + quad * (-1.0 / 2.0);
}

/* emulator.c:672-704: h.beta + k*.C^-1 y - k*.C^-1 (H beta) */
double makeEmulatedMean(gsl_matrix *inverse_cov_matrix, gsl_vector *training_vector, gsl_vector *kplus_vector,
                        gsl_vector *h_vector, gsl_matrix *h_matrix, gsl_vector *beta_vector, int nmodel_points)
{
	const int N = nmodel_points, nreg = (int)beta_vector->size;
	double *U = apply_cinv(inverse_cov_matrix, kplus_vector, h_matrix, N, 0, "makeEmulatedMean");   /* C^-1 k* */
	double mean = 0.0;
	for (int a = 0; a < nreg; a++) mean += gsl_vector_get(h_vector, a) * gsl_vector_get(beta_vector, a);
	for (int i = 0; i < N; i++) {
		double r = gsl_vector_get(training_vector, i);
		for (int a = 0; a < nreg; a++) r -= gsl_matrix_get(h_matrix, i, a) * gsl_vector_get(beta_vector, a);
		mean += U[i] * r;
	}
	free(U);
	return mean;
}

/* emulator.c:720-785: kappa - k*.C^-1 k* + q^T (H^T C^-1 H)^-1 q,  q = h - H^T C^-1 k* */
double makeEmulatedVariance(gsl_matrix *inverse_cov_matrix, gsl_vector *kplus_vector, gsl_vector *h_vector,
                            gsl_matrix *h_matrix, double kappa, int nmodel_points, int nregression_fns)
{
	const int N = nmodel_points, nreg = nregression_fns;
	double *W = apply_cinv(inverse_cov_matrix, kplus_vector, h_matrix, N, nreg, "makeEmulatedVariance");
	double kck = 0.0;
	for (int i = 0; i < N; i++) kck += gsl_vector_get(kplus_vector, i) * W[i];
	double *A = (double *)calloc((size_t)nreg * nreg, sizeof(double)), *q = (double *)calloc((size_t)nreg, sizeof(double));
	for (int a = 0; a < nreg; a++) {
		double hk = 0.0;
		for (int i = 0; i < N; i++) hk += gsl_matrix_get(h_matrix, i, a) * W[i];            /* (H^T C^-1 k*)_a */
		q[a] = gsl_vector_get(h_vector, a) - hk;
		for (int c = 0; c < nreg; c++) {
			double s = 0.0;
			for (int i = 0; i < N; i++) s += gsl_matrix_get(h_matrix, i, a) * W[(size_t)(1 + c) * N + i];
			A[a * nreg + c] = s;
		}
	}
	if (!small_spd_inverse(A, nreg)) {
		fprintf(stderr, "trying to cholesky a non postive def matrix, sorry...\n");          /* emulator.c:751-766 */
		gpemu_host_exit(1);
	}
	double reg = 0.0;
	for (int a = 0; a < nreg; a++) {
		double t = 0.0;
		for (int c = 0; c < nreg; c++) t += A[a * nreg + c] * q[c];
		reg += q[a] * t;
	}
	free(W); free(A); free(q);
	return kappa - kck + reg;
}

/* maxmultimin.c:571-608: -1/2 trace(C^-1 dC) + 1/2 y^T C^-1 dC C^-1 y */
double getGradientCn(gsl_matrix *dCdtheta, gsl_matrix *cinverse, gsl_vector *training_vector, int nmodel_points, int nthetas)
{
	(void)nthetas;
	const int N = nmodel_points;
	gpemu_ctx *ctx = gpemu_host_scratch_ctx("getGradientCn");
	double trace = 0.0;
	int rc = gpemu_trace_product(ctx, N, cinverse->data, (int)cinverse->tda, dCdtheta->data, (int)dCdtheta->tda, &trace);
	if (rc) die_ll(ctx, rc, "getGradientCn");
	double *alpha = apply_cinv(cinverse, training_vector, NULL, N, 0, "getGradientCn");          /* C^-1 y */
	double *w = (double *)malloc(sizeof(double) * (size_t)N);
	rc = gpemu_symm_apply(ctx, N, dCdtheta->data, (int)dCdtheta->tda, 1, alpha, w);                /* dC alpha */
	if (rc) die_ll(ctx, rc, "getGradientCn");
	double quad = 0.0;
	for (int i = 0; i < N; i++) quad += alpha[i] * w[i];
	free(alpha); free(w);
	return -0.5 * trace + 0.5 * quad;
}

/* emulator_struct.c:63-118: the same functions on the emulator_struct's own model pointers */
void makeHMatrix_es(gsl_matrix *h_matrix, emulator_struct *e)
{
	makeHMatrix_fnptr(h_matrix, e->model->xmodel, e->nmodel_points, e->nparams, e->nregression_fns, e->model->makeHVector);
}
void makeCovMatrix_es(gsl_matrix *cov_matrix, emulator_struct *e)
{
	makeCovMatrix_fnptr(cov_matrix, e->model->xmodel, e->model->thetas, e->nmodel_points, e->nthetas, e->nparams,
	                    e->model->covariance_fn);
}
void makeKVector_es(gsl_vector *kvector, gsl_vector *point, emulator_struct *e)
{
	makeKVector_fnptr(kvector, e->model->xmodel, point, e->model->thetas, e->nmodel_points, e->nthetas, e->nparams,
	                  e->model->covariance_fn);
}
void estimateBeta_es(gsl_vector *beta_vector, emulator_struct *e)
{
	estimateBeta(beta_vector, e->h_matrix, e->cinverse, e->model->training_vector, e->nmodel_points, e->nregression_fns);
}

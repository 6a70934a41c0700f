/*
 * covreg.c -- the function-pointer targets of modelstruct (covariance_fn, makeHVector,
 * makeGradMatLength) and the matrix/vector builders of libEmu/emulator.h and
 * libEmu/regression.h.
 *
 * The scalar covariance functions evaluate ONE element (callers use them for kappa =
 * c(x*,x*), emulator_struct.c:135); which of the three a modelstruct points at selects the
 * device kernel.  Whole matrices / vectors are produced on the GPU (gpemu_cov_matrix,
 * gpemu_kvectors, gpemu_derivative_gauss).  gradFnMulti itself never materialises dC/dtheta
 * (device_bridge.c); the derivative-matrix builders below serve callers that want the matrix.
 */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "libemu.h"
#include "gpemu.h"

/* libEmu/emulator.c:101-152 */
double covariance_fn_gaussian(gsl_vector *xm, gsl_vector *xn, gsl_vector *thetas, int nthetas, int nparams)
{
	(void)nthetas;
	int same = 0;
	double expo = 0.0;
	for (int i = 0; i < nparams; i++) {
		double r = exp(gsl_vector_get(thetas, i + 2));
		double dist = fabs(gsl_vector_get(xm, i) - gsl_vector_get(xn, i));
		expo += (-1.0 / 2.0) * dist * dist / (r * r);
		if (dist < 0.0000000001) same++;
	}
	double c = exp(expo) * exp(gsl_vector_get(thetas, 0));
	if (same == nparams) c += exp(gsl_vector_get(thetas, 1));
	return c;
}

/* GPEMU_MODE_MATERN_LOG (gpemu.h; GPEMU_MATERN_FIXED=1, --matern_fixed, or recorded in the snapshot): amplitude and
 * nugget on the log scale, so that the scalar functions agree with the matrices the device builds in that mode.  The
 * flag lives in ONE place for the whole host layer (gpemu_host_modes, device_bridge.c). */
static int matern_log_scale(void) { return (gpemu_host_modes() & GPEMU_MODE_MATERN_LOG) != 0; }

static double matern_dist(gsl_vector *xm, gsl_vector *xn, int nparams, int *same)
{
	double d2 = 0.0;
	*same = 0;
	for (int i = 0; i < nparams; i++) {
		double t = fabs(gsl_vector_get(xm, i) - gsl_vector_get(xn, i));
		d2 += t * t;
		if (t < 0.0000000000000001) (*same)++;
	}
	return sqrt(d2);
}

/* libEmu/emulator.c:344-386 (amplitude and nugget are NOT exponentiated) */
double covariance_fn_matern_three(gsl_vector *xm, gsl_vector *xn, gsl_vector *thetas, int nthetas, int nparams)
{
	(void)nthetas;
	int same;
	const int ls = matern_log_scale();
	const double amp = ls ? exp(gsl_vector_get(thetas, 0)) : gsl_vector_get(thetas, 0);
	const double nugget = ls ? exp(gsl_vector_get(thetas, 1)) : gsl_vector_get(thetas, 1);
	const double rho = exp(gsl_vector_get(thetas, 2)), root3 = 1.732050808;
	const double dist = matern_dist(xm, xn, nparams, &same);
	double c = (dist > 0.0) ? amp * (1 + root3 * (dist / rho)) * exp(-root3 * (dist / rho)) : amp;
	if (same == nparams) c += nugget;
	return c;
}

/* libEmu/emulator.c:438-480 */
double covariance_fn_matern_five(gsl_vector *xm, gsl_vector *xn, gsl_vector *thetas, int nthetas, int nparams)
{
	(void)nthetas;
	int same;
	const int ls = matern_log_scale();
	const double amp = ls ? exp(gsl_vector_get(thetas, 0)) : gsl_vector_get(thetas, 0);
	const double nugget = ls ? exp(gsl_vector_get(thetas, 1)) : gsl_vector_get(thetas, 1);
	const double rho = exp(gsl_vector_get(thetas, 2)), root5 = 2.236067978;
	const double dist = matern_dist(xm, xn, nparams, &same);
	const double s = dist / rho;
	double c = (dist > 0.0) ? amp * (1 + root5 * s + (5.0 / 3.0) * s * s) * exp(-root5 * s) : amp;
	if (same == nparams) c += nugget;
	return c;
}

/* libEmu/emulator.c:173-209, materialised on the device (gradFnMulti itself never forms it) */
extern gpemu_ctx *gpemu_host_scratch_ctx(const char *where);
void derivative_l_gauss(gsl_matrix *dCdTheta, gsl_matrix *xmodel, double thetaLength, int index, int nmodel_points, int nparams)
{
	(void)nparams;
	const int N = nmodel_points, col = index - 2;         /* nthetasConstant = 2 (:178-179) */
	double *xc = (double *)malloc(sizeof(double) * (size_t)N);
	for (int i = 0; i < N; i++) xc[i] = gsl_matrix_get(xmodel, i, col);
	gpemu_ctx *ctx = gpemu_host_scratch_ctx("derivative_l_gauss");
	int rc = gpemu_derivative_gauss(ctx, N, xc, thetaLength, dCdTheta->data, (int)dCdTheta->tda);
	free(xc);
	if (rc) { fprintf(stderr, "derivative_l_gauss: gpemu error %d: %s\n", rc, gpemu_last_error(ctx)); gpemu_host_exit(EXIT_FAILURE); }
}
/* libEmu/emulator.c:401-433.  Literal: `rtemp` is never reset between (i,j) pairs (:410 vs :423-425), so element
 * (i,j) depends on every element before it in row-major order, and thetaLength is the raw (log-scale) value.  The
 * recurrence r <- sqrt(r + |x_i - x_j|^2) is a dependent chain of N^2 steps: it runs here, on the host, in the
 * reference's order (N^2 (d + sqrt + exp) scalar work, about 3 s at N = 8192).  The device never uses these
 * matrices: its Matern gradient is the analytic one behind GPEMU_EXACT_GRAD (gpemu.h). */
void derivative_l_matern_three(gsl_matrix *dCdTheta, gsl_matrix *xmodel, double thetaLength, int index, int nmodel_points,
                               int nparams)
{
	(void)index;                                          /* must be 2 (:403) */
	const double root3 = 1.732050808;
	double rtemp = 0.0;
	const double thetaLCubed = thetaLength * thetaLength * thetaLength;
	for (int i = 0; i < nmodel_points; i++)
		for (int j = 0; j < nmodel_points; j++) {
			for (int p = 0; p < nparams; p++) {
				const double x_temp = gsl_matrix_get(xmodel, i, p), y_temp = gsl_matrix_get(xmodel, j, p);
				rtemp += (x_temp - y_temp) * (x_temp - y_temp);
			}
			rtemp = sqrt(rtemp);
			gsl_matrix_set(dCdTheta, i, j, 3.0 * exp(-root3 * rtemp / thetaLength) * (rtemp * rtemp / thetaLCubed));
		}
}

/* libEmu/emulator.c:497-532 (same carried rtemp; constants 2.2360680, 3.72768, 1.66667 as written there) */
void derivative_l_matern_five(gsl_matrix *dCdTheta, gsl_matrix *xmodel, double thetaLength, int index, int nmodel_points,
                              int nparams)
{
	(void)index;
	const double root5 = 2.2360680;
	double rtemp = 0.0, rsq;
	const double thetaLCubed = thetaLength * thetaLength * thetaLength;
	for (int i = 0; i < nmodel_points; i++)
		for (int j = 0; j < nmodel_points; j++) {
			for (int p = 0; p < nparams; p++) {
				const double x_temp = gsl_matrix_get(xmodel, i, p), y_temp = gsl_matrix_get(xmodel, j, p);
				rtemp += (x_temp - y_temp) * (x_temp - y_temp);
			}
			rsq = rtemp;
			rtemp = sqrt(rtemp);
			gsl_matrix_set(dCdTheta, i, j,
			               (rsq / (thetaLCubed)) * exp(-root5 * rtemp / thetaLength) * (3.72768 * rtemp + 1.66667 * thetaLength));
		}
}

int gpemu_host_kind_of(double (*fn)(gsl_vector *, gsl_vector *, gsl_vector *, int, int))
{
	if (fn == &covariance_fn_gaussian) return GPEMU_POWEREXP;
	if (fn == &covariance_fn_matern_three) return GPEMU_MATERN32;
	if (fn == &covariance_fn_matern_five) return GPEMU_MATERN52;
	return 0;
}

extern int gpemu_host_device(void);

static void die_gpemu(gpemu_ctx *ctx, int rc, const char *where)
{
	fprintf(stderr, "%s: gpemu error %d: %s\n", where, rc, ctx ? gpemu_last_error(ctx) : "no context");
	gpemu_host_exit(EXIT_FAILURE);
}

static double *pack_matrix(const gsl_matrix *m)
{
	double *p = (double *)malloc(sizeof(double) * m->size1 * m->size2);
	for (size_t i = 0; i < m->size1; i++)
		for (size_t j = 0; j < m->size2; j++) p[i * m->size2 + j] = m->data[i * m->tda + j];
	return p;
}

static double *pack_vector(const gsl_vector *v)
{
	double *p = (double *)malloc(sizeof(double) * v->size);
	for (size_t i = 0; i < v->size; i++) p[i] = v->data[i * v->stride];
	return p;
}

/* libEmu/emulator.c:636-653: the full N x N matrix, filled on the GPU */
void makeCovMatrix_fnptr(gsl_matrix *cov_matrix, gsl_matrix *xmodel, gsl_vector *thetas, int nmodel_points, int nthetas,
                         int nparams, double (*covariance_fn_ptr)(gsl_vector *, gsl_vector *, gsl_vector *, int, int))
{
	const int kind = gpemu_host_kind_of(covariance_fn_ptr);
	if (!kind) { fprintf(stderr, "makeCovMatrix_fnptr: unknown covariance function (no device kernel)\n"); gpemu_host_exit(EXIT_FAILURE); }
	gpemu_ctx *ctx = NULL;
	int rc = gpemu_ctx_create(&ctx, gpemu_host_device());
	if (rc) die_gpemu(NULL, rc, "makeCovMatrix_fnptr");
	double *X = pack_matrix(xmodel), *th = pack_vector(thetas);
	double *y = (double *)calloc((size_t)nmodel_points, sizeof(double));
	double *out = (double *)malloc(sizeof(double) * (size_t)nmodel_points * nmodel_points);
	rc = gpemu_set_model(ctx, kind, 0, nmodel_points, nparams, X, y);
	if (!rc) rc = gpemu_cov_matrix(ctx, th, nthetas, out);
	if (rc) die_gpemu(ctx, rc, "makeCovMatrix_fnptr");
	for (int i = 0; i < nmodel_points; i++)
		for (int j = 0; j < nmodel_points; j++) gsl_matrix_set(cov_matrix, i, j, out[(size_t)i * nmodel_points + j]);
	free(X); free(th); free(y); free(out);
	gpemu_ctx_destroy(ctx);
}

/* libEmu/emulator.c:578-593 (with the 1e-10 clamp), on the GPU */
void makeKVector_fnptr(gsl_vector *kvector, gsl_matrix *xmodel, gsl_vector *xnew, gsl_vector *thetas, int nmodel_points,
                       int nthetas, int nparams,
                       double (*covariance_fn_ptr)(gsl_vector *, gsl_vector *, gsl_vector *, int, int))
{
	const int kind = gpemu_host_kind_of(covariance_fn_ptr);
	if (!kind) { fprintf(stderr, "makeKVector_fnptr: unknown covariance function (no device kernel)\n"); gpemu_host_exit(EXIT_FAILURE); }
	gpemu_ctx *ctx = NULL;
	int rc = gpemu_ctx_create(&ctx, gpemu_host_device());
	if (rc) die_gpemu(NULL, rc, "makeKVector_fnptr");
	double *X = pack_matrix(xmodel), *th = pack_vector(thetas), *q = pack_vector(xnew);
	double *y = (double *)calloc((size_t)nmodel_points, sizeof(double));
	double *out = (double *)malloc(sizeof(double) * (size_t)nmodel_points);
	rc = gpemu_set_model(ctx, kind, 0, nmodel_points, nparams, X, y);
	if (!rc) rc = gpemu_kvectors(ctx, th, nthetas, 1, q, out);
	if (rc) die_gpemu(ctx, rc, "makeKVector_fnptr");
	for (int i = 0; i < nmodel_points; i++) gsl_vector_set(kvector, i, out[i]);
	free(X); free(th); free(q); free(y); free(out);
	gpemu_ctx_destroy(ctx);
}

/* libEmu/regression.c:9-67 */
void makeHVector_trivial(gsl_vector *h, gsl_vector *x, int nparams)
{
	(void)x; (void)nparams;
	gsl_vector_set_zero(h);
	gsl_vector_set(h, 0, 1);
}

static void hvec_order(gsl_vector *h, gsl_vector *x, int nparams, int order)
{
	gsl_vector_set(h, 0, 1);
	for (int p = 1; p <= order; p++)
		for (int i = 0; i < nparams; i++) {
			double v = gsl_vector_get(x, i), w = v;
			for (int q = 1; q < p; q++) w *= v;
			gsl_vector_set(h, (p - 1) * nparams + i + 1, w);
		}
}
void makeHVector_linear(gsl_vector *h, gsl_vector *x, int nparams) { hvec_order(h, x, nparams, 1); }
void makeHVector_quadratic(gsl_vector *h, gsl_vector *x, int nparams) { hvec_order(h, x, nparams, 2); }
void makeHVector_cubic(gsl_vector *h, gsl_vector *x, int nparams) { hvec_order(h, x, nparams, 3); }

/* libEmu/regression.c:100-112 */
void makeHMatrix_fnptr(gsl_matrix *h_matrix, gsl_matrix *xmodel, int nmodel_points, int nparams, int nregression_fns,
                       void (*makeHVector_ptr)(gsl_vector *, gsl_vector *, int))
{
	(void)nregression_fns;
	for (int i = 0; i < nmodel_points; i++) {
		gsl_vector_view xr = gsl_matrix_row(xmodel, i);
		gsl_vector_view hr = gsl_matrix_row(h_matrix, i);
		makeHVector_ptr(&hr.vector, &xr.vector, nparams);
	}
}

/* ---- the process-wide function pointers of the reference's headers (emulator.h:13, regression.h:32,
 * maxmultimin.h:26) and the entry points that dispatch on them (emulator.c:548-562, 607-623; regression.c:77-91).
 * set_global_ptrs (modelstruct.c) sets them together with the model's own pointers; as in the reference, two models
 * with different kernels in one process make these -- and only these -- entry points ambiguous. */
double (*covariance_fn)(gsl_vector *, gsl_vector *, gsl_vector *, int, int) = NULL;
void (*makeHVector)(gsl_vector *h_vector, gsl_vector *x_location, int nparams) = NULL;
void (*makeGradMatLength)(gsl_matrix *dCdTheta, gsl_matrix *xmodel, double thetaLength, int index, int nmodel_points,
                          int nparams) = NULL;

static void need_global(const void *p, const char *who)
{
	if (!p) { fprintf(stderr, "%s: set_global_ptrs has not been called\n", who); gpemu_host_exit(EXIT_FAILURE); }
}

void makeCovMatrix(gsl_matrix *cov_matrix, gsl_matrix *xmodel, gsl_vector *thetas, int nmodel_points, int nthetas, int nparams)
{
	need_global((const void *)covariance_fn, "makeCovMatrix");
	makeCovMatrix_fnptr(cov_matrix, xmodel, thetas, nmodel_points, nthetas, nparams, covariance_fn);
}

void makeKVector(gsl_vector *kvector, gsl_matrix *xmodel, gsl_vector *xnew, gsl_vector *thetas, int nmodel_points,
                 int nthetas, int nparams)
{
	need_global((const void *)covariance_fn, "makeKVector");
	makeKVector_fnptr(kvector, xmodel, xnew, thetas, nmodel_points, nthetas, nparams, covariance_fn);
}

void makeHMatrix(gsl_matrix *h_matrix, gsl_matrix *xmodel, int nmodel_points, int nparams, int nregression_fns)
{
	need_global((const void *)makeHVector, "makeHMatrix");
	makeHMatrix_fnptr(h_matrix, xmodel, nmodel_points, nparams, nregression_fns, makeHVector);
}

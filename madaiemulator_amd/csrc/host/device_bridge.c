/*
 * device_bridge.c -- the reference's likelihood / gradient / prediction entry points
 * (libEmu/maxmultimin.c:148-618, emulator_struct.c:13-143) implemented on the device
 * library.  A gpemu_ctx (one HIP stream + HBM workspace) is cached per `params` pointer
 * (likelihood side) or per emulator_struct (prediction side): the reference's per-thread
 * deep copies (estimate_threaded.c:57-68) become per-thread device contexts, so
 * evalFnMulti / gradFnMulti stay re-entrant across threads with distinct params.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>
#include "libemu.h"
#include "gpemu.h"

extern int gpemu_host_kind_of(double (*fn)(gsl_vector *, gsl_vector *, gsl_vector *, int, int));

static int g_device = 0;
void gpemu_host_set_device(int device) { g_device = device; }
int gpemu_host_device(void) { return g_device; }

/* ---------------------------------------------------------------- registry */
struct entry {
	const void *key;
	gpemu_ctx *ctx;
	const double *xdata, *ydata;      /* identity of the uploaded data */
	int N, d, kind, order;
	struct entry *next;
};
static struct entry *g_entries = NULL;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;

static void die(gpemu_ctx *ctx, int rc, const char *where)
{
	fprintf(stderr, "%s: gpemu error %d: %s\n", where, rc, ctx ? gpemu_last_error(ctx) : "no usable HIP device");
	exit(EXIT_FAILURE);
}

static struct entry *lookup(const void *key, int create)
{
	struct entry *e;
	pthread_mutex_lock(&g_lock);
	for (e = g_entries; e; e = e->next)
		if (e->key == key) break;
	if (!e && create) {
		e = (struct entry *)calloc(1, sizeof *e);
		e->key = key;
		int rc = gpemu_ctx_create(&e->ctx, g_device);
		if (rc) { pthread_mutex_unlock(&g_lock); die(NULL, rc, "gpemu_ctx_create"); }
		e->next = g_entries;
		g_entries = e;
	}
	pthread_mutex_unlock(&g_lock);
	return e;
}

void gpemu_host_release(void *key)
{
	pthread_mutex_lock(&g_lock);
	struct entry **pp = &g_entries;
	while (*pp) {
		if ((*pp)->key == key) {
			struct entry *e = *pp;
			*pp = e->next;
			gpemu_ctx_destroy(e->ctx);
			free(e);
			break;
		}
		pp = &(*pp)->next;
	}
	pthread_mutex_unlock(&g_lock);
}

static double *pack_matrix(const gsl_matrix *m)
{
	double *p = (double *)malloc(sizeof(double) * m->size1 * m->size2);
	for (size_t i = 0; i < m->size1; i++)
		memcpy(p + i * m->size2, m->data + i * m->tda, sizeof(double) * m->size2);
	return p;
}
static double *pack_vector(const gsl_vector *v)
{
	double *p = (double *)malloc(sizeof(double) * v->size);
	for (size_t i = 0; i < v->size; i++) p[i] = v->data[i * v->stride];
	return p;
}

/* make sure the model's design / training vector are the ones resident in this entry's HBM */
static gpemu_ctx *bind_model(const void *key, modelstruct *m, const char *where)
{
	struct entry *e = lookup(key, 1);
	const optstruct *o = m->options;
	const int kind = gpemu_host_kind_of(m->covariance_fn);
	if (!kind) { fprintf(stderr, "%s: unknown covariance function (no device kernel)\n", where); exit(EXIT_FAILURE); }
	if (e->xdata != m->xmodel->data || e->N != o->nmodel_points || e->d != o->nparams || e->kind != kind ||
	    e->order != o->regression_order) {
		double *X = pack_matrix(m->xmodel), *y = pack_vector(m->training_vector);
		int rc = gpemu_set_model(e->ctx, kind, o->regression_order, o->nmodel_points, o->nparams, X, y);
		free(X); free(y);
		if (rc) die(e->ctx, rc, where);
		e->xdata = m->xmodel->data; e->ydata = m->training_vector->data;
		e->N = o->nmodel_points; e->d = o->nparams; e->kind = kind; e->order = o->regression_order;
	} else if (e->ydata != m->training_vector->data) {
		double *y = pack_vector(m->training_vector);
		int rc = gpemu_set_training(e->ctx, y);
		free(y);
		if (rc) die(e->ctx, rc, where);
		e->ydata = m->training_vector->data;
	}
	return e->ctx;
}

/* theta_local = [0, theta_less_amp...]  (maxmultimin.c:311-313) */
static double *full_thetas(const gsl_vector *less_amp, int nthetas)
{
	double *t = (double *)malloc(sizeof(double) * (size_t)nthetas);
	t[0] = 0.0;
	for (int i = 1; i < nthetas; i++) t[i] = gsl_vector_get(less_amp, i - 1);
	return t;
}

static void note_not_pd(const char *who, const double *th, int nthetas)
{
	/* the reference dumps the whole matrix to chol-err.dat (maxmultimin.c:327-343); here only the thetas are
	 * written -- the N x N matrix lives in HBM and can be regenerated with makeCovMatrix_fnptr */
	fprintf(stderr, "%s\n", who);
	fprintf(stderr, "trying to cholesky a non postive def matrix, sorry...\n");
	fprintf(stderr, "thetas dumped to chol-err.dat\n");
	FILE *f = fopen("chol-err.dat", "w");
	if (f) {
		fprintf(f, "#thetas: ");
		for (int i = 0; i < nthetas; i++) fprintf(f, "%lf\t", th[i]);
		fprintf(f, "\n");
		fclose(f);
	}
}

/* libEmu/maxmultimin.c:288-394 */
double evalFnMulti(const gsl_vector *theta_vec_less_amp, void *params_in)
{
	struct estimate_thetas_params *params = (struct estimate_thetas_params *)params_in;
	const int nthetas = params->options->nthetas;
	gpemu_ctx *ctx = bind_model(params, params->the_model, "evalFnMulti");
	double *th = full_thetas(theta_vec_less_amp, nthetas);
	double val = GSL_NAN;
	int info = 0;
	int rc = gpemu_loglik(ctx, th, nthetas, &val, NULL, NULL, NULL, NULL, &info);
	if (rc == GPEMU_ERR_NOT_PD) {
		note_not_pd("evalFnMulti", th, nthetas);
		val = GSL_NAN;
	} else if (rc == GPEMU_ERR_REGRESSION) {
		fprintf(stderr, "# err: estimateBeta\n# trying to cholesky a non postive def matrix, sorry...\n");
		exit(1);                                          /* regression.c:134-160 */
	} else if (rc) {
		die(ctx, rc, "evalFnMulti");
	}
	free(th);
	return val;
}

/* evalFnMulti over a list of theta rows (npts x (nthetas-1), each {nugget, lengths...}): the rows are independent
 * evaluations of one model (libRbind/rbind.c:626-724 callEvalLhoodList; the restarts of maxWithMultiMin), so they
 * go to the device in lock-step batches (gpemu_loglik_batch) instead of one factorisation at a time.  answer[i] is
 * what evalFnMulti would have returned for row i. */
void evalFnMultiList(const gsl_matrix *theta_rows_less_amp, void *params_in, double *answer)
{
	struct estimate_thetas_params *params = (struct estimate_thetas_params *)params_in;
	const int nthetas = params->options->nthetas;
	const int npts = (int)theta_rows_less_amp->size1;
	gpemu_ctx *ctx = bind_model(params, params->the_model, "evalFnMultiList");
	int maxb = 16;                                         /* workspace: maxb * (N+64) * N * 8 bytes */
	const char *e = getenv("GPEMU_HOST_BATCH");
	if (e && atoi(e) >= 1) maxb = atoi(e) > GPEMU_MAX_BATCH ? GPEMU_MAX_BATCH : atoi(e);
	double *th = (double *)calloc((size_t)maxb * nthetas, sizeof(double));
	int *status = (int *)malloc(sizeof(int) * (size_t)maxb);
	for (int p0 = 0; p0 < npts; p0 += maxb) {
		const int nb = npts - p0 < maxb ? npts - p0 : maxb;
		for (int b = 0; b < nb; b++) {
			th[(size_t)b * nthetas] = 0.0;                    /* theta_local[0] = 0 (maxmultimin.c:322) */
			for (int i = 1; i < nthetas; i++)
				th[(size_t)b * nthetas + i] = gsl_matrix_get(theta_rows_less_amp, p0 + b, i - 1);
		}
		int rc = gpemu_loglik_batch(ctx, nb, th, nthetas, answer + p0, NULL, NULL, NULL, NULL, NULL, status);
		if (rc) die(ctx, rc, "evalFnMultiList");
		for (int b = 0; b < nb; b++) {
			if (status[b] == GPEMU_ERR_NOT_PD) {
				note_not_pd("evalFnMulti", th + (size_t)b * nthetas, nthetas);
				answer[p0 + b] = GSL_NAN;
			} else if (status[b] == GPEMU_ERR_REGRESSION) {
				fprintf(stderr, "# err: estimateBeta\n# trying to cholesky a non postive def matrix, sorry...\n");
				exit(1);                                      /* regression.c:134-160 */
			} else if (status[b]) {
				die(ctx, status[b], "evalFnMultiList");
			}
		}
	}
	free(status); free(th);
}

static void grad_failure(gpemu_ctx *ctx, int rc, const double *th, int nthetas)
{
	if (rc == GPEMU_ERR_NOT_PD) {
		note_not_pd("gradFnMulti", th, nthetas);
		exit(EXIT_FAILURE);                               /* maxmultimin.c:495 */
	}
	die(ctx, rc, "gradFnMulti");
}

/* libEmu/maxmultimin.c:416-550 */
void gradFnMulti(const gsl_vector *theta_vec_less_amp, void *params_in, gsl_vector *grad_vec)
{
	struct estimate_thetas_params *params = (struct estimate_thetas_params *)params_in;
	const int nthetas = params->options->nthetas;
	gpemu_ctx *ctx = bind_model(params, params->the_model, "gradFnMulti");
	double *th = full_thetas(theta_vec_less_amp, nthetas);
	double *g = (double *)malloc(sizeof(double) * (size_t)(nthetas - 1));
	int info = 0;
	int rc = gpemu_grad(ctx, th, nthetas, g, &info);
	if (rc) grad_failure(ctx, rc, th, nthetas);
	for (int i = 0; i < nthetas - 1; i++) gsl_vector_set(grad_vec, i, g[i]);
	free(g); free(th);
}

/* libEmu/maxmultimin.c:615-618, one factorisation instead of two */
void evalFnGradMulti(const gsl_vector *theta_vec, void *params_in, double *fnval, gsl_vector *grad_vec)
{
	struct estimate_thetas_params *params = (struct estimate_thetas_params *)params_in;
	const int nthetas = params->options->nthetas;
	gpemu_ctx *ctx = bind_model(params, params->the_model, "evalFnGradMulti");
	double *th = full_thetas(theta_vec, nthetas);
	double *g = (double *)malloc(sizeof(double) * (size_t)(nthetas - 1));
	int info = 0;
	int rc = gpemu_loglik_grad(ctx, th, nthetas, fnval, NULL, NULL, g, &info);
	if (rc == GPEMU_ERR_NOT_PD) {
		/* The reference would return GSL_NAN from evalFnMulti and then exit(EXIT_FAILURE) inside gradFnMulti
		 * (maxmultimin.c:349,495) -- a line-search trial point that is numerically not positive definite kills
		 * the whole training run.  Here the pair reports NaN value AND NaN gradient and the line search backs off. */
		note_not_pd("evalFnGradMulti", th, nthetas);
		*fnval = GSL_NAN;
		for (int i = 0; i < nthetas - 1; i++) gsl_vector_set(grad_vec, i, GSL_NAN);
		free(g); free(th);
		return;
	}
	if (rc) grad_failure(ctx, rc, th, nthetas);
	for (int i = 0; i < nthetas - 1; i++) gsl_vector_set(grad_vec, i, g[i]);
	free(g); free(th);
}

/* libEmu/maxmultimin.c:148-201: sigma^2 at {nug, lengths...}; NaN if the matrix is not positive definite */
double estimateSigmaFull(gsl_vector *thetas_less_amp, void *params_in)
{
	struct estimate_thetas_params *params = (struct estimate_thetas_params *)params_in;
	const int nthetas = params->options->nthetas;
	gpemu_ctx *ctx = bind_model(params, params->the_model, "estimateSigmaFull");
	double *th = full_thetas(thetas_less_amp, nthetas);
	double s2 = GSL_NAN;
	int info = 0;
	int rc = gpemu_loglik(ctx, th, nthetas, NULL, &s2, NULL, NULL, NULL, &info);
	if (rc == GPEMU_ERR_NOT_PD) { note_not_pd("estSigmaFull", th, nthetas); s2 = GSL_NAN; }
	else if (rc) die(ctx, rc, "estimateSigmaFull");
	free(th);
	return s2;
}

/* ---------------------------------------------------------------- prediction */

/* emulator_struct.c:13-37: factor once, keep everything the sweep needs resident in HBM.  cinverse / beta /
 * h_matrix are filled on the host as well because they are public fields of the struct. */
emulator_struct *alloc_emulator_struct(modelstruct *model)
{
	emulator_struct *e = (emulator_struct *)malloc(sizeof(emulator_struct));
	e->nparams = model->options->nparams;
	e->nmodel_points = model->options->nmodel_points;
	e->nregression_fns = model->options->nregression_fns;
	e->nthetas = model->options->nthetas;
	e->model = model;
	e->cinverse = gsl_matrix_alloc(e->nmodel_points, e->nmodel_points);
	e->beta_vector = gsl_vector_alloc(e->nregression_fns);
	e->h_matrix = gsl_matrix_alloc(e->nmodel_points, e->nregression_fns);

	gpemu_ctx *ctx = bind_model(e, model, "alloc_emulator_struct");
	double *th = pack_vector(model->thetas);
	double *beta = (double *)malloc(sizeof(double) * (size_t)e->nregression_fns);
	int info = 0;
	int rc = gpemu_predict_setup(ctx, th, e->nthetas, beta, &info);
	if (rc == GPEMU_ERR_NOT_PD) {
		fprintf(stderr, "trying to cholesky a non postive def matrix, in emulate-fns.c sorry...\n");
		exit(1);                                          /* emulate-fns.c:282-285 */
	}
	if (rc) die(ctx, rc, "alloc_emulator_struct");
	for (int a = 0; a < e->nregression_fns; a++) gsl_vector_set(e->beta_vector, a, beta[a]);
	makeHMatrix_fnptr(e->h_matrix, model->xmodel, e->nmodel_points, e->nparams, e->nregression_fns, model->makeHVector);
	if (!getenv("GPEMU_SKIP_CINVERSE")) {
		rc = gpemu_get_cinverse(ctx, e->cinverse->data);
		if (rc) die(ctx, rc, "alloc_emulator_struct(cinverse)");
	}
	free(th); free(beta);
	return e;
}

void free_emulator_struct(emulator_struct *e)
{
	if (!e) return;
	gpemu_host_release(e);
	gsl_matrix_free(e->cinverse);
	gsl_matrix_free(e->h_matrix);
	gsl_vector_free(e->beta_vector);
	free(e);
}

void emulate_points(emulator_struct *e, gsl_matrix *points, double *mean, double *variance)
{
	struct entry *en = lookup(e, 0);
	if (!en) { fprintf(stderr, "emulate_point: emulator_struct was not made by alloc_emulator_struct\n"); exit(EXIT_FAILURE); }
	double *q = pack_matrix(points);
	int rc = gpemu_predict_batch(en->ctx, (int)points->size1, q, mean, variance);
	free(q);
	if (rc) die(en->ctx, rc, "emulate_point");
}

/* emulator_struct.c:124-143 */
void emulate_point(emulator_struct *e, gsl_vector *point, double *mean, double *variance)
{
	gsl_matrix view;
	double *q = pack_vector(point);
	view.size1 = 1; view.size2 = point->size; view.tda = point->size; view.data = q; view.block = NULL; view.owner = 0;
	emulate_points(e, &view, mean, variance);
	free(q);
}

/*
 * rbind_compat.c -- the stateless libRbind entry points, without R: same flat .C()-style signatures (pointers to
 * scalars, column-major arrays as R passes them).
 *   callEvalLhoodList  libRbind/rbind.c:626-724   a ready-made batch of independent likelihood evaluations
 *   callEstimate       libRbind/rbind.c:35-98     estimate_thetas on flat arrays
 *   callEmulateAtList  libRbind/rbind.c:121-190   mean / variance at a list of points for given thetas
 *   callEmulateAtPt    libRbind/rbind.c:214-290   ... at one point
 *   setupEmulateMC / callEmulateMC / freeEmulateMC               rbind.c:299-460   one emulator kept between calls (MCMC)
 *   setupEmulateMCMulti / callEmulateMCMulti / freeEmulateMCMulti rbind.c:483-600  nydims independent emulators of one design
 * The rows of pointList are independent evaluations of one model: they go through evalFnMultiList, i.e. through
 * lock-step batches of GPU factorisations (gpemu_loglik_batch).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libemu.h"

/* rbind.c:840-855: input is an (ny x nx) matrix flattened column by column */
static void column_major_to_matrix(gsl_matrix *m, const double *input, int nx, int ny)
{
	for (int i = 0; i < nx; i++)
		for (int j = 0; j < ny; j++) gsl_matrix_set(m, j, i, input[j + ny * i]);
}

void callEvalLhoodList(double *xmodel_in, int *nparams_in, double *pointList_in, int *nevalPoints_in,
                       double *training_in, int *nmodelPoints_in, int *nthetas_in, double *answer,
                       int *cov_fn_index_in, int *regression_order_in)
{
	const int N = *nmodelPoints_in, d = *nparams_in, npts = *nevalPoints_in;
	gsl_matrix *x = gsl_matrix_alloc(N, d);
	gsl_vector *y = gsl_vector_alloc(N);
	column_major_to_matrix(x, xmodel_in, d, N);
	for (int i = 0; i < N; i++) gsl_vector_set(y, i, training_in[i]);
	modelstruct *model = alloc_modelstruct_2(x, y, *cov_fn_index_in, *regression_order_in);
	const int nthetas = model->options->nthetas;        /* the reference trusts *nthetas_in to equal this */
	(void)nthetas_in;
	gsl_matrix *pts = gsl_matrix_alloc(npts, nthetas);
	column_major_to_matrix(pts, pointList_in, nthetas, npts);
	struct estimate_thetas_params params;
	memset(&params, 0, sizeof params);
	params.options = model->options;
	params.the_model = model;
	/* evalFnMulti reads nthetas-1 entries {nugget, lengths...} from the start of each row (rbind.c:704-712);
	 * evalFnMultiList does the same for every row of pts (its last column is never read) */
	evalFnMultiList(pts, &params, answer);
	gpemu_host_release(&params);
	gsl_matrix_free(pts);
	gsl_matrix_free(model->xmodel);
	free_modelstruct_2(model);
	gsl_matrix_free(x);
	gsl_vector_free(y);
}

/* a model on flat R arrays; the caller frees with free_flat_model */
static modelstruct *flat_model(double *xmodel_in, int d, double *training_in, int N, int cov_fn_index, int regression_order,
                               gsl_matrix **x_out, gsl_vector **y_out)
{
	gsl_matrix *x = gsl_matrix_alloc(N, d);
	gsl_vector *y = gsl_vector_alloc(N);
	column_major_to_matrix(x, xmodel_in, d, N);
	for (int i = 0; i < N; i++) gsl_vector_set(y, i, training_in[i]);
	*x_out = x; *y_out = y;
	return alloc_modelstruct_2(x, y, cov_fn_index, regression_order);
}

static void free_flat_model(modelstruct *model, gsl_matrix *x, gsl_vector *y)
{
	gsl_matrix_free(model->xmodel);
	free_modelstruct_2(model);
	gsl_matrix_free(x);
	gsl_vector_free(y);
}

/* rbind.c:35-98 (the fixed-nugget arguments are accepted and, as in the reference, overridden: :62-64) */
void callEstimate(double *xmodel_in, int *nparams_in, double *training_in, int *nmodelpts, int *nthetas_in, double *final_thetas,
                  int *use_fixed_nugget, double *fixed_nugget_in, int *cov_fn_index_in, int *regression_order_in)
{
	(void)use_fixed_nugget; (void)fixed_nugget_in;
	gsl_matrix *x; gsl_vector *y;
	modelstruct *model = flat_model(xmodel_in, *nparams_in, training_in, *nmodelpts, *cov_fn_index_in, *regression_order_in, &x, &y);
	estimate_thetas_threaded(model, model->options);
	const int n = *nthetas_in < model->options->nthetas ? *nthetas_in : model->options->nthetas;
	for (int i = 0; i < n; i++) final_thetas[i] = gsl_vector_get(model->thetas, i);
	free_flat_model(model, x, y);
}

/* rbind.c:121-190 */
void callEmulateAtList(double *xmodel_in, int *nparams_in, double *points_in, int *nemupoints, double *training_in,
                       int *nmodelpts, double *thetas_in, int *nthetas_in, double *final_emulated_y,
                       double *final_emulated_variance, int *cov_fn_index_in, int *regression_order_in)
{
	gsl_matrix *x; gsl_vector *y;
	const int d = *nparams_in, M = *nemupoints;
	modelstruct *model = flat_model(xmodel_in, d, training_in, *nmodelpts, *cov_fn_index_in, *regression_order_in, &x, &y);
	const int n = *nthetas_in < model->options->nthetas ? *nthetas_in : model->options->nthetas;
	for (int i = 0; i < n; i++) gsl_vector_set(model->thetas, i, thetas_in[i]);
	gsl_matrix *pts = gsl_matrix_alloc(M, d);
	column_major_to_matrix(pts, points_in, d, M);
	emulator_struct *e = alloc_emulator_struct(model);
	emulate_points(e, pts, final_emulated_y, final_emulated_variance);
	free_emulator_struct(e);
	gsl_matrix_free(pts);
	free_flat_model(model, x, y);
}

/* rbind.c:214-290 */
void callEmulateAtPt(double *xmodel_in, int *nparams_in, double *point_in, double *training_in, int *nmodelpts,
                     double *thetas_in, int *nthetas_in, double *final_emulated_y, double *final_emulated_variance,
                     int *cov_fn_index_in, int *regression_order_in)
{
	int one = 1;
	callEmulateAtList(xmodel_in, nparams_in, point_in, &one, training_in, nmodelpts, thetas_in, nthetas_in, final_emulated_y,
	                  final_emulated_variance, cov_fn_index_in, regression_order_in);
}

/* ---- the emulator kept between calls (rbind.c:299-460).  The reference keeps C, C^-1, beta and H on the host and runs
 * emulateQuick per call; here the factor stays in HBM behind an emulator_struct and a call is one emulate_point. ---- */
struct emulateMCData { modelstruct *model; emulator_struct *emu; gsl_matrix *x; gsl_vector *y; };
static struct emulateMCData emuMCData;
static struct emulateMCData *emuMCDataMulti;

static void mc_setup(struct emulateMCData *m, double *xmodel_in, int d, double *training_in, int N, const double *thetas,
                     int nthetas_in, int cov_fn_index, int regression_order)
{
	m->model = flat_model(xmodel_in, d, training_in, N, cov_fn_index, regression_order, &m->x, &m->y);
	const int n = nthetas_in < m->model->options->nthetas ? nthetas_in : m->model->options->nthetas;
	for (int i = 0; i < n; i++) gsl_vector_set(m->model->thetas, i, thetas[i]);
	m->emu = alloc_emulator_struct(m->model);
}

static void mc_free(struct emulateMCData *m)
{
	if (!m->model) return;
	free_emulator_struct(m->emu);
	free_flat_model(m->model, m->x, m->y);
	memset(m, 0, sizeof *m);
}

void setupEmulateMC(double *xmodel_in, int *nparams_in, double *training_in, int *nmodelpts, double *thetas_in, int *nthetas_in,
                    int *cov_fn_index_in, int *regression_order_in)
{
	mc_free(&emuMCData);
	mc_setup(&emuMCData, xmodel_in, *nparams_in, training_in, *nmodelpts, thetas_in, *nthetas_in, *cov_fn_index_in,
	         *regression_order_in);
}

void callEmulateMC(double *point_in, double *mean_out, double *var_out)
{
	if (!emuMCData.emu) { fprintf(stderr, "callEmulateMC: setupEmulateMC has not been called\n"); gpemu_host_exit(EXIT_FAILURE); }   /* rbind.c:416-419 asserts */
	const int d = emuMCData.model->options->nparams;
	gsl_vector *pt = gsl_vector_alloc(d);
	for (int i = 0; i < d; i++) gsl_vector_set(pt, i, point_in[i]);
	emulate_point(emuMCData.emu, pt, mean_out, var_out);
	gsl_vector_free(pt);
}

void freeEmulateMC(void) { mc_free(&emuMCData); }

static int emuMCDataMulti_n;

/* rbind.c:483-527: training_in is (nmodelpts x nydims), thetas_in (nydims x nthetas), both flattened column by column */
void setupEmulateMCMulti(double *xmodel_in, int *nparams_in, double *training_in, int *nydims_in, int *nmodelpts_in,
                         double *thetas_in, int *nthetas_in, int *cov_fn_index_in, int *regression_order_in)
{
	const int nydims = *nydims_in, N = *nmodelpts_in, nthetas = *nthetas_in;
	if (emuMCDataMulti) { int n = emuMCDataMulti_n; freeEmulateMCMulti(&n); }
	emuMCDataMulti = (struct emulateMCData *)calloc((size_t)nydims, sizeof *emuMCDataMulti);
	emuMCDataMulti_n = nydims;
	double *th = (double *)malloc(sizeof(double) * (size_t)nthetas);
	for (int c = 0; c < nydims; c++) {
		for (int i = 0; i < nthetas; i++) th[i] = thetas_in[c + nydims * i];
		mc_setup(&emuMCDataMulti[c], xmodel_in, *nparams_in, training_in + (size_t)N * c, N, th, nthetas, *cov_fn_index_in,
		         *regression_order_in);
	}
	free(th);
}

/* rbind.c:535-576; the nydims queries are all enqueued before the first is collected */
void callEmulateMCMulti(double *point_in, int *nydims_in, double *final_mean, double *final_var)
{
	const int nydims = *nydims_in;
	if (!emuMCDataMulti || nydims > emuMCDataMulti_n) {
		fprintf(stderr, "callEmulateMCMulti: setupEmulateMCMulti has not been called for %d outputs\n", nydims);
		gpemu_host_exit(EXIT_FAILURE);
	}
	const int d = emuMCDataMulti[0].model->options->nparams;
	gsl_matrix *pt = gsl_matrix_alloc(1, d);
	for (int i = 0; i < d; i++) gsl_matrix_set(pt, 0, i, point_in[i]);
	for (int c = 0; c < nydims; c++) emulate_points_enqueue(emuMCDataMulti[c].emu, pt);
	for (int c = 0; c < nydims; c++) emulate_points_collect(emuMCDataMulti[c].emu, 1, &final_mean[c], &final_var[c]);
	gsl_matrix_free(pt);
}

void freeEmulateMCMulti(int *nydims_in)
{
	if (!emuMCDataMulti) return;
	(void)nydims_in;                                    /* the reference trusts it to equal the count given at setup */
	for (int c = 0; c < emuMCDataMulti_n; c++) mc_free(&emuMCDataMulti[c]);
	free(emuMCDataMulti);
	emuMCDataMulti = NULL;
	emuMCDataMulti_n = 0;
}

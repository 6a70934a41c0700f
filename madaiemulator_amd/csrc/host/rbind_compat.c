/*
 * rbind_compat.c -- the stateless libRbind entry points, without R: same flat .C()-style signatures (pointers to
 * scalars, column-major arrays as R passes them).
 *   callEvalLhoodList  libRbind/rbind.c:626-724   a ready-made batch of independent likelihood evaluations
 *   callEstimate       libRbind/rbind.c:35-98     estimate_thetas on flat arrays
 *   callEmulateAtList  libRbind/rbind.c:121-190   mean / variance at a list of points for given thetas
 *   callEmulateAtPt    libRbind/rbind.c:214-290   ... at one point
 * (setupEmulateMC / callEmulateMC keep a process-global emulator between calls and are not mirrored.)
 * The rows of pointList are independent evaluations of one model: they go through evalFnMultiList, i.e. through
 * lock-step batches of GPU factorisations (gpemu_loglik_batch).
 */
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

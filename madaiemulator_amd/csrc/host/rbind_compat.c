/*
 * rbind_compat.c -- the one libRbind entry point that is a ready-made batch of independent
 * likelihood evaluations (libRbind/rbind.c:626-724 callEvalLhoodList), without R: same flat
 * .C()-style signature (pointers to scalars, column-major arrays as R passes them).
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

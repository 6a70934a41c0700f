/*
 * modelstruct.c -- scalar-GP containers, search ranges and the per-model section of the
 * MODEL_SNAPSHOT_FILE (modelstruct.c:188-467, optstruct.c:142-250 of the reference).
 * The snapshot grammar is reproduced field for field: "%d\n" ints,
 * "%.17lf " reals; the loader takes any whitespace layout.
 */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "libemu.h"

/* modelstruct.c:188-213: smallest |x_{j+1,k} - x_{j,k}| over consecutive rows, floored at 1e-5 */
gsl_vector *fill_sample_scales_vec(gsl_matrix *xmodel)
{
	if (xmodel == NULL) return NULL;
	const int n = (int)xmodel->size1, d = (int)xmodel->size2;
	gsl_vector *scales = gsl_vector_alloc(d);
	for (int k = 0; k < d; k++) {
		double lo = (n > 1) ? fabs(gsl_matrix_get(xmodel, 1, k) - gsl_matrix_get(xmodel, 0, k)) : 0.0;
		for (int j = 1; j < n - 1; j++) {
			const double v = fabs(gsl_matrix_get(xmodel, j + 1, k) - gsl_matrix_get(xmodel, j, k));
			if (v < lo) lo = v;
		}
		if (lo < 1.0e-5) lo = 1.0e-5;
		gsl_vector_set(scales, k, lo);
	}
	return scales;
}

/* modelstruct.c:220-259: the three function pointers of the model (the process-wide twins of the
 * reference's headers are gone: everything here dispatches on the modelstruct) */
void set_global_ptrs(modelstruct *model)
{
	switch (model->options->regression_order) {
	case 1: model->makeHVector = &makeHVector_linear; break;
	case 2: model->makeHVector = &makeHVector_quadratic; break;
	case 3: model->makeHVector = &makeHVector_cubic; break;
	default: model->makeHVector = &makeHVector_trivial;
	}
	switch (model->options->cov_fn_index) {
	case MATERN32:
		model->covariance_fn = &covariance_fn_matern_three;
		model->makeGradMatLength = &derivative_l_matern_three;
		break;
	case MATERN52:
		model->covariance_fn = &covariance_fn_matern_five;
		model->makeGradMatLength = &derivative_l_matern_five;
		break;
	default:
		model->covariance_fn = &covariance_fn_gaussian;
		model->makeGradMatLength = &derivative_l_gauss;
	}
	/* the reference sets its process-wide pointers at the same place (modelstruct.c:222-262) */
	covariance_fn = model->covariance_fn;
	makeHVector = model->makeHVector;
	makeGradMatLength = model->makeGradMatLength;
}

/* optstruct.c:142-250 */
void setup_optimization_ranges(optstruct *options, modelstruct *the_model)
{
	const double rangeMinLog = 0.0001, rangeMinNugget = -5.0, rangeMaxNugget = -2.0, bigRANGE = 10.0;
	double rangeMin, rangeMax;
	options->grad_ranges = gsl_matrix_alloc(options->nthetas, 2);
	if (options->cov_fn_index == POWEREXPCOVFN) { rangeMin = rangeMinLog; rangeMax = 5; }
	else { rangeMin = 0; rangeMax = bigRANGE; }
	gsl_matrix_set(options->grad_ranges, 0, 0, rangeMinLog);
	gsl_matrix_set(options->grad_ranges, 0, 1, rangeMax);
	gsl_matrix_set(options->grad_ranges, 1, 0, rangeMinNugget);
	gsl_matrix_set(options->grad_ranges, 1, 1, rangeMaxNugget);
	for (int i = 2; i < options->nthetas; i++) {
		if (options->use_data_scales) {
			const double s = gsl_vector_get(the_model->sample_scales, i - 2);
			if (options->cov_fn_index == POWEREXPCOVFN) {
				rangeMin = 0.5 * log(s);
				rangeMax = log(25 * exp(rangeMin));
			} else {
				rangeMin = 0.5 * s;                       /* rangeMax stays at bigRANGE */
			}
			if (rangeMin > rangeMax) {
				fprintf(stderr, "#ranges failed\n");
				printf("# %d ranges: %lf %lf\n", i, rangeMin, rangeMax);
				printf("# sampleScale: %lf\n", s);
				gpemu_host_exit(EXIT_FAILURE);
			}
		}
		gsl_matrix_set(options->grad_ranges, i, 0, rangeMin);
		gsl_matrix_set(options->grad_ranges, i, 1, rangeMax);
	}
	if (options->fixed_nugget_mode == 1) {
		const double leeway = 0.20 * options->fixed_nugget;
		gsl_matrix_set(options->grad_ranges, 1, 0, rangeMinNugget);
		gsl_matrix_set(options->grad_ranges, 1, 1, options->fixed_nugget + leeway);
		printf("# (reset) %d ranges: %lf %lf (nugget)\n", 1, gsl_matrix_get(options->grad_ranges, 1, 0),
		       gsl_matrix_get(options->grad_ranges, 1, 1));
	}
	printf("# grad ranges (logged):\n");
	for (int i = 0; i < options->nthetas; i++) {
		const double low = gsl_matrix_get(options->grad_ranges, i, 0), high = gsl_matrix_get(options->grad_ranges, i, 1);
		if (i == 0) printf("# %d ranges: %lf %lf (scale)\n", i, low, high);
		if (i == 1) printf("# %d ranges: %lf %lf (nugget)\n", i, low, high);
		else if (i > 1) printf("# %d ranges: %lf %lf\n", i, low, high);
	}
}

/* modelstruct.c:282-353 */
modelstruct *alloc_modelstruct_2(gsl_matrix *xmodel, gsl_vector *training_vector, int cov_fn_index, int regression_order)
{
	assert(training_vector->size == xmodel->size1);
	assert(training_vector->size > 0);
	assert(xmodel->size2 > 0);
	const int n = (int)xmodel->size1, d = (int)xmodel->size2;
	if (regression_order < 0 || regression_order > 3) regression_order = 0;
	int nthetas;
	if (cov_fn_index == MATERN32 || cov_fn_index == MATERN52) nthetas = 3;
	else { cov_fn_index = POWEREXPCOVFN; nthetas = d + 2; }

	modelstruct *model = (modelstruct *)malloc(sizeof(modelstruct));
	optstruct *o = (optstruct *)malloc(sizeof(optstruct));
	model->options = o;
	o->nparams = d; o->nmodel_points = n; o->nthetas = nthetas;
	o->cov_fn_index = cov_fn_index; o->regression_order = regression_order;
	o->nregression_fns = 1 + regression_order * d;
	o->nemulate_points = 0; o->use_data_scales = 1; o->fixed_nugget_mode = 0; o->fixed_nugget = 0;
	set_global_ptrs(model);
	model->xmodel = gsl_matrix_alloc(n, d);
	gsl_matrix_memcpy(model->xmodel, xmodel);
	model->training_vector = training_vector;           /* borrowed, as in the reference (:340-341) */
	model->thetas = gsl_vector_alloc(nthetas);
	model->sample_scales = fill_sample_scales_vec(model->xmodel);
	setup_optimization_ranges(o, model);
	return model;
}

void free_modelstruct_2(modelstruct *model)
{
	gsl_vector_free(model->thetas);
	gsl_vector_free(model->sample_scales);
	gsl_matrix_free(model->options->grad_ranges);
	free(model->options);
	free(model);
}

/* modelstruct.c:375-409 */
void dump_modelstruct_2(FILE *fptr, modelstruct *m)
{
	const optstruct *o = m->options;
	fprintf(fptr, "%d\n", o->nthetas);
	fprintf(fptr, "%d\n", o->nparams);
	fprintf(fptr, "%d\n", o->nmodel_points);
	fprintf(fptr, "%d\n", o->nemulate_points);
	fprintf(fptr, "%d\n", o->regression_order);
	fprintf(fptr, "%d\n", o->nregression_fns);
	fprintf(fptr, "%d\n", o->fixed_nugget_mode);
	fprintf(fptr, "%.17lf\n", o->fixed_nugget);
	fprintf(fptr, "%d\n", o->cov_fn_index);
	fprintf(fptr, "%d\n", o->use_data_scales);
	for (int i = 0; i < o->nthetas; i++)
		fprintf(fptr, "%.17lf %.17lf\n", gsl_matrix_get(o->grad_ranges, i, 0), gsl_matrix_get(o->grad_ranges, i, 1));
	for (int i = 0; i < o->nmodel_points; i++) {
		for (int j = 0; j < o->nparams; j++) fprintf(fptr, "%.17lf ", gsl_matrix_get(m->xmodel, i, j));
		fprintf(fptr, "\n");
	}
	for (int i = 0; i < o->nmodel_points; i++) fprintf(fptr, "%.17lf ", gsl_vector_get(m->training_vector, i));
	fprintf(fptr, "\n");
	for (int i = 0; i < o->nthetas; i++) fprintf(fptr, "%.17lf ", gsl_vector_get(m->thetas, i));
	fprintf(fptr, "\n");
	for (int i = 0; i < o->nparams; i++) fprintf(fptr, "%.17lf ", gsl_vector_get(m->sample_scales, i));
	fprintf(fptr, "\n");
}

static int rd_int(FILE *f) { int v = 0; if (fscanf(f, "%d%*c", &v) != 1) { fprintf(stderr, "snapshot: read error\n"); gpemu_host_exit(EXIT_FAILURE); } return v; }
static double rd_dbl(FILE *f) { double v = 0; if (fscanf(f, "%lf%*c", &v) != 1) { fprintf(stderr, "snapshot: read error\n"); gpemu_host_exit(EXIT_FAILURE); } return v; }

/* modelstruct.c:419-467 */
modelstruct *load_modelstruct_2(FILE *fptr)
{
	modelstruct *m = (modelstruct *)malloc(sizeof(modelstruct));
	optstruct *o = (optstruct *)malloc(sizeof(optstruct));
	m->options = o;
	o->nthetas = rd_int(fptr);
	o->nparams = rd_int(fptr);
	o->nmodel_points = rd_int(fptr);
	o->nemulate_points = rd_int(fptr);
	o->regression_order = rd_int(fptr);
	o->nregression_fns = rd_int(fptr);
	o->fixed_nugget_mode = rd_int(fptr);
	o->fixed_nugget = rd_dbl(fptr);
	o->cov_fn_index = rd_int(fptr);
	o->use_data_scales = rd_int(fptr);
	o->grad_ranges = gsl_matrix_alloc(o->nthetas, 2);
	for (int i = 0; i < o->nthetas; i++) {
		gsl_matrix_set(o->grad_ranges, i, 0, rd_dbl(fptr));
		gsl_matrix_set(o->grad_ranges, i, 1, rd_dbl(fptr));
	}
	m->xmodel = gsl_matrix_alloc(o->nmodel_points, o->nparams);
	for (int i = 0; i < o->nmodel_points; i++)
		for (int j = 0; j < o->nparams; j++) gsl_matrix_set(m->xmodel, i, j, rd_dbl(fptr));
	m->training_vector = gsl_vector_alloc(o->nmodel_points);
	for (int i = 0; i < o->nmodel_points; i++) gsl_vector_set(m->training_vector, i, rd_dbl(fptr));
	m->thetas = gsl_vector_alloc(o->nthetas);
	for (int i = 0; i < o->nthetas; i++) gsl_vector_set(m->thetas, i, rd_dbl(fptr));
	m->sample_scales = gsl_vector_alloc(o->nparams);
	for (int i = 0; i < o->nparams; i++) gsl_vector_set(m->sample_scales, i, rd_dbl(fptr));
	set_global_ptrs(m);
	return m;
}

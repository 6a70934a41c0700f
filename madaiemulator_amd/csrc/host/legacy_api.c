/*
 * legacy_api.c -- the remaining entry points of the reference's INSTALLED headers (CMakeLists.txt:99,
 * src/CMakeLists.txt:50-51), so that a consumer built against those headers links libEmuMI in place of libEmu:
 *
 *   optstruct.h:90-96        free_optstruct, copy_optstruct, dump_optstruct, load_optstruct, setup_cov_fn, setup_regression
 *   modelstruct.h            alloc_modelstruct, free_modelstruct, copy_modelstruct, fill_modelstruct, dump_modelstruct,
 *                            load_modelstruct (the older, optstruct-sized form beside alloc_modelstruct_2 and friends)
 *   resultstruct.h           alloc_resultstruct, free_resultstruct, copy_resultstruct, fill_resultstruct
 *   libEmu/emulate-fns.h     emulateAtPoint, emulateAtPointList, emulateQuick, emulate_ith_location, emulate_model_results
 *   libEmu/emulator.h        print_matrix, initialise_new_x
 *   libEmu/estimate_threaded.h  setup_params, fprintPt, estimate_thread_function
 *
 * Callers in the reference: libRbind (src/libRbind/rbind.c:61-62, 84-87, 689-691), estimate_threaded.c:61.  None of them is
 * on the hot path; what computes below goes through the same device entries as everything else in this layer (one
 * factorisation per call where the reference factors per call, emulator_struct underneath).
 * Not provided: covariance_fn_gaussian_exact -- declared in libEmu/emulator.h:23, defined nowhere in the reference.
 */
#include <assert.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libemu.h"

/* ---------------------------------------------------------------- optstruct.h */

/* optstruct.c:8-10: the struct itself belongs to the caller */
void free_optstruct(optstruct *opts) { gsl_matrix_free(opts->grad_ranges); }

/* optstruct.c:12-31: every scalar field, and a grad_ranges matrix of dst's own (allocated here) */
void copy_optstruct(optstruct *dst, optstruct *src)
{
	gsl_matrix *ranges = gsl_matrix_alloc(src->nthetas, 2);
	gsl_matrix_memcpy(ranges, src->grad_ranges);
	*dst = *src;
	dst->grad_ranges = ranges;
}

/* optstruct.c:38-67: nregression_fns from the order, and the process-wide makeHVector pointer */
void setup_regression(optstruct *opts)
{
	assert(opts->regression_order < 4 && opts->regression_order > -1);
	assert(opts->nparams > 0);
	static void (*const basis[4])(gsl_vector *, gsl_vector *, int) = {&makeHVector_trivial, &makeHVector_linear, &makeHVector_quadratic,
	                                                                  &makeHVector_cubic};
	opts->nregression_fns = 1 + opts->regression_order * opts->nparams;
	makeHVector = basis[opts->regression_order];
}

/* optstruct.c:81-119: the process-wide covariance / derivative pointers, nthetas forced to what the function takes */
void setup_cov_fn(optstruct *options)
{
	int want;
	switch (options->cov_fn_index) {
	case MATERN32:
		covariance_fn = &covariance_fn_matern_three; makeGradMatLength = &derivative_l_matern_three; want = 3;
		break;
	case MATERN52:
		covariance_fn = &covariance_fn_matern_five; makeGradMatLength = &derivative_l_matern_five; want = 3;
		break;
	case POWEREXPCOVFN:
		covariance_fn = &covariance_fn_gaussian; makeGradMatLength = &derivative_l_gauss; want = options->nparams + 2;
		break;
	default:
		printf("err: cov_fn_index set to unsupported value %d\n", options->cov_fn_index);
		gpemu_host_exit(1);
	}
	if (options->nthetas != want)
		fprintf(stderr, "# (warn) setup_cov_fn has changed nthetas from %d, potential memory errors\n", options->nthetas);
	options->nthetas = want;
}

/* optstruct.c:257-274 */
void dump_optstruct(FILE *fptr, optstruct *opts)
{
	fprintf(fptr, "%d\n%d\n%d\n%d\n%d\n%d\n%d\n", opts->nthetas, opts->nparams, opts->nmodel_points, opts->nemulate_points,
	        opts->regression_order, opts->nregression_fns, opts->fixed_nugget_mode);
	fprintf(fptr, "%lf\n", opts->fixed_nugget);
	fprintf(fptr, "%d\n%d\n", opts->cov_fn_index, opts->use_data_scales);
	for (int i = 0; i < opts->nthetas; i++)
		fprintf(fptr, "%lf\t%lf\n", gsl_matrix_get(opts->grad_ranges, i, 0), gsl_matrix_get(opts->grad_ranges, i, 1));
}

/* optstruct.c:280-305.  The reference reads nthetas TWICE (:283-284) although dump_optstruct writes it once, so its own
 * load(dump(x)) shifts every field by one; here the fields are read in the order dump_optstruct writes them. */
void load_optstruct(FILE *fptr, optstruct *opts)
{
	int ok = fscanf(fptr, "%d %d %d %d %d %d %d", &opts->nthetas, &opts->nparams, &opts->nmodel_points, &opts->nemulate_points,
	                &opts->regression_order, &opts->nregression_fns, &opts->fixed_nugget_mode) == 7;
	ok = ok && fscanf(fptr, "%lf %d %d", &opts->fixed_nugget, &opts->cov_fn_index, &opts->use_data_scales) == 3;
	if (!ok || opts->nthetas < 1) { fprintf(stderr, "load_optstruct: read error\n"); gpemu_host_exit(EXIT_FAILURE); }
	opts->grad_ranges = gsl_matrix_alloc(opts->nthetas, 2);
	for (int i = 0; i < opts->nthetas; i++) {
		double lo, hi;
		if (fscanf(fptr, "%lf %lf", &lo, &hi) != 2) { fprintf(stderr, "load_optstruct: read error\n"); gpemu_host_exit(EXIT_FAILURE); }
		gsl_matrix_set(opts->grad_ranges, i, 0, lo);
		gsl_matrix_set(opts->grad_ranges, i, 1, hi);
	}
}

/* ---------------------------------------------------------------- modelstruct.h (the optstruct-sized form) */

/* modelstruct.c:12-19 */
void alloc_modelstruct(modelstruct *the_model, optstruct *options)
{
	the_model->xmodel = gsl_matrix_alloc(options->nmodel_points, options->nparams);
	the_model->training_vector = gsl_vector_alloc(options->nmodel_points);
	the_model->thetas = gsl_vector_alloc(options->nthetas);
	the_model->sample_scales = gsl_vector_alloc(options->nparams);
	the_model->options = NULL;
}

/* modelstruct.c:24-30: the four containers and the options struct (not its grad_ranges: free_optstruct does that) */
void free_modelstruct(modelstruct *the_model)
{
	gsl_matrix_free(the_model->xmodel);
	gsl_vector_free(the_model->training_vector);
	gsl_vector_free(the_model->thetas);
	gsl_vector_free(the_model->sample_scales);
	free(the_model->options);
}

/* modelstruct.c:35-49: dst's containers exist already (alloc_modelstruct); options are deep-copied when src has them */
void copy_modelstruct(modelstruct *dst, modelstruct *src)
{
	gsl_matrix_memcpy(dst->xmodel, src->xmodel);
	gsl_vector_memcpy(dst->training_vector, src->training_vector);
	gsl_vector_memcpy(dst->thetas, src->thetas);
	gsl_vector_memcpy(dst->sample_scales, src->sample_scales);
	if (src->options != NULL) {
		dst->options = (optstruct *)malloc(sizeof(optstruct));
		copy_optstruct(dst->options, src->options);
	}
	dst->makeHVector = src->makeHVector;
	dst->covariance_fn = src->covariance_fn;
	dst->makeGradMatLength = src->makeGradMatLength;
}

/* modelstruct.c:57-82: "%lf " fields -- design rows on lines, the three vectors behind each other */
void dump_modelstruct(FILE *fptr, modelstruct *the_model, optstruct *opts)
{
	for (int i = 0; i < opts->nmodel_points; i++) {
		for (int j = 0; j < opts->nparams; j++) fprintf(fptr, "%lf ", gsl_matrix_get(the_model->xmodel, i, j));
		fprintf(fptr, "\n");
	}
	for (int i = 0; i < opts->nmodel_points; i++) fprintf(fptr, "%lf ", gsl_vector_get(the_model->training_vector, i));
	for (int i = 0; i < opts->nthetas; i++) fprintf(fptr, "%lf ", gsl_vector_get(the_model->thetas, i));
	for (int i = 0; i < opts->nparams; i++) fprintf(fptr, "%lf ", gsl_vector_get(the_model->sample_scales, i));
}

static double next_double(FILE *fptr, const char *who)
{
	double v;
	if (fscanf(fptr, "%lf ", &v) != 1) { fprintf(stderr, "%s: read error\n", who); gpemu_host_exit(EXIT_FAILURE); }
	return v;
}

/* modelstruct.c:88-125: allocates the containers from the sizes in opts, then reads what dump_modelstruct wrote */
void load_modelstruct(FILE *fptr, modelstruct *the_model, optstruct *opts)
{
	alloc_modelstruct(the_model, opts);
	for (int i = 0; i < opts->nmodel_points; i++)
		for (int j = 0; j < opts->nparams; j++) gsl_matrix_set(the_model->xmodel, i, j, next_double(fptr, "load_modelstruct"));
	for (int i = 0; i < opts->nmodel_points; i++) gsl_vector_set(the_model->training_vector, i, next_double(fptr, "load_modelstruct"));
	for (int i = 0; i < opts->nthetas; i++) gsl_vector_set(the_model->thetas, i, next_double(fptr, "load_modelstruct"));
	for (int i = 0; i < opts->nparams; i++) gsl_vector_set(the_model->sample_scales, i, next_double(fptr, "load_modelstruct"));
}

/* modelstruct.c:133-181: line i of input_data = nparams design values and the training value, blank or tab separated (the
 * lines are cut up in place, as strtok does there); sample_scales = the smallest |x_{j+1,k} - x_{j,k}| over consecutive rows,
 * WITHOUT the 1e-5 floor of fill_sample_scales_vec */
void fill_modelstruct(modelstruct *the_model, optstruct *options, char **input_data)
{
	const int n = options->nmodel_points, d = options->nparams;
	for (int i = 0; i < n; i++) {
		char *save = NULL, *tok = strtok_r(input_data[i], "\t ", &save);
		for (int j = 0; j <= d; j++) {
			double v = 0.0;
			assert(tok != NULL);
			sscanf(tok, "%lg", &v);
			if (j < d) gsl_matrix_set(the_model->xmodel, i, j, v);
			else gsl_vector_set(the_model->training_vector, i, v);
			tok = strtok_r(NULL, "\t ", &save);
		}
	}
	for (int k = 0; k < d; k++) {
		double lo = HUGE_VAL, sum = 0.0;
		for (int j = 0; j + 1 < n; j++) {
			const double dx = fabs(gsl_matrix_get(the_model->xmodel, j + 1, k) - gsl_matrix_get(the_model->xmodel, j, k));
			if (dx < lo) lo = dx;
			sum += dx;
		}
		gsl_vector_set(the_model->sample_scales, k, lo);
		fprintf(stderr, "# param %d min-value %lf average %lf\n", k, lo, n > 1 ? sum / (n - 1) : 0.0);
	}
}

/* ---------------------------------------------------------------- resultstruct.h */

void alloc_resultstruct(resultstruct *res, optstruct *opts)
{
	res->new_x = gsl_matrix_alloc(opts->nemulate_points, opts->nparams);
	res->emulated_mean = gsl_vector_alloc(opts->nemulate_points);
	res->emulated_var = gsl_vector_alloc(opts->nemulate_points);
	res->options = opts;
}

void free_resultstruct(resultstruct *res)
{
	gsl_matrix_free(res->new_x);
	gsl_vector_free(res->emulated_mean);
	gsl_vector_free(res->emulated_var);
}

void copy_resultstruct(resultstruct *dst, resultstruct *src)
{
	gsl_matrix_memcpy(dst->new_x, src->new_x);
	gsl_vector_memcpy(dst->emulated_mean, src->emulated_mean);
	gsl_vector_memcpy(dst->emulated_var, src->emulated_var);
	dst->options = src->options;
}

/* resultstruct.c:45-68: line i of input_data = the nparams coordinates of query point i (cut up in place) */
void fill_resultstruct(resultstruct *res, optstruct *options, char **input_data)
{
	for (int i = 0; i < options->nemulate_points; i++) {
		char *save = NULL, *tok = strtok_r(input_data[i], "\t ", &save);
		for (int j = 0; j < options->nparams; j++) {
			double v = 0.0;
			assert(tok != NULL);
			printf("%s\n", tok);
			sscanf(tok, "%lg", &v);
			gsl_matrix_set(res->new_x, i, j, v);
			tok = strtok_r(NULL, "\t ", &save);
		}
	}
	fprintf(stderr, "fill_resultsruct: matrix: %d x %d\n", options->nemulate_points, options->nparams);
	print_matrix(res->new_x, options->nemulate_points, options->nparams);
}

/* ---------------------------------------------------------------- libEmu/emulator.h */

/* emulator.c:42-52: nx rows of ny values on stderr */
void print_matrix(gsl_matrix *m, int nx, int ny)
{
	for (int i = 0; i < nx; i++) {
		for (int j = 0; j < ny; j++) fprintf(stderr, "%g ", gsl_matrix_get(m, i, j));
		fprintf(stderr, "\n");
	}
}

/* emulator.c:793-817: a regular lattice of query points on [emulate_min, emulate_max), 1 or 2 parameters only */
void initialise_new_x(gsl_matrix *new_x, int nparams, int nemulate_points, double emulate_min, double emulate_max)
{
	if (nparams == 1) {
		const double step = (emulate_max - emulate_min) / ((double)nemulate_points);
		for (int i = 0; i < nemulate_points; i++) gsl_matrix_set(new_x, i, 0, step * ((double)i) + emulate_min);
	} else if (nparams == 2) {
		const int side = (int)floor(sqrt(nemulate_points));
		const double step = (emulate_max - emulate_min) / ((double)side);
		for (int i = 0; i < side; i++)
			for (int j = 0; j < side; j++) {
				gsl_matrix_set(new_x, i * side + j, 0, step * ((double)i) + emulate_min);
				gsl_matrix_set(new_x, i * side + j, 1, step * ((double)j) + emulate_min);
			}
	} else {
		fprintf(stderr, "oops there's no support for %d'd problems yet!\n", nparams);
	}
}

/* ---------------------------------------------------------------- libEmu/emulate-fns.h
 * The reference builds C, its inverse, H and beta on every call (emulate-fns.c:13-265) and then walks the points one at a
 * time through makeKVector / makeEmulatedMean / makeEmulatedVariance.  Here a call makes one emulator_struct (one device
 * factorisation: alloc_emulator_struct) and the points go through the batched sweep (emulate_points). */

/* the model as alloc_emulator_struct wants it: the caller's options and the process-wide function pointers these
 * entries use in the reference (makeCovMatrix / makeHMatrix / makeKVector of emulator.h, regression.h) */
static modelstruct view_with_globals(modelstruct *the_model, optstruct *options, int use_model_cov)
{
	modelstruct v = *the_model;
	v.options = options;
	v.makeHVector = makeHVector ? makeHVector : the_model->makeHVector;
	if (!use_model_cov || !v.covariance_fn) v.covariance_fn = covariance_fn ? covariance_fn : the_model->covariance_fn;
	v.makeGradMatLength = makeGradMatLength ? makeGradMatLength : the_model->makeGradMatLength;
	if (!v.makeHVector || !v.covariance_fn) { fprintf(stderr, "emulate-fns: set_global_ptrs / setup_cov_fn / setup_regression have not been called\n"); gpemu_host_exit(EXIT_FAILURE); }
	return v;
}

static void emulate_list(modelstruct *the_model, optstruct *options, gsl_matrix *points, int npoints, double *mean, double *var,
                         int use_model_cov)
{
	modelstruct v = view_with_globals(the_model, options, use_model_cov);
	emulator_struct *e = gpemu_host_alloc_emulator(&v, 0);     /* nobody reads the struct's host copy of C^-1 here */
	gsl_matrix rows = *points;
	rows.size1 = (size_t)npoints;
	emulate_points(e, &rows, mean, var);
	free_emulator_struct(e);
}

/* emulate-fns.c:75-130: options->nemulate_points rows of point_list (the model's own covariance function, :107) */
void emulateAtPointList(modelstruct *the_model, gsl_matrix *point_list, optstruct *options, double *the_mean, double *the_variance)
{
	emulate_list(the_model, options, point_list, options->nemulate_points, the_mean, the_variance, 1);
}

/* emulate-fns.c:138-190 */
void emulateAtPoint(modelstruct *the_model, gsl_vector *the_point, optstruct *options, double *the_mean, double *the_variance)
{
	gsl_matrix one;
	double *q = (double *)malloc(sizeof(double) * (size_t)options->nparams);
	for (int k = 0; k < options->nparams; k++) q[k] = gsl_vector_get(the_point, k);
	one.size1 = 1; one.size2 = (size_t)options->nparams; one.tda = one.size2; one.data = q; one.block = NULL; one.owner = 0;
	emulate_list(the_model, options, &one, 1, the_mean, the_variance, 0);
	free(q);
}

/* emulate-fns.c:197-224: the caller's own C^-1, H and beta (host matrices): the host-matrix entries of lowlevel.c */
void emulateQuick(modelstruct *the_model, gsl_vector *the_point, optstruct *options, double *mean_out, double *var_out,
                  gsl_matrix *h_matrix, gsl_matrix *cinverse, gsl_vector *beta_vector)
{
	gsl_vector *kplus = gsl_vector_alloc(options->nmodel_points), *h_vector = gsl_vector_alloc(options->nregression_fns);
	makeKVector(kplus, the_model->xmodel, the_point, the_model->thetas, options->nmodel_points, options->nthetas, options->nparams);
	makeHVector(h_vector, the_point, options->nparams);
	*mean_out = makeEmulatedMean(cinverse, the_model->training_vector, kplus, h_vector, h_matrix, beta_vector, options->nmodel_points);
	const double kappa = covariance_fn(the_point, the_point, the_model->thetas, options->nthetas, options->nparams);
	*var_out = makeEmulatedVariance(cinverse, kplus, h_vector, h_matrix, kappa, options->nmodel_points, options->nregression_fns);
	gsl_vector_free(kplus);
	gsl_vector_free(h_vector);
}

/* emulate-fns.c:230-262: row i of results->new_x with the caller's C^-1, H and beta */
void emulate_ith_location(modelstruct *the_model, optstruct *options, resultstruct *results, int i, gsl_matrix *h_matrix,
                          gsl_matrix *cinverse, gsl_vector *beta_vector)
{
	gsl_vector_view row = gsl_matrix_row(results->new_x, (size_t)i);
	double m, v;
	emulateQuick(the_model, &row.vector, options, &m, &v, h_matrix, cinverse, beta_vector);
	gsl_vector_set(results->emulated_mean, i, m);
	gsl_vector_set(results->emulated_var, i, v);
}

/* emulate-fns.c:13-62: the points of results->new_x; prints the regression coefficients and the first coordinates as the
 * reference does */
void emulate_model_results(modelstruct *the_model, optstruct *options, resultstruct *results)
{
	const int np = options->nemulate_points;
	modelstruct v = view_with_globals(the_model, options, 0);
	emulator_struct *e = gpemu_host_alloc_emulator(&v, 0);
	fprintf(stderr, "regression cpts: ");
	for (int a = 0; a < options->nregression_fns; a++) fprintf(stderr, "%g ", gsl_vector_get(e->beta_vector, a));
	fprintf(stderr, "\n");
	for (int i = 0; i < np; i++) printf("%g\n", gsl_matrix_get(results->new_x, i, 0));
	double *m = (double *)malloc(sizeof(double) * (size_t)np), *s = (double *)malloc(sizeof(double) * (size_t)np);
	gsl_matrix rows = *results->new_x;
	rows.size1 = (size_t)np;
	emulate_points(e, &rows, m, s);
	for (int i = 0; i < np; i++) { gsl_vector_set(results->emulated_mean, i, m[i]); gsl_vector_set(results->emulated_var, i, s[i]); }
	free(m); free(s);
	free_emulator_struct(e);
}

/* ---------------------------------------------------------------- libEmu/estimate_threaded.h */

/* estimate_threaded.c:57-68: per-thread deep copies (the caller has allocated the structs the entries point at) */
void setup_params(struct estimate_thetas_params *params_array, modelstruct *the_model, optstruct *options, int nthreads, int max_tries)
{
	for (int i = 0; i < nthreads; i++) {
		copy_optstruct(params_array[i].options, options);
		alloc_modelstruct(params_array[i].the_model, options);
		copy_modelstruct(params_array[i].the_model, the_model);
		params_array[i].max_tries = max_tries;
	}
}

/* estimate_threaded.c:337-345: a thread id as hex bytes */
void fprintPt(FILE *f, pthread_t pt)
{
	const unsigned char *b = (const unsigned char *)(const void *)&pt;
	fprintf(f, "0x");
	for (size_t i = 0; i < sizeof pt; i++) fprintf(f, "%02x", (unsigned)b[i]);
}

/* estimate_threaded.c:239-334 is the body of the reference's pool threads and reads that file's globals (job counter, best
 * thetas): outside estimate_thetas_threaded it is one restart job on the caller's params -- maxWithMultiMin, the thread's own
 * best kept in params->my_best -- which is what a caller who starts it on a thread of its own gets here. */
void *estimate_thread_function(void *args)
{
	struct estimate_thetas_params *params = (struct estimate_thetas_params *)args;
	maxWithMultiMin(params);
	if (params->lhood_current > params->my_best) params->my_best = params->lhood_current;
	printf("# thread: ");
	fprintPt(stdout, pthread_self());
	printf(" is done\n");
	return NULL;
}

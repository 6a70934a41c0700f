/* the helper entry points of the reference's installed headers (csrc/host/legacy_api.c), called the way libRbind and
 * estimate_threaded.c call them (src/libRbind/rbind.c:61-62, 84-87, 689-691; libEmu/estimate_threaded.c:57-68).  No device
 * work: containers, deep copies, dump -> load round trips, the process-wide function pointers.  Exit status 0 = all held. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "libemu.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(int argc, char **argv)
{
	if (argc < 2) return 2;
	const int N = 7, d = 2;
	optstruct o;
	memset(&o, 0, sizeof o);
	o.nparams = d; o.nmodel_points = N; o.nemulate_points = 4; o.regression_order = 2; o.cov_fn_index = POWEREXPCOVFN;
	o.nthetas = 99; o.fixed_nugget = 0.25; o.use_data_scales = 1;
	setup_cov_fn(&o);                                        /* optstruct.c:81-119 */
	CHECK(o.nthetas == d + 2 && covariance_fn == &covariance_fn_gaussian && makeGradMatLength == &derivative_l_gauss);
	setup_regression(&o);                                    /* optstruct.c:38-67 */
	CHECK(o.nregression_fns == 1 + 2 * d && makeHVector == &makeHVector_quadratic);
	optstruct om = o;
	om.cov_fn_index = MATERN52;
	setup_cov_fn(&om);
	CHECK(om.nthetas == 3 && covariance_fn == &covariance_fn_matern_five && makeGradMatLength == &derivative_l_matern_five);
	setup_cov_fn(&o);

	modelstruct m;
	alloc_modelstruct(&m, &o);                               /* modelstruct.c:12-19 */
	CHECK(m.xmodel->size1 == (size_t)N && m.xmodel->size2 == (size_t)d && m.training_vector->size == (size_t)N);
	CHECK(m.thetas->size == (size_t)(d + 2) && m.sample_scales->size == (size_t)d && m.options == NULL);
	char *lines[7];
	char buf[7][64];
	for (int i = 0; i < N; i++) {
		snprintf(buf[i], sizeof buf[i], "%g\t%g %g", 0.1 * i * i, 1.0 - 0.05 * i, sin(0.7 * i));
		lines[i] = buf[i];
	}
	fill_modelstruct(&m, &o, lines);                         /* modelstruct.c:133-181 */
	CHECK(fabs(gsl_matrix_get(m.xmodel, 3, 0) - 0.9) < 1e-15 && fabs(gsl_matrix_get(m.xmodel, 6, 1) - 0.7) < 1e-15);
	CHECK(fabs(gsl_vector_get(m.training_vector, 2) - sin(1.4)) < 1e-5);      /* "%g": six digits */
	CHECK(fabs(gsl_vector_get(m.sample_scales, 0) - 0.1) < 1e-12 && fabs(gsl_vector_get(m.sample_scales, 1) - 0.05) < 1e-12);
	for (int t = 0; t < d + 2; t++) gsl_vector_set(m.thetas, t, 0.5 - t);
	m.options = (optstruct *)malloc(sizeof(optstruct));
	o.grad_ranges = gsl_matrix_alloc(o.nthetas, 2);
	for (int t = 0; t < o.nthetas; t++) { gsl_matrix_set(o.grad_ranges, t, 0, -1.0 - t); gsl_matrix_set(o.grad_ranges, t, 1, 2.5 + t); }
	copy_optstruct(m.options, &o);                           /* optstruct.c:12-31 */
	CHECK(m.options->grad_ranges != o.grad_ranges && gsl_matrix_get(m.options->grad_ranges, 3, 1) == 5.5);
	CHECK(m.options->nregression_fns == o.nregression_fns && m.options->fixed_nugget == 0.25 && m.options->use_data_scales == 1);
	set_global_ptrs(&m);

	/* the per-thread deep copies of estimate_threaded.c:57-68 */
	struct estimate_thetas_params P[2];
	for (int i = 0; i < 2; i++) {
		memset(&P[i], 0, sizeof P[i]);
		P[i].options = (optstruct *)malloc(sizeof(optstruct));
		P[i].the_model = (modelstruct *)malloc(sizeof(modelstruct));
	}
	setup_params(P, &m, &o, 2, 11);
	for (int i = 0; i < 2; i++) {
		CHECK(P[i].max_tries == 11 && P[i].the_model->xmodel != m.xmodel && P[i].the_model->options != m.options);
		CHECK(!memcmp(P[i].the_model->xmodel->data, m.xmodel->data, sizeof(double) * N * d));
		CHECK(gsl_vector_get(P[i].the_model->thetas, 2) == -1.5 && P[i].the_model->covariance_fn == m.covariance_fn);
		CHECK(P[i].the_model->makeHVector == &makeHVector_quadratic && P[i].options->grad_ranges != o.grad_ranges);
		CHECK(gsl_matrix_get(P[i].the_model->options->grad_ranges, 1, 0) == -2.0);
	}
	gsl_matrix_set(P[0].the_model->xmodel, 0, 0, 42.0);      /* a copy, not a view */
	CHECK(gsl_matrix_get(m.xmodel, 0, 0) == 0.0 && gsl_matrix_get(P[1].the_model->xmodel, 0, 0) == 0.0);

	/* dump -> load round trips ("%lf": six decimals) */
	FILE *f = fopen(argv[1], "w");
	dump_optstruct(f, &o);
	dump_modelstruct(f, &m, &o);
	fclose(f);
	optstruct o2;
	modelstruct m2;
	f = fopen(argv[1], "r");
	load_optstruct(f, &o2);
	CHECK(o2.nthetas == o.nthetas && o2.nparams == d && o2.nmodel_points == N && o2.nemulate_points == 4 && o2.regression_order == 2);
	CHECK(o2.nregression_fns == o.nregression_fns && o2.cov_fn_index == POWEREXPCOVFN && o2.use_data_scales == 1 && o2.fixed_nugget == 0.25);
	CHECK(gsl_matrix_get(o2.grad_ranges, 2, 0) == -3.0 && gsl_matrix_get(o2.grad_ranges, 2, 1) == 4.5);
	load_modelstruct(f, &m2, &o2);
	fclose(f);
	for (int i = 0; i < N; i++) {
		CHECK(fabs(gsl_vector_get(m2.training_vector, i) - gsl_vector_get(m.training_vector, i)) < 1e-6);
		for (int k = 0; k < d; k++) CHECK(fabs(gsl_matrix_get(m2.xmodel, i, k) - gsl_matrix_get(m.xmodel, i, k)) < 1e-6);
	}
	CHECK(gsl_vector_get(m2.thetas, 3) == -2.5 && fabs(gsl_vector_get(m2.sample_scales, 1) - 0.05) < 1e-6);

	/* resultstruct + the lattice of query points (emulator.c:793-817) */
	resultstruct r, r2;
	alloc_resultstruct(&r, &o);
	alloc_resultstruct(&r2, &o);
	CHECK(r.new_x->size1 == 4 && r.new_x->size2 == (size_t)d && r.options == &o);
	initialise_new_x(r.new_x, d, 4, 0.0, 1.0);
	CHECK(gsl_matrix_get(r.new_x, 3, 0) == 0.5 && gsl_matrix_get(r.new_x, 3, 1) == 0.5 && gsl_matrix_get(r.new_x, 1, 1) == 0.5);
	gsl_vector_set(r.emulated_mean, 2, 7.0);
	copy_resultstruct(&r2, &r);
	CHECK(gsl_vector_get(r2.emulated_mean, 2) == 7.0 && gsl_matrix_get(r2.new_x, 2, 0) == 0.5);
	gsl_matrix *one = gsl_matrix_alloc(5, 1);
	initialise_new_x(one, 1, 5, 1.0, 2.0);
	CHECK(fabs(gsl_matrix_get(one, 4, 0) - 1.8) < 1e-15);

	free_resultstruct(&r); free_resultstruct(&r2);
	for (int i = 0; i < 2; i++) { free_optstruct(P[i].the_model->options); free_modelstruct(P[i].the_model); free_optstruct(P[i].options); }
	free_optstruct(m.options);
	free_modelstruct(&m);
	free_optstruct(&o);
	printf("legacy api ok\n");
	return 0;
}

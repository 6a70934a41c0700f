/* Drives the host-side libEmu mirror (csrc/host/libemu.h) the way the reference's own callers do
 * (gsl_multimin callbacks on an estimate_thetas_params; alloc_emulator_struct + emulate_point) and prints
 * the results for tests/test_host_api.py to compare with the oracle.
 *
 *   host_api_driver eval  INPUT_MODEL_FILE cov_fn order theta_less_amp...
 *   host_api_driver emu   INPUT_MODEL_FILE cov_fn order QUERY_FILE theta_full...
 *   host_api_driver roundtrip SNAPSHOT_IN SNAPSHOT_OUT
 *   host_api_driver multi SNAPSHOT QUERY_FILE
 *   host_api_driver lowlevel INPUT_MODEL_FILE cov_fn order QUERY_FILE theta_full...   (the host-matrix interface)
 *   host_api_driver rewrite INPUT_MODEL_FILE cov_fn order theta_less_amp...   (the caller rewrites its buffers in place)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "libemu.h"

static int read_model(const char *name, gsl_matrix **x, gsl_matrix **y)
{
	FILE *in = fopen(name, "r");
	int nt, d, n;
	if (!in || fscanf(in, "%d %d %d", &nt, &d, &n) != 3) return 0;
	*x = gsl_matrix_alloc(n, d);
	*y = gsl_matrix_alloc(n, nt);
	for (int i = 0; i < n; i++) for (int j = 0; j < d; j++) if (fscanf(in, "%lf", gsl_matrix_ptr(*x, i, j)) != 1) return 0;
	for (int i = 0; i < n; i++) for (int j = 0; j < nt; j++) if (fscanf(in, "%lf", gsl_matrix_ptr(*y, i, j)) != 1) return 0;
	fclose(in);
	return 1;
}

static gsl_matrix *read_queries(const char *name, int d)
{
	FILE *in = fopen(name, "r");
	double v, *buf = NULL;
	size_t n = 0, cap = 0;
	while (in && fscanf(in, "%lf", &v) == 1) {
		if (n == cap) { cap = cap ? 2 * cap : 1024; buf = (double *)realloc(buf, cap * sizeof(double)); }
		buf[n++] = v;
	}
	if (in) fclose(in);
	gsl_matrix *q = gsl_matrix_alloc(n / d, d);
	memcpy(q->data, buf, (n / d) * d * sizeof(double));
	free(buf);
	return q;
}

int main(int argc, char **argv)
{
	if (argc < 3) return 2;
	if (!strcmp(argv[1], "roundtrip")) {
		FILE *in = fopen(argv[2], "r");
		multi_modelstruct *m = load_multi_modelstruct(in);
		fclose(in);
		FILE *out = fopen(argv[3], "w");
		dump_multi_modelstruct(out, m);
		fclose(out);
		printf("nt %d nr %d N %d d %d\n", m->nt, m->nr, m->nmodel_points, m->nparams);
		return 0;
	}
	if (!strcmp(argv[1], "multi")) {
		FILE *in = fopen(argv[2], "r");
		multi_modelstruct *m = load_multi_modelstruct(in);
		fclose(in);
		multi_emulator *e = alloc_multi_emulator(m);
		gsl_matrix *q = read_queries(argv[3], m->nparams);
		gsl_vector *mean = gsl_vector_alloc(m->nt), *var = gsl_vector_alloc(m->nt), *pt = gsl_vector_alloc(m->nparams);
		for (size_t i = 0; i < q->size1; i++) {
			for (int k = 0; k < m->nparams; k++) gsl_vector_set(pt, k, gsl_matrix_get(q, i, k));
			emulate_point_multi(e, pt, mean, var);
			printf("pred");
			for (int t = 0; t < m->nt; t++) printf(" %.17g %.17g", gsl_vector_get(mean, t), gsl_vector_get(var, t));
			printf("\n");
		}
		return 0;
	}
	gsl_matrix *x, *ymat;
	if (!read_model(argv[2], &x, &ymat)) return 3;
	const int cov = atoi(argv[3]), order = atoi(argv[4]);
	gsl_vector *y = gsl_vector_alloc(x->size1);
	for (size_t i = 0; i < x->size1; i++) gsl_vector_set(y, i, gsl_matrix_get(ymat, i, 0));
	modelstruct *model = alloc_modelstruct_2(x, y, cov, order);
	const int nthetas = model->options->nthetas;

	if (!strcmp(argv[1], "eval")) {
		struct estimate_thetas_params params;
		memset(&params, 0, sizeof params);
		params.options = model->options;
		params.the_model = model;
		gsl_vector *th = gsl_vector_alloc(nthetas - 1), *g = gsl_vector_alloc(nthetas - 1), *g2 = gsl_vector_alloc(nthetas - 1);
		for (int i = 0; i < nthetas - 1; i++) gsl_vector_set(th, i, atof(argv[5 + i]));
		const double f = evalFnMulti(th, &params);
		printf("evalFnMulti %.17g\n", f);
		printf("estimateSigmaFull %.17g\n", estimateSigmaFull(th, &params));
		if (cov == POWEREXPCOVFN) {
			gradFnMulti(th, &params, g);
			printf("gradFnMulti");
			for (int i = 0; i < nthetas - 1; i++) printf(" %.17g", gsl_vector_get(g, i));
			printf("\n");
			double f2;
			evalFnGradMulti(th, &params, &f2, g2);
			printf("evalFnGradMulti %.17g", f2);
			for (int i = 0; i < nthetas - 1; i++) printf(" %.17g", gsl_vector_get(g2, i));
			printf("\n");
		}
		gpemu_host_release(&params);
	} else if (!strcmp(argv[1], "rewrite")) {
		/* the reference re-reads the model on every call (maxmultimin.c:317): a caller may rewrite training_vector->data or
		 * xmodel->data IN PLACE between calls (same pointers) and must get the likelihood of the new data */
		struct estimate_thetas_params params;
		memset(&params, 0, sizeof params);
		params.options = model->options;
		params.the_model = model;
		gsl_vector *th = gsl_vector_alloc(nthetas - 1), *g = gsl_vector_alloc(nthetas - 1);
		for (int i = 0; i < nthetas - 1; i++) gsl_vector_set(th, i, atof(argv[5 + i]));
		printf("step0 %.17g %.17g\n", evalFnMulti(th, &params), estimateSigmaFull(th, &params));
		for (size_t i = 0; i < model->training_vector->size; i++)           /* same buffer, new outputs */
			model->training_vector->data[i * model->training_vector->stride] = 0.5 * model->training_vector->data[i * model->training_vector->stride] + 0.01 * (double)(i % 7);
		printf("step1 %.17g %.17g\n", evalFnMulti(th, &params), estimateSigmaFull(th, &params));
		for (size_t i = 0; i < model->xmodel->size1; i++)                   /* same buffer, new design */
			for (size_t k = 0; k < model->xmodel->size2; k++)
				model->xmodel->data[i * model->xmodel->tda + k] = 0.9 * model->xmodel->data[i * model->xmodel->tda + k] + 0.003 * (double)((i + 3 * k) % 11);
		double f2;
		evalFnGradMulti(th, &params, &f2, g);
		printf("step2 %.17g %.17g\n", f2, estimateSigmaFull(th, &params));
		printf("step2b %.17g\n", evalFnMulti(th, &params));
		{
			long v, vg, c, r, e;
			gpemu_host_eval_stats(&v, &vg, &c, &r, &e);
			printf("evalstats %ld %ld %ld\n", v, vg, c);
		}
		gpemu_host_release(&params);
	} else if (!strcmp(argv[1], "train")) {
		/* estimate_thetas_threaded on the model, then the best thetas and -logL at them (evalFnMulti) */
		setup_optimization_ranges(model->options, model);
		estimate_thetas_threaded(model, model->options);
		printf("thetas");
		for (int i = 0; i < nthetas; i++) printf(" %.17g", gsl_vector_get(model->thetas, i));
		printf("\n");
		struct estimate_thetas_params params;
		memset(&params, 0, sizeof params);
		params.options = model->options;
		params.the_model = model;
		gsl_vector *th = gsl_vector_alloc(nthetas - 1);
		for (int i = 0; i < nthetas - 1; i++) gsl_vector_set(th, i, gsl_vector_get(model->thetas, i + 1));
		printf("neglogl %.17g\n", evalFnMulti(th, &params));
		{
			long runs, conv, noprog, fb;
			double gn;
			gpemu_host_search_stats(&runs, &conv, &noprog, &fb, &gn);
			printf("search %ld %ld %ld %ld %.17g\n", runs, conv, noprog, fb, gn);
			{
				long v, vg, c, r, e;
				gpemu_host_eval_stats(&v, &vg, &c, &r, &e);
				printf("evalstats %ld %ld %ld %ld %ld\n", v, vg, c, r, e);
			}
			if (cov == POWEREXPCOVFN || getenv("GPEMU_EXACT_GRAD")) {
				gsl_vector *g = gsl_vector_alloc(nthetas - 1);
				gradFnMulti(th, &params, g);
				printf("grad_at_best");
				for (int i = 0; i < nthetas - 1; i++) printf(" %.17g", gsl_vector_get(g, i));
				printf("\n");
			}
		}
		gpemu_host_release(&params);
	} else if (!strcmp(argv[1], "lowlevel")) {
		/* the call sequence of the reference's R bindings (libRbind/rbind.c:121-210): N x N matrices in host memory */
		const int N = (int)x->size1, d = (int)x->size2, nreg = model->options->nregression_fns;
		gsl_matrix *q = read_queries(argv[5], d);
		for (int i = 0; i < nthetas; i++) gsl_vector_set(model->thetas, i, atof(argv[6 + i]));
		gsl_matrix *c = gsl_matrix_alloc(N, N), *cinv = gsl_matrix_alloc(N, N), *H = gsl_matrix_alloc(N, nreg);
		makeCovMatrix(c, model->xmodel, model->thetas, N, nthetas, d);      /* global-pointer twin (emulator.c:607) */
		double det = 0.0;
		chol_inverse_cov_matrix(model->options, c, cinv, &det);
		printf("logdet %.17g\n", log(det));
		makeHMatrix(H, model->xmodel, N, d, nreg);                            /* regression.c:77 */
		gsl_vector *beta = gsl_vector_alloc(nreg);
		estimateBeta(beta, H, cinv, model->training_vector, N, nreg);
		printf("beta");
		for (int a = 0; a < nreg; a++) printf(" %.17g", gsl_vector_get(beta, a));
		printf("\n");
		printf("loglik %.17g\n", getLogLikelyhood(cinv, det, model->xmodel, model->training_vector, model->thetas, H, N,
		                                          nthetas, d, nreg, model->makeHVector));
		struct estimate_thetas_params params;
		memset(&params, 0, sizeof params);
		params.options = model->options; params.the_model = model; params.h_matrix = H;
		printf("sigma2 %.17g\n", estimateSigma(cinv, &params));
		{
			/* getGradientCn with dC := C itself: -1/2 trace(C^-1 C) + 1/2 y^T C^-1 C C^-1 y = -N/2 + 1/2 y^T C^-1 y */
			gsl_matrix *c2 = gsl_matrix_alloc(N, N);
			makeCovMatrix(c2, model->xmodel, model->thetas, N, nthetas, d);
			printf("gradcn %.17g\n", getGradientCn(c2, cinv, model->training_vector, N, nthetas));
			gsl_matrix_free(c2);
		}
		gsl_vector *k = gsl_vector_alloc(N), *h = gsl_vector_alloc(nreg), *pt = gsl_vector_alloc(d);
		for (size_t i = 0; i < q->size1; i++) {
			for (int kk = 0; kk < d; kk++) gsl_vector_set(pt, kk, gsl_matrix_get(q, i, kk));
			makeKVector(k, model->xmodel, pt, model->thetas, N, nthetas, d);    /* emulator.c:548 */
			makeHVector(h, pt, d);
			const double kappa = covariance_fn(pt, pt, model->thetas, nthetas, d);
			const double m = makeEmulatedMean(cinv, model->training_vector, k, h, H, beta, N);
			const double v = makeEmulatedVariance(cinv, k, h, H, kappa, N, nreg);
			printf("pred %.17g %.17g\n", m, v);
		}
	} else if (!strcmp(argv[1], "emu")) {
		gsl_matrix *q = read_queries(argv[5], (int)x->size2);
		for (int i = 0; i < nthetas; i++) gsl_vector_set(model->thetas, i, atof(argv[6 + i]));
		emulator_struct *e = alloc_emulator_struct(model);
		printf("beta");
		for (int a = 0; a < e->nregression_fns; a++) printf(" %.17g", gsl_vector_get(e->beta_vector, a));
		printf("\n");
		printf("cinverse00 %.17g %.17g\n", gsl_matrix_get(e->cinverse, 0, 0),
		       gsl_matrix_get(e->cinverse, e->nmodel_points - 1, 0));
		{
			/* the emulator_struct.c:63-118 wrappers */
			gsl_vector *b2 = gsl_vector_alloc(e->nregression_fns), *kv = gsl_vector_alloc(e->nmodel_points);
			gsl_vector_view q0 = gsl_matrix_row(q, 0);
			estimateBeta_es(b2, e);
			printf("beta_es");
			for (int a = 0; a < e->nregression_fns; a++) printf(" %.17g", gsl_vector_get(b2, a));
			printf("\n");
			makeKVector_es(kv, &q0.vector, e);
			printf("kvec_es %.17g %.17g\n", gsl_vector_get(kv, 0), gsl_vector_get(kv, e->nmodel_points - 1));
			gsl_matrix *hm = gsl_matrix_alloc(e->nmodel_points, e->nregression_fns);
			makeHMatrix_es(hm, e);
			printf("hmat_es %.17g\n", gsl_matrix_get(hm, e->nmodel_points - 1, e->nregression_fns - 1));
		}
		gsl_vector *pt = gsl_vector_alloc(x->size2);
		for (size_t i = 0; i < q->size1; i++) {
			double m, v;
			for (size_t k = 0; k < x->size2; k++) gsl_vector_set(pt, k, gsl_matrix_get(q, i, k));
			emulate_point(e, pt, &m, &v);
			printf("pred %.17g %.17g\n", m, v);
		}
		double *mm = (double *)malloc(sizeof(double) * q->size1), *vv = (double *)malloc(sizeof(double) * q->size1);
		emulate_points(e, q, mm, vv);
		for (size_t i = 0; i < q->size1; i++) printf("batch %.17g %.17g\n", mm[i], vv[i]);
		{
			/* libEmu/emulate-fns.h (legacy_api.c): the list form, the one-point form (process-wide pointers), the
			 * caller's-matrices form and the resultstruct form give what emulate_point gives */
			optstruct *o = model->options;
			const int keep = o->nemulate_points;
			o->nemulate_points = (int)q->size1;
			emulateAtPointList(model, q, o, mm, vv);
			for (size_t i = 0; i < q->size1; i++) printf("atlist %.17g %.17g\n", mm[i], vv[i]);
			gsl_vector_view q0 = gsl_matrix_row(q, 0);
			double m1, v1, m2, v2;
			emulateAtPoint(model, &q0.vector, o, &m1, &v1);
			emulateQuick(model, &q0.vector, o, &m2, &v2, e->h_matrix, e->cinverse, e->beta_vector);
			printf("atpoint %.17g %.17g\nquick %.17g %.17g\n", m1, v1, m2, v2);
			resultstruct res;
			alloc_resultstruct(&res, o);
			gsl_matrix_memcpy(res.new_x, q);
			fflush(stdout);
			FILE *keep_out = stdout;
			stdout = stderr;                              /* (emulate_model_results prints the first coordinates on stdout) */
			emulate_model_results(model, o, &res);
			stdout = keep_out;
			for (size_t i = 0; i < q->size1; i++) printf("results %.17g %.17g\n", gsl_vector_get(res.emulated_mean, i), gsl_vector_get(res.emulated_var, i));
			emulate_ith_location(model, o, &res, 1, e->h_matrix, e->cinverse, e->beta_vector);
			printf("ith %.17g %.17g\n", gsl_vector_get(res.emulated_mean, 1), gsl_vector_get(res.emulated_var, 1));
			free_resultstruct(&res);
			o->nemulate_points = keep;
		}
		free_emulator_struct(e);
	}
	return 0;
}

/* Host-side logic of the libEmu mirror that needs no GPU: the mt19937 generator behind gsl_rng_default, the
 * regression basis, the PCA decomposition of a multi-output training set and the snapshot writer / reader.
 *   host_cpu_driver INPUT_MODEL_FILE SNAPSHOT_OUT SNAPSHOT_OUT2 [DERIV_OUT]
 * DERIV_OUT: raw doubles, derivative_l_matern_three then derivative_l_matern_five of the design at thetaLength 0.7
 * (libEmu/emulator.c:401-433, 497-532: the literal sequential recurrence, host code)
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
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

int main(int argc, char **argv)
{
	if (argc < 4) return 2;
	/* mt19937, init_genrand(5489): the generator's published first outputs */
	gsl_rng *r = gsl_rng_alloc(gsl_rng_default);
	gsl_rng_set(r, 5489UL);
	printf("mt");
	for (int i = 0; i < 4; i++) printf(" %lu", gsl_rng_get(r));
	printf("\n");
	gsl_rng_set(r, 5489UL);
	printf("uniform %.17g\n", gsl_rng_uniform(r));
	gsl_rng_free(r);

	gsl_matrix *x, *y;
	if (!read_model(argv[1], &x, &y)) return 3;
	/* regression basis at the first design point, orders 0..3 */
	gsl_vector_view row = gsl_matrix_row(x, 0);
	void (*hv[4])(gsl_vector *, gsl_vector *, int) = {makeHVector_trivial, makeHVector_linear, makeHVector_quadratic,
	                                                  makeHVector_cubic};
	for (int order = 0; order < 4; order++) {
		const int nreg = 1 + order * (int)x->size2;
		gsl_vector *h = gsl_vector_alloc(nreg);
		hv[order](h, &row.vector, (int)x->size2);
		printf("h%d", order);
		for (int i = 0; i < nreg; i++) printf(" %.17g", gsl_vector_get(h, i));
		printf("\n");
		gsl_vector_free(h);
	}
	/* PCA of the outputs + snapshot round trip of the untrained model */
	multi_modelstruct *m = alloc_multimodelstruct(x, y, POWEREXPCOVFN, 1, 0.99);
	printf("nr %d\n", m->nr);
	printf("evals");
	for (int i = 0; i < m->nr; i++) printf(" %.17g", gsl_vector_get(m->pca_evals_r, i));
	printf("\n");
	for (int t = 0; t < m->nt; t++) {
		printf("evec");
		for (int i = 0; i < m->nr; i++) printf(" %.17g", gsl_matrix_get(m->pca_evecs_r, t, i));
		printf("\n");
	}
	printf("z0");
	for (int i = 0; i < m->nr; i++) printf(" %.17g", gsl_matrix_get(m->pca_zmatrix, 0, i));
	printf("\n");
	FILE *out = fopen(argv[2], "w");
	dump_multi_modelstruct(out, m);
	fclose(out);
	FILE *in = fopen(argv[2], "r");
	multi_modelstruct *m2 = load_multi_modelstruct(in);
	fclose(in);
	out = fopen(argv[3], "w");
	dump_multi_modelstruct(out, m2);
	fclose(out);
	printf("loaded nt %d nr %d N %d d %d\n", m2->nt, m2->nr, m2->nmodel_points, m2->nparams);
	if (argc > 4) {
		const int n = (int)x->size1;
		gsl_matrix *dC = gsl_matrix_alloc(n, n);
		FILE *df = fopen(argv[4], "wb");
		if (!df) return 4;
		derivative_l_matern_three(dC, x, 0.7, 2, n, (int)x->size2);
		for (int i = 0; i < n; i++) fwrite(gsl_matrix_ptr(dC, i, 0), sizeof(double), (size_t)n, df);
		derivative_l_matern_five(dC, x, 0.7, 2, n, (int)x->size2);
		for (int i = 0; i < n; i++) fwrite(gsl_matrix_ptr(dC, i, 0), sizeof(double), (size_t)n, df);
		fclose(df);
		gsl_matrix_free(dC);
		printf("deriv %d\n", n);
	}
	return 0;
}

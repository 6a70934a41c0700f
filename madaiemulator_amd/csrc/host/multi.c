/*
 * multi.c -- multi-output models: PCA of the N x t training matrix, nr independent scalar GPs,
 * the MODEL_SNAPSHOT_FILE, and back-projection of predictions
 * (multi_modelstruct.c:38-507, multivar_support.c:20-157 of the reference).
 *
 * gsl_eigen_symmv is replaced by a cyclic Jacobi eigen-solver (nt is the number of outputs, tens
 * at most); eigenvector signs are whatever the solver returns, as with GSL -- only back-projected
 * outputs are comparable between implementations.
 */
#include <assert.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>
#include "libemu.h"
#include "gpemu.h"

double vector_elt_sum(gsl_vector *vec, int nstop)
{
	assert(nstop >= 0);
	assert((unsigned)nstop <= vec->size);
	double sum = 0.0;
	for (int i = 0; i < nstop; i++) sum += gsl_vector_get(vec, i);
	return sum;
}

/* symmetric eigen-decomposition A = V diag(w) V^T by cyclic Jacobi rotations; A is n x n row-major and destroyed */
static void jacobi_eigen(double *A, int n, double *w, double *V)
{
	for (int i = 0; i < n; i++)
		for (int j = 0; j < n; j++) V[i * n + j] = (i == j) ? 1.0 : 0.0;
	for (int sweep = 0; sweep < 100; sweep++) {
		double off = 0.0;
		for (int i = 0; i < n; i++)
			for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
		if (off < 1e-300) break;
		for (int p = 0; p < n - 1; p++)
			for (int q = p + 1; q < n; q++) {
				const double apq = A[p * n + q];
				if (apq == 0.0) continue;
				const double theta = (A[q * n + q] - A[p * n + p]) / (2.0 * apq);
				const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
				const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
				for (int k = 0; k < n; k++) {
					const double akp = A[k * n + p], akq = A[k * n + q];
					A[k * n + p] = c * akp - s * akq;
					A[k * n + q] = s * akp + c * akq;
				}
				for (int k = 0; k < n; k++) {
					const double apk = A[p * n + k], aqk = A[q * n + k];
					A[p * n + k] = c * apk - s * aqk;
					A[q * n + k] = s * apk + c * aqk;
				}
				for (int k = 0; k < n; k++) {
					const double vkp = V[k * n + p], vkq = V[k * n + q];
					V[k * n + p] = c * vkp - s * vkq;
					V[k * n + q] = s * vkp + c * vkq;
				}
			}
	}
	for (int i = 0; i < n; i++) w[i] = A[i * n + i];
}

/* multi_modelstruct.c:172-338 */
void gen_pca_decomp(multi_modelstruct *m, double vfrac)
{
	const int nt = m->nt, N = m->nmodel_points;
	double *ysub = (double *)malloc(sizeof(double) * (size_t)N * nt);
	double *cov = (double *)calloc((size_t)nt * nt, sizeof(double));
	double *w = (double *)malloc(sizeof(double) * (size_t)nt), *V = (double *)malloc(sizeof(double) * (size_t)nt * nt);
	for (int i = 0; i < nt; i++) {
		printf("# y(%d) mean: %lf\n", i, gsl_vector_get(m->training_mean, i));
		for (int j = 0; j < N; j++) ysub[j * nt + i] = gsl_matrix_get(m->training_matrix, j, i) - gsl_vector_get(m->training_mean, i);
	}
	for (int a = 0; a < nt; a++)
		for (int b = 0; b < nt; b++) {
			double s = 0.0;
			for (int j = 0; j < N; j++) s += ysub[j * nt + a] * ysub[j * nt + b];
			cov[a * nt + b] = s * (1.0 / (double)N);
		}
	jacobi_eigen(cov, nt, w, V);
	/* descending eigenvalues, eigenvectors in columns (gsl_eigen_symmv_sort ... GSL_EIGEN_SORT_VAL_DESC) */
	int *ord = (int *)malloc(sizeof(int) * (size_t)nt);
	for (int i = 0; i < nt; i++) ord[i] = i;
	for (int i = 0; i < nt; i++)
		for (int j = i + 1; j < nt; j++)
			if (w[ord[j]] > w[ord[i]]) { int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
	gsl_vector *evals = gsl_vector_alloc(nt);
	for (int i = 0; i < nt; i++) gsl_vector_set(evals, i, w[ord[i]]);
	const double total_variance = vector_elt_sum(evals, nt);
	/* the reference's loop (:267-272): frac is the share of the FIRST i values, tested before i is advanced,
	 * so one component more than needed is kept, capped at nt-1 */
	int i = 0;
	double frac = 0.0;
	while (frac < vfrac && (i + 1) < nt) {
		frac = (1.0 / total_variance) * vector_elt_sum(evals, i);
		i++;
	}
	m->nr = i;
	if (nt == 1) m->nr = 1;
	m->pca_evals_r = gsl_vector_alloc(m->nr);
	m->pca_evecs_r = gsl_matrix_alloc(nt, m->nr);
	fprintf(stderr, "# nr: %d frac: %lf\n", m->nr, frac);
	for (int r = 0; r < m->nr; r++) {
		gsl_vector_set(m->pca_evals_r, r, gsl_vector_get(evals, r));
		for (int t = 0; t < nt; t++) gsl_matrix_set(m->pca_evecs_r, t, r, V[t * nt + ord[r]]);
	}
	/* Z = Ysub U_r diag(lambda_r^-1/2) */
	m->pca_zmatrix = gsl_matrix_alloc(N, m->nr);
	for (int j = 0; j < N; j++)
		for (int r = 0; r < m->nr; r++) {
			double s = 0.0;
			for (int t = 0; t < nt; t++) s += ysub[j * nt + t] * gsl_matrix_get(m->pca_evecs_r, t, r);
			gsl_matrix_set(m->pca_zmatrix, j, r, s * (1.0 / sqrt(gsl_vector_get(m->pca_evals_r, r))));
		}
	gsl_vector_free(evals);
	free(ord); free(ysub); free(cov); free(w); free(V);
}

/* multi_modelstruct.c:121-146 */
void gen_pca_model_array(multi_modelstruct *m)
{
	m->pca_model_array = (modelstruct **)malloc(sizeof(modelstruct *) * (size_t)m->nr);
	for (int i = 0; i < m->nr; i++) {
		gsl_vector *z = gsl_vector_alloc(m->nmodel_points);
		for (int j = 0; j < m->nmodel_points; j++) gsl_vector_set(z, j, gsl_matrix_get(m->pca_zmatrix, j, i));
		m->pca_model_array[i] = alloc_modelstruct_2(m->xmodel, z, m->cov_fn_index, m->regression_order);
	}
}

/* multi_modelstruct.c:38-110 */
multi_modelstruct *alloc_multimodelstruct(gsl_matrix *xmodel_in, gsl_matrix *training_matrix_in, int cov_fn_index,
                                          int regression_order, double varfrac)
{
	assert(training_matrix_in->size1 == xmodel_in->size1);
	assert(training_matrix_in->size1 > 0);
	assert(training_matrix_in->size2 > 0);
	assert(xmodel_in->size2 > 0);
	if (regression_order < 0 || regression_order > 3) regression_order = 0;
	if (varfrac < 0 || varfrac > 1) varfrac = 0.95;
	if (cov_fn_index != MATERN32 && cov_fn_index != MATERN52) cov_fn_index = POWEREXPCOVFN;
	multi_modelstruct *m = (multi_modelstruct *)malloc(sizeof(multi_modelstruct));
	m->nt = (int)training_matrix_in->size2;
	m->nr = 0;
	m->nmodel_points = (int)xmodel_in->size1;
	m->nparams = (int)xmodel_in->size2;
	m->xmodel = xmodel_in;
	m->training_matrix = training_matrix_in;
	m->training_mean = gsl_vector_alloc(m->nt);
	m->regression_order = regression_order;
	m->cov_fn_index = cov_fn_index;
	for (int i = 0; i < m->nt; i++) {
		gsl_vector_view col = gsl_matrix_column(m->training_matrix, i);
		gsl_vector_set(m->training_mean, i, vector_elt_sum(&col.vector, m->nmodel_points) / (double)m->nmodel_points);
	}
	gen_pca_decomp(m, varfrac);
	gen_pca_model_array(m);
	return m;
}

/* multi_modelstruct.c:346-401 */
void dump_multi_modelstruct(FILE *fptr, multi_modelstruct *m)
{
	assert(fptr);
	fprintf(fptr, "%d\n", m->nt);
	fprintf(fptr, "%d\n", m->nr);
	fprintf(fptr, "%d\n", m->nparams);
	fprintf(fptr, "%d\n", m->nmodel_points);
	fprintf(fptr, "%d\n", m->cov_fn_index);
	fprintf(fptr, "%d\n", m->regression_order);
	for (int i = 0; i < m->nmodel_points; i++) {
		for (int j = 0; j < m->nparams; j++) fprintf(fptr, "%.17lf ", gsl_matrix_get(m->xmodel, i, j));
		fprintf(fptr, "\n");
	}
	for (int i = 0; i < m->nmodel_points; i++) {
		for (int j = 0; j < m->nt; j++) fprintf(fptr, "%.17lf ", gsl_matrix_get(m->training_matrix, i, j));
		fprintf(fptr, "\n");
	}
	for (int i = 0; i < m->nr; i++) fprintf(fptr, "%.17lf ", gsl_vector_get(m->pca_evals_r, i));
	fprintf(fptr, "\n");
	for (int i = 0; i < m->nt; i++) {
		for (int j = 0; j < m->nr; j++) fprintf(fptr, "%.17lf ", gsl_matrix_get(m->pca_evecs_r, i, j));
		fprintf(fptr, "\n");
	}
	for (int i = 0; i < m->nmodel_points; i++) {
		for (int j = 0; j < m->nr; j++) fprintf(fptr, "%.17lf ", gsl_matrix_get(m->pca_zmatrix, i, j));
		fprintf(fptr, "\n");
	}
	for (int i = 0; i < m->nr; i++) dump_modelstruct_2(fptr, m->pca_model_array[i]);
	/* Not in the reference's grammar, and invisible to its loader (which stops reading after the last modelstruct,
	 * multi_modelstruct.c:417-472): a Matern model trained with amplitude and nugget on the LOG scale says so.  Queried
	 * in the raw form its thetas would silently mean amp = theta0 instead of e^theta0.  Literal-mode snapshots (and every
	 * pow-exp one) carry no such line: they stay byte for byte what the reference writes. */
	if (m->cov_fn_index != POWEREXPCOVFN && (gpemu_host_modes() & 2 /* GPEMU_MODE_MATERN_LOG */))
		fprintf(fptr, "#gpemu matern_log_scale 1\n");
}

static int rd_int(FILE *f) { int v = 0; if (fscanf(f, "%d%*c", &v) != 1) { fprintf(stderr, "snapshot: read error\n"); gpemu_host_exit(EXIT_FAILURE); } return v; }
static double rd_dbl(FILE *f) { double v = 0; if (fscanf(f, "%lf%*c", &v) != 1) { fprintf(stderr, "snapshot: read error\n"); gpemu_host_exit(EXIT_FAILURE); } return v; }

/* multi_modelstruct.c:406-472 */
multi_modelstruct *load_multi_modelstruct(FILE *fptr)
{
	multi_modelstruct *m = (multi_modelstruct *)malloc(sizeof(multi_modelstruct));
	m->nt = rd_int(fptr);
	m->nr = rd_int(fptr);
	m->nparams = rd_int(fptr);
	m->nmodel_points = rd_int(fptr);
	m->cov_fn_index = rd_int(fptr);
	m->regression_order = rd_int(fptr);
	const int N = m->nmodel_points, nt = m->nt, nr = m->nr, d = m->nparams;
	m->xmodel = gsl_matrix_alloc(N, d);
	m->training_matrix = gsl_matrix_alloc(N, nt);
	m->training_mean = gsl_vector_alloc(nt);
	m->pca_model_array = (modelstruct **)malloc(sizeof(modelstruct *) * (size_t)nr);
	m->pca_evals_r = gsl_vector_alloc(nr);
	m->pca_evecs_r = gsl_matrix_alloc(nt, nr);
	m->pca_zmatrix = gsl_matrix_alloc(N, nr);
	for (int i = 0; i < N; i++) for (int j = 0; j < d; j++) gsl_matrix_set(m->xmodel, i, j, rd_dbl(fptr));
	for (int i = 0; i < N; i++) for (int j = 0; j < nt; j++) gsl_matrix_set(m->training_matrix, i, j, rd_dbl(fptr));
	for (int i = 0; i < nr; i++) gsl_vector_set(m->pca_evals_r, i, rd_dbl(fptr));
	for (int i = 0; i < nt; i++) for (int j = 0; j < nr; j++) gsl_matrix_set(m->pca_evecs_r, i, j, rd_dbl(fptr));
	for (int i = 0; i < N; i++) for (int j = 0; j < nr; j++) gsl_matrix_set(m->pca_zmatrix, i, j, rd_dbl(fptr));
	for (int i = 0; i < nr; i++) m->pca_model_array[i] = load_modelstruct_2(fptr);
	{
		/* the trailer dump_multi_modelstruct writes for log-scale Matern models: switch the layer to that mode */
		char line[128];
		int recorded = 0;
		while (fgets(line, sizeof line, fptr))
			if (!strncmp(line, "#gpemu matern_log_scale 1", 25)) recorded = 1;
		if (m->cov_fn_index != POWEREXPCOVFN) {
			const int have = (gpemu_host_modes() & 2) != 0;
			if (recorded && !have) {
				fprintf(stderr, "# snapshot: Matern thetas are on the log scale (trained with --matern_fixed): using that mode\n");
				gpemu_host_set_modes(gpemu_host_modes() | 2);
			} else if (!recorded && have) {
				fprintf(stderr, "# snapshot: no log-scale record in this Matern snapshot, but GPEMU_MATERN_FIXED is set: thetas are taken on the log scale\n");
			}
		}
	}
	for (int i = 0; i < nt; i++) {
		gsl_vector_view col = gsl_matrix_column(m->training_matrix, i);
		gsl_vector_set(m->training_mean, i, vector_elt_sum(&col.vector, N) / (double)N);
	}
	return m;
}

void free_multimodelstruct(multi_modelstruct *m)
{
	gsl_vector_free(m->training_mean);
	for (int i = 0; i < m->nr; i++) {
		gsl_vector_free(m->pca_model_array[i]->training_vector);
		gsl_matrix_free(m->pca_model_array[i]->xmodel);
		free_modelstruct_2(m->pca_model_array[i]);
	}
	free(m->pca_model_array);
	gsl_matrix_free(m->xmodel);
	gsl_matrix_free(m->training_matrix);
	gsl_vector_free(m->pca_evals_r);
	gsl_matrix_free(m->pca_evecs_r);
	gsl_matrix_free(m->pca_zmatrix);
	free(m);
}

/* multivar_support.c:20-28: the nr scalar GPs are independent; the reference trains them one after the other.  Here a
 * list of components is trained by a pool of host threads: component list[p] belongs to device slot p mod S
 * (S = gpemu_host_device_slots(), or the one device the caller is pinned to), and on every slot up to C components are
 * trained SIDE BY SIDE (GPEMU_COMPONENTS_PER_SLOT; default: as many of the slot's components as fit the device's free
 * memory with two lock-step groups each, at most 8) -- a search leaves the device idle between the rounds of its lock-step
 * groups (the BFGS arithmetic of its host threads, the latency-bound panel chains of its small batches), which the other
 * components' rounds fill.  Each thread runs the whole restart search of one component after the other on its slot's
 * device; no exchange between the threads -- the results meet in the modelstructs.  A component's search depends on the
 * seed, its data and the lock-step group size alone, not on which slot or thread ran it or what ran beside it, so the thetas
 * (and the snapshot) are the ones the serial loop gives. */
struct component_pool {
	multi_modelstruct *m;
	const int *list;             /* component indices of this slot, in order */
	int n, next;
	pthread_mutex_t mu;
	int device, share;
};

static void *component_main(void *arg)
{
	struct component_pool *P = (struct component_pool *)arg;
	gpemu_host_thread_device(P->device);
	gpemu_host_thread_share(P->share);
	for (;;) {
		pthread_mutex_lock(&P->mu);
		const int p = P->next < P->n ? P->next++ : -1;
		pthread_mutex_unlock(&P->mu);
		if (p < 0) break;
		modelstruct *c = P->m->pca_model_array[P->list[p]];
		estimate_thetas_threaded(c, c->options);
	}
	return NULL;
}

static void train_components(multi_modelstruct *m, const int *list, int nlist)
{
	if (nlist < 1) return;
	const int pinned = gpemu_host_thread_device_get();
	int nslots = pinned >= 0 ? 1 : gpemu_host_device_slots();
	if (nslots > nlist) nslots = nlist;
	int want = 0;                                                /* components side by side on one slot: 0 = by memory */
	const char *e = getenv("GPEMU_COMPONENTS_PER_SLOT");
	if (e && atoi(e) > 0) want = atoi(e);
	struct component_pool *pools = (struct component_pool *)calloc((size_t)nslots, sizeof *pools);
	int *order = (int *)malloc(sizeof(int) * (size_t)nlist), *nthreads = (int *)calloc((size_t)nslots, sizeof(int));
	int filled = 0, total_threads = 0;
	for (int s = 0; s < nslots; s++) {
		struct component_pool *P = &pools[s];
		P->m = m; P->list = order + filled; P->next = 0;
		pthread_mutex_init(&P->mu, NULL);
		P->device = pinned >= 0 ? pinned : gpemu_host_slot_device(s);
		for (int p = s; p < nlist; p += nslots) order[filled++] = list[p];
		P->n = (int)(order + filled - P->list);
	}
	for (int s = 0; s < nslots; s++) {
		/* slots of this call on the same physical device share its memory (GPEMU_DEVICES=0,0,0: three slots, one GPU) */
		int same = 0;
		for (int t = 0; t < nslots; t++) same += pools[t].device == pools[s].device;
		int c = want;
		if (!c) {
			const optstruct *o = m->pca_model_array[pools[s].list[0]]->options;
			const double need = 2.0 * gpemu_host_group_bytes(o->nmodel_points, 16);
			size_t fr = 0, tot = 0;
			c = 8;
			if (gpemu_device_memory(pools[s].device, &fr, &tot) == GPEMU_OK && fr > 0)
				while (c > 1 && (double)c * same * need > 0.9 * (double)fr) c--;
		}
		if (c > pools[s].n) c = pools[s].n;
		nthreads[s] = c;
		pools[s].share = c * same;
		total_threads += c;
	}
	if (total_threads == 1) {
		/* one component at a time on the caller's own thread, which stays as it is: unpinned, the search of a lone
		 * component deals its lock-step groups to EVERY device slot (estimate_thetas_threaded) */
		for (int p = 0; p < pools[0].n; p++) {
			modelstruct *c = m->pca_model_array[pools[0].list[p]];
			estimate_thetas_threaded(c, c->options);
		}
	} else {
		pthread_t *tid = (pthread_t *)calloc((size_t)total_threads, sizeof *tid);
		int t = 0;
		for (int s = 0; s < nslots; s++)
			for (int k = 0; k < nthreads[s]; k++, t++)
				if (pthread_create(&tid[t], NULL, component_main, &pools[s])) { perror("pthread_create"); gpemu_host_exit(EXIT_FAILURE); }
		for (t = 0; t < total_threads; t++)
			if (pthread_join(tid[t], NULL)) { perror("pthread_join"); gpemu_host_exit(EXIT_FAILURE); }
		free(tid);
	}
	for (int s = 0; s < nslots; s++) pthread_mutex_destroy(&pools[s].mu);
	free(pools); free(order); free(nthreads);
}

/* one process per GPU (ranks.c): rank r trains the components c = r, r + W, ...; the thetas of all components meet in ONE
 * all-gather (nthetas doubles per component, the shares padded to the same length); every rank then holds the whole model,
 * rank 0 writes it.  A single-output model has nothing to deal here: estimate_thetas_threaded deals its run list instead. */
static int g_components_over_ranks = 0;
int gpemu_host_components_over_ranks(void) { return g_components_over_ranks; }

static void estimate_multi_ranks(multi_modelstruct *m, FILE *outfp)
{
	const int world = gpemu_host_world_size(), rank = gpemu_host_rank();
	const int nthetas = (int)m->pca_model_array[0]->thetas->size;
	const int share = (m->nr + world - 1) / world;                  /* components per rank, padded */
	int *mine = (int *)malloc(sizeof(int) * (size_t)(share > 0 ? share : 1)), nmine = 0;
	for (int i = rank; i < m->nr; i += world) mine[nmine++] = i;
	g_components_over_ranks = 1;
	train_components(m, mine, nmine);
	g_components_over_ranks = 0;
	free(mine);
	double *send = (double *)calloc((size_t)share * nthetas, sizeof(double));
	double *recv = (double *)calloc((size_t)share * nthetas * world, sizeof(double));
	for (int j = 0, i = rank; i < m->nr; i += world, j++)
		for (int t = 0; t < nthetas; t++) send[(size_t)j * nthetas + t] = gsl_vector_get(m->pca_model_array[i]->thetas, t);
	gpemu_host_allgather(send, share * nthetas, recv);
	for (int i = 0; i < m->nr; i++) {
		const double *src = recv + ((size_t)(i % world) * share + (size_t)(i / world)) * nthetas;
		for (int t = 0; t < nthetas; t++) gsl_vector_set(m->pca_model_array[i]->thetas, t, src[t]);
	}
	free(send); free(recv);
	if (rank == 0) dump_multi_modelstruct(outfp, m);
}

void estimate_multi(multi_modelstruct *m, FILE *outfp)
{
	if (gpemu_host_world_size() > 1) {
		if (m->nr > 1) { estimate_multi_ranks(m, outfp); return; }
		/* one component: its run list is dealt to the ranks inside estimate_thetas_threaded; every rank returns the same thetas */
		estimate_thetas_threaded(m->pca_model_array[0], m->pca_model_array[0]->options);
		if (gpemu_host_rank() == 0) dump_multi_modelstruct(outfp, m);
		return;
	}
	int *all = (int *)malloc(sizeof(int) * (size_t)m->nr);
	for (int i = 0; i < m->nr; i++) all[i] = i;
	train_components(m, all, m->nr);
	free(all);
	dump_multi_modelstruct(outfp, m);
}

/* multivar_support.c:30-52 */
multi_emulator *alloc_multi_emulator(multi_modelstruct *m)
{
	multi_emulator *e = (multi_emulator *)malloc(sizeof(multi_emulator));
	e->nt = m->nt; e->nr = m->nr; e->nparams = m->nparams; e->nmodel_points = m->nmodel_points;
	e->nregression_fns = m->pca_model_array[0]->options->nregression_fns;
	e->nthetas = m->pca_model_array[0]->options->nthetas;
	e->model = m;
	e->emu_struct_array = (emulator_struct **)malloc(sizeof(emulator_struct *) * (size_t)e->nr);
	/* component i lives on device slot i mod S: emulate_points_multi starts all of them before it waits for the first.  The
	 * components that share a device are set up by ONE lock-step factorisation (gpemu_host_alloc_emulators) instead of the
	 * reference's one alloc_emulator_struct after the other: eight single-matrix chains are eight times the latency-bound
	 * panel chain, a batch of eight pays it once. */
	const int pinned = gpemu_host_thread_device_get();
	const int fill_cinverse = getenv("GPEMU_SKIP_CINVERSE") == NULL;
	modelstruct **models = (modelstruct **)malloc(sizeof(modelstruct *) * (size_t)e->nr);
	emulator_struct **made = (emulator_struct **)malloc(sizeof(emulator_struct *) * (size_t)e->nr);
	int *which = (int *)malloc(sizeof(int) * (size_t)e->nr);
	char *done = (char *)calloc((size_t)e->nr, 1);
	for (int first = 0; first < e->nr; first++) {
		if (done[first]) continue;
		const int dev = pinned >= 0 ? pinned : gpemu_host_slot_device(first);
		int n = 0;
		for (int i = first; i < e->nr; i++)              /* every component whose slot sits on this device */
			if (!done[i] && (pinned >= 0 || gpemu_host_slot_device(i) == dev)) { models[n] = m->pca_model_array[i]; which[n++] = i; done[i] = 1; }
		gpemu_host_thread_device(dev);
		if (fill_cinverse) {
			/* the reference's public emulator_struct.cinverse is wanted (library callers): every component is set up on its
			 * own -- the explicit N x N inverse comes out of a component's OWN factorisation workspace, and downloading and
			 * mirroring it dwarfs what a batch saves */
			for (int k = 0; k < n; k++) made[k] = gpemu_host_alloc_emulator(models[k], 1);
		} else {
			for (int b0 = 0; b0 < n; b0 += 16) {           /* (workspace: 16 x (2 Np + 64) x Np x 8 bytes per call) */
				const int nb = n - b0 < 16 ? n - b0 : 16;
				gpemu_host_alloc_emulators(models + b0, nb, 0, made + b0);
			}
		}
		for (int k = 0; k < n; k++) e->emu_struct_array[which[k]] = made[k];
	}
	gpemu_host_thread_device(pinned);
	free(which);
	free(models); free(made); free(done);
	return e;
}

void free_multi_emulator(multi_emulator *e)
{
	for (int i = 0; i < e->nr; i++) free_emulator_struct(e->emu_struct_array[i]);
	free(e->emu_struct_array);
	free_multimodelstruct(e->model);
	free(e);
}

/* npoints x nr PCA-space results -> npoints x nt observable-space results (multivar_support.c:126-151) */
static void backproject(const multi_emulator *emu, int npoints, const double *mp, const double *vp, double *mean_out, double *var_out)
{
	const int nt = emu->nt, nr = emu->nr;
	const multi_modelstruct *m = emu->model;
	for (int q = 0; q < npoints; q++)
		for (int i = 0; i < nt; i++) {
			double ms = 0.0, vs = 0.0;
			for (int j = 0; j < nr; j++) {
				const double u = gsl_matrix_get(m->pca_evecs_r, i, j), lam = gsl_vector_get(m->pca_evals_r, j);
				ms += u * sqrt(lam) * mp[(size_t)q * nr + j];
				vs += pow(u, 2.0) * lam * vp[(size_t)q * nr + j];
			}
			mean_out[(size_t)q * nt + i] = gsl_vector_get(m->training_mean, i) + ms;
			var_out[(size_t)q * nt + i] = vs;
		}
}

void emulate_points_multi(multi_emulator *emu, gsl_matrix *points, int pca_space, double *mean_out, double *var_out)
{
	const int np = (int)points->size1, nr = emu->nr;
	double *mp = (double *)malloc(sizeof(double) * (size_t)np * nr), *vp = (double *)malloc(sizeof(double) * (size_t)np * nr);
	double *mc = (double *)malloc(sizeof(double) * (size_t)np), *vc = (double *)malloc(sizeof(double) * (size_t)np);
	/* the nr scalar emulators are independent (multivar_support.c:116 loops over them): each has its own device
	 * context, so all of them are started before the first result is waited for */
	for (int c = 0; c < nr; c++) emulate_points_enqueue(emu->emu_struct_array[c], points);
	for (int c = 0; c < nr; c++) {
		emulate_points_collect(emu->emu_struct_array[c], np, mc, vc);
		for (int q = 0; q < np; q++) { mp[(size_t)q * nr + c] = mc[q]; vp[(size_t)q * nr + c] = vc[q]; }
	}
	if (pca_space) {
		memcpy(mean_out, mp, sizeof(double) * (size_t)np * nr);
		memcpy(var_out, vp, sizeof(double) * (size_t)np * nr);
	} else {
		backproject(emu, np, mp, vp, mean_out, var_out);
	}
	free(mp); free(vp); free(mc); free(vc);
}

static void one_point(multi_emulator *emu, gsl_vector *the_point, gsl_vector *the_mean, gsl_vector *the_variance, int pca_space)
{
	gsl_matrix view;
	const int n = pca_space ? emu->nr : emu->nt;
	double *q = (double *)malloc(sizeof(double) * the_point->size);
	double *mo = (double *)malloc(sizeof(double) * (size_t)n), *vo = (double *)malloc(sizeof(double) * (size_t)n);
	for (size_t i = 0; i < the_point->size; i++) q[i] = gsl_vector_get(the_point, i);
	view.size1 = 1; view.size2 = the_point->size; view.tda = the_point->size; view.data = q; view.block = NULL; view.owner = 0;
	emulate_points_multi(emu, &view, pca_space, mo, vo);
	for (int i = 0; i < n; i++) { gsl_vector_set(the_mean, i, mo[i]); gsl_vector_set(the_variance, i, vo[i]); }
	free(q); free(mo); free(vo);
}

/* multivar_support.c:78-86 */
void emulate_point_multi_pca(multi_emulator *emu, gsl_vector *the_point, gsl_vector *the_mean, gsl_vector *the_variance)
{
	one_point(emu, the_point, the_mean, the_variance, 1);
}

/* multivar_support.c:103-157 */
void emulate_point_multi(multi_emulator *emu, gsl_vector *the_point, gsl_vector *the_mean, gsl_vector *the_variance)
{
	one_point(emu, the_point, the_mean, the_variance, 0);
}

/*
 * optimizer.c -- the search driver above the device likelihood: random restarts + a BFGS
 * minimiser of -logL (libEmu/maxmultimin.c:47-136, 633-807; libEmu/estimate_threaded.c:78-334).
 *
 * gsl_multimin's vector_bfgs2 is not available (no GSL): the minimiser below is a textbook
 * BFGS with a strong-Wolfe bracketing/zoom line search (Nocedal & Wright alg. 3.5/3.6, cubic
 * interpolation), driven with the reference's constants: first step 1.5, line-search
 * sigma 0.5 (rho 0.01), stop at |g| < 0.1 or after 30 iterations, "no progress" ends a run.
 * Trajectories are not comparable with GSL's (the reference seeds from /dev/urandom anyway).
 * Every function/gradient pair is ONE device call (evalFnGradMulti shares the factorisation).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <unistd.h>
#include <pthread.h>
#include <sys/time.h>
#include "libemu.h"
#include "gpemu.h"

/* The reference starts its arg-max over restarts at -2000 (maxmultimin.c:38,60,110) and over threads at -20000
 * (estimate_threaded.c:5,34): a model whose log-likelihood never exceeds that (any N of a few thousand: logL scales
 * with N) cannot be trained there -- it ends with "maximisation didn't work at all" and thetas = 0.  The floors are
 * scale-dependent constants, not semantics: here the arg-max starts at -infinity, which gives the reference's
 * result whenever the reference produced one. */
#define SCREWUPVALUE (-HUGE_VAL)
#define SCREWUPVALUE_T (-HUGE_VAL)

/* search statistics (gpemu_host_search_stats): what the minimiser runs of the last estimate_thetas_threaded calls did */
static long g_stat_runs = 0, g_stat_converged = 0, g_stat_noprogress = 0, g_stat_fallbacks = 0, g_stat_iters = 0;
static double g_stat_best_gnorm = -1.0;         /* |gradient| at the end of the winning run of the LAST search */
static __thread double tls_run_gnorm = -1.0, tls_best_gnorm = -1.0;
static __thread long *tls_runs = NULL, *tls_iters = NULL;      /* the counters of the search this thread works for */
void gpemu_host_search_stats(long *runs, long *converged, long *noprogress, long *ls_fallbacks, double *best_gnorm)
{
	if (runs) *runs = __sync_fetch_and_add(&g_stat_runs, 0);
	if (converged) *converged = __sync_fetch_and_add(&g_stat_converged, 0);
	if (noprogress) *noprogress = __sync_fetch_and_add(&g_stat_noprogress, 0);
	if (ls_fallbacks) *ls_fallbacks = __sync_fetch_and_add(&g_stat_fallbacks, 0);
	if (best_gnorm) *best_gnorm = g_stat_best_gnorm;
}

static unsigned long g_seed = 0;
static int g_nthreads = 0, g_restarts = 50;
void gpemu_host_set_seed(unsigned long seed) { g_seed = seed; }
void gpemu_host_set_search(int nthreads, int restarts_per_job)
{
	if (nthreads > 0) g_nthreads = nthreads;
	if (restarts_per_job > 0) g_restarts = restarts_per_job;
}

/* libEmu/maxmultimin.c:789-804 */
void set_random_init_value(gsl_rng *rand, gsl_vector *x, gsl_matrix *ranges, int nthetas)
{
	for (int i = 0; i < nthetas; i++) {
		const double lo = gsl_matrix_get(ranges, i, 0), hi = gsl_matrix_get(ranges, i, 1);
		gsl_vector_set(x, i, gsl_rng_uniform(rand) * (hi - lo) + lo);
	}
}

static int env_flag(const char *name) { const char *e = getenv(name); return e && atoi(e) > 0; }

/* ------------------------------------------------------------------ BFGS */
struct fdf {
	void (*fdf)(const gsl_vector *, void *, double *, gsl_vector *);
	void *args;
	int n;
	int nevals;
	double nugget_floor;     /* GPEMU_NUGGET_FLOOR of THIS run (below); -inf: none */
};

static double dot(const double *a, const double *b, int n) { double s = 0; for (int i = 0; i < n; i++) s += a[i] * b[i]; return s; }

/* phi(alpha) = f(x + alpha p), dphi = g(x + alpha p).p ; NaN/inf values are reported as +inf so the search backs off */
/* GPEMU_NUGGET_FLOOR (not in the reference; unset = no floor): a lower wall for the log nugget, the first of the optimised
 * thetas (maxmultimin.c:281).  Training data without noise drive the unbounded search to nugget -> 0 and a numerically
 * singular model (the reference's own test/uni-2d-param: e^-33); with the wall a trial point below it counts as unusable --
 * exactly as one whose matrix does not factor -- and the line search backs off.  A START point below the wall (the search
 * box of the log nugget is [-5, -2], optstruct.c:142-250: only a floor above -5 can cut into it) is moved onto the wall:
 * evaluated where it was drawn, every trial of its run would count as unusable and the run would end where it began.
 * The floor is read per run and travels in the run's own struct fdf: concurrent searches (the component threads of
 * estimate_multi) share nothing. */
static double nugget_floor_from_env(void)
{
	const char *env = getenv("GPEMU_NUGGET_FLOOR");
	return (env && *env) ? atof(env) : -HUGE_VAL;
}

static double phi(struct fdf *F, const double *x, const double *p, double alpha, double *gout, double *dphi, gsl_vector *xt, gsl_vector *gt)
{
	double f;
	for (int i = 0; i < F->n; i++) gsl_vector_set(xt, i, x[i] + alpha * p[i]);
	if (gsl_vector_get(xt, 0) < F->nugget_floor) {
		for (int i = 0; i < F->n; i++) gout[i] = 0.0;
		*dphi = INFINITY;
		return INFINITY;
	}
	F->fdf(xt, F->args, &f, gt);
	F->nevals++;
	for (int i = 0; i < F->n; i++) gout[i] = gsl_vector_get(gt, i);
	*dphi = dot(gout, p, F->n);
	/* a trial point is unusable when the value OR the slope is not finite (the literal derivative formula overflows
	 * to inf * 0 for extreme length scales while the likelihood itself is still finite) */
	if (isnan(f) || isinf(f) || isnan(*dphi) || isinf(*dphi)) { f = INFINITY; *dphi = INFINITY; }
	return f;
}

static double cubic_min(double a, double fa, double da, double b, double fb, double db)
{
	/* minimiser of the cubic through (a,fa,da),(b,fb,db); falls back to bisection */
	const double d1 = da + db - 3.0 * (fa - fb) / (a - b);
	const double rad = d1 * d1 - da * db;
	if (!(rad >= 0.0) || isinf(fb) || isinf(db)) return 0.5 * (a + b);
	const double d2 = (b > a ? 1.0 : -1.0) * sqrt(rad);
	double t = b - (b - a) * (db + d2 - d1) / (db - da + 2.0 * d2);
	const double lo = fmin(a, b), hi = fmax(a, b), w = hi - lo;
	if (!(t > lo + 0.1 * w && t < hi - 0.1 * w)) t = 0.5 * (a + b);
	return t;
}

/* strong Wolfe line search; returns 0 and (alpha,f,g) on success, 1 if no acceptable step was found */
static int line_search(struct fdf *F, const double *x, double f0, const double *g0, const double *p, double alpha1,
                       double rho, double sigma, double *alpha_out, double *f_out, double *g_out, gsl_vector *xt, gsl_vector *gt)
{
	const int n = F->n;
	const double d0 = dot(g0, p, n);
	if (!(d0 < 0.0)) return 1;
	double a_prev = 0.0, f_prev = f0, d_prev = d0;
	double a = alpha1, fa, da;
	double lo = 0, flo = f0, dlo = d0, hi = 0, fhi = 0, dhi = 0;
	int bracketed = 0;
	double *g = (double *)malloc(sizeof(double) * (size_t)n);
	/* lowest trial value seen, in case no point passes the Armijo/Wolfe tests: the reference's gradient is the
	 * literal one-coordinate formula (emulator.c:173-209), which for nparams > 1 is NOT the derivative of the
	 * likelihood -- its slope d0 overstates the decrease by orders of magnitude and the sufficient-decrease test can
	 * then reject every step that does lower f.  Such a step is taken rather than ending the run. */
	double best_a = 0.0, best_f = f0;
	double *best_g = (double *)malloc(sizeof(double) * (size_t)n);
#define NOTE_TRIAL() do { if (fa < best_f) { best_f = fa; best_a = a; memcpy(best_g, g, sizeof(double) * (size_t)n); } } while (0)
#define LS_RETURN(code) do { free(g); free(best_g); return (code); } while (0)
	const int debug = getenv("GPEMU_OPT_DEBUG") != NULL;
	/* With the exact gradient (GPEMU_EXACT_GRAD, gpemu.h) the search is an ordinary strong-Wolfe search on a smooth
	 * function and is given the sections it needs: -logL has plateaus (all length scales large) next to walls (a
	 * vanishing nugget), so a bracket can be 10^5 times wider than the acceptable interval.  With the literal
	 * gradient the tests cannot be trusted (see above) and the search gives up early. */
	const int exact = (gpemu_host_modes() & GPEMU_MODE_EXACT_GRAD) != 0;
	const int max_zoom = exact ? 48 : 16;
	/* no trial point further than 6 log-units from x in any hyper-parameter (a factor 400 in a length scale) */
	double pmax = 0.0;
	for (int i = 0; i < n; i++) if (fabs(p[i]) > pmax) pmax = fabs(p[i]);
	const double amax = pmax > 0.0 ? 6.0 / pmax : HUGE_VAL;
	if (a > amax) a = amax;
	if (debug) fprintf(stderr, "#   ls f0 %.10g d0 %.4g alpha1 %.4g\n", f0, d0, alpha1);
	for (int it = 0; it < 12 && !bracketed; it++) {
		fa = phi(F, x, p, a, g, &da, xt, gt);
		if (debug) fprintf(stderr, "#   ls expand a %.4g fa %.10g da %.4g\n", a, fa, da);
		NOTE_TRIAL();
		if (fa > f0 + rho * a * d0 || (it > 0 && fa >= f_prev)) {
			lo = a_prev; flo = f_prev; dlo = d_prev; hi = a; fhi = fa; dhi = da; bracketed = 1; break;
		}
		if (fabs(da) <= -sigma * d0) { *alpha_out = a; *f_out = fa; memcpy(g_out, g, sizeof(double) * (size_t)n); LS_RETURN(0); }
		if (da >= 0.0) { lo = a; flo = fa; dlo = da; hi = a_prev; fhi = f_prev; dhi = d_prev; bracketed = 1; break; }
		if (a >= amax) {
			/* sufficient decrease at the step bound, still descending: take the bounded step (More-Thuente's
			 * "step at the upper bound"), the next iteration searches from there */
			*alpha_out = a; *f_out = fa; memcpy(g_out, g, sizeof(double) * (size_t)n); LS_RETURN(0);
		}
		a_prev = a; f_prev = fa; d_prev = da;
		a *= 2.5;
		if (a > amax) a = amax;
	}
	if (!bracketed) goto fallback;
	for (int it = 0; it < max_zoom; it++) {
		a = cubic_min(lo, flo, dlo, hi, fhi, dhi);
		/* a wall at the far end (value not finite, or orders of magnitude above the drop on offer): the acceptable
		 * points hug lo, halving the bracket would take dozens of sections */
		if (isinf(fhi) || fhi - f0 > 1e3 * fmax(fabs(d0 * (hi - lo)), fabs(f0 - flo))) a = lo + 0.1 * (hi - lo);
		fa = phi(F, x, p, a, g, &da, xt, gt);
		if (debug) fprintf(stderr, "#   ls zoom [%.4g,%.4g] a %.4g fa %.10g da %.4g\n", lo, hi, a, fa, da);
		NOTE_TRIAL();
		if (fa > f0 + rho * a * d0 || fa >= flo) {
			hi = a; fhi = fa; dhi = da;
		} else {
			if (fabs(da) <= -sigma * d0) { *alpha_out = a; *f_out = fa; memcpy(g_out, g, sizeof(double) * (size_t)n); LS_RETURN(0); }
			if (da * (hi - lo) >= 0.0) { hi = lo; fhi = flo; dhi = dlo; }
			lo = a; flo = fa; dlo = da;
		}
		if (fabs(hi - lo) < 1e-10 * fmax(1.0, fabs(lo))) break;
		if (!exact && best_f < f0 && it >= 5) break;   /* six sections without an acceptable point: take the best decrease */
	}
	/* accept the best sufficient-decrease point seen, if any */
fallback:
	if (best_f < f0) {
		if (debug) fprintf(stderr, "#   ls no Wolfe point: taking the lowest trial a %.4g f %.10g\n", best_a, best_f);
		__sync_fetch_and_add(&g_stat_fallbacks, 1);
		*alpha_out = best_a; *f_out = best_f; memcpy(g_out, best_g, sizeof(double) * (size_t)n);
		LS_RETURN(0);
	}
	LS_RETURN(1);
#undef NOTE_TRIAL
#undef LS_RETURN
}

/* libEmu/maxmultimin.c:633-778 */
int doOptimizeMultiMin(double (*fn)(const gsl_vector *, void *),
                       void (*gradientFn)(const gsl_vector *, void *, gsl_vector *),
                       void (*fnGradFn)(const gsl_vector *, void *, double *, gsl_vector *),
                       gsl_vector *thetaInit, gsl_vector *thetaFinal, void *args)
{
	(void)fn; (void)gradientFn;
	struct estimate_thetas_params *params = (struct estimate_thetas_params *)args;
	const int nthetas = params->options->nthetas;
	const int n = nthetas - 1;                 /* the amplitude is set by the data (estimateSigma) */
	const int stepmax = 30;
	const double stepSizeInit = 1.5, tolerance = 0.5, epsAbs = 0.1;
	int status = GSL_CONTINUE, stepcount = 0;

	struct fdf F = {fnGradFn, args, n, 0, nugget_floor_from_env()};
	double *x = (double *)malloc(sizeof(double) * (size_t)n), *g = (double *)malloc(sizeof(double) * (size_t)n);
	double *p = (double *)malloc(sizeof(double) * (size_t)n), *gn = (double *)malloc(sizeof(double) * (size_t)n);
	double *s = (double *)malloc(sizeof(double) * (size_t)n), *yv = (double *)malloc(sizeof(double) * (size_t)n);
	double *Hy = (double *)malloc(sizeof(double) * (size_t)n);
	double *H = (double *)calloc((size_t)n * n, sizeof(double));
	gsl_vector *xt = gsl_vector_alloc(n), *gt = gsl_vector_alloc(n);
	for (int i = 0; i < n; i++) { x[i] = gsl_vector_get(thetaInit, i + 1); H[i * n + i] = 1.0; }
	if (x[0] < F.nugget_floor) x[0] = F.nugget_floor;              /* a start point below the wall starts ON it */

	double f, dd;
	{
		double zero_p[64] = {0};
		double *zp = (n <= 64) ? zero_p : (double *)calloc((size_t)n, sizeof(double));
		f = phi(&F, x, zp, 0.0, g, &dd, xt, gt);
		if (zp != zero_p) free(zp);
	}
	const int debug = getenv("GPEMU_OPT_DEBUG") != NULL;
	if (isinf(f)) status = GSL_ENOPROG;
	while (status == GSL_CONTINUE && stepcount < stepmax) {
		const double gnorm = sqrt(dot(g, g, n));
		if (debug) fprintf(stderr, "# bfgs it %d f %.10g |g| %.4g evals %d\n", stepcount, f, gnorm, F.nevals);
		if (gnorm < epsAbs) { status = GSL_SUCCESS; break; }
		for (int i = 0; i < n; i++) { double t = 0; for (int j = 0; j < n; j++) t -= H[i * n + j] * g[j]; p[i] = t; }
		double alpha1 = 1.0;
		if (stepcount == 0) { const double pn = sqrt(dot(p, p, n)); alpha1 = stepSizeInit / (pn > 0 ? pn : 1.0); }
		if (!(dot(g, p, n) < 0.0)) {            /* not a descent direction: reset to steepest descent */
			memset(H, 0, sizeof(double) * (size_t)n * n);
			for (int i = 0; i < n; i++) { H[i * n + i] = 1.0; p[i] = -g[i]; }
			alpha1 = stepSizeInit / (gnorm > 0 ? gnorm : 1.0);
		}
		double alpha, fnew;
		if (line_search(&F, x, f, g, p, alpha1, 0.01, tolerance, &alpha, &fnew, gn, xt, gt)) {
			status = GSL_ENOPROG;               /* maxmultimin.c:703-707: no progress ends the run */
			if (debug) fprintf(stderr, "# bfgs line search failed at it %d (alpha1 %.4g, %d evals)\n", stepcount, alpha1, F.nevals);
			break;
		}
		for (int i = 0; i < n; i++) { s[i] = alpha * p[i]; yv[i] = gn[i] - g[i]; x[i] += s[i]; g[i] = gn[i]; }
		f = fnew;
		const double sy = dot(s, yv, n);
		if (sy > 1e-12 * sqrt(dot(s, s, n) * dot(yv, yv, n))) {
			/* H <- (I - s y^T/sy) H (I - y s^T/sy) + s s^T/sy */
			for (int i = 0; i < n; i++) { double t = 0; for (int j = 0; j < n; j++) t += H[i * n + j] * yv[j]; Hy[i] = t; }
			const double yHy = dot(yv, Hy, n);
			for (int i = 0; i < n; i++)
				for (int j = 0; j < n; j++)
					H[i * n + j] += (1.0 + yHy / sy) * s[i] * s[j] / sy - (Hy[i] * s[j] + s[i] * Hy[j]) / sy;
		}
		stepcount++;
	}
	if (status == GSL_CONTINUE && sqrt(dot(g, g, n)) < epsAbs) status = GSL_SUCCESS;
	tls_run_gnorm = sqrt(dot(g, g, n));
	__sync_fetch_and_add(&g_stat_runs, 1);
	__sync_fetch_and_add(&g_stat_iters, stepcount);
	if (tls_runs) __sync_fetch_and_add(tls_runs, 1);
	if (tls_iters) __sync_fetch_and_add(tls_iters, stepcount);
	if (status == GSL_SUCCESS) __sync_fetch_and_add(&g_stat_converged, 1);
	if (status == GSL_ENOPROG) __sync_fetch_and_add(&g_stat_noprogress, 1);
	if (stepcount == stepmax) fprintf(stderr, "# (error) multimin: no converge at stepmax %d\n", stepmax);

	for (int i = 0; i < n; i++) gsl_vector_set(xt, i, x[i]);
	const double sigma_final = log(estimateSigmaFull(xt, args));       /* maxmultimin.c:757 */
	gsl_vector_set(thetaFinal, 0, sigma_final);
	for (int i = 0; i < n; i++) gsl_vector_set(thetaFinal, i + 1, x[i]);
	free(x); free(g); free(p); free(gn); free(s); free(yv); free(Hy); free(H);
	gsl_vector_free(xt); gsl_vector_free(gt);
	return status;
}

/* libEmu/maxmultimin.c:47-136 */
void maxWithMultiMin(struct estimate_thetas_params *params)
{
	const int nthetas = params->options->nthetas;
	const int N = params->options->nmodel_points, d = params->options->nparams, nreg = params->options->nregression_fns;
	int tries = 0, success_count = 0;
	double bestLHood = SCREWUPVALUE;
	gsl_vector *xInit = gsl_vector_alloc(nthetas), *xFinal = gsl_vector_calloc(nthetas);
	gsl_vector *xBest = gsl_vector_calloc(nthetas), *xTest = gsl_vector_alloc(nthetas - 1);

	params->h_matrix = gsl_matrix_alloc(N, nreg);                       /* left for the caller to free (:75) */
	makeHMatrix_fnptr(params->h_matrix, params->the_model->xmodel, N, d, nreg, params->the_model->makeHVector);

	while (tries < params->max_tries) {
		set_random_init_value(params->random_number, xInit, params->options->grad_ranges, nthetas);
		const int status = doOptimizeMultiMin(&evalFnMulti, &gradFnMulti, &evalFnGradMulti, xInit, xFinal, params);
		if (status == GSL_SUCCESS) success_count++;
		for (int i = 0; i < nthetas - 1; i++) gsl_vector_set(xTest, i, gsl_vector_get(xFinal, i + 1));
		const double likelihood = -1 * evalFnMulti(xTest, params);
		if (likelihood > bestLHood && (isnan(likelihood) == 0 && isinf(likelihood) == 0)) {
			bestLHood = likelihood;
			gsl_vector_memcpy(xBest, xFinal);
			tls_best_gnorm = tls_run_gnorm;
		}
		tries++;
		gsl_vector_set_zero(xFinal);
	}
	if (bestLHood == SCREWUPVALUE) fprintf(stderr, "maximisation didn't work at all, relax your ranges\n");
	gsl_vector_memcpy(params->the_model->thetas, xBest);
	params->lhood_current = bestLHood;
	params->success_count = success_count;
	gsl_vector_free(xInit); gsl_vector_free(xFinal); gsl_vector_free(xBest); gsl_vector_free(xTest);
}

/* ------------------------------------------------------------------ restart pool */
int get_number_cpus(void)
{
	long n = sysconf(_SC_NPROCESSORS_ONLN);
	if (n < 1) n = 1;
	printf("# NCPUS: %ld\n", n);
	return (int)n;
}

static unsigned long seed_noblock(void)
{
	unsigned long seed = 0;
	FILE *f = fopen("/dev/urandom", "r");
	if (f) { if (fread(&seed, sizeof seed, 1, f) != 1) seed = 0; fclose(f); }
	if (!seed) { struct timeval tv; gettimeofday(&tv, 0); seed = (unsigned long)(tv.tv_sec + tv.tv_usec); }
	return seed;
}

struct pool {
	pthread_mutex_t result_lock;
	int total_runs, nthreads, next_run;
	int rank, world, local_runs;     /* ranks.c: runs rank, rank + world, ... are this process's (world = 1: all of them) */
	unsigned long seed;
	gsl_vector *best_thetas;
	double best_likelyhood_val;
	int best_run;
	double best_gnorm;
	long evals[5];                   /* device_bridge.c gpemu_host_thread_counters: this search's own evaluation counters */
	long runs, iters;                /* BFGS runs and iterations of this search */
};

struct worker { struct pool *pool; struct estimate_thetas_params params; int id; int in_group; int device; gsl_vector *best_thetas; };

/* The search is a LIST OF RUNS, r = 0 .. njobs * restarts - 1, each one maxWithMultiMin restart with its own generator
 * seeded from (seed, r): the list -- how many runs, where each starts -- depends on GPEMU_JOBS / GPEMU_RESTARTS /
 * GPEMU_SEED alone, not on how many threads, lock-step groups or GPUs the runs are dealt to, and the arg-max breaks ties
 * by run index: with a fixed seed a 1-GPU and an 8-GPU machine train the same thetas and write the same snapshot.
 * A thread that has finished a run takes the NEXT run of the list (the reference's job counter, estimate_threaded.c:
 * 295-306): the lock-step groups stay full until the list is exhausted, whichever runs happen to stop early.
 * (The reference's amount of search depends on the core count, estimate_threaded.c:97-113; its seeds on /dev/urandom.) */
static void *worker_main(void *arg)
{
	struct worker *w = (struct worker *)arg;
	struct pool *P = w->pool;
	gpemu_host_thread_device(w->device);         /* contexts this thread creates live on its slot's device */
	extern void gpemu_host_thread_counters(long *five);
	gpemu_host_thread_counters(P->evals);
	tls_runs = &P->runs; tls_iters = &P->iters;
	for (;;) {
		pthread_mutex_lock(&P->result_lock);
		/* (one process per GPU, ranks.c: this rank owns the runs rank, rank + W, ... of the list) */
		const int mine = P->next_run < P->local_runs ? P->next_run++ : -1;
		pthread_mutex_unlock(&P->result_lock);
		if (mine < 0) break;
		const int run = P->rank + P->world * mine;
		gsl_rng_set(w->params.random_number, P->seed ? P->seed + 7919UL * (unsigned long)run : seed_noblock());
		if (w->params.h_matrix) { gsl_matrix_free(w->params.h_matrix); w->params.h_matrix = NULL; }
		maxWithMultiMin(&w->params);             /* max_tries = 1: one restart */
		const double val = w->params.lhood_current;
		if (val > w->params.my_best) { w->params.my_best = val; gsl_vector_memcpy(w->best_thetas, w->params.the_model->thetas); }
		pthread_mutex_lock(&P->result_lock);
		if (val > P->best_likelyhood_val || (val == P->best_likelyhood_val && run < P->best_run)) {
			gsl_vector_memcpy(P->best_thetas, w->params.the_model->thetas);
			P->best_likelyhood_val = val;
			P->best_run = run;
			P->best_gnorm = tls_best_gnorm;
			printf("# worker %d won with %g\n", w->id, val);
		}
		pthread_mutex_unlock(&P->result_lock);
	}
	if (w->in_group) gpemu_host_group_leave(&w->params);
	gpemu_host_thread_counters(NULL);
	tls_runs = tls_iters = NULL;
	return NULL;
}

/* How a run list is dealt to host threads, lock-step groups and device slots (pure arithmetic, no device: the CPU
 * tests check it through this entry).  total_runs BFGS runs; groups of up to `lockstep` threads; `per_slot` groups per
 * device slot; nslots slots.  Returns the number of groups (<= cap) and fills *nthreads and, per group, its thread range
 * [lo, hi) and its slot (group g works for slot g mod nslots).  Full groups first -- 20 runs on one slot are a group of
 * 16 and a group of 4, not two of 10: a batch of 16 costs less per evaluation than two of 10 side by side -- unless that
 * would leave a slot without work: then the threads are spread evenly over one group per slot. */
int gpemu_host_plan_groups(int total_runs, int lockstep, int per_slot, int nslots, int *nthreads_out, int *lo, int *hi, int *slot,
                           int cap)
{
	if (total_runs < 1 || lockstep < 2 || per_slot < 1 || nslots < 1) return 0;
	long want = (long)lockstep * per_slot * nslots;
	int nthreads = want > total_runs ? total_runs : (int)want;
	const int full = (nthreads + lockstep - 1) / lockstep;
	int ngroups = full;
	if (ngroups < nslots) ngroups = nslots < nthreads ? nslots : nthreads;
	if (ngroups > cap) return 0;
	const int even = ngroups != full;            /* spread over more slots than full groups would use */
	for (int g = 0; g < ngroups; g++) {
		if (even) { lo[g] = (int)((long)g * nthreads / ngroups); hi[g] = (int)((long)(g + 1) * nthreads / ngroups); }
		else { lo[g] = g * lockstep; hi[g] = lo[g] + lockstep < nthreads ? lo[g] + lockstep : nthreads; }
		slot[g] = g % nslots;
	}
	if (nthreads_out) *nthreads_out = nthreads;
	return ngroups;
}

/* every group holds a value+gradient workspace on its slot's device: lockstep x (2 Np + 64) x Np x 8 bytes plus up to 10 GB
 * of C^-1 corners (gpemu.h) */
double gpemu_host_group_bytes(int nmodel_points, int lockstep)
{
	const double Np = 64.0 * ceil(nmodel_points / 64.0);
	const double corner = (Np + 64.0) * (Np + 64.0) * 8.0;
	double corners = floor(10.0e9 / corner);
	if (corners < 1.0) corners = 1.0;
	if (corners > lockstep) corners = lockstep;
	return lockstep * (2.0 * Np + 64.0) * Np * 8.0 + corners * corner + 64.0e6;
}

/* libEmu/estimate_threaded.c:78-237.  The reference starts one pthread per CPU, each running jobs of 50 restarts
 * on its own copy of the model.  Here the restarts run as LOCK-STEP GROUPS (default): up to 16 host threads per group,
 * each an ordinary sequential BFGS run, share one device context; whenever all threads of a group have asked for a
 * likelihood (+gradient) the requests go to the GPU as one batch (device_bridge.c).  TWO groups work for each device
 * slot (GPEMU_GROUPS_PER_SLOT): while one group's host threads do their BFGS arithmetic -- and while its batch runs
 * its latency-bound panel chain -- the other group's batch keeps the matrix cores busy (the device form of the
 * reference keeping every core busy, estimate_threaded.c:172-188).  The runs of the list above are dealt round-robin to
 * the threads, the threads in contiguous shares to the groups, the groups to the device slots
 * (gpemu_host_device_slots(): every visible GPU unless GPEMU_DEVICES / GPEMU_DEVICE says otherwise -- set it on a
 * shared node) -- no communication between groups until the arg-max under the result mutex
 * (estimate_threaded.c:308-313).  A caller that already works for one slot (a component thread of estimate_multi)
 * keeps the whole search on that slot.  GPEMU_LOCKSTEP=1 (or a Matern model without the corrected gradient) gives
 * the older scheme: GPEMU_NTHREADS workers with one device context each. */
void estimate_thetas_threaded(modelstruct *the_model, optstruct *options)
{
	int nthreads = g_nthreads > 0 ? g_nthreads : 1;
	const char *env = getenv("GPEMU_NTHREADS");
	if (env && atoi(env) > 0) nthreads = atoi(env);
	int njobs = nthreads;
	env = getenv("GPEMU_JOBS");
	if (env && atoi(env) > 0) njobs = atoi(env);
	if (njobs < nthreads) njobs = nthreads;
	int restarts = g_restarts;
	env = getenv("GPEMU_RESTARTS");
	if (env && atoi(env) > 0) restarts = atoi(env);
	int lockstep = 16;
	env = getenv("GPEMU_LOCKSTEP");
	if (env && atoi(env) > 0) lockstep = atoi(env);
	if (lockstep > 64) lockstep = 64;
	int per_slot = 2;
	env = getenv("GPEMU_GROUPS_PER_SLOT");
	if (env && atoi(env) > 0) per_slot = atoi(env) > 8 ? 8 : atoi(env);
	else if (lockstep > 1) {
		/* every group holds a value+gradient workspace on its slot's device: lockstep x (2 Np + 64) x Np x 8 bytes plus up
		 * to 10 GB of C^-1 corners (gpemu.h).  Two groups per slot are the default only while they fit the device's FREE
		 * memory with a margin (at N ~ 24 000 one group of 16 needs ~155 GB: one fits 288 GB, two do not); the result of
		 * a search does not depend on the number of groups, only its speed does. */
		const double need = gpemu_host_group_bytes(options->nmodel_points, lockstep);
		const int dev0 = gpemu_host_thread_device_get() >= 0 ? gpemu_host_thread_device_get() : gpemu_host_slot_device(0);
		const int nsl = gpemu_host_thread_device_get() >= 0 ? 1 : gpemu_host_device_slots();
		int sharing = 0;                               /* slots that live on the same physical device as slot 0 */
		for (int s_ = 0; s_ < nsl; s_++) sharing += (gpemu_host_thread_device_get() >= 0 ? dev0 : gpemu_host_slot_device(s_)) == dev0;
		sharing *= gpemu_host_thread_share_get();      /* ... and searches that run beside this one on it (estimate_multi) */
		size_t fr = 0, tot = 0;
		if (gpemu_device_memory(dev0, &fr, &tot) == GPEMU_OK && fr > 0)
			while (per_slot > 1 && (double)per_slot * sharing * need > 0.9 * (double)fr) per_slot--;
		if (per_slot < 2 && getenv("GPEMU_SEARCH_STATS"))
			fprintf(stderr, "# search: one lock-step group per device slot (a group needs %.1f GB, %.1f GB are free)\n", need / 1e9, (double)fr / 1e9);
	}
	/* the batched gradient exists for pow-exp (literal or exact) and for Matern with the corrected forms (gpemu.h modes) */
	if (options->cov_fn_index != POWEREXPCOVFN && gpemu_host_modes() != (GPEMU_MODE_EXACT_GRAD | GPEMU_MODE_MATERN_LOG)) lockstep = 1;
	const int total = njobs * restarts;             /* the run list */
	/* one process per GPU (ranks.c): a single-output model's run list is dealt to the ranks (a multi-output model deals its
	 * components instead, estimate_multi, and every rank then runs whole lists) */
	extern int gpemu_host_components_over_ranks(void);
	const int world = gpemu_host_components_over_ranks() ? 1 : gpemu_host_world_size();
	const int rank = world > 1 ? gpemu_host_rank() : 0;
	const int local_total = total > rank ? (total - rank + world - 1) / world : 0;
	/* device slots this search may use */
	const int pinned = gpemu_host_thread_device_get();
	const int nslots = pinned >= 0 ? 1 : gpemu_host_device_slots();
	int ngroups = 0;
	int glo[512], ghi[512], gslot[512];
	if (lockstep > 1) {
		ngroups = gpemu_host_plan_groups(local_total > 0 ? local_total : 1, lockstep, per_slot, nslots, &nthreads, glo, ghi, gslot, 512);
		if (ngroups < 1) { fprintf(stderr, "estimate_thetas_threaded: cannot lay out the lock-step groups\n"); gpemu_host_exit(EXIT_FAILURE); }
	} else if (nthreads > local_total) nthreads = local_total > 0 ? local_total : 1;
	unsigned long seed = g_seed;
	env = getenv("GPEMU_SEED");
	if (env && atol(env) > 0) seed = (unsigned long)atol(env);
	struct timeval tv0;
	gettimeofday(&tv0, 0);

	struct pool P;
	pthread_mutex_init(&P.result_lock, NULL);
	P.total_runs = total; P.nthreads = nthreads; P.next_run = 0; P.seed = seed;
	P.rank = rank; P.world = world; P.local_runs = local_total;
	P.best_thetas = gsl_vector_calloc(options->nthetas);
	P.best_likelyhood_val = SCREWUPVALUE_T;
	P.best_run = total;
	P.best_gnorm = -1.0;
	memset(P.evals, 0, sizeof P.evals);
	P.runs = P.iters = 0;

	struct worker *W = (struct worker *)calloc((size_t)nthreads, sizeof *W);
	pthread_t *tid = (pthread_t *)calloc((size_t)nthreads, sizeof *tid);
	for (int i = 0; i < nthreads; i++) {
		W[i].pool = &P; W[i].id = i;
		/* private modelstruct: own thetas; design / training data are read-only and shared */
		modelstruct *m = (modelstruct *)malloc(sizeof(modelstruct));
		*m = *the_model;
		m->thetas = gsl_vector_calloc(options->nthetas);
		W[i].params.the_model = m;
		W[i].params.options = options;
		W[i].params.max_tries = 1;
		W[i].params.my_best = SCREWUPVALUE_T;
		W[i].best_thetas = gsl_vector_calloc(options->nthetas);
		W[i].params.h_matrix = NULL;
		W[i].params.random_number = gsl_rng_alloc(gsl_rng_default);
	}
	void **groups = NULL;
	if (lockstep > 1) {
		/* group g = threads [glo[g], ghi[g]) on device slot gslot[g] (gpemu_host_plan_groups) */
		groups = (void **)calloc((size_t)ngroups, sizeof(void *));
		struct estimate_thetas_params **members = (struct estimate_thetas_params **)malloc(sizeof(void *) * (size_t)nthreads);
		for (int g = 0; g < ngroups; g++) {
			const int lo = glo[g], hi = ghi[g];
			const int dev = pinned >= 0 ? pinned : gpemu_host_slot_device(gslot[g]);
			for (int i = lo; i < hi; i++) { members[i - lo] = &W[i].params; W[i].in_group = 1; W[i].device = dev; }
			gpemu_host_thread_device(dev);            /* the group's context is created on the creator's device */
			groups[g] = gpemu_host_group_create(members, hi - lo);
			gpemu_host_thread_device(pinned);
			if (!groups[g]) { fprintf(stderr, "estimate_thetas_threaded: cannot create the lock-step group\n"); gpemu_host_exit(EXIT_FAILURE); }
		}
		free(members);
	} else {
		for (int i = 0; i < nthreads; i++) W[i].device = pinned >= 0 ? pinned : gpemu_host_slot_device(i);
	}
	for (int i = 0; i < nthreads; i++)
		if (pthread_create(&tid[i], NULL, worker_main, &W[i])) { perror("pthread_create"); gpemu_host_exit(EXIT_FAILURE); }
	for (int i = 0; i < nthreads; i++)
		if (pthread_join(tid[i], NULL)) { perror("pthread_join"); gpemu_host_exit(EXIT_FAILURE); }

	printf("-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=\n");
	for (int i = 0; i < nthreads; i++) {
		printf("thread(%d)\tlog-L: %lf\tthetas:", i, W[i].params.my_best);
		for (int j = 0; j < options->nthetas; j++) printf("%lf ", exp(gsl_vector_get(W[i].best_thetas, j)));
		printf("\n");
	}
	printf("-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=-=\n");
	for (int g = 0; g < ngroups; g++) gpemu_host_group_destroy(groups[g]);
	free(groups);
	for (int i = 0; i < nthreads; i++) {
		gpemu_host_release(&W[i].params);
		gsl_rng_free(W[i].params.random_number);
		if (W[i].params.h_matrix) gsl_matrix_free(W[i].params.h_matrix);
		gsl_vector_free(W[i].best_thetas);
		gsl_vector_free(W[i].params.the_model->thetas);
		free(W[i].params.the_model);
	}
	if (world > 1) {
		/* the arg-max of estimate_threaded.c:308-313 across processes: ONE all-gather of (best value, its run index, its
		 * thetas) per rank; the larger value wins, ties go to the lower run index -- what one process decides run by run */
		const int nt = options->nthetas, len = 2 + nt;
		double *send = (double *)calloc((size_t)len, sizeof(double)), *recv = (double *)calloc((size_t)len * world, sizeof(double));
		send[0] = P.best_likelyhood_val; send[1] = (double)P.best_run;
		for (int t = 0; t < nt; t++) send[2 + t] = gsl_vector_get(P.best_thetas, t);
		gpemu_host_allgather(send, len, recv);
		int win = -1;
		for (int r = 0; r < world; r++) {
			const double *q = recv + (size_t)r * len;
			if ((int)q[1] >= total) continue;                           /* that rank had no successful run */
			if (win < 0 || q[0] > recv[(size_t)win * len] || (q[0] == recv[(size_t)win * len] && q[1] < recv[(size_t)win * len + 1])) win = r;
		}
		if (win >= 0) {
			P.best_likelyhood_val = recv[(size_t)win * len];
			P.best_run = (int)recv[(size_t)win * len + 1];
			for (int t = 0; t < nt; t++) gsl_vector_set(P.best_thetas, t, recv[(size_t)win * len + 2 + t]);
		}
		free(send); free(recv);
	}
	gsl_vector_memcpy(the_model->thetas, P.best_thetas);
	g_stat_best_gnorm = P.best_gnorm;
	if (env_flag("GPEMU_SEARCH_STATS")) {
		/* one line for bench.py / a curious user: what this search cost on the device */
		struct timeval tv1;
		gettimeofday(&tv1, 0);
		const double secs = (double)(tv1.tv_sec - tv0.tv_sec) + 1e-6 * (double)(tv1.tv_usec - tv0.tv_usec);
		fprintf(stderr, "# search stats: runs %ld threads %d groups %d slots %d value_grad_evals %ld value_evals %ld cached %ld rounds %ld "
		        "round_elements %ld iterations %ld seconds %.3f best %.10g\n", P.runs, nthreads, ngroups, nslots,
		        P.evals[1], P.evals[0], P.evals[2], P.evals[3], P.evals[4], P.iters, secs, P.best_likelyhood_val);
	}
	gsl_vector_free(P.best_thetas);
	pthread_mutex_destroy(&P.result_lock);
	free(W); free(tid);
}

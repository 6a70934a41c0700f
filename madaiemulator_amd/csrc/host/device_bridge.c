/*
 * device_bridge.c -- the reference's likelihood / gradient / prediction entry points
 * (libEmu/maxmultimin.c:148-618, emulator_struct.c:13-143) implemented on the device
 * library.  A gpemu_ctx (one HIP stream + HBM workspace) is cached per `params` pointer
 * (likelihood side) or per emulator_struct (prediction side): the reference's per-thread
 * deep copies (estimate_threaded.c:57-68) become per-thread device contexts, so
 * evalFnMulti / gradFnMulti stay re-entrant across threads with distinct params.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <pthread.h>
#include <time.h>
#include "libemu.h"
#include "gpemu.h"

static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

extern int gpemu_host_kind_of(double (*fn)(gsl_vector *, gsl_vector *, gsl_vector *, int, int));

/* ---------------------------------------------------------------- devices
 * The reference spreads its work over the host's CPUs (estimate_threaded.c:97: one pthread per core).  Here the unit is
 * the GPU: a list of device SLOTS (GPEMU_DEVICES=0,1,2,... ; a device may be listed more than once: two contexts on
 * it; default = every visible device, or the one device pinned by gpemu_host_set_device / GPEMU_DEVICE).  Independent
 * pieces of work -- the PCA components of estimate_multi (multivar_support.c:20-28), the restart groups of
 * estimate_thetas_threaded (estimate_threaded.c:101-113), the component emulators of alloc_multi_emulator -- are dealt
 * to the slots; a host thread that works for one slot says so with gpemu_host_thread_device() and every context it
 * creates lands there. */
static int g_device = -1;                       /* pinned device (gpemu_host_set_device), -1: none */
static int g_slots[64], g_nslots = 0;
static pthread_once_t g_slots_once = PTHREAD_ONCE_INIT;
static __thread int tls_device = -1;            /* device of the slot this thread works for, -1: slot 0 */
static __thread int tls_share = 1;              /* searches that run side by side on this thread's device (estimate_multi) */

static void slots_init(void)
{
	const char *e = getenv("GPEMU_DEVICES");
	if (e && *e) {
		const char *p = e;
		while (*p && g_nslots < 64) {
			char *end;
			long v = strtol(p, &end, 10);
			if (end == p) break;
			if (v >= 0) g_slots[g_nslots++] = (int)v;
			p = (*end == ',') ? end + 1 : end;
		}
	}
	if (!g_nslots && g_device >= 0) g_slots[g_nslots++] = g_device;
	if (!g_nslots) {
		int n = gpemu_device_count();
		if (n > 64) n = 64;
		for (int i = 0; i < n; i++) g_slots[g_nslots++] = i;
	}
	if (!g_nslots) g_slots[g_nslots++] = 0;      /* no device: gpemu_ctx_create will say so */
}

void gpemu_host_set_device(int device) { g_device = device; }
int gpemu_host_device_slots(void) { pthread_once(&g_slots_once, slots_init); return g_nslots; }
int gpemu_host_slot_device(int slot) { pthread_once(&g_slots_once, slots_init); return g_slots[((slot % g_nslots) + g_nslots) % g_nslots]; }
void gpemu_host_thread_device(int device) { tls_device = device; }
int gpemu_host_thread_device_get(void) { return tls_device; }
void gpemu_host_thread_share(int n) { tls_share = n > 1 ? n : 1; }
int gpemu_host_thread_share_get(void) { return tls_share; }
int gpemu_host_device(void) { return tls_device >= 0 ? tls_device : gpemu_host_slot_device(0); }

/* ---------------------------------------------------------------- modes
 * ONE source of truth for the corrected forms of gpemu.h (GPEMU_MODE_EXACT_GRAD, GPEMU_MODE_MATERN_LOG) in the host
 * layer: read from the environment once (GPEMU_EXACT_GRAD / GPEMU_MATERN_FIXED), changed by gpemu_host_set_modes (the
 * CLI's --exact_gradient / --matern_fixed, a snapshot that records the log-scale mode), and applied to EVERY device
 * context the layer creates or holds.  The scalar covariance functions (covreg.c: kappa = c(x*,x*)), the optimiser's
 * line search and the device kernels therefore always agree on what the thetas mean. */
static int g_modes = -1;
struct entry;
static void apply_modes_to_entries(int flags);
void gpemu_host_warm_wait(void);

int gpemu_host_modes(void)
{
	if (g_modes < 0) {
		const char *eg = getenv("GPEMU_EXACT_GRAD"), *mf = getenv("GPEMU_MATERN_FIXED");
		int m = 0;
		if (eg && atoi(eg) > 0) m |= GPEMU_MODE_EXACT_GRAD;
		if (mf && atoi(mf) > 0) m |= GPEMU_MODE_MATERN_LOG;
		__sync_val_compare_and_swap(&g_modes, -1, m);
	}
	return g_modes;
}

void gpemu_host_set_modes(int flags)
{
	flags &= GPEMU_MODE_EXACT_GRAD | GPEMU_MODE_MATERN_LOG;
	(void)gpemu_host_modes();
	g_modes = flags;
	apply_modes_to_entries(flags);
}

/* ---------------------------------------------------------------- registry */
/* the last VCACHE_N (theta -> value, sigma^2, status) results of one caller (a `params`): evalFnMulti and estimateSigmaFull
 * are pure functions of theta and the model, and the search asks for both at the point its last evalFnGradMulti call
 * already evaluated (maxmultimin.c:98-103, 757: the value of the final thetas and their sigma^2) -- those two calls per
 * run are answered from here instead of two more factorisations.  Exact theta bits only; dropped when the model data
 * change. */
#define VCACHE_N 64   /* a run's final point was accepted at most one (failed) line search ago: <= 60 evaluations */
struct vcache {
	int n, next, nthetas;
	double th[VCACHE_N][GPEMU_MAX_PARAMS + 2];
	double val[VCACHE_N], sigma2[VCACHE_N];
	int status[VCACHE_N];
};

struct entry {
	const void *key;
	gpemu_ctx *ctx;
	const double *xdata, *ydata;      /* where the uploaded data came from ... */
	unsigned long long xsum, ysum;    /* ... and a 64-bit checksum of every value: callers rewrite the buffers in place */
	int N, d, kind, order;
	struct vcache vc;                 /* (callers with a context of their own; group members have theirs in the group) */
	struct entry *next;
};

static void vcache_clear(struct vcache *c) { c->n = 0; c->next = 0; }
static int vcache_find(const struct vcache *c, const double *th, int nthetas)
{
	if (c->nthetas != nthetas) return -1;
	for (int i = 0; i < c->n; i++)
		if (!memcmp(c->th[i] + 1, th + 1, sizeof(double) * (size_t)(nthetas - 1))) return i;     /* theta[0] is ignored by every entry */
	return -1;
}
static void vcache_put(struct vcache *c, const double *th, int nthetas, double val, double sigma2, int status)
{
	if (nthetas > GPEMU_MAX_PARAMS + 2) return;
	if (c->nthetas != nthetas) { c->n = 0; c->next = 0; c->nthetas = nthetas; }
	int i = vcache_find(c, th, nthetas);
	if (i < 0) { i = c->next; c->next = (c->next + 1) % VCACHE_N; if (c->n < VCACHE_N) c->n++; }
	memcpy(c->th[i], th, sizeof(double) * (size_t)nthetas);
	c->val[i] = val; c->sigma2[i] = sigma2; c->status[i] = status;
}

/* evaluation counters of the whole process (gpemu_host_eval_stats): device evaluations by kind and cache answers */
static long g_n_value = 0, g_n_valgrad = 0, g_n_cached = 0, g_n_rounds = 0, g_n_round_elems = 0;
/* ... and of ONE search: the threads of a search (optimizer.c worker_main) point this at their pool's five counters
 * {value, value+gradient, cached, rounds, round elements}, so that searches running side by side (the component threads of
 * estimate_multi) each report their own figures; a lock-step round is counted by the member that sends it, which belongs to
 * the group's search */
static __thread long *tls_counters = NULL;
void gpemu_host_thread_counters(long *five) { tls_counters = five; }
#define COUNT(global_, idx_, n_) do { __sync_fetch_and_add(&(global_), (n_)); if (tls_counters) __sync_fetch_and_add(&tls_counters[idx_], (n_)); } while (0)
void gpemu_host_eval_stats(long *value_evals, long *valgrad_evals, long *cached, long *rounds, long *round_elements)
{
	if (value_evals) *value_evals = __sync_fetch_and_add(&g_n_value, 0);
	if (valgrad_evals) *valgrad_evals = __sync_fetch_and_add(&g_n_valgrad, 0);
	if (cached) *cached = __sync_fetch_and_add(&g_n_cached, 0);
	if (rounds) *rounds = __sync_fetch_and_add(&g_n_rounds, 0);
	if (round_elements) *round_elements = __sync_fetch_and_add(&g_n_round_elems, 0);
}
static struct entry *g_entries = NULL;
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;

static void die(gpemu_ctx *ctx, int rc, const char *where) __attribute__((noreturn));
static void die(gpemu_ctx *ctx, int rc, const char *where)
{
	/* (message and exit behind ONE gate, fatal.c: a device error usually hits several host threads at once) */
	gpemu_host_fatal("%s: gpemu error %d: %s\n", where, rc,
	                 ctx ? gpemu_last_error(ctx) : rc == GPEMU_ERR_NO_DEVICE ? "no usable HIP device" : "(reported by the lock-step group's device context)");
}

static struct entry *lookup(const void *key, int create)
{
	struct entry *e;
	pthread_mutex_lock(&g_lock);
	for (e = g_entries; e; e = e->next)
		if (e->key == key) break;
	if (!e && create) {
		gpemu_host_warm_wait();
		e = (struct entry *)calloc(1, sizeof *e);
		e->key = key;
		int rc = gpemu_ctx_create(&e->ctx, gpemu_host_device());
		if (rc) { pthread_mutex_unlock(&g_lock); die(NULL, rc, "gpemu_ctx_create"); }
		gpemu_set_mode(e->ctx, gpemu_host_modes());      /* the layer's modes, not whatever the environment says now */
		e->next = g_entries;
		g_entries = e;
	}
	pthread_mutex_unlock(&g_lock);
	return e;
}

static void apply_modes_to_entries(int flags)
{
	pthread_mutex_lock(&g_lock);
	for (struct entry *e = g_entries; e; e = e->next) {
		if (gpemu_get_mode(e->ctx) != flags) {
			gpemu_set_mode(e->ctx, flags);
			vcache_clear(&e->vc);                        /* the thetas mean something else now */
		}
	}
	pthread_mutex_unlock(&g_lock);
}

/* one scratch device context per host thread for the model-less low-level entries (lowlevel.c) */
gpemu_ctx *gpemu_host_scratch_ctx(const char *where)
{
	static __thread char tls_key;
	struct entry *e = lookup(&tls_key, 1);
	if (!e || !e->ctx) { fprintf(stderr, "%s: no device context\n", where); gpemu_host_exit(EXIT_FAILURE); }
	return e->ctx;
}

void gpemu_host_release(void *key)
{
	pthread_mutex_lock(&g_lock);
	struct entry **pp = &g_entries;
	while (*pp) {
		/* (key + 1: the second context evalFnMultiList keeps for a params) */
		if ((*pp)->key == key || (*pp)->key == (const void *)((const char *)key + 1)) {
			struct entry *e = *pp;
			*pp = e->next;
			gpemu_ctx_destroy(e->ctx);
			free(e);
			continue;
		}
		pp = &(*pp)->next;
	}
	pthread_mutex_unlock(&g_lock);
}

static double *pack_matrix(const gsl_matrix *m)
{
	double *p = (double *)malloc(sizeof(double) * m->size1 * m->size2);
	for (size_t i = 0; i < m->size1; i++)
		memcpy(p + i * m->size2, m->data + i * m->tda, sizeof(double) * m->size2);
	return p;
}
static double *pack_vector(const gsl_vector *v)
{
	double *p = (double *)malloc(sizeof(double) * v->size);
	for (size_t i = 0; i < v->size; i++) p[i] = v->data[i * v->stride];
	return p;
}

/* 64-bit checksum of every element (multiply-xor, position dependent): N*(d+1) doubles, microseconds */
static unsigned long long sum_doubles(unsigned long long h, const double *p, size_t n, size_t stride)
{
	const unsigned long long K = 0xFF51AFD7ED558CCDull;
	for (size_t i = 0; i < n; i++) {
		unsigned long long w;
		memcpy(&w, p + i * stride, sizeof w);
		h = (h ^ w) * K;
		h ^= h >> 29;
	}
	return h;
}
static unsigned long long sum_matrix(const gsl_matrix *m)
{
	unsigned long long h = 0x9E3779B97F4A7C15ull;
	for (size_t i = 0; i < m->size1; i++) h = sum_doubles(h, m->data + i * m->tda, m->size2, 1);
	return h;
}
static unsigned long long sum_vector(const gsl_vector *v) { return sum_doubles(0xBF58476D1CE4E5B9ull, v->data, v->size, v->stride); }

/* make sure the model's design / training vector are the ones resident in this entry's HBM.  The reference re-reads the
 * model on every call (maxmultimin.c:317 fills the matrix from the_model->xmodel each time), so a caller may rewrite
 * xmodel->data or training_vector->data in place between calls (libRbind-style loops, an MCMC re-fit): pointer identity
 * says nothing -- the resident copy is kept only while a checksum over ALL values agrees.  *changed (optional) is set
 * when anything was uploaded. */
static struct entry *bind_model_entry(const void *key, modelstruct *m, const char *where, int *changed)
{
	struct entry *e = lookup(key, 1);
	const optstruct *o = m->options;
	const int kind = gpemu_host_kind_of(m->covariance_fn);
	if (!kind) { fprintf(stderr, "%s: unknown covariance function (no device kernel)\n", where); gpemu_host_exit(EXIT_FAILURE); }
	const unsigned long long xsum = sum_matrix(m->xmodel), ysum = sum_vector(m->training_vector);
	if (changed) *changed = 0;
	if (e->xdata != m->xmodel->data || e->xsum != xsum || e->N != o->nmodel_points || e->d != o->nparams || e->kind != kind ||
	    e->order != o->regression_order) {
		double *X = pack_matrix(m->xmodel), *y = pack_vector(m->training_vector);
		int rc = gpemu_set_model(e->ctx, kind, o->regression_order, o->nmodel_points, o->nparams, X, y);
		free(X); free(y);
		if (rc) die(e->ctx, rc, where);
		e->xdata = m->xmodel->data; e->ydata = m->training_vector->data;
		e->xsum = xsum; e->ysum = ysum;
		e->N = o->nmodel_points; e->d = o->nparams; e->kind = kind; e->order = o->regression_order;
		vcache_clear(&e->vc);
		if (changed) *changed = 1;
	} else if (e->ydata != m->training_vector->data || e->ysum != ysum) {
		double *y = pack_vector(m->training_vector);
		int rc = gpemu_set_training(e->ctx, y);
		free(y);
		if (rc) die(e->ctx, rc, where);
		e->ydata = m->training_vector->data; e->ysum = ysum;
		vcache_clear(&e->vc);
		if (changed) *changed = 1;
	}
	return e;
}

static gpemu_ctx *bind_model(const void *key, modelstruct *m, const char *where)
{
	return bind_model_entry(key, m, where, NULL)->ctx;
}

/* theta_local = [0, theta_less_amp...]  (maxmultimin.c:311-313) */
static double *full_thetas(const gsl_vector *less_amp, int nthetas)
{
	double *t = (double *)malloc(sizeof(double) * (size_t)nthetas);
	t[0] = 0.0;
	for (int i = 1; i < nthetas; i++) t[i] = gsl_vector_get(less_amp, i - 1);
	return t;
}

static void note_not_pd(const char *who, const double *th, int nthetas)
{
	/* the reference dumps the whole matrix to chol-err.dat (maxmultimin.c:327-343); here only the thetas are
	 * written -- the N x N matrix lives in HBM and can be regenerated with makeCovMatrix_fnptr */
	fprintf(stderr, "%s\n", who);
	fprintf(stderr, "trying to cholesky a non postive def matrix, sorry...\n");
	fprintf(stderr, "thetas dumped to chol-err.dat\n");
	FILE *f = fopen("chol-err.dat", "w");
	if (f) {
		fprintf(f, "#thetas: ");
		for (int i = 0; i < nthetas; i++) fprintf(f, "%lf\t", th[i]);
		fprintf(f, "\n");
		fclose(f);
	}
}

/* ---------------------------------------------------------------- lock-step groups
 * The restarts of a search are independent BFGS runs that each ask for one likelihood (+gradient) at a time
 * (maxmultimin.c:82-119; one pthread each in estimate_threaded.c).  A group makes n such threads share ONE device
 * context: a thread deposits its theta and sleeps; the last one to arrive (or to leave) sends all deposited thetas to
 * the device as one lock-step batch (gpemu_loglik_batch / gpemu_loglik_grad_batch) and wakes the others.  The
 * minimiser code is unchanged -- the batching happens underneath evalFnMulti / evalFnGradMulti. */
struct group {
	pthread_mutex_t mu;
	pthread_cond_t cv;
	int nlive, narrived, cap, nthetas;
	int device;                  /* device of the slot the group works for (captured from its creator) */
	unsigned long generation;
	modelstruct *model;
	const void **members;
	int nmembers;
	double *th, *val, *sigma2, *grad;
	int *want_grad, *status, *who;   /* who[slot]: member index of the request deposited in that slot */
	struct vcache *vc;               /* one per member */
	/* what the members' caches were computed FOR: checksums of the model's design and training values and the layer's
	 * mode flags at the last device round (a caller may rewrite the buffers in place or switch modes while the group lives) */
	unsigned long long vc_xsum, vc_ysum;
	int vc_modes, vc_bound;
	struct group *next;
};
static struct group *g_groups = NULL;

static struct group *find_group(const void *params)
{
	struct group *G;
	pthread_mutex_lock(&g_lock);
	for (G = g_groups; G; G = G->next) {
		for (int i = 0; i < G->nmembers; i++)
			if (G->members[i] == params) { pthread_mutex_unlock(&g_lock); return G; }
	}
	pthread_mutex_unlock(&g_lock);
	return NULL;
}

void *gpemu_host_group_create(struct estimate_thetas_params **members, int n)
{
	if (n < 1 || n > GPEMU_MAX_BATCH) return NULL;
	struct group *G = (struct group *)calloc(1, sizeof *G);
	pthread_mutex_init(&G->mu, NULL);
	pthread_cond_init(&G->cv, NULL);
	G->nlive = n; G->cap = n; G->nmembers = n;
	G->device = gpemu_host_device();
	G->nthetas = members[0]->options->nthetas;
	G->model = members[0]->the_model;
	G->members = (const void **)malloc(sizeof(void *) * (size_t)n);
	for (int i = 0; i < n; i++) G->members[i] = members[i];
	G->th = (double *)calloc((size_t)n * G->nthetas, sizeof(double));
	G->val = (double *)calloc((size_t)n, sizeof(double));
	G->sigma2 = (double *)calloc((size_t)n, sizeof(double));
	G->grad = (double *)calloc((size_t)n * G->nthetas, sizeof(double));
	G->want_grad = (int *)calloc((size_t)n, sizeof(int));
	G->status = (int *)calloc((size_t)n, sizeof(int));
	G->who = (int *)calloc((size_t)n, sizeof(int));
	G->vc = (struct vcache *)calloc((size_t)n, sizeof(struct vcache));
	pthread_mutex_lock(&g_lock);
	G->next = g_groups;
	g_groups = G;
	pthread_mutex_unlock(&g_lock);
	return G;
}

void gpemu_host_group_destroy(void *group)
{
	struct group *G = (struct group *)group;
	if (!G) return;
	pthread_mutex_lock(&g_lock);
	for (struct group **pp = &g_groups; *pp; pp = &(*pp)->next)
		if (*pp == G) { *pp = G->next; break; }
	pthread_mutex_unlock(&g_lock);
	gpemu_host_release(G);                       /* the shared device context is keyed by the group */
	pthread_mutex_destroy(&G->mu);
	pthread_cond_destroy(&G->cv);
	free(G->members); free(G->th); free(G->val); free(G->sigma2); free(G->grad); free(G->want_grad); free(G->status);
	free(G->who); free(G->vc);
	free(G);
}

/* GPEMU_FAULT_ENQUEUE=K (test hook): the K-th value+gradient round of the process's lock-step groups (1-based) reports a
 * device error instead of enqueueing -- the way to see what a device failure in the middle of a threaded search does to the
 * process (message, status 1: fatal.c) without having to break a GPU */
static int fault_injected(void)
{
	static long seen = 0;
	const char *e = getenv("GPEMU_FAULT_ENQUEUE");
	if (!e || atol(e) < 1) return 0;
	return __sync_add_and_fetch(&seen, 1) == atol(e);
}

/* all live members have deposited a request: the value+gradient requests go to the device as ONE lock-step batch; the
 * value-only requests of the same round (rare: the two end-of-run calls are answered from the members' caches) as a
 * second one, enqueued right behind it on the same stream before the first is collected.  Results into the slots and
 * into the members' caches.  Called with G->mu held. */
static void group_run_round(struct group *G)
{
	const int n = G->narrived, nt = G->nthetas;
	const int saved_device = tls_device;          /* the round runs on whichever member arrived last: use the group's device */
	tls_device = G->device;
	int changed = 0;
	struct entry *bound = bind_model_entry(G, G->model, "lock-step group", &changed);
	gpemu_ctx *ctx = bound->ctx;
	tls_device = saved_device;
	if (changed || !G->vc_bound || G->vc_modes != gpemu_host_modes())
		for (int i = 0; i < G->nmembers; i++) vcache_clear(&G->vc[i]);
	G->vc_xsum = bound->xsum; G->vc_ysum = bound->ysum; G->vc_modes = gpemu_host_modes(); G->vc_bound = 1;
	double *th = (double *)malloc(sizeof(double) * (size_t)n * nt * 2);
	double *val = (double *)malloc(sizeof(double) * (size_t)n), *s2 = (double *)malloc(sizeof(double) * (size_t)n);
	double *gr = (double *)malloc(sizeof(double) * (size_t)n * nt);
	int *st = (int *)malloc(sizeof(int) * (size_t)n), *idx = (int *)malloc(sizeof(int) * (size_t)n * 2);
	int m[2] = {0, 0};                            /* pass 1: value + gradient, pass 0: value only */
	for (int pass = 1; pass >= 0; pass--)
		for (int i = 0; i < n; i++)
			if (G->want_grad[i] == pass) {
				memcpy(th + ((size_t)pass * n + m[pass]) * nt, G->th + (size_t)i * nt, sizeof(double) * (size_t)nt);
				idx[pass * n + m[pass]++] = i;
			}
	/* the larger batch first: the context sizes its per-batch result slots on the way up, and growing them between the
	 * two enqueues would drop the first batch's entry of the result ring */
	const int first = m[1] >= m[0] ? 1 : 0;
	int rc = 0;
	for (int k = 0; k < 2 && !rc; k++) {
		const int pass = k == 0 ? first : 1 - first;
		if (!m[pass]) continue;
		if (pass && fault_injected()) die(NULL, GPEMU_ERR_HIP, "lock-step group (GPEMU_FAULT_ENQUEUE: injected device error)");
		rc = pass ? gpemu_loglik_grad_batch_enqueue(ctx, m[1], th + (size_t)n * nt, nt) : gpemu_loglik_batch_enqueue(ctx, m[0], th, nt);
	}
	if (rc) die(ctx, rc, "lock-step group");
	for (int k = 0; k < 2; k++) {
		const int pass = k == 0 ? first : 1 - first;
		if (!m[pass]) continue;
		const int back = (k == 0 && m[1 - pass]) ? 1 : 0;       /* the first of two enqueued batches sits one entry back */
		rc = pass ? gpemu_loglik_grad_batch_collect_back(ctx, back, m[1], val, s2, NULL, gr, NULL, st)
		          : gpemu_loglik_batch_collect_back(ctx, back, m[0], val, s2, NULL, NULL, NULL, NULL, st);
		if (rc) die(ctx, rc, "lock-step group");
		for (int j = 0; j < m[pass]; j++) {
			const int i = idx[pass * n + j];
			G->val[i] = val[j]; G->sigma2[i] = s2[j]; G->status[i] = st[j];
			if (pass == 1) memcpy(G->grad + (size_t)i * nt, gr + (size_t)j * (nt - 1), sizeof(double) * (size_t)(nt - 1));
			vcache_put(&G->vc[G->who[i]], G->th + (size_t)i * nt, nt, val[j], s2[j], st[j]);
		}
	}
	COUNT(g_n_valgrad, 1, m[1]);
	COUNT(g_n_value, 0, m[0]);
	COUNT(g_n_rounds, 3, 1);
	COUNT(g_n_round_elems, 4, n);
	free(th); free(val); free(s2); free(gr); free(st); free(idx);
}

static int group_member_index(const struct group *G, const void *params)
{
	for (int i = 0; i < G->nmembers; i++)
		if (G->members[i] == params) return i;
	return 0;
}

/* one member's request; returns the device status of ITS element */
static int group_eval(struct group *G, const void *params, const double *th, int want_grad, double *val, double *sigma2, double *grad)
{
	/* (the checksums of a cache answer are taken BEFORE the group's mutex: N*(d+1) doubles per value-only request, and the
	 * members of a group would otherwise queue behind each other's sums) */
	unsigned long long xs = 0, ys = 0;
	if (!want_grad) { xs = sum_matrix(G->model->xmodel); ys = sum_vector(G->model->training_vector); }
	pthread_mutex_lock(&G->mu);
	const int me = group_member_index(G, params);
	if (!want_grad) {
		/* value / sigma^2 at a point this member has already evaluated: no device work, no round -- provided the cached
		 * numbers still belong to the model as it is NOW (same checksums of design and training values, same modes: the
		 * check bind_model_entry makes in front of every device round is made here in front of every cache answer) */
		if (G->vc_bound && (G->vc_modes != gpemu_host_modes() || G->vc_xsum != xs || G->vc_ysum != ys)) {
			for (int i = 0; i < G->nmembers; i++) vcache_clear(&G->vc[i]);
			G->vc_bound = 0;
		}
		const int c = G->vc_bound ? vcache_find(&G->vc[me], th, G->nthetas) : -1;
		if (c >= 0) {
			if (val) *val = G->vc[me].val[c];
			if (sigma2) *sigma2 = G->vc[me].sigma2[c];
			const int st = G->vc[me].status[c];
			pthread_mutex_unlock(&G->mu);
			COUNT(g_n_cached, 2, 1);
			return st;
		}
	}
	const int slot = G->narrived++;
	memcpy(G->th + (size_t)slot * G->nthetas, th, sizeof(double) * (size_t)G->nthetas);
	G->want_grad[slot] = want_grad ? 1 : 0;
	G->who[slot] = me;
	const unsigned long gen = G->generation;
	if (G->narrived == G->nlive) {
		group_run_round(G);
		G->narrived = 0;
		G->generation++;
		pthread_cond_broadcast(&G->cv);
	} else {
		while (G->generation == gen) pthread_cond_wait(&G->cv, &G->mu);
	}
	if (val) *val = G->val[slot];
	if (sigma2) *sigma2 = G->sigma2[slot];
	if (grad) memcpy(grad, G->grad + (size_t)slot * G->nthetas, sizeof(double) * (size_t)(G->nthetas - 1));
	const int st = G->status[slot];
	pthread_mutex_unlock(&G->mu);
	return st;
}

/* a member's thread has finished its restarts: the others no longer wait for it */
void gpemu_host_group_leave(void *params)
{
	struct group *G = find_group(params);
	if (!G) return;
	pthread_mutex_lock(&G->mu);
	G->nlive--;
	if (G->nlive > 0 && G->narrived == G->nlive) {
		group_run_round(G);
		G->narrived = 0;
		G->generation++;
		pthread_cond_broadcast(&G->cv);
	}
	pthread_mutex_unlock(&G->mu);
	pthread_mutex_lock(&g_lock);
	for (int i = 0; i < G->nmembers; i++)
		if (G->members[i] == params) G->members[i] = NULL;
	pthread_mutex_unlock(&g_lock);
}

/* value / sigma^2 of a caller with a context of its own: from its cache when this theta has just been evaluated */
static int own_value(struct estimate_thetas_params *params, const char *where, const double *th, int nthetas, double *val,
                     double *sigma2, gpemu_ctx **ctx_out)
{
	struct entry *e = bind_model_entry(params, params->the_model, where, NULL);
	*ctx_out = e->ctx;
	const int c = vcache_find(&e->vc, th, nthetas);
	if (c >= 0) {
		if (val) *val = e->vc.val[c];
		if (sigma2) *sigma2 = e->vc.sigma2[c];
		COUNT(g_n_cached, 2, 1);
		return e->vc.status[c];
	}
	double v = GSL_NAN, s2 = GSL_NAN;
	int info = 0;
	const int rc = gpemu_loglik(e->ctx, th, nthetas, &v, &s2, NULL, NULL, NULL, &info);
	COUNT(g_n_value, 0, 1);
	if (rc == GPEMU_OK || rc == GPEMU_ERR_NOT_PD) vcache_put(&e->vc, th, nthetas, v, s2, rc);
	if (val) *val = v;
	if (sigma2) *sigma2 = s2;
	return rc;
}

/* libEmu/maxmultimin.c:288-394 */
double evalFnMulti(const gsl_vector *theta_vec_less_amp, void *params_in)
{
	struct estimate_thetas_params *params = (struct estimate_thetas_params *)params_in;
	const int nthetas = params->options->nthetas;
	struct group *G = find_group(params);
	gpemu_ctx *ctx = NULL;
	double *th = full_thetas(theta_vec_less_amp, nthetas);
	double val = GSL_NAN;
	int rc = G ? group_eval(G, params, th, 0, &val, NULL, NULL) : own_value(params, "evalFnMulti", th, nthetas, &val, NULL, &ctx);
	if (rc == GPEMU_ERR_NOT_PD) {
		note_not_pd("evalFnMulti", th, nthetas);
		val = GSL_NAN;
	} else if (rc == GPEMU_ERR_REGRESSION) {
		fprintf(stderr, "# err: estimateBeta\n# trying to cholesky a non postive def matrix, sorry...\n");
		gpemu_host_exit(1);                                          /* regression.c:134-160 */
	} else if (rc) {
		die(ctx, rc, "evalFnMulti");
	}
	free(th);
	return val;
}

/* evalFnMulti over a list of theta rows (npts x (nthetas-1), each {nugget, lengths...}): the rows are independent
 * evaluations of one model (libRbind/rbind.c:626-724 callEvalLhoodList; the restarts of maxWithMultiMin), so they
 * go to the device in lock-step batches (gpemu_loglik_batch) instead of one factorisation at a time.  answer[i] is
 * what evalFnMulti would have returned for row i. */
void evalFnMultiList(const gsl_matrix *theta_rows_less_amp, void *params_in, double *answer)
{
	struct estimate_thetas_params *params = (struct estimate_thetas_params *)params_in;
	const int nthetas = params->options->nthetas;
	const int npts = (int)theta_rows_less_amp->size1;
	int maxb = 16;                                         /* workspace: maxb * (N+64) * N * 8 bytes per context */
	const char *e = getenv("GPEMU_HOST_BATCH");
	if (e && atoi(e) >= 1) maxb = atoi(e) > GPEMU_MAX_BATCH ? GPEMU_MAX_BATCH : atoi(e);
	/* the blocks of the list alternate between TWO device contexts, each block enqueued before the previous block of the
	 * same context is collected: one block's panel chain and the host's finishing run beside the other's trailing updates
	 * (what bench.py's headline region does with its two contexts).  A short list stays on one context. */
	const int nblocks = (npts + maxb - 1) / maxb;
	const int nctx = nblocks >= 2 ? 2 : 1;
	gpemu_ctx *ctx[2];
	ctx[0] = bind_model(params, params->the_model, "evalFnMultiList");
	ctx[1] = nctx > 1 ? bind_model((const char *)params + 1, params->the_model, "evalFnMultiList") : NULL;
	double *th = (double *)calloc((size_t)nctx * maxb * nthetas, sizeof(double));
	int *status = (int *)malloc(sizeof(int) * (size_t)maxb);
	int pend_p0[2] = {-1, -1}, pend_nb[2] = {0, 0};
	for (int blk = 0; blk <= nblocks + nctx - 1; blk++) {
		const int k = blk % nctx;
		if (pend_p0[k] >= 0) {                             /* collect this context's previous block */
			const int p0 = pend_p0[k], nb = pend_nb[k];
			const double *tk = th + (size_t)k * maxb * nthetas;
			int rc = gpemu_loglik_batch_collect_back(ctx[k], 0, nb, answer + p0, NULL, NULL, NULL, NULL, NULL, status);
			if (rc) die(ctx[k], rc, "evalFnMultiList");
			for (int b = 0; b < nb; b++) {
				if (status[b] == GPEMU_ERR_NOT_PD) {
					note_not_pd("evalFnMulti", tk + (size_t)b * nthetas, nthetas);
					answer[p0 + b] = GSL_NAN;
				} else if (status[b] == GPEMU_ERR_REGRESSION) {
					fprintf(stderr, "# err: estimateBeta\n# trying to cholesky a non postive def matrix, sorry...\n");
					gpemu_host_exit(1);                                      /* regression.c:134-160 */
				} else if (status[b]) {
					die(ctx[k], status[b], "evalFnMultiList");
				}
			}
			pend_p0[k] = -1;
		}
		if (blk >= nblocks) continue;
		const int p0 = blk * maxb, nb = npts - p0 < maxb ? npts - p0 : maxb;
		double *tk = th + (size_t)k * maxb * nthetas;
		for (int b = 0; b < nb; b++) {
			tk[(size_t)b * nthetas] = 0.0;                    /* theta_local[0] = 0 (maxmultimin.c:322) */
			for (int i = 1; i < nthetas; i++)
				tk[(size_t)b * nthetas + i] = gsl_matrix_get(theta_rows_less_amp, p0 + b, i - 1);
		}
		int rc = gpemu_loglik_batch_enqueue(ctx[k], nb, tk, nthetas);
		if (rc) die(ctx[k], rc, "evalFnMultiList");
		COUNT(g_n_value, 0, nb);
		pend_p0[k] = p0; pend_nb[k] = nb;
	}
	free(status); free(th);
}

static void grad_failure(gpemu_ctx *ctx, int rc, const double *th, int nthetas)
{
	if (rc == GPEMU_ERR_NOT_PD) {
		note_not_pd("gradFnMulti", th, nthetas);
		gpemu_host_exit(EXIT_FAILURE);                               /* maxmultimin.c:495 */
	}
	die(ctx, rc, "gradFnMulti");
}

/* libEmu/maxmultimin.c:416-550 */
void gradFnMulti(const gsl_vector *theta_vec_less_amp, void *params_in, gsl_vector *grad_vec)
{
	struct estimate_thetas_params *params = (struct estimate_thetas_params *)params_in;
	const int nthetas = params->options->nthetas;
	struct group *G = find_group(params);
	gpemu_ctx *ctx = G ? NULL : bind_model(params, params->the_model, "gradFnMulti");
	double *th = full_thetas(theta_vec_less_amp, nthetas);
	double *g = (double *)malloc(sizeof(double) * (size_t)(nthetas - 1));
	int info = 0;
	int rc = G ? group_eval(G, params, th, 1, NULL, NULL, g) : gpemu_grad(ctx, th, nthetas, g, &info);
	if (!G) COUNT(g_n_valgrad, 1, 1);
	if (rc) grad_failure(ctx, rc, th, nthetas);
	for (int i = 0; i < nthetas - 1; i++) gsl_vector_set(grad_vec, i, g[i]);
	free(g); free(th);
}

/* libEmu/maxmultimin.c:615-618, one factorisation instead of two */
void evalFnGradMulti(const gsl_vector *theta_vec, void *params_in, double *fnval, gsl_vector *grad_vec)
{
	struct estimate_thetas_params *params = (struct estimate_thetas_params *)params_in;
	const int nthetas = params->options->nthetas;
	struct group *G = find_group(params);
	struct entry *en = G ? NULL : bind_model_entry(params, params->the_model, "evalFnGradMulti", NULL);
	gpemu_ctx *ctx = en ? en->ctx : NULL;
	double *th = full_thetas(theta_vec, nthetas);
	double *g = (double *)malloc(sizeof(double) * (size_t)(nthetas - 1));
	int info = 0;
	double s2 = GSL_NAN;
	int rc = G ? group_eval(G, params, th, 1, fnval, NULL, g) : gpemu_loglik_grad(ctx, th, nthetas, fnval, &s2, NULL, g, &info);
	if (en) {
		COUNT(g_n_valgrad, 1, 1);
		if (rc == GPEMU_OK || rc == GPEMU_ERR_NOT_PD) vcache_put(&en->vc, th, nthetas, *fnval, s2, rc);
	}
	if (rc == GPEMU_ERR_NOT_PD) {
		/* The reference would return GSL_NAN from evalFnMulti and then gpemu_host_exit(EXIT_FAILURE) inside gradFnMulti
		 * (maxmultimin.c:349,495) -- a line-search trial point that is numerically not positive definite kills
		 * the whole training run.  Here the pair reports NaN value AND NaN gradient and the line search backs off. */
		note_not_pd("evalFnGradMulti", th, nthetas);
		*fnval = GSL_NAN;
		for (int i = 0; i < nthetas - 1; i++) gsl_vector_set(grad_vec, i, GSL_NAN);
		free(g); free(th);
		return;
	}
	if (rc) grad_failure(ctx, rc, th, nthetas);
	for (int i = 0; i < nthetas - 1; i++) gsl_vector_set(grad_vec, i, g[i]);
	free(g); free(th);
}

/* libEmu/maxmultimin.c:148-201: sigma^2 at {nug, lengths...}; NaN if the matrix is not positive definite */
double estimateSigmaFull(gsl_vector *thetas_less_amp, void *params_in)
{
	struct estimate_thetas_params *params = (struct estimate_thetas_params *)params_in;
	const int nthetas = params->options->nthetas;
	struct group *G = find_group(params);
	gpemu_ctx *ctx = NULL;
	double *th = full_thetas(thetas_less_amp, nthetas);
	double s2 = GSL_NAN;
	int rc = G ? group_eval(G, params, th, 0, NULL, &s2, NULL) : own_value(params, "estimateSigmaFull", th, nthetas, NULL, &s2, &ctx);
	if (rc == GPEMU_ERR_NOT_PD) { note_not_pd("estSigmaFull", th, nthetas); s2 = GSL_NAN; }
	else if (rc) die(ctx, rc, "estimateSigmaFull");
	free(th);
	return s2;
}

/* ---------------------------------------------------------------- prediction */

/* emulator_struct.c:13-37: factor once, keep everything the sweep needs resident in HBM.  cinverse / beta /
 * h_matrix are filled on the host as well because they are public fields of the struct. */
/* Would alloc_emulator_struct refuse this model?  1: the covariance matrix at model->thetas (amplitude included) is not
 * numerically positive definite; 0: it factors.  The search factors the matrix at amplitude 1 (maxmultimin.c:311) and the
 * emulator the one at the estimated amplitude with the same nugget: noise-free training data drive the unbounded BFGS of the
 * reference towards nugget -> 0, where the first can still factor and the second no longer does.  The CLI warns at
 * training time instead of leaving the discovery to interactive_mode. */
int gpemu_host_emulator_setup_fails(modelstruct *model)
{
	int key = 0;                                                 /* (its address is the key) a context of its own, released below */
	gpemu_ctx *ctx = bind_model(&key, model, "gpemu_host_emulator_setup_fails");
	double *th = pack_vector(model->thetas);
	double *beta = (double *)malloc(sizeof(double) * (size_t)model->options->nregression_fns);
	int info = 0;
	const int rc = gpemu_predict_setup(ctx, th, model->options->nthetas, beta, &info);
	free(th); free(beta);
	gpemu_host_release(&key);
	return rc == GPEMU_ERR_NOT_PD;
}

emulator_struct *alloc_emulator_struct(modelstruct *model)
{
	return gpemu_host_alloc_emulator(model, getenv("GPEMU_SKIP_CINVERSE") == NULL);
}

/* alloc_emulator_struct for the n components of a multi-output model (multivar_support.c:30-52 loops it) with ONE lock-step
 * factorisation for all of them (gpemu_predict_setup_batch): same design, covariance function and regression order, a
 * training vector and thetas of its own each.  out[c] is what alloc_emulator_struct(models[c]) returns, bit for bit.  The
 * contexts live on the calling thread's device. */
void gpemu_host_alloc_emulators(modelstruct **models, int n, int fill_cinverse, emulator_struct **out)
{
	const int trace = getenv("GPEMU_SETUP_TRACE") != NULL;
	const double t0 = now_s();
	gpemu_ctx **ctxs = (gpemu_ctx **)malloc(sizeof(gpemu_ctx *) * (size_t)n);
	const int nthetas = models[0]->options->nthetas, nreg = models[0]->options->nregression_fns;
	double *th = (double *)malloc(sizeof(double) * (size_t)n * nthetas), *beta = (double *)malloc(sizeof(double) * (size_t)n * nreg);
	int *status = (int *)calloc((size_t)n, sizeof(int));
	for (int c = 0; c < n; c++) {
		modelstruct *model = models[c];
		emulator_struct *e = (emulator_struct *)malloc(sizeof(emulator_struct));
		e->nparams = model->options->nparams;
		e->nmodel_points = model->options->nmodel_points;
		e->nregression_fns = model->options->nregression_fns;
		e->nthetas = model->options->nthetas;
		e->model = model;
		e->cinverse = gsl_matrix_alloc(e->nmodel_points, e->nmodel_points);
		e->beta_vector = gsl_vector_alloc(e->nregression_fns);
		e->h_matrix = gsl_matrix_alloc(e->nmodel_points, e->nregression_fns);
		out[c] = e;
		ctxs[c] = bind_model(e, model, "alloc_multi_emulator");
		if (e->nthetas != nthetas || e->nregression_fns != nreg)
			gpemu_host_fatal("alloc_multi_emulator: the components of a multi-output model share covariance function and regression order\n");
		for (int t = 0; t < nthetas; t++) th[(size_t)c * nthetas + t] = gsl_vector_get(model->thetas, t);
	}
	const double t1 = now_s();
	int rc = gpemu_predict_setup_batch(ctxs, n, th, nthetas, beta, NULL, status);
	if (rc == GPEMU_ERR_NOT_PD) {
		fprintf(stderr, "trying to cholesky a non postive def matrix, in emulate-fns.c sorry...\n");
		gpemu_host_exit(1);                               /* emulate-fns.c:282-285 */
	}
	if (rc) die(ctxs[0], rc, "alloc_multi_emulator");
	const double t2 = now_s();
	for (int c = 0; c < n; c++) {
		emulator_struct *e = out[c];
		for (int a = 0; a < nreg; a++) gsl_vector_set(e->beta_vector, a, beta[(size_t)c * nreg + a]);
		makeHMatrix_fnptr(e->h_matrix, e->model->xmodel, e->nmodel_points, e->nparams, e->nregression_fns, e->model->makeHVector);
		if (fill_cinverse) {
			rc = gpemu_get_cinverse(ctxs[c], e->cinverse->data);
			if (rc) die(ctxs[c], rc, "alloc_multi_emulator(cinverse)");
		}
	}
	if (trace)
		fprintf(stderr, "# setup trace (%d components, one lock-step batch): host_allocs+contexts+uploads %.3f ms  predict_setup_batch %.3f ms  "
		        "h_matrices+cinverse %.3f ms\n", n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (now_s() - t2) * 1e3);
	free(ctxs); free(th); free(beta); free(status);
}

/* What a process pays once before its first device result (the HIP runtime's start, the loading of the device code, tables)
 * paid on a thread of its own while the caller reads its input: gpemu_host_warm_start() starts it (GPEMU_WARM_START=0: not),
 * gpemu_host_warm_wait() waits for it -- every entry of this file that makes a context does. */
static pthread_t g_warm_thread;
static volatile int g_warm_state = 0;           /* 0: none, 1: running, 2: joined */
static pthread_mutex_t g_warm_mu = PTHREAD_MUTEX_INITIALIZER;

static void *warm_main(void *arg)
{
	/* (the device is looked up HERE: counting the devices is already a call into the HIP runtime and starts it -- on the
	 * caller's thread that would be the very wait this thread exists to take off it.  A failure shows up again, with its
	 * message, at the first real call.) */
	(void)arg;
	(void)gpemu_warm_start(gpemu_host_device());
	return NULL;
}

void gpemu_host_warm_start(void)
{
	const char *e = getenv("GPEMU_WARM_START");
	if (e && atoi(e) == 0) return;
	pthread_mutex_lock(&g_warm_mu);
	if (g_warm_state == 0 && pthread_create(&g_warm_thread, NULL, warm_main, NULL) == 0) g_warm_state = 1;
	pthread_mutex_unlock(&g_warm_mu);
}

void gpemu_host_warm_wait(void)
{
	if (g_warm_state != 1) return;
	pthread_mutex_lock(&g_warm_mu);
	if (g_warm_state == 1) { pthread_join(g_warm_thread, NULL); g_warm_state = 2; }
	pthread_mutex_unlock(&g_warm_mu);
}

emulator_struct *gpemu_host_alloc_emulator(modelstruct *model, int fill_cinverse)
{
	const double t0 = now_s();
	emulator_struct *e = (emulator_struct *)malloc(sizeof(emulator_struct));
	e->nparams = model->options->nparams;
	e->nmodel_points = model->options->nmodel_points;
	e->nregression_fns = model->options->nregression_fns;
	e->nthetas = model->options->nthetas;
	e->model = model;
	e->cinverse = gsl_matrix_alloc(e->nmodel_points, e->nmodel_points);
	e->beta_vector = gsl_vector_alloc(e->nregression_fns);
	e->h_matrix = gsl_matrix_alloc(e->nmodel_points, e->nregression_fns);

	/* GPEMU_SETUP_TRACE=1: host timestamps of the phases below on stderr (where an emulator's start-up time goes) */
	const int trace = getenv("GPEMU_SETUP_TRACE") != NULL;
	struct entry *en0 = lookup(e, 1);                 /* device context (the first one of a process also starts the HIP runtime) */
	(void)en0;
	const double t1 = now_s();
	gpemu_ctx *ctx = bind_model(e, model, "alloc_emulator_struct");
	const double t2 = now_s();
	double *th = pack_vector(model->thetas);
	double *beta = (double *)malloc(sizeof(double) * (size_t)e->nregression_fns);
	int info = 0;
	int rc = gpemu_predict_setup(ctx, th, e->nthetas, beta, &info);
	const double t3 = now_s();
	if (rc == GPEMU_ERR_NOT_PD) {
		fprintf(stderr, "trying to cholesky a non postive def matrix, in emulate-fns.c sorry...\n");
		gpemu_host_exit(1);                                          /* emulate-fns.c:282-285 */
	}
	if (rc) die(ctx, rc, "alloc_emulator_struct");
	for (int a = 0; a < e->nregression_fns; a++) gsl_vector_set(e->beta_vector, a, beta[a]);
	makeHMatrix_fnptr(e->h_matrix, model->xmodel, e->nmodel_points, e->nparams, e->nregression_fns, model->makeHVector);
	if (fill_cinverse) {
		rc = gpemu_get_cinverse(ctx, e->cinverse->data);
		if (rc) die(ctx, rc, "alloc_emulator_struct(cinverse)");
	}
	free(th); free(beta);
	if (trace)
		fprintf(stderr, "# setup trace: host_allocs+context %.3f ms  upload_model %.3f ms  predict_setup %.3f ms  h_matrix+cinverse %.3f ms\n",
		        (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (now_s() - t3) * 1e3);
	return e;
}

void free_emulator_struct(emulator_struct *e)
{
	if (!e) return;
	gpemu_host_release(e);
	gsl_matrix_free(e->cinverse);
	gsl_matrix_free(e->h_matrix);
	gsl_vector_free(e->beta_vector);
	free(e);
}

void emulate_points(emulator_struct *e, gsl_matrix *points, double *mean, double *variance)
{
	struct entry *en = lookup(e, 0);
	if (!en) { fprintf(stderr, "emulate_point: emulator_struct was not made by alloc_emulator_struct\n"); gpemu_host_exit(EXIT_FAILURE); }
	double *q = pack_matrix(points);
	int rc = gpemu_predict_batch(en->ctx, (int)points->size1, q, mean, variance);
	free(q);
	if (rc) die(en->ctx, rc, "emulate_point");
}

/* the two halves of emulate_points: all components of a multi-output emulator enqueue, then all collect */
void emulate_points_enqueue(emulator_struct *e, gsl_matrix *points)
{
	struct entry *en = lookup(e, 0);
	if (!en) { fprintf(stderr, "emulate_point: emulator_struct was not made by alloc_emulator_struct\n"); gpemu_host_exit(EXIT_FAILURE); }
	double *q = pack_matrix(points);
	int rc = gpemu_predict_batch_enqueue(en->ctx, (int)points->size1, q);
	free(q);
	if (rc) die(en->ctx, rc, "emulate_point");
}

void emulate_points_collect(emulator_struct *e, int npoints, double *mean, double *variance)
{
	struct entry *en = lookup(e, 0);
	if (!en) { fprintf(stderr, "emulate_point: emulator_struct was not made by alloc_emulator_struct\n"); gpemu_host_exit(EXIT_FAILURE); }
	int rc = gpemu_predict_batch_collect(en->ctx, npoints, mean, variance);
	if (rc) die(en->ctx, rc, "emulate_point");
}

/* emulator_struct.c:124-143 */
void emulate_point(emulator_struct *e, gsl_vector *point, double *mean, double *variance)
{
	gsl_matrix view;
	double *q = pack_vector(point);
	view.size1 = 1; view.size2 = point->size; view.tda = point->size; view.data = q; view.block = NULL; view.owner = 0;
	emulate_points(e, &view, mean, variance);
	free(q);
}

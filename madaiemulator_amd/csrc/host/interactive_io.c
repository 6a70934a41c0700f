/*
 * interactive_io.c -- the request/response loop of interactive_mode (src/interactive_emulator.c:398-440 of the
 * reference) as a three-stage pipeline, so that a stream of query points keeps the device busy:
 *
 *   reader thread   stdin -> points: tokenising + strtod (text, the reference's fscanf("%lf%*c") framing, :420) or raw
 *                   doubles (binary, the reference's BINARY_INTERACTIVE_MODE framing: fread of sizeof(double), :418)
 *   caller's thread the batch through `fn` (emulate_points_multi: every PCA component's device context at once)
 *   writer thread   results -> stdout: "%.17f\n" per number (:434-435) or raw doubles (:431-432), one flush per batch
 *
 * The reference answers one point per loop turn and flushes after it (:440).  Here the points that are ALREADY waiting on
 * stdin are answered as one device batch: the first point of a batch is waited for, further ones are taken only while
 * input is immediately available; when input runs dry and the device stage is idle the batch goes out at once (a lone
 * point -- an MCMC driver that waits for each answer -- is answered immediately), while the device stage is busy the
 * batch keeps growing.  Results leave in input order.  The strtod / "%.17f" conversions of a large batch are dealt to a
 * few helper threads (they are the bound of the text protocol: ~100 ns per number read, ~200 ns per number written).
 */
#define _GNU_SOURCE
#include <errno.h>
#include <poll.h>
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/eventfd.h>
#include <time.h>
#include <unistd.h>
#include "libemu.h"

#define IO_BATCH_MAX 16384            /* points per device batch */
#define IO_SLOTS 4                    /* batches in flight between the three stages */
#define IO_RAW ((size_t)1 << 22)      /* raw input buffer */
#define IO_HELPERS_MAX 7
#define IO_PAR_MIN 4096               /* numbers in a batch from which the helpers are woken */

struct io_slot { int np, last; double *pts, *mean, *var; };

struct io_queue { struct io_slot *item[IO_SLOTS]; int head, count; pthread_mutex_t mu; pthread_cond_t cv; };

static void q_init(struct io_queue *q) { memset(q, 0, sizeof *q); pthread_mutex_init(&q->mu, NULL); pthread_cond_init(&q->cv, NULL); }
static void q_push(struct io_queue *q, struct io_slot *s)
{
	pthread_mutex_lock(&q->mu);
	q->item[(q->head + q->count) % IO_SLOTS] = s;        /* (at most IO_SLOTS slots exist: never full) */
	q->count++;
	pthread_cond_signal(&q->cv);
	pthread_mutex_unlock(&q->mu);
}
static struct io_slot *q_pop(struct io_queue *q)
{
	pthread_mutex_lock(&q->mu);
	while (q->count == 0) pthread_cond_wait(&q->cv, &q->mu);
	struct io_slot *s = q->item[q->head];
	q->head = (q->head + 1) % IO_SLOTS;
	q->count--;
	pthread_mutex_unlock(&q->mu);
	return s;
}

/* ---- fork/join over a few persistent helper threads: fn(arg, part, nparts), part 0 on the calling thread ---- */
struct io_pool;
struct io_helper { struct io_pool *pool; int idx; pthread_t tid; };
struct io_pool {
	int nhelp, stop, left;
	unsigned gen;
	void (*fn)(void *, int, int);
	void *arg;
	pthread_mutex_t mu;
	pthread_cond_t go, done;
	struct io_helper h[IO_HELPERS_MAX];
};

static void *pool_main(void *a)
{
	struct io_helper *me = (struct io_helper *)a;
	struct io_pool *p = me->pool;
	unsigned seen = 0;
	for (;;) {
		pthread_mutex_lock(&p->mu);
		while (!p->stop && p->gen == seen) pthread_cond_wait(&p->go, &p->mu);
		if (p->stop) { pthread_mutex_unlock(&p->mu); return NULL; }
		seen = p->gen;
		void (*fn)(void *, int, int) = p->fn;
		void *arg = p->arg;
		pthread_mutex_unlock(&p->mu);
		fn(arg, me->idx + 1, p->nhelp + 1);
		pthread_mutex_lock(&p->mu);
		if (--p->left == 0) pthread_cond_signal(&p->done);
		pthread_mutex_unlock(&p->mu);
	}
}

static void pool_start(struct io_pool *p, int nhelp)
{
	memset(p, 0, sizeof *p);
	pthread_mutex_init(&p->mu, NULL);
	pthread_cond_init(&p->go, NULL);
	pthread_cond_init(&p->done, NULL);
	if (nhelp > IO_HELPERS_MAX) nhelp = IO_HELPERS_MAX;
	for (int i = 0; i < nhelp; i++) {
		p->h[i].pool = p; p->h[i].idx = i;
		if (pthread_create(&p->h[i].tid, NULL, pool_main, &p->h[i])) break;
		p->nhelp++;
	}
}

static void pool_stop(struct io_pool *p)
{
	pthread_mutex_lock(&p->mu);
	p->stop = 1;
	pthread_cond_broadcast(&p->go);
	pthread_mutex_unlock(&p->mu);
	for (int i = 0; i < p->nhelp; i++) pthread_join(p->h[i].tid, NULL);
}

static void pool_run(struct io_pool *p, void (*fn)(void *, int, int), void *arg, int serial)
{
	if (serial || p->nhelp == 0) { fn(arg, 0, 1); return; }
	pthread_mutex_lock(&p->mu);
	p->fn = fn; p->arg = arg; p->left = p->nhelp; p->gen++;
	pthread_cond_broadcast(&p->go);
	pthread_mutex_unlock(&p->mu);
	fn(arg, 0, p->nhelp + 1);
	pthread_mutex_lock(&p->mu);
	while (p->left) pthread_cond_wait(&p->done, &p->mu);
	pthread_mutex_unlock(&p->mu);
}

static double now_s(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* ---- the pipeline ---- */
struct io_state {
	int fd_in, fd_out, efd, d, nout, nprint, binary, nhelp;
	gpemu_points_fn fn;
	void *user;
	struct io_queue free_q, work_q, out_q;
	int inflight;                       /* batches handed to the device stage and not finished (atomic) */
	int write_failed;
	struct gpemu_io_stats st;
};

static unsigned char sep_tab[256];

/* tokens of raw[pos, len): offsets of the COMPLETE ones (a token that touches the end of the buffer is complete only at
 * end of input) into off[0 .. maxtok); returns the new read position */
static size_t scan_tokens(const char *raw, size_t pos, size_t len, int eof, uint32_t *off, int maxtok, int *ntok)
{
	int n = 0;
	while (n < maxtok) {
		while (pos < len && sep_tab[(unsigned char)raw[pos]]) pos++;
		size_t e = pos;
		while (e < len && !sep_tab[(unsigned char)raw[e]]) e++;
		if (e == pos || (e == len && !eof)) break;
		off[n++] = (uint32_t)pos;
		pos = e;
	}
	*ntok = n;
	return pos;
}

struct conv_job { const char *raw; const uint32_t *off; int ntok; double *dst; int first_bad; pthread_mutex_t mu; };

static void conv_part(void *a, int part, int nparts)
{
	struct conv_job *j = (struct conv_job *)a;
	const int lo = (int)((long)j->ntok * part / nparts), hi = (int)((long)j->ntok * (part + 1) / nparts);
	int bad = -1;
	for (int i = lo; i < hi; i++) {
		char *endp;
		j->dst[i] = strtod(j->raw + j->off[i], &endp);
		if (endp == j->raw + j->off[i]) { bad = i; break; }      /* not a number: input ends here, as fscanf's would */
	}
	if (bad >= 0) {
		pthread_mutex_lock(&j->mu);
		if (j->first_bad < 0 || bad < j->first_bad) j->first_bad = bad;
		pthread_mutex_unlock(&j->mu);
	}
}

static void *reader_main(void *a)
{
	struct io_state *S = (struct io_state *)a;
	const int d = S->d;
	const long cap = (long)IO_BATCH_MAX * d;
	char *raw = (char *)malloc(IO_RAW + 1);
	uint32_t *off = S->binary ? NULL : (uint32_t *)malloc(sizeof(uint32_t) * (size_t)cap);
	size_t len = 0, pos = 0;
	int eof = 0, input_done = 0;
	struct io_pool pool;
	pool_start(&pool, S->binary ? 0 : S->nhelp);
	while (!input_done) {
		struct io_slot *s = q_pop(&S->free_q);
		long cnt = 0;                                  /* numbers of this batch so far */
		for (;;) {
			const double t0 = now_s();
			if (S->binary) {
				long take = (long)((len - pos) / sizeof(double));
				if (take > cap - cnt) take = cap - cnt;
				memcpy(s->pts + cnt, raw + pos, (size_t)take * sizeof(double));
				pos += (size_t)take * sizeof(double);
				cnt += take;
			} else {
				int ntok = 0;
				raw[len] = 0;
				const size_t npos = scan_tokens(raw, pos, len, eof, off, (int)(cap - cnt), &ntok);
				struct conv_job job = {raw, off, ntok, s->pts + cnt, -1, PTHREAD_MUTEX_INITIALIZER};
				pool_run(&pool, conv_part, &job, ntok < IO_PAR_MIN);
				pos = npos;
				if (job.first_bad >= 0) { cnt += job.first_bad; input_done = 1; }
				else cnt += ntok;
			}
			S->st.parse_seconds += now_s() - t0;
			if (input_done || cnt == cap) break;
			if (eof) { input_done = 1; break; }
			/* more input is needed.  Inside a point (or before the first one) wait for it; with a whole number of points
			 * in hand take only what is there already -- unless the device stage is busy anyway */
			if (cnt > 0 && cnt % d == 0) {
				const int busy = __atomic_load_n(&S->inflight, __ATOMIC_ACQUIRE) > 0;
				struct pollfd pf[2] = {{S->fd_in, POLLIN, 0}, {S->efd, POLLIN, 0}};
				const int pr = poll(pf, 2, busy ? -1 : 0);
				if (pr < 0 && errno != EINTR) { input_done = 1; break; }
				if (!(pr > 0 && (pf[0].revents & (POLLIN | POLLHUP | POLLERR)))) {
					if (pr > 0 && (pf[1].revents & POLLIN)) { uint64_t v; if (read(S->efd, &v, sizeof v) < 0) { /* drained */ } }
					if (__atomic_load_n(&S->inflight, __ATOMIC_ACQUIRE) == 0) break;       /* idle device, nothing waiting: answer now */
					continue;
				}
			}
			if (pos > 0) { memmove(raw, raw + pos, len - pos); len -= pos; pos = 0; }
			if (len >= IO_RAW) { input_done = 1; break; }                                  /* a 4 MB "number": give up like a failed scan */
			const ssize_t n = read(S->fd_in, raw + len, IO_RAW - len);
			if (n < 0 && errno == EINTR) continue;
			if (n <= 0) eof = 1;                                                          /* (tokens still buffered are flushed by the next pass) */
			else len += (size_t)n;
		}
		s->np = (int)(cnt / d);                            /* a trailing partial point is dropped (r < expected_r, :423) */
		s->last = input_done;
		if (s->np > S->st.max_batch) S->st.max_batch = s->np;
		__atomic_add_fetch(&S->inflight, 1, __ATOMIC_ACQ_REL);
		q_push(&S->work_q, s);
	}
	pool_stop(&pool);
	free(raw); free(off);
	return NULL;
}

/* "%.17f\n" of mean and variance, nprint pairs per point (pairs beyond nout are zeros: pca-space output keeps nt pairs) */
struct fmt_job { struct io_state *S; struct io_slot *s; char *buf[IO_HELPERS_MAX + 1]; size_t cap[IO_HELPERS_MAX + 1], len[IO_HELPERS_MAX + 1]; };

static void fmt_part(void *a, int part, int nparts)
{
	struct fmt_job *j = (struct fmt_job *)a;
	const struct io_state *S = j->S;
	const int np = j->s->np, nout = S->nout, nprint = S->nprint;
	const int lo = (int)((long)np * part / nparts), hi = (int)((long)np * (part + 1) / nparts);
	size_t used = 0;
	for (int q = lo; q < hi; q++)
		for (int i = 0; i < nprint; i++) {
			if (j->cap[part] - used < 800) {                    /* two numbers of up to 328 characters each */
				j->cap[part] = j->cap[part] * 2 + 4096;
				j->buf[part] = (char *)realloc(j->buf[part], j->cap[part]);
				if (!j->buf[part]) { perror("realloc"); gpemu_host_exit(EXIT_FAILURE); }
			}
			const double m = i < nout ? j->s->mean[(size_t)q * nout + i] : 0.0, v = i < nout ? j->s->var[(size_t)q * nout + i] : 0.0;
			used += (size_t)snprintf(j->buf[part] + used, j->cap[part] - used, "%.17f\n%.17f\n", m, v);
		}
	j->len[part] = used;
}

static int write_all(int fd, const char *p, size_t n)
{
	while (n > 0) {
		const ssize_t w = write(fd, p, n);
		if (w < 0 && errno == EINTR) continue;
		if (w <= 0) return -1;
		p += w; n -= (size_t)w;
	}
	return 0;
}

static void *writer_main(void *a)
{
	struct io_state *S = (struct io_state *)a;
	struct io_pool pool;
	pool_start(&pool, S->binary ? 0 : S->nhelp);
	struct fmt_job job;
	memset(&job, 0, sizeof job);
	job.S = S;
	double *bin = S->binary ? (double *)malloc(sizeof(double) * 2 * (size_t)IO_BATCH_MAX * S->nprint) : NULL;
	for (;;) {
		struct io_slot *s = q_pop(&S->out_q);
		const int last = s->last;
		if (s->np > 0 && !S->write_failed) {
			const double t0 = now_s();
			if (S->binary) {
				for (int q = 0; q < s->np; q++)
					for (int i = 0; i < S->nprint; i++) {
						bin[2 * ((size_t)q * S->nprint + i)] = i < S->nout ? s->mean[(size_t)q * S->nout + i] : 0.0;
						bin[2 * ((size_t)q * S->nprint + i) + 1] = i < S->nout ? s->var[(size_t)q * S->nout + i] : 0.0;
					}
				if (write_all(S->fd_out, (const char *)bin, sizeof(double) * 2 * (size_t)s->np * S->nprint)) S->write_failed = 1;
			} else {
				job.s = s;
				const int serial = (long)s->np * S->nprint * 2 < IO_PAR_MIN;
				pool_run(&pool, fmt_part, &job, serial);
				const int nparts = serial ? 1 : pool.nhelp + 1;
				for (int p = 0; p < nparts && !S->write_failed; p++)
					if (write_all(S->fd_out, job.buf[p], job.len[p])) S->write_failed = 1;
			}
			S->st.format_seconds += now_s() - t0;
		}
		q_push(&S->free_q, s);
		if (last) break;
	}
	pool_stop(&pool);
	for (int p = 0; p <= IO_HELPERS_MAX; p++) free(job.buf[p]);
	free(bin);
	return NULL;
}

int gpemu_host_interactive_loop(int fd_in, int fd_out, int nparams, int nout, int nprint, int binary, gpemu_points_fn fn,
                                void *user, struct gpemu_io_stats *stats)
{
	if (nparams < 1 || nout < 1 || nprint < nout || !fn) return -1;
	struct io_state *S = (struct io_state *)calloc(1, sizeof *S);
	if (!sep_tab[' ']) { const char *sp = " \t\r\n,;"; for (; *sp; sp++) sep_tab[(unsigned char)*sp] = 1; }
	S->fd_in = fd_in; S->fd_out = fd_out; S->d = nparams; S->nout = nout; S->nprint = nprint; S->binary = binary != 0;
	S->fn = fn; S->user = user;
	S->efd = eventfd(0, EFD_NONBLOCK);
	if (S->efd < 0) { free(S); return -1; }
	{
		/* helper threads per conversion stage: GPEMU_IO_THREADS, default 2 (on top of the stage's own thread) */
		const char *e = getenv("GPEMU_IO_THREADS");
		S->nhelp = e ? atoi(e) : 2;
		if (S->nhelp < 0) S->nhelp = 0;
		if (S->nhelp > IO_HELPERS_MAX) S->nhelp = IO_HELPERS_MAX;
	}
	q_init(&S->free_q); q_init(&S->work_q); q_init(&S->out_q);
	struct io_slot slots[IO_SLOTS];
	for (int i = 0; i < IO_SLOTS; i++) {
		slots[i].np = 0; slots[i].last = 0;
		slots[i].pts = (double *)malloc(sizeof(double) * (size_t)IO_BATCH_MAX * nparams);
		slots[i].mean = (double *)malloc(sizeof(double) * (size_t)IO_BATCH_MAX * nout);
		slots[i].var = (double *)malloc(sizeof(double) * (size_t)IO_BATCH_MAX * nout);
		if (!slots[i].pts || !slots[i].mean || !slots[i].var) { perror("malloc"); gpemu_host_exit(EXIT_FAILURE); }
		q_push(&S->free_q, &slots[i]);
	}
	const double t0 = now_s();
	pthread_t rd, wr;
	if (pthread_create(&rd, NULL, reader_main, S) || pthread_create(&wr, NULL, writer_main, S)) { perror("pthread_create"); gpemu_host_exit(EXIT_FAILURE); }
	for (;;) {
		struct io_slot *s = q_pop(&S->work_q);
		const int last = s->last;
		if (s->np > 0) {
			const double c0 = now_s();
			fn(user, s->np, s->pts, s->mean, s->var);
			S->st.compute_seconds += now_s() - c0;
			S->st.points += s->np;
			S->st.batches++;
		}
		__atomic_sub_fetch(&S->inflight, 1, __ATOMIC_ACQ_REL);
		{ const uint64_t one = 1; if (write(S->efd, &one, sizeof one) < 0) { /* counter saturated: the reader is awake anyway */ } }
		q_push(&S->out_q, s);
		if (last) break;
	}
	pthread_join(rd, NULL);
	pthread_join(wr, NULL);
	S->st.wall_seconds = now_s() - t0;
	if (stats) *stats = S->st;
	const int failed = S->write_failed;
	for (int i = 0; i < IO_SLOTS; i++) { free(slots[i].pts); free(slots[i].mean); free(slots[i].var); }
	close(S->efd);
	free(S);
	return failed ? -2 : 0;
}

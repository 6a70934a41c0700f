/*
 * ranks.c -- the host layer as ONE PROCESS PER GPU (BASELINE.json north_star: "one GPU per shard with a
 * single RCCL gather over xGMI at the end").  Inside one process libEmuMI already spreads its independent work over the
 * GPUs of a node by host threads (device slots, device_bridge.c); this file is the process-per-GPU form of the same two axes:
 *
 *   PCA components   estimate_multi (multivar_support.c:20-28): component c is trained by rank c mod W; one all-gather of
 *                    nthetas doubles per component at the end; rank 0 writes the snapshot.
 *   restart runs     estimate_thetas_threaded (estimate_threaded.c:101-113) for a single-output model: run r of the run
 *                    list belongs to rank r mod W; one all-gather of (best value, its run index, its thetas) per rank
 *                    replaces the mutex-guarded arg-max of estimate_threaded.c:308-313 across ranks.
 *
 * A run's start point and generator depend on (seed, run index) alone and ties go to the lower run index, so W ranks return
 * the thetas ONE process returns, bit for bit, and rank 0's snapshot is the serial one.
 *
 * Ranks are plain processes of ONE node started by the user or a launcher: GPEMU_RANK / GPEMU_WORLD_SIZE name them,
 * GPEMU_LOCAL_RANK (default: the rank) picks the GPU unless GPEMU_DEVICE / GPEMU_DEVICES does, GPEMU_RENDEZVOUS_DIR is a
 * directory every rank can reach.  The gather is RCCL (gpemu_rccl_comm_allgather, xGMI between the GPUs of the node);
 * GPEMU_GATHER=file exchanges the same few doubles through files of that directory instead -- for ranks that SHARE a device
 * (two RCCL ranks cannot), i.e. for rehearsing the path on a one-GPU machine.
 *
 * How the ranks find each other and what happens when one of them does not make it (round 5):
 *
 *   run nonce   Every file of a run carries the nonce rank 0 drew for THIS run (pid, start time) in its name, so whatever an
 *               earlier run -- finished, crashed or still going -- left in the directory is never read: `run_id` (rank 0,
 *               atomically replaced) names the nonce; rank r answers with `ack_<nonce>_<r>` holding a nonce of its own; rank 0
 *               answers all of them with `go_<nonce>` listing the nonces it has seen (and the ncclUniqueId).  A rank goes on
 *               only when a `go` lists ITS OWN fresh nonce (a stale `go` of a crashed run cannot), rank 0 only with an `ack`
 *               for its own fresh nonce from every rank.
 *   when        At START-UP (gpemu_host_rank_device, the CLI's first call), where the wait measures launch skew
 *               (GPEMU_RENDEZVOUS_WAIT_S, default 120 s) and not the training time of the slowest rank; the RCCL communicator
 *               is made right there too.  A library user who never calls it gets the same handshake at the first gather.
 *   failures    A rank that ends through the layer's exit path (fatal.c) drops `failed_<nonce>_<rank>`; a rank that is
 *               killed leaves a pid that no longer exists.  A watchdog thread in every rank looks for both every 200 ms from
 *               the handshake on and ends its own process with a message -- whether the main thread computes, waits for a
 *               file or sits inside ncclCommInitRank / ncclAllGather -- so nobody waits for a rank that is gone; there is no
 *               other deadline on the gather, for either transport (training may take hours).
 *   cleaning    Rank 0 removes a gather's files once every rank has marked it read, and the run's own files at its end
 *               (gpemu_host_ranks_finish, the CLI's last call).
 */
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <signal.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <time.h>
#include <unistd.h>
#include "libemu.h"
#include "gpemu.h"

static int env_int(const char *name, int dflt)
{
	const char *v = getenv(name);
	return (v && *v) ? atoi(v) : dflt;
}

int gpemu_host_world_size(void)
{
	const int w = env_int("GPEMU_WORLD_SIZE", 1);
	return w > 1 ? w : 1;
}

int gpemu_host_rank(void)
{
	const int w = gpemu_host_world_size(), r = env_int("GPEMU_RANK", 0);
	if (r < 0 || r >= w) gpemu_host_fatal("GPEMU_RANK %d outside [0, GPEMU_WORLD_SIZE = %d)\n", r, w);
	return r;
}

#define MAX_WORLD 64
#define NONCE_LEN 48

static pthread_mutex_t g_mu = PTHREAD_MUTEX_INITIALIZER;
static int g_joined = 0;                    /* the handshake of this process has succeeded (atomic: read by the watchdog, the exit hook) */
static char g_dir[3072];
static char g_run[NONCE_LEN];               /* rank 0's nonce: the name of this run */
static long g_peer_pid[MAX_WORLD];
static void *g_comm = NULL;                 /* RCCL communicator (transport rccl) */
static int g_use_files = 0;
static unsigned g_gather_seq = 0;
static int g_finished = 0;                  /* this rank's part in the run is over (atomic: set by the main thread, read by the watchdog) */
#define LOAD(x) __atomic_load_n(&(x), __ATOMIC_ACQUIRE)
#define STORE(x, v) __atomic_store_n(&(x), (v), __ATOMIC_RELEASE)
static int g_pid_check = 1;                 /* GPEMU_RANK_PID_CHECK=0: the ranks do not share a pid namespace (containers) */

static void sleep_ms(int ms) { struct timespec t = {ms / 1000, (long)(ms % 1000) * 1000000L}; nanosleep(&t, NULL); }
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec; }

static void make_nonce(char *out)
{
	struct timespec t;
	clock_gettime(CLOCK_REALTIME, &t);
	snprintf(out, NONCE_LEN, "%ld-%lx%05lx", (long)getpid(), (unsigned long)t.tv_sec, (unsigned long)(t.tv_nsec / 1000) & 0xfffffUL);
}
static long nonce_pid(const char *nonce) { return atol(nonce); }

/* whole small files, written under a temporary name and renamed: a reader sees all of one or nothing */
static int write_file(const char *path, const void *data, size_t len)
{
	char tmp[4200];
	snprintf(tmp, sizeof tmp, "%s.tmp%ld", path, (long)getpid());
	FILE *f = fopen(tmp, "wb");
	if (!f) return -1;
	const int ok = fwrite(data, 1, len, f) == len;
	if (fclose(f) || !ok || rename(tmp, path)) { unlink(tmp); return -1; }
	return 0;
}
static long read_file(const char *path, void *data, size_t cap)
{
	FILE *f = fopen(path, "rb");
	if (!f) return -1;
	const size_t got = fread(data, 1, cap, f);
	fclose(f);
	return (long)got;
}

static void drop_failed_marker(int status)
{
	(void)status;
	if (!LOAD(g_joined) || LOAD(g_finished)) return;
	char path[4200];
	snprintf(path, sizeof path, "%s/failed_%s_%d", g_dir, g_run, gpemu_host_rank());
	const int fd = open(path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
	if (fd >= 0) close(fd);
}

/* a process that has ended but has not been reaped by its parent (a launcher that waits for its ranks one after the other)
 * still answers kill(pid, 0): /proc says what it is */
static int pid_is_zombie(long pid)
{
	char path[64], buf[512];
	snprintf(path, sizeof path, "/proc/%ld/stat", pid);
	FILE *f = fopen(path, "r");
	if (!f) return 0;
	const size_t n = fread(buf, 1, sizeof buf - 1, f);
	fclose(f);
	buf[n] = 0;
	const char *p = strrchr(buf, ')');          /* "pid (comm) S ...": the state letter follows the LAST ')' */
	return p && p[1] == ' ' && (p[2] == 'Z' || p[2] == 'X');
}

/* is any other rank of this run known to be gone?  (its marker, or -- same node -- its process) */
static int a_peer_is_gone(int world, int me, int *which, const char **how)
{
	char path[4200];
	for (int r = 0; r < world; r++) {
		if (r == me) continue;
		snprintf(path, sizeof path, "%s/failed_%s_%d", g_dir, g_run, r);
		if (access(path, F_OK) == 0) { *which = r; *how = "ended with an error"; return 1; }
		if (g_pid_check && g_peer_pid[r] > 0 && ((kill((pid_t)g_peer_pid[r], 0) != 0 && errno == ESRCH) || pid_is_zombie(g_peer_pid[r]))) {
			/* a rank that has finished its part leaves `left_<run>_<r>` before it goes: that exit is not a failure */
			snprintf(path, sizeof path, "%s/left_%s_%d", g_dir, g_run, r);
			if (access(path, F_OK) == 0) continue;
			*which = r; *how = "is no longer running";
			return 1;
		}
	}
	return 0;
}

void gpemu_host_ranks_finish(void);

static void *watchdog_main(void *arg)
{
	(void)arg;
	const int world = gpemu_host_world_size(), me = gpemu_host_rank();
	while (!LOAD(g_finished)) {
		int which = -1;
		const char *how = "";
		if (a_peer_is_gone(world, me, &which, &how) && !LOAD(g_finished))
			gpemu_host_fatal("rank %d: rank %d of this run %s -- not waiting for it (GPEMU_RENDEZVOUS_DIR %s, run %s)\n", me, which, how, g_dir, g_run);
		sleep_ms(200);
	}
	return NULL;
}

/* the start-up handshake described in the header; idempotent */
static void join_run(void)
{
	pthread_mutex_lock(&g_mu);
	if (LOAD(g_joined)) { pthread_mutex_unlock(&g_mu); return; }
	const int world = gpemu_host_world_size(), rank = gpemu_host_rank();
	if (world > MAX_WORLD) gpemu_host_fatal("GPEMU_WORLD_SIZE %d: at most %d ranks (one per GPU of a node)\n", world, MAX_WORLD);
	const char *dir = getenv("GPEMU_RENDEZVOUS_DIR");
	if (!dir || !*dir) gpemu_host_fatal("GPEMU_WORLD_SIZE > 1 needs GPEMU_RENDEZVOUS_DIR (a directory every rank can reach)\n");
	snprintf(g_dir, sizeof g_dir, "%s", dir);
	const char *how = getenv("GPEMU_GATHER");
	g_use_files = how && !strcmp(how, "file");
	const double wait_s = getenv("GPEMU_RENDEZVOUS_WAIT_S") ? atof(getenv("GPEMU_RENDEZVOUS_WAIT_S")) : 120.0;
	const double t0 = now_s();
	char path[4200], mine[NONCE_LEN];
	/* go file: world nonces of NONCE_LEN bytes each, then the ncclUniqueId */
	const size_t go_len = (size_t)world * NONCE_LEN + GPEMU_RCCL_ID_BYTES;
	char *go = (char *)calloc(1, go_len);
	make_nonce(mine);
	if (rank == 0) {
		snprintf(g_run, sizeof g_run, "%s", mine);
		if (!g_use_files) {
			char err[512] = "";
			const int rc = gpemu_rccl_unique_id(go + (size_t)world * NONCE_LEN, err, sizeof err);
			if (rc) gpemu_host_fatal("rank 0: no RCCL unique id (%d): %s\n", rc, err);
		}
		snprintf(path, sizeof path, "%s/run_id", g_dir);
		if (write_file(path, g_run, NONCE_LEN)) gpemu_host_fatal("rank 0: cannot write %s: %s\n", path, strerror(errno));
		memcpy(go, g_run, NONCE_LEN);
		for (int r = 1; r < world; r++) {
			snprintf(path, sizeof path, "%s/ack_%s_%d", g_dir, g_run, r);
			for (;;) {
				char theirs[NONCE_LEN];
				if (read_file(path, theirs, NONCE_LEN) == NONCE_LEN) { theirs[NONCE_LEN - 1] = 0; memcpy(go + (size_t)r * NONCE_LEN, theirs, NONCE_LEN); break; }
				if (now_s() - t0 > wait_s)
					gpemu_host_fatal("rank 0: rank %d has not joined run %s within %.0f s (GPEMU_RENDEZVOUS_WAIT_S; directory %s)\n", r, g_run, wait_s, g_dir);
				sleep_ms(2);
			}
		}
		snprintf(path, sizeof path, "%s/go_%s", g_dir, g_run);
		if (write_file(path, go, go_len)) gpemu_host_fatal("rank 0: cannot write %s: %s\n", path, strerror(errno));
	} else {
		char acked[NONCE_LEN] = "";
		for (;;) {
			char cur[NONCE_LEN];
			snprintf(path, sizeof path, "%s/run_id", g_dir);
			if (read_file(path, cur, NONCE_LEN) == NONCE_LEN) {
				cur[NONCE_LEN - 1] = 0;
				if (strcmp(cur, acked)) {                    /* a run we have not answered yet (the first, or rank 0 has only now replaced a stale one) */
					if (acked[0]) { snprintf(path, sizeof path, "%s/ack_%s_%d", g_dir, acked, rank); unlink(path); }
					snprintf(path, sizeof path, "%s/ack_%s_%d", g_dir, cur, rank);
					if (write_file(path, mine, NONCE_LEN)) gpemu_host_fatal("rank %d: cannot write %s: %s\n", rank, path, strerror(errno));
					memcpy(acked, cur, NONCE_LEN);
				}
				snprintf(path, sizeof path, "%s/go_%s", g_dir, acked);
				if (read_file(path, go, go_len) == (long)go_len && !strncmp(go + (size_t)rank * NONCE_LEN, mine, NONCE_LEN)) {
					memcpy(g_run, acked, NONCE_LEN);
					break;                                   /* rank 0 of THIS launch has seen THIS process */
				}
			}
			if (now_s() - t0 > wait_s)
				gpemu_host_fatal("rank %d: rank 0 has not opened a run within %.0f s (GPEMU_RENDEZVOUS_WAIT_S; directory %s%s)\n", rank, wait_s, g_dir,
				                 acked[0] ? "; a run_id is there, but nobody answers for it: left by an earlier run?" : "");
			sleep_ms(2);
		}
	}
	for (int r = 0; r < world; r++) g_peer_pid[r] = nonce_pid(go + (size_t)r * NONCE_LEN);
	g_pid_check = env_int("GPEMU_RANK_PID_CHECK", 1) != 0;
	STORE(g_joined, 1);
	gpemu_host_on_exit(drop_failed_marker);
	atexit(gpemu_host_ranks_finish);             /* a caller that never says it has finished: a regular exit says it for it */
	pthread_t wd;
	if (pthread_create(&wd, NULL, watchdog_main, NULL) == 0) pthread_detach(wd);
	if (!g_use_files) {
		char err[512] = "";
		const int rc = gpemu_rccl_comm_create(gpemu_host_device(), rank, world, go + (size_t)world * NONCE_LEN, &g_comm, err, sizeof err);
		if (rc) gpemu_host_fatal("rank %d: RCCL communicator (%d): %s\n", rank, rc, err);
	}
	free(go);
	pthread_mutex_unlock(&g_mu);
}

/* the GPU of this rank when nothing else pins one: GPEMU_LOCAL_RANK (default: the rank) modulo the visible devices; and the
 * ranks' rendezvous, at start-up */
void gpemu_host_rank_device(void)
{
	if (gpemu_host_world_size() <= 1) return;
	if (!getenv("GPEMU_DEVICE") && !getenv("GPEMU_DEVICES")) {
		const int n = gpemu_device_count();
		if (n > 0) gpemu_host_set_device(env_int("GPEMU_LOCAL_RANK", gpemu_host_rank()) % n);
	}
	join_run();
}

/* recv[r * count + i] = rank r's send[i] on every rank */
void gpemu_host_allgather(const double *send, int count, double *recv)
{
	const int world = gpemu_host_world_size(), rank = gpemu_host_rank();
	if (world == 1) { memcpy(recv, send, sizeof(double) * (size_t)count); return; }
	join_run();
	const unsigned seq = g_gather_seq++;
	char path[4200];
	if (!g_use_files) {
		char err[512] = "";
		const int rc = gpemu_rccl_comm_allgather(g_comm, send, count, recv, err, sizeof err);
		if (rc) gpemu_host_fatal("rank %d: RCCL all-gather failed (%d): %s\n", rank, rc, err);
		return;
	}
	/* every rank writes its share, then reads everybody's; no deadline -- the watchdog ends the wait if a rank is gone */
	snprintf(path, sizeof path, "%s/gather_%s_%u_%d.bin", g_dir, g_run, seq, rank);
	if (write_file(path, send, sizeof(double) * (size_t)count)) gpemu_host_fatal("rank %d: cannot write %s: %s\n", rank, path, strerror(errno));
	for (int r = 0; r < world; r++) {
		snprintf(path, sizeof path, "%s/gather_%s_%u_%d.bin", g_dir, g_run, seq, r);
		while (read_file(path, recv + (size_t)r * count, sizeof(double) * (size_t)count) != (long)(sizeof(double) * (size_t)count)) sleep_ms(2);
	}
	snprintf(path, sizeof path, "%s/read_%s_%u_%d", g_dir, g_run, seq, rank);
	if (write_file(path, "", 0)) gpemu_host_fatal("rank %d: cannot write %s: %s\n", rank, path, strerror(errno));
	if (rank == 0) {
		/* everybody has read everything: the gather's files go */
		for (int r = 0; r < world; r++) {
			snprintf(path, sizeof path, "%s/read_%s_%u_%d", g_dir, g_run, seq, r);
			while (access(path, F_OK) != 0) sleep_ms(2);
		}
		for (int r = 0; r < world; r++) {
			snprintf(path, sizeof path, "%s/read_%s_%u_%d", g_dir, g_run, seq, r);
			unlink(path);
			snprintf(path, sizeof path, "%s/gather_%s_%u_%d.bin", g_dir, g_run, seq, r);
			unlink(path);
		}
	}
}

/* the end of this rank's part in the run (the CLI's last call; harmless without a run): the communicator goes, the rank says
 * that its exit is a regular one, rank 0 waits for the others to have said so and removes the run's files */
void gpemu_host_ranks_finish(void)
{
	if (gpemu_host_world_size() <= 1 || !LOAD(g_joined) || LOAD(g_finished)) return;
	const int world = gpemu_host_world_size(), rank = gpemu_host_rank();
	char path[4200];
	if (g_comm) { gpemu_rccl_comm_destroy(g_comm); g_comm = NULL; }
	if (rank != 0) STORE(g_finished, 1);             /* (before the marker: rank 0 may be gone a moment after it appears) */
	snprintf(path, sizeof path, "%s/left_%s_%d", g_dir, g_run, rank);
	(void)write_file(path, "", 0);
	if (rank == 0) {
		for (int r = 1; r < world; r++) {
			snprintf(path, sizeof path, "%s/left_%s_%d", g_dir, g_run, r);
			while (access(path, F_OK) != 0) sleep_ms(2);         /* (the watchdog still runs: a rank that dies here ends the wait) */
		}
		STORE(g_finished, 1);
		for (int r = 0; r < world; r++) {
			snprintf(path, sizeof path, "%s/left_%s_%d", g_dir, g_run, r); unlink(path);
			snprintf(path, sizeof path, "%s/ack_%s_%d", g_dir, g_run, r); unlink(path);
		}
		snprintf(path, sizeof path, "%s/go_%s", g_dir, g_run); unlink(path);
		char cur[NONCE_LEN];
		snprintf(path, sizeof path, "%s/run_id", g_dir);
		if (read_file(path, cur, NONCE_LEN) == NONCE_LEN && !strncmp(cur, g_run, NONCE_LEN)) unlink(path);   /* (not a newer run's) */
	}
	STORE(g_finished, 1);
}

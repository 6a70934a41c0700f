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
 * Ranks are plain processes started by the user or a launcher: GPEMU_RANK / GPEMU_WORLD_SIZE name them, GPEMU_LOCAL_RANK
 * (default: the rank) picks the GPU unless GPEMU_DEVICE / GPEMU_DEVICES does, GPEMU_RENDEZVOUS_DIR is a directory every
 * rank can reach.  The gather itself is gpemu_rccl_allgather (RCCL, xGMI between the GPUs of a node); GPEMU_GATHER=file
 * exchanges the same few doubles through files of that directory instead -- for ranks that SHARE a device (two RCCL ranks
 * cannot), i.e. for rehearsing the path on a one-GPU machine.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <time.h>
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
	if (r < 0 || r >= w) { fprintf(stderr, "GPEMU_RANK %d outside [0, GPEMU_WORLD_SIZE = %d)\n", r, w); gpemu_host_exit(EXIT_FAILURE); }
	return r;
}

/* the GPU of this rank when nothing else pins one: GPEMU_LOCAL_RANK (default: the rank) modulo the visible devices */
void gpemu_host_rank_device(void)
{
	if (gpemu_host_world_size() <= 1 || getenv("GPEMU_DEVICE") || getenv("GPEMU_DEVICES")) return;
	const int n = gpemu_device_count();
	if (n > 0) gpemu_host_set_device(env_int("GPEMU_LOCAL_RANK", gpemu_host_rank()) % n);
}

static unsigned g_gather_seq = 0;

static void sleep_ms(int ms) { struct timespec t = {ms / 1000, (long)(ms % 1000) * 1000000L}; nanosleep(&t, NULL); }

/* recv[r * count + i] = rank r's send[i] on every rank */
void gpemu_host_allgather(const double *send, int count, double *recv)
{
	const int world = gpemu_host_world_size(), rank = gpemu_host_rank();
	if (world == 1) { memcpy(recv, send, sizeof(double) * (size_t)count); return; }
	const char *dir = getenv("GPEMU_RENDEZVOUS_DIR");
	if (!dir || !*dir) { fprintf(stderr, "GPEMU_WORLD_SIZE > 1 needs GPEMU_RENDEZVOUS_DIR (a directory every rank can reach)\n"); gpemu_host_exit(EXIT_FAILURE); }
	const unsigned seq = g_gather_seq++;
	const char *how = getenv("GPEMU_GATHER");
	char path[4096];
	if (how && !strcmp(how, "file")) {
		/* every rank writes its share under a temporary name, renames it, then waits for everybody else's */
		char tmp[4200];
		snprintf(path, sizeof path, "%s/gather_%u_%d.bin", dir, seq, rank);
		snprintf(tmp, sizeof tmp, "%s.tmp", path);
		FILE *f = fopen(tmp, "wb");
		if (!f || fwrite(send, sizeof(double), (size_t)count, f) != (size_t)count) { perror(tmp); gpemu_host_exit(EXIT_FAILURE); }
		fclose(f);
		if (rename(tmp, path)) { perror(path); gpemu_host_exit(EXIT_FAILURE); }
		for (int r = 0; r < world; r++) {
			snprintf(path, sizeof path, "%s/gather_%u_%d.bin", dir, seq, r);
			int waited = 0;
			for (;;) {
				f = fopen(path, "rb");
				if (f) {
					const size_t got = fread(recv + (size_t)r * count, sizeof(double), (size_t)count, f);
					fclose(f);
					if (got == (size_t)count) break;
				}
				if ((waited += 5) > 3600 * 1000) { fprintf(stderr, "rank %d never delivered %s\n", r, path); gpemu_host_exit(EXIT_FAILURE); }
				sleep_ms(5);
			}
		}
		return;
	}
	char err[512] = "";
	snprintf(path, sizeof path, "%s/rccl_id_%u", dir, seq);
	const int rc = gpemu_rccl_allgather(gpemu_host_device(), rank, world, path, send, count, recv, err, sizeof err);
	if (rc) { fprintf(stderr, "RCCL all-gather failed (%d): %s\n", rc, err); gpemu_host_exit(EXIT_FAILURE); }
}

/* Drives gpemu_host_interactive_loop (csrc/host/interactive_io.c: interactive_mode's reader -> device -> writer pipeline)
 * with a stand-in for the device stage, so that the framing, batching and ordering logic is tested without a GPU:
 *   io_loop_driver D NOUT NPRINT BINARY [DELAY_US]
 * mean_i = (i + 1) * sum_k x_k, variance_i = x_0 * x_0 + i; DELAY_US: sleep per batch (a busy device stage).
 * stderr: "stats points batches max_batch". */
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>
#include "libemu.h"

struct fake { int d, nout, delay_us; };

static void fake_points(void *user, int np, const double *pts, double *mean, double *var)
{
	const struct fake *f = (const struct fake *)user;
	for (int q = 0; q < np; q++) {
		double s = 0.0;
		for (int k = 0; k < f->d; k++) s += pts[(size_t)q * f->d + k];
		for (int i = 0; i < f->nout; i++) {
			mean[(size_t)q * f->nout + i] = (double)(i + 1) * s;
			var[(size_t)q * f->nout + i] = pts[(size_t)q * f->d] * pts[(size_t)q * f->d] + (double)i;
		}
	}
	if (f->delay_us > 0) usleep((useconds_t)f->delay_us);
}

int main(int argc, char **argv)
{
	if (argc < 5) return 2;
	struct fake f = {atoi(argv[1]), atoi(argv[2]), argc > 5 ? atoi(argv[5]) : 0};
	struct gpemu_io_stats st;
	const int rc = gpemu_host_interactive_loop(STDIN_FILENO, STDOUT_FILENO, f.d, f.nout, atoi(argv[3]), atoi(argv[4]), fake_points, &f, &st);
	fprintf(stderr, "stats %ld %ld %d\n", st.points, st.batches, st.max_batch);
	return rc == 0 ? 0 : 1;
}

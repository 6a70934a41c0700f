/*
 * interactive_emulator -- command-line front end, drop-in for the reference's
 * src/interactive_emulator.c (same three modes, options, file formats and stdout protocol):
 *
 *   interactive_emulator estimate_thetas INPUT_MODEL_FILE MODEL_SNAPSHOT_FILE [OPTIONS]
 *   interactive_emulator interactive_mode MODEL_SNAPSHOT_FILE [OPTIONS]
 *   interactive_emulator print_thetas MODEL_SNAPSHOT_FILE
 *
 * Kept quirks: --covariance_fn is atoi'd and compared with POWEREXPCOVFN=1 /
 * MATERN32=2 / MATERN52=3 (0 and 1 both mean power-exponential); --pca_variance falls through
 * into --pca_output and --quiet.
 *
 * interactive_mode is a request/response pipe (flush after every answer, :440 of the reference).
 * Points that are ALREADY waiting on stdin are answered as one GPU batch; a lone point is
 * answered immediately, so MCMC drivers that wait for each answer never dead-lock.  Reading/parsing,
 * the device and formatting/writing run as a three-stage pipeline (interactive_io.c); --binary selects
 * the reference's BINARY_INTERACTIVE_MODE framing (raw doubles in and out) at run time.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <getopt.h>
#include <unistd.h>
#include <assert.h>
#include <math.h>
#include <time.h>
#include "libemu.h"

struct cmdLineOpts {
	int regOrder, covFn, quietFlag, pcaOutputFlag, binaryFlag;
	double pca_variance;
	char *run_mode, *inputfile, *statefile;
};

static const char useage[] =
	"useage:\n"
	"  interactive_emulator estimate_thetas INPUT_MODEL_FILE MODEL_SNAPSHOT_FILE [OPTIONS]\n"
	"or\n"
	"  interactive_emulator interactive_mode MODEL_SNAPSHOT_FILE [OPTIONS]\n"
	"or\n"
	"  interactive_emulator print_thetas MODEL_SNAPSHOT_FILE\n"
	"\n"
	"INPUT_MODEL_FILE can be \"-\" to read from standard input.\n"
	"\n"
	"Options which only influence estimate_thetas:\n"
	"  --regression_order=0..3   (const, linear, quadratic, cubic)\n"
	"  --covariance_fn=0|1 (POWER_EXPONENTIAL)  2 (MATERN32)  3 (MATERN52)\n"
	"  (-v FRAC) --pca_variance=FRAC : keep PCA components up to variance fraction FRAC\n"
	"options which influence interactive_mode:\n"
	"  (-q) --quiet: run without any extraneous output\n"
	"  (-z) --pca_output: emulator output is left in the pca space\n"
	"  --binary: points are read and results written as raw doubles (the reference's BINARY_INTERACTIVE_MODE build)\n"
	"general options:\n"
	"  -h -? print this dialogue\n"
	"environment: GPEMU_DEVICE (HIP device), GPEMU_SEED, GPEMU_RESTARTS, GPEMU_JOBS, GPEMU_LOCKSTEP (restart threads\n"
	"sharing one device context, default 16), GPEMU_NTHREADS (with GPEMU_LOCKSTEP=1: threads with a context each)\n";

static int perr(const char *s) { fprintf(stderr, "%s\n", s); return EXIT_FAILURE; }

static double wall_s(void)
{
	struct timespec t;
	clock_gettime(CLOCK_MONOTONIC, &t);
	return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}

/* interactive_emulator.c:212-251 of the reference: "nt d N", N*d design values, N*nt outputs */
static int open_model_file(const char *name, gsl_matrix **xmodel_ptr, gsl_matrix **training_ptr)
{
	FILE *in = (!strcmp(name, "-") || !strcmp(name, "stdin")) ? stdin : fopen(name, "r");
	if (!in) return 0;
	int nt = 0, d = 0, n = 0;
	if (fscanf(in, "%d%*c", &nt) != 1 || fscanf(in, "%d%*c", &d) != 1 || fscanf(in, "%d%*c", &n) != 1) return 0;
	assert(nt > 0 && d > 0 && n > 0);
	gsl_matrix *x = gsl_matrix_alloc(n, d), *y = gsl_matrix_alloc(n, nt);
	for (int i = 0; i < n; i++) for (int j = 0; j < d; j++) if (fscanf(in, "%lf%*c", gsl_matrix_ptr(x, i, j)) != 1) return 0;
	for (int i = 0; i < n; i++) for (int j = 0; j < nt; j++) if (fscanf(in, "%lf%*c", gsl_matrix_ptr(y, i, j)) != 1) return 0;
	if (in != stdin) fclose(in);
	*xmodel_ptr = x; *training_ptr = y;
	return 1;
}

static int estimate_thetas(struct cmdLineOpts *o)
{
	gsl_matrix *xmodel = NULL, *training = NULL;
	double varfrac = 0.95;
	const double t_start = wall_s();
	/* one process per GPU (GPEMU_RANK / GPEMU_WORLD_SIZE, ranks.c): every rank trains its share and ends with the whole
	 * model; rank 0 alone writes MODEL_SNAPSHOT_FILE.  The ranks meet here, before anything else (ranks.c join_run). */
	gpemu_host_rank_device();
	gpemu_host_warm_start();                             /* the HIP runtime starts while the input file is being parsed */
	if (!open_model_file(o->inputfile, &xmodel, &training)) return perr("Input File read failed.");
	const double t_read = wall_s();
	FILE *out = fopen(gpemu_host_rank() == 0 ? o->statefile : "/dev/null", "w");
	if (!out) return perr("Opening statefile failed.");
	if (o->pca_variance <= 1.0 && o->pca_variance > 0) varfrac = o->pca_variance;
	if (o->covFn < 0 || o->covFn > 3) { fprintf(stderr, "#ERROR cov_fn_index %d not supported\n", o->covFn); exit(EXIT_FAILURE); }
	if (o->regOrder < 0 || o->regOrder > 3) { fprintf(stderr, "#ERROR regression_order %d not supported\n", o->regOrder); exit(EXIT_FAILURE); }
	multi_modelstruct *model = alloc_multimodelstruct(xmodel, training, o->covFn, o->regOrder, varfrac);
	if (!model) return perr("Failed to allocated multi_modelstruct.\n");
	const double t_alloc = wall_s();
	estimate_multi(model, out);
	fclose(out);
	const double t_train = wall_s();
	gpemu_host_ranks_finish();                       /* (ranks.c: the one gather is behind us) */
	if (gpemu_host_rank() == 0 && !getenv("GPEMU_NO_SNAPSHOT_CHECK")) {
		/* (not in the reference, which lets interactive_mode find out) */
		for (int i = 0; i < model->nr; i++)
			if (gpemu_host_emulator_setup_fails(model->pca_model_array[i]))
				fprintf(stderr, "# warning: component %d: the covariance matrix at the trained thetas (amplitude e^%.3f, nugget e^%.3f) is numerically "
				        "singular -- interactive_mode will refuse this snapshot (\"trying to cholesky a non postive def matrix\").  Training data without "
				        "noise drive the search, which is unbounded as in the reference, towards nugget -> 0; GPEMU_NUGGET_FLOOR=<log nugget>, e.g. -12, "
				        "gives the search a lower wall.\n", i,
				        gsl_vector_get(model->pca_model_array[i]->thetas, 0), gsl_vector_get(model->pca_model_array[i]->thetas, 1));
	}
	if (getenv("GPEMU_SEARCH_STATS"))
		fprintf(stderr, "# cli phases: rendezvous_read_input_s %.3f pca_alloc_s %.3f train_and_dump_s %.3f snapshot_check_s %.3f\n",
		        t_read - t_start, t_alloc - t_read, t_train - t_alloc, wall_s() - t_train);
	free_multimodelstruct(model);
	return EXIT_SUCCESS;
}

/* one batch of waiting points through every component's emulator (multivar_support.c:103-157, batched) */
struct emu_call { multi_emulator *emu; int pca_space, d; };

static void emu_points(void *user, int np, const double *pts, double *mean, double *var)
{
	struct emu_call *c = (struct emu_call *)user;
	gsl_matrix view;
	view.size1 = (size_t)np; view.size2 = (size_t)c->d; view.tda = (size_t)c->d; view.data = (double *)pts; view.block = NULL; view.owner = 0;
	emulate_points_multi(c->emu, &view, c->pca_space, mean, var);
}

static int interactive_mode(struct cmdLineOpts *o)
{
	FILE *fp = fopen(o->statefile, "r");
	if (!fp) return perr("Error opening file");
	const double t_start = wall_s();
	gpemu_host_warm_start();                             /* the HIP runtime starts while the snapshot is being parsed */
	/* nobody in this program reads emulator_struct.cinverse (the reference's interactive_mode does not either): spare every
	 * component the N x N download (512 MB at N = 8192) -- the library keeps filling it for callers that link libEmu */
	setenv("GPEMU_SKIP_CINVERSE", "1", 0);
	multi_modelstruct *model = load_multi_modelstruct(fp);
	fclose(fp);
	const double t_loaded = wall_s();
	multi_emulator *emu = alloc_multi_emulator(model);
	const double t_ready = wall_s();
	const int d = model->nparams, nt = model->nt;
	const int nout = o->pcaOutputFlag ? model->nr : nt;
	FILE *out = stdout;
	if (!o->quietFlag) {
		fprintf(out, "%d\n", d);
		for (int i = 0; i < d; i++) fprintf(out, "%s%d\n", "param_", i);
		fprintf(out, "%d\n", 2 * nt);
		for (int i = 0; i < nt; i++) fprintf(out, "%s_%d\n%s_%d\n", "mean", i, "variance", i);
	}
	fflush(out);
	/* the loop itself (interactive_emulator.c:414-441 of the reference): points already waiting on stdin are answered as
	 * one device batch, a lone point at once; reading/parsing, the device and formatting/writing overlap (interactive_io.c).
	 * The reference always prints nt pairs; in pca mode entries nr..nt-1 are whatever its vectors held: zeros here. */
	struct emu_call call = {emu, o->pcaOutputFlag, d};
	struct gpemu_io_stats st;
	const int rc = gpemu_host_interactive_loop(STDIN_FILENO, STDOUT_FILENO, d, nout, nt, o->binaryFlag, emu_points, &call, &st);
	const char *want = getenv("GPEMU_IO_STATS");
	if (want && atoi(want) > 0)
		fprintf(stderr, "# interactive stats: points %ld batches %ld max_batch %d parse_s %.6f device_s %.6f format_s %.6f wall_s %.6f "
		        "load_snapshot_s %.6f alloc_multi_emulator_s %.6f components %d\n",
		        st.points, st.batches, st.max_batch, st.parse_seconds, st.compute_seconds, st.format_seconds, st.wall_seconds,
		        t_loaded - t_start, t_ready - t_loaded, model->nr);
	free_multi_emulator(emu);
	return rc == 0 ? 0 : EXIT_FAILURE;
}

static int print_thetas(struct cmdLineOpts *o)
{
	FILE *fp = fopen(o->statefile, "r");
	if (!fp) return perr("Error opening file");
	multi_modelstruct *model = load_multi_modelstruct(fp);
	fclose(fp);
	const int nr = model->nr, d = model->nparams, N = model->nmodel_points;
	const int nthetas = (int)model->pca_model_array[0]->thetas->size;
	double vartot = 0;
	printf("#-- EMULATOR LENGTH SCALES (thetas) IN PCA SPACE -- #\n");
	for (int i = 0; i < nr; i++) vartot += gsl_vector_get(model->pca_evals_r, i);
	printf("#-- id\tpca-var\tScale\tNugget");
	for (int i = 0; i < d; i++) printf("\tlength_%d", i);
	printf(" -- #\n");
	for (int i = 0; i < nr; i++) {
		printf("%d\t", i);
		printf("%lf\t", gsl_vector_get(model->pca_evals_r, i) / vartot);
		for (int j = 0; j < nthetas; j++) printf("%lf\t", exp(gsl_vector_get(model->pca_model_array[i]->thetas, j)));
		printf("\n");
	}
	for (int i = 0; i < nr; i++) {
		char name[256];
		snprintf(name, sizeof name, "pca_emu_summary_%d.dat", i);
		fp = fopen(name, "w");
		if (!fp) continue;
		for (int j = 0; j < N; j++) {
			for (int k = 0; k < d; k++) fprintf(fp, "%lf\t", gsl_matrix_get(model->pca_model_array[i]->xmodel, j, k));
			fprintf(fp, "%lf\n", gsl_vector_get(model->pca_model_array[i]->training_vector, j));
		}
		fclose(fp);
	}
	return 0;
}

static struct cmdLineOpts *global_opt_parse(int argc, char **argv)
{
	static const char *optString = "r:c:v:zqh?";
	static const struct option longOpts[] = {
		{"regression_order", required_argument, NULL, 'r'}, {"covariance_fn", required_argument, NULL, 'c'},
		{"pca_variance", required_argument, NULL, 'v'},      {"pca_output", no_argument, NULL, 'z'},
		{"quiet", no_argument, NULL, 'q'},                    {"help", no_argument, NULL, 'h'},
		/* not in the reference: the corrected forms of gpemu.h (same as GPEMU_EXACT_GRAD=1 / GPEMU_MATERN_FIXED=1) */
		{"exact_gradient", no_argument, NULL, 1001},          {"matern_fixed", no_argument, NULL, 1002},
		/* the reference's compile-time BINARY_INTERACTIVE_MODE (interactive_emulator.c:119,392-396,418-438) as a run-time flag */
		{"binary", no_argument, NULL, 1003},
		{NULL, no_argument, NULL, 0}};
	struct cmdLineOpts *o = (struct cmdLineOpts *)calloc(1, sizeof *o);
	o->pca_variance = 0.99;
	int idx, opt;
	while ((opt = getopt_long(argc, argv, optString, longOpts, &idx)) != -1) {
		switch (opt) {
		case 'r': o->regOrder = atoi(optarg); break;
		case 'c': o->covFn = atoi(optarg); break;
		case 'v':
			o->pca_variance = atof(optarg);
			if (o->pca_variance < 0.0 || o->pca_variance > 1.0) {
				fprintf(stderr, "# err pca_variance argument given incorrect value: %lf\n", o->pca_variance);
				o->pca_variance = 0.95;
				fprintf(stderr, "# using default value: %lf\n", o->pca_variance);
			}
			fprintf(stderr, "# var-frac: %lf\n", o->pca_variance);
			/* fall through (as the reference does) */
		case 'z': o->pcaOutputFlag = 1; /* fall through */
		case 'q': o->quietFlag = 1; break;
		case 1001: gpemu_host_set_modes(gpemu_host_modes() | 1 /* GPEMU_MODE_EXACT_GRAD */); break;
		case 1002: gpemu_host_set_modes(gpemu_host_modes() | 2 /* GPEMU_MODE_MATERN_LOG */); break;
		case 1003: o->binaryFlag = 1; break;
		case 'h':
		case '?': exit(perr(useage));
		default: break;
		}
	}
	if (optind >= argc) exit(perr(useage));
	o->run_mode = argv[optind];
	if (!strcmp(o->run_mode, "estimate_thetas")) {
		if (optind + 2 >= argc) exit(perr(useage));
		o->inputfile = argv[optind + 1];
		o->statefile = argv[optind + 2];
	} else {
		if (optind + 1 >= argc) exit(perr(useage));
		o->statefile = argv[optind + 1];
	}
	return o;
}

int main(int argc, char **argv)
{
	if (argc < 3) return perr(useage);
	const char *dev = getenv("GPEMU_DEVICE");
	if (dev) gpemu_host_set_device(atoi(dev));
	struct cmdLineOpts *o = global_opt_parse(argc, argv);
	int rc;
	if (!strcmp(o->run_mode, "estimate_thetas")) rc = estimate_thetas(o);
	else if (!strcmp(o->run_mode, "interactive_mode")) rc = interactive_mode(o);
	else if (!strcmp(o->run_mode, "print_thetas")) rc = print_thetas(o);
	else { free(o); return perr(useage); }
	free(o);
	return rc;
}

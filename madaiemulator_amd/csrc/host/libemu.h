/*
 * libemu.h -- host-side (C99) mirror of the reference's libEmu / emulator
 * interface for the GP hot path, implemented on the MI355X device library
 * (include/gpemu.h).  Same names, argument meaning and error behaviour as the
 * reference headers, one header instead of eleven:
 *
 *   optstruct.h:25-88        struct optstruct            (field order kept)
 *   modelstruct.h:28-98      struct modelstruct          (field order kept)
 *   emulator_struct.h:20-29  struct emulator_struct      (field order kept)
 *   multi_modelstruct.h:18-61, multivar_support.h:11-20
 *   libEmu/estimate_threaded.h:18-29  struct estimate_thetas_params
 *   libEmu/{emulator,regression,estimator-fns,maxmultimin,emulate-fns}.h prototypes
 *
 * What differs from the reference, by design:
 *   - every O(N^2)/O(N^3) operation runs on the GPU; there is no CPU path for them;
 *   - log det C is 2*sum(log L_ii) instead of log((prod L_ii)^2) (the product under/overflows);
 *   - evalFnGradMulti shares ONE factorisation between value and gradient;
 *   - emulate_points() is added: a batch of query points per call.
 */
#ifndef GPEMU_LIBEMU_H
#define GPEMU_LIBEMU_H

#include <stdio.h>
#include <pthread.h>
#include "gsl_compat.h"

#ifdef __cplusplus
extern "C" {
#endif

#define POWEREXPCOVFN 1
#define MATERN32 2
#define MATERN52 3

struct modelstruct;

typedef struct optstruct {
	int nthetas;
	int nparams;
	int nmodel_points;
	int nemulate_points;
	int regression_order;
	int nregression_fns;
	int fixed_nugget_mode;
	double fixed_nugget;
	int cov_fn_index;
	int use_data_scales;
	gsl_matrix *grad_ranges;          /* nthetas x 2 search box */
} optstruct;

typedef struct modelstruct {
	gsl_matrix *xmodel;               /* nmodel_points x nparams */
	gsl_vector *training_vector;      /* nmodel_points */
	gsl_vector *thetas;               /* nthetas */
	gsl_vector *sample_scales;        /* nparams */
	struct optstruct *options;
	void (*makeHVector)(gsl_vector *h_vector, gsl_vector *x_location, int nparams);
	double (*covariance_fn)(gsl_vector *, gsl_vector *, gsl_vector *, int, int);
	void (*makeGradMatLength)(gsl_matrix *dCdTheta, gsl_matrix *xmodel, double thetaLength, int index,
	                          int nmodel_points, int nparams);
} modelstruct;

typedef struct emulator_struct {
	int nparams;
	int nmodel_points;
	int nregression_fns;
	int nthetas;
	struct modelstruct *model;
	gsl_matrix *cinverse;
	gsl_vector *beta_vector;
	gsl_matrix *h_matrix;
} emulator_struct;

typedef struct multi_modelstruct {
	int nt;
	int nr;
	int nparams;
	int nmodel_points;
	int cov_fn_index;
	int regression_order;
	gsl_matrix *xmodel;
	gsl_matrix *training_matrix;
	gsl_vector *training_mean;
	modelstruct **pca_model_array;
	gsl_vector *pca_evals_r;
	gsl_matrix *pca_evecs_r;
	gsl_matrix *pca_zmatrix;
} multi_modelstruct;

typedef struct multi_emulator {
	int nt;
	int nr;
	int nparams;
	int nmodel_points;
	int nregression_fns;
	int nthetas;
	multi_modelstruct *model;
	emulator_struct **emu_struct_array;
} multi_emulator;

struct estimate_thetas_params {
	struct optstruct *options;
	struct modelstruct *the_model;
	gsl_rng *random_number;
	gsl_matrix *h_matrix;
	int max_tries;
	int success_count;
	double lhood_current;
	double my_best;
};

/* ---- libEmu/emulator.h ------------------------------------------------- */
double covariance_fn_gaussian(gsl_vector *xm, gsl_vector *xn, gsl_vector *thetas, int nthetas, int nparams);
double covariance_fn_matern_three(gsl_vector *xm, gsl_vector *xn, gsl_vector *thetas, int nthetas, int nparams);
double covariance_fn_matern_five(gsl_vector *xm, gsl_vector *xn, gsl_vector *thetas, int nthetas, int nparams);
void derivative_l_gauss(gsl_matrix *dCdTheta, gsl_matrix *xmodel, double thetaLength, int index, int nmodel_points, int nparams);
void derivative_l_matern_three(gsl_matrix *dCdTheta, gsl_matrix *xmodel, double thetaLength, int index, int nmodel_points, int nparams);
void derivative_l_matern_five(gsl_matrix *dCdTheta, gsl_matrix *xmodel, double thetaLength, int index, int nmodel_points, int nparams);
void makeCovMatrix_fnptr(gsl_matrix *cov_matrix, gsl_matrix *xmodel, gsl_vector *thetas, int nmodel_points, int nthetas,
                         int nparams, double (*covariance_fn_ptr)(gsl_vector *, gsl_vector *, gsl_vector *, int, int));
void makeKVector_fnptr(gsl_vector *kvector, gsl_matrix *xmodel, gsl_vector *xnew, gsl_vector *thetas, int nmodel_points,
                       int nthetas, int nparams,
                       double (*covariance_fn_ptr)(gsl_vector *, gsl_vector *, gsl_vector *, int, int));

/* the process-wide pointers of the reference's headers and the entry points that use them (set by set_global_ptrs) */
extern double (*covariance_fn)(gsl_vector *, gsl_vector *, gsl_vector *, int, int);
extern void (*makeHVector)(gsl_vector *h_vector, gsl_vector *x_location, int nparams);
extern void (*makeGradMatLength)(gsl_matrix *dCdTheta, gsl_matrix *xmodel, double thetaLength, int index, int nmodel_points,
                                 int nparams);
void makeCovMatrix(gsl_matrix *cov_matrix, gsl_matrix *xmodel, gsl_vector *thetas, int nmodel_points, int nthetas, int nparams);
void makeKVector(gsl_vector *kvector, gsl_matrix *xmodel, gsl_vector *xnew, gsl_vector *thetas, int nmodel_points, int nthetas,
                 int nparams);
void makeHMatrix(gsl_matrix *h_matrix, gsl_matrix *xmodel, int nmodel_points, int nparams, int nregression_fns);

void print_matrix(gsl_matrix *m, int nx, int ny);
void initialise_new_x(gsl_matrix *new_x, int nparams, int nemulate_points, double emulate_min, double emulate_max);

/* ---- libEmu/regression.h ------------------------------------------------ */
void makeHVector_trivial(gsl_vector *h_vector, gsl_vector *x_location, int nparams);
void makeHVector_linear(gsl_vector *h_vector, gsl_vector *x_location, int nparams);
void makeHVector_quadratic(gsl_vector *h_vector, gsl_vector *x_location, int nparams);
void makeHVector_cubic(gsl_vector *h_vector, gsl_vector *x_location, int nparams);
void makeHMatrix_fnptr(gsl_matrix *h_matrix, gsl_matrix *xmodel, int nmodel_points, int nparams, int nregression_fns,
                       void (*makeHVector_ptr)(gsl_vector *, gsl_vector *, int));

/* ---- host-matrix interface below evalFnMulti / emulate_point (what libRbind calls; lowlevel.c) ------------- */
void chol_inverse_cov_matrix(optstruct *options, gsl_matrix *temp_matrix, gsl_matrix *result_matrix, double *final_determinant_c);
void estimateBeta(gsl_vector *beta_vector, gsl_matrix *h_matrix, gsl_matrix *cinverse, gsl_vector *trainingvector,
                  int nmodel_points, int nregression_fns);
double estimateSigma(gsl_matrix *cinverse, void *params_in);
double getLogLikelyhood(gsl_matrix *cinverse, double det_cmatrix, gsl_matrix *xmodel, gsl_vector *trainingvector,
                        gsl_vector *thetas, gsl_matrix *h_matrix, int nmodel_points, int nthetas, int nparams,
                        int nregression_fns, void (*makeHVector)(gsl_vector *, gsl_vector *, int));
double makeEmulatedMean(gsl_matrix *inverse_cov_matrix, gsl_vector *training_vector, gsl_vector *kplus_vector,
                        gsl_vector *h_vector, gsl_matrix *h_matrix, gsl_vector *beta_vector, int nmodel_points);
double makeEmulatedVariance(gsl_matrix *inverse_cov_matrix, gsl_vector *kplus_vector, gsl_vector *h_vector,
                            gsl_matrix *h_matrix, double kappa, int nmodel_points, int nregression_fns);
double getGradientCn(gsl_matrix *dCdtheta, gsl_matrix *cinverse, gsl_vector *training_vector, int nmodel_points, int nthetas);

/* ---- libEmu/maxmultimin.h ------------------------------------------------ */
double evalFnMulti(const gsl_vector *theta_vec_less_amp, void *params_in);
/* extension: evalFnMulti for every row of a matrix (the first nthetas-1 entries of each row are read), factored in
 * lock-step batches on the device */
void evalFnMultiList(const gsl_matrix *theta_rows_less_amp, void *params_in, double *answer);
void gradFnMulti(const gsl_vector *theta_vec_less_amp, void *params_in, gsl_vector *grad_vec);
void evalFnGradMulti(const gsl_vector *theta_vec, void *params, double *fnval, gsl_vector *grad_vec);
double estimateSigmaFull(gsl_vector *thetas_less_amp, void *params_in);
void maxWithMultiMin(struct estimate_thetas_params *params);
int doOptimizeMultiMin(double (*fn)(const gsl_vector *, void *),
                       void (*gradientFn)(const gsl_vector *, void *, gsl_vector *),
                       void (*fnGradFn)(const gsl_vector *, void *, double *, gsl_vector *),
                       gsl_vector *thetaInit, gsl_vector *thetaFinal, void *args);
void set_random_init_value(gsl_rng *rand, gsl_vector *x, gsl_matrix *ranges, int nthetas);

/* ---- libEmu/estimate_threaded.h ------------------------------------------ */
void estimate_thetas_threaded(modelstruct *the_model, optstruct *options);
int get_number_cpus(void);
void setup_params(struct estimate_thetas_params *params_array, modelstruct *the_model, optstruct *options, int nthreads, int max_tries);
void *estimate_thread_function(void *args);
void fprintPt(FILE *f, pthread_t pt);

/* ---- resultstruct.h:20-36 + libEmu/emulate-fns.h:13-27 (legacy_api.c) ------- */
typedef struct resultstruct {
	gsl_matrix *new_x;                /* nemulate_points x nparams */
	gsl_vector *emulated_mean;
	gsl_vector *emulated_var;
	optstruct *options;
	modelstruct *model;
} resultstruct;
void alloc_resultstruct(resultstruct *res, optstruct *opts);
void free_resultstruct(resultstruct *res);
void copy_resultstruct(resultstruct *dst, resultstruct *src);
void fill_resultstruct(resultstruct *res, optstruct *options, char **input_data);
void emulate_model_results(modelstruct *the_model, optstruct *options, resultstruct *results);
void emulateAtPoint(modelstruct *the_model, gsl_vector *the_point, optstruct *options, double *the_mean, double *the_variance);
void emulateAtPointList(modelstruct *the_model, gsl_matrix *point_list, optstruct *options, double *the_mean, double *the_variance);
void emulateQuick(modelstruct *the_model, gsl_vector *the_point, optstruct *options, double *mean_out, double *var_out,
                  gsl_matrix *h_matrix, gsl_matrix *cinverse, gsl_vector *beta_vector);
void emulate_ith_location(modelstruct *the_model, optstruct *options, resultstruct *results, int i, gsl_matrix *h_matrix,
                          gsl_matrix *cinverse, gsl_vector *beta_vector);

/* ---- modelstruct.h / optstruct.h ------------------------------------------ */
/* the older, optstruct-sized forms and the optstruct helpers (legacy_api.c; callers: libRbind, estimate_threaded.c:57-68) */
void alloc_modelstruct(modelstruct *the_model, optstruct *options);
void free_modelstruct(modelstruct *the_model);
void copy_modelstruct(modelstruct *dst, modelstruct *src);
void fill_modelstruct(modelstruct *the_model, optstruct *options, char **input_data);
void dump_modelstruct(FILE *fptr, modelstruct *the_model, optstruct *opts);
void load_modelstruct(FILE *fptr, modelstruct *the_model, optstruct *opts);
void free_optstruct(optstruct *opts);
void copy_optstruct(optstruct *dst, optstruct *src);
void dump_optstruct(FILE *fptr, optstruct *opts);
void load_optstruct(FILE *fptr, optstruct *opts);
void setup_cov_fn(optstruct *opts);
void setup_regression(optstruct *opts);
modelstruct *alloc_modelstruct_2(gsl_matrix *xmodel, gsl_vector *training_vector, int cov_fn_index, int regression_order);
void free_modelstruct_2(modelstruct *model);
void dump_modelstruct_2(FILE *fptr, modelstruct *the_model);
modelstruct *load_modelstruct_2(FILE *fptr);
void set_global_ptrs(modelstruct *model);
gsl_vector *fill_sample_scales_vec(gsl_matrix *xmodel);
void setup_optimization_ranges(optstruct *options, modelstruct *the_model);

/* ---- emulator_struct.h ------------------------------------------------------ */
emulator_struct *alloc_emulator_struct(modelstruct *model);
void free_emulator_struct(emulator_struct *e);
void emulate_point(emulator_struct *e, gsl_vector *point, double *mean, double *variance);
void makeHMatrix_es(gsl_matrix *h_matrix, emulator_struct *e);
void makeCovMatrix_es(gsl_matrix *cov_matrix, emulator_struct *e);
void makeKVector_es(gsl_vector *kvector, gsl_vector *point, emulator_struct *e);
void estimateBeta_es(gsl_vector *beta_vector, emulator_struct *e);
/* alloc_emulator_struct with the say on the host copy of C^-1 (a public field, N x N doubles downloaded and mirrored:
 * most of the call's time at large N): 0 leaves e->cinverse allocated but unfilled for callers that never read it */
emulator_struct *gpemu_host_alloc_emulator(modelstruct *model, int fill_cinverse);
/* ... for the n components of a multi-output model at once: one lock-step factorisation (gpemu_predict_setup_batch) */
void gpemu_host_alloc_emulators(modelstruct **models, int n, int fill_cinverse, emulator_struct **out);
/* the process's one-off device start-up costs on a thread of its own while the caller reads its input (device_bridge.c) */
void gpemu_host_warm_start(void);
void gpemu_host_warm_wait(void);
/* extension: npoints query rows (npoints x nparams), mean/variance arrays of npoints */
void emulate_points(emulator_struct *e, gsl_matrix *points, double *mean, double *variance);
/* emulate_points in two halves (device work runs in between): used to query all PCA components at the same time */
void emulate_points_enqueue(emulator_struct *e, gsl_matrix *points);
void emulate_points_collect(emulator_struct *e, int npoints, double *mean, double *variance);

/* ---- multi_modelstruct.h / multivar_support.h ---------------------------------- */
multi_modelstruct *alloc_multimodelstruct(gsl_matrix *xmodel_in, gsl_matrix *training_matrix_in, int cov_fn_index,
                                          int regression_order, double varfrac);
void gen_pca_decomp(multi_modelstruct *m, double vfrac);
void gen_pca_model_array(multi_modelstruct *m);
void dump_multi_modelstruct(FILE *fptr, multi_modelstruct *m);
multi_modelstruct *load_multi_modelstruct(FILE *fptr);
double vector_elt_sum(gsl_vector *vec, int nstop);
void free_multimodelstruct(multi_modelstruct *m);
multi_emulator *alloc_multi_emulator(multi_modelstruct *model);
void free_multi_emulator(multi_emulator *e);
void estimate_multi(multi_modelstruct *m, FILE *outfp);
void emulate_point_multi(multi_emulator *emu, gsl_vector *the_point, gsl_vector *the_mean, gsl_vector *the_variance);
void emulate_point_multi_pca(multi_emulator *emu, gsl_vector *the_point, gsl_vector *the_mean, gsl_vector *the_variance);
/* extension: batched form of the two calls above; outputs are npoints x nt (or x nr) row-major */
void emulate_points_multi(multi_emulator *emu, gsl_matrix *points, int pca_space, double *mean_out, double *var_out);

/* ---- libRbind/rbind.h: the batched likelihood entry point, R-free (rbind.c:626-724) ---------------- */
void callEvalLhoodList(double *xmodel_in, int *nparams_in, double *pointList_in, int *nevalPoints_in,
                       double *training_in, int *nmodelPoints_in, int *nthetas_in, double *answer,
                       int *cov_fn_index_in, int *regression_order_in);
void callEstimate(double *xmodel_in, int *nparams_in, double *training_in, int *nmodelpts, int *nthetas_in, double *final_thetas,
                  int *use_fixed_nugget, double *fixed_nugget_in, int *cov_fn_index_in, int *regression_order_in);
void callEmulateAtList(double *xmodel_in, int *nparams_in, double *points_in, int *nemupoints, double *training_in,
                       int *nmodelpts, double *thetas_in, int *nthetas_in, double *final_emulated_y,
                       double *final_emulated_variance, int *cov_fn_index_in, int *regression_order_in);
void callEmulateAtPt(double *xmodel_in, int *nparams_in, double *point_in, double *training_in, int *nmodelpts,
                     double *thetas_in, int *nthetas_in, double *final_emulated_y, double *final_emulated_variance,
                     int *cov_fn_index_in, int *regression_order_in);
/* one emulator (or nydims of them on one design) kept between calls -- rbind.c:299-600 */
void setupEmulateMC(double *xmodel_in, int *nparams_in, double *training_in, int *nmodelpts, double *thetas_in, int *nthetas_in,
                    int *cov_fn_index_in, int *regression_order_in);
void callEmulateMC(double *point_in, double *mean_out, double *var_out);
void freeEmulateMC(void);
void setupEmulateMCMulti(double *xmodel_in, int *nparams_in, double *training_in, int *nydims_in, int *nmodelpts_in,
                         double *thetas_in, int *nthetas_in, int *cov_fn_index_in, int *regression_order_in);
void callEmulateMCMulti(double *point_in, int *nydims_in, double *final_mean, double *final_var);
void freeEmulateMCMulti(int *nydims_in);

/* ---- knobs of this implementation (not in the reference) ------------------------ */
/* How the layer ends the process where the reference calls exit(1) (fatal.c): single entry -- the first caller wins, later
 * ones sleep --, message, flush, _exit(status): no atexit handler or static destructor (the HIP runtime's) runs beside the
 * host threads that are still inside device calls.  gpemu_host_fatal prints its message INSIDE the gate (one message however
 * many threads fail together) and leaves with EXIT_FAILURE. */
void gpemu_host_exit(int status) __attribute__((noreturn));
void gpemu_host_fatal(const char *fmt, ...) __attribute__((noreturn, format(printf, 1, 2)));
void gpemu_host_on_exit(void (*hook)(int status));
void gpemu_host_set_device(int device);          /* pin the whole process to ONE HIP device (before the first device call) */
/* device slots: GPEMU_DEVICES=0,1,... (repeats allowed), else every visible device, else the pinned one.  Independent
 * work (PCA components, restart groups, component emulators) is dealt to the slots; a thread working for a slot
 * declares it with gpemu_host_thread_device(device) and every context created from that thread lives there. */
int gpemu_host_device_slots(void);
int gpemu_host_slot_device(int slot);
void gpemu_host_thread_device(int device);       /* -1: back to slot 0 */
int gpemu_host_thread_device_get(void);
int gpemu_host_device(void);                     /* device a context created by the calling thread would get */
/* n searches run side by side on the calling thread's device (the component threads of estimate_multi): the search this
 * thread starts sizes its lock-step groups for 1/n of the device's free memory */
void gpemu_host_thread_share(int n);
int gpemu_host_thread_share_get(void);
/* device bytes ONE lock-step group of `lockstep` value+gradient evaluations at nmodel_points holds (optimizer.c) */
double gpemu_host_group_bytes(int nmodel_points, int lockstep);
void gpemu_host_set_seed(unsigned long seed);    /* 0 = /dev/urandom as the reference (estimate_threaded.c:159) */
void gpemu_host_set_search(int nthreads, int restarts_per_job);   /* defaults: 1 thread (1 GPU stream), 50 restarts */
/* what the BFGS runs did so far in this process: runs, runs that ended at |g| < 0.1, runs that ended with "no progress",
 * line searches that fell back to "lowest trial value" (no Wolfe point), |g| at the end of the winning run of the last search */
void gpemu_host_search_stats(long *runs, long *converged, long *noprogress, long *ls_fallbacks, double *best_gnorm);
/* device evaluations of this process so far: value-only, value+gradient, requests answered from a caller's cache of its
 * last results, lock-step rounds and the requests they carried (GPEMU_SEARCH_STATS=1 prints the figures of a search) */
void gpemu_host_eval_stats(long *value_evals, long *valgrad_evals, long *cached, long *rounds, long *round_elements);
/* the corrected forms of gpemu.h (GPEMU_MODE_EXACT_GRAD = 1, GPEMU_MODE_MATERN_LOG = 2) for the WHOLE host layer: first
 * read from GPEMU_EXACT_GRAD / GPEMU_MATERN_FIXED, set here (the CLI's --exact_gradient / --matern_fixed; a snapshot that
 * records the log-scale mode), applied to every device context the layer holds or creates */
int gpemu_host_modes(void);
void gpemu_host_set_modes(int flags);
void gpemu_host_release(void *params_or_emulator); /* drop the device context cached for a params / emulator pointer */
/* lock-step group: n restart threads (one struct estimate_thetas_params each, same model) share one device context;
 * their concurrent evalFnMulti / gradFnMulti / evalFnGradMulti / estimateSigmaFull calls are gathered into device
 * batches.  A member's thread calls gpemu_host_group_leave(params) when it will make no further calls. */
/* how estimate_thetas_threaded deals a run list to threads, groups and slots (pure arithmetic; optimizer.c) */
int gpemu_host_plan_groups(int total_runs, int lockstep, int per_slot, int nslots, int *nthreads_out, int *lo, int *hi, int *slot,
                           int cap);
void *gpemu_host_group_create(struct estimate_thetas_params **members, int n);
void gpemu_host_group_leave(void *params);
void gpemu_host_group_destroy(void *group);

/* one process per GPU (ranks.c): GPEMU_RANK / GPEMU_WORLD_SIZE / GPEMU_LOCAL_RANK / GPEMU_RENDEZVOUS_DIR.  estimate_multi deals
 * the PCA components, estimate_thetas_threaded (single-output models) the runs of the run list to the ranks; the one
 * collective is an all-gather of a few doubles per rank (RCCL: gpemu_rccl_allgather; GPEMU_GATHER=file for ranks that share
 * a device).  Rank 0 writes the snapshot; every rank ends with the full model. */
/* 1 when alloc_emulator_struct would refuse the model at its present thetas ("trying to cholesky a non postive def matrix"),
 * without exiting: estimate_thetas warns with it when a search has ended on a numerically singular model */
int gpemu_host_emulator_setup_fails(modelstruct *model);
int gpemu_host_world_size(void);
int gpemu_host_rank(void);
void gpemu_host_rank_device(void);
void gpemu_host_allgather(const double *send, int count, double *recv);
/* this rank's part in the run is over (after the last gather; idempotent, also run by a regular exit()): the RCCL communicator
 * goes, the exit is marked as a regular one, rank 0 removes the run's files from the rendezvous directory */
void gpemu_host_ranks_finish(void);

/* interactive_mode's request/response loop (src/interactive_emulator.c:398-440) as a reader -> device -> writer pipeline
 * (interactive_io.c): reads points of nparams numbers from fd_in -- text, the reference's fscanf("%lf%*c") framing, or raw
 * doubles (binary != 0: the reference's BINARY_INTERACTIVE_MODE framing, :418-432) --, hands the points that are already
 * waiting to `fn` as one batch (mean / var: npoints x nout, row-major) and writes nprint (mean, variance) pairs per point
 * ("%.17f\n" each, or raw doubles; pairs beyond nout are zeros) in input order, one flush per batch.  A lone point is
 * answered at once.  Returns 0 at end of input, -2 when fd_out failed. */
typedef void (*gpemu_points_fn)(void *user, int npoints, const double *points, double *mean, double *var);
struct gpemu_io_stats { long points, batches; int max_batch; double parse_seconds, compute_seconds, format_seconds, wall_seconds; };
int gpemu_host_interactive_loop(int fd_in, int fd_out, int nparams, int nout, int nprint, int binary, gpemu_points_fn fn,
                                void *user, struct gpemu_io_stats *stats);

#ifdef __cplusplus
}
#endif
#endif

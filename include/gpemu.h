/*
 * gpemu.h -- C-ABI of the MI355X (gfx950) device library for the
 * MADAIEmulator GP hot path (libgpemu_hip.so).
 *
 * Plain C: opaque handle, plain pointers and sizes, int status returns.  No
 * GSL and no torch types cross this boundary.  Every entry point names the
 * reference interface it stands in for (paths relative to the reference's
 * src/ directory).  All matrices are row-major, element (i,j) at a[i*ld+j]
 * (the gsl_matrix layout); all arithmetic is IEEE fp64.
 *
 * Threading: a gpemu_ctx owns one HIP stream and its own HBM workspace; one
 * ctx per host thread (this is what the reference's per-thread
 * estimate_thetas_params deep copy becomes, libEmu/estimate_threaded.c:57-68).
 * Different ctx objects may be used concurrently.
 *
 * There is no CPU fallback: every compute entry fails with
 * GPEMU_ERR_NO_DEVICE / GPEMU_ERR_HIP when no gfx950 device is usable.
 */
#ifndef GPEMU_H
#define GPEMU_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* covariance-function index, optstruct.h:12-14 */
#define GPEMU_POWEREXP 1
#define GPEMU_MATERN32 2
#define GPEMU_MATERN52 3

#define GPEMU_MAX_PARAMS 64   /* largest design dimension d accepted */

/* status codes */
#define GPEMU_OK              0
#define GPEMU_ERR_ARG         1   /* bad argument / model not set */
#define GPEMU_ERR_NO_DEVICE   2   /* no HIP device */
#define GPEMU_ERR_HIP         3   /* a HIP runtime call failed; see gpemu_last_error */
#define GPEMU_ERR_NOT_PD      4   /* Cholesky met a pivot <= 0 (GSL_EDOM in the reference) */
#define GPEMU_ERR_REGRESSION  5   /* H^T C^-1 H not positive definite (regression.c:134-160) */
#define GPEMU_ERR_STATE       6   /* call order (e.g. predict before predict_setup) */

typedef struct gpemu_ctx gpemu_ctx;

/* ---- context ------------------------------------------------------- */
int  gpemu_ctx_create(gpemu_ctx **out, int device);
void gpemu_ctx_destroy(gpemu_ctx *ctx);
const char *gpemu_last_error(const gpemu_ctx *ctx);
const char *gpemu_version(void);
int  gpemu_device_count(void);
/* free / total HBM of a device in bytes (hipMemGetInfo): the host layer sizes its lock-step groups with it.  A likelihood
 * batch of nb evaluations holds nb * (N + 64) * N * 8 bytes, a value+gradient batch nb * (2 N + 64) * N * 8 plus up to
 * 10 GB of C^-1 corners, per context. */
int  gpemu_device_memory(int device, size_t *free_bytes, size_t *total_bytes);

/* ---- the single collective of a multi-process run (north_star: "one GPU per shard with a single RCCL gather
 * over xGMI at the end"): all-gather of `count` doubles per rank over RCCL -- what the mutex-guarded arg-max of
 * libEmu/estimate_threaded.c:308-313 and the serial component loop of multivar_support.c:20-28 become when the
 * independent restarts / PCA components run one process per GPU (csrc/host/ranks.c).  Host buffers:
 * recv[r * count + i] = rank r's send[i].  librccl is opened at run time; the ncclUniqueId goes from rank 0 to the
 * others through the file `id_path` (a fresh name per call, in a directory every rank can reach).  errbuf (optional)
 * receives the message of a failure. */
int gpemu_rccl_allgather(int device, int rank, int world, const char *id_path, const double *send, int count,
                         double *recv, char *errbuf, size_t errlen);
/* The same gather in three steps, for ranks that meet when they START (csrc/host/ranks.c: the communicator exists before
 * the training begins, so a rank never sits in a rendezvous for as long as the slowest rank trains):
 *   gpemu_rccl_unique_id       rank 0 makes the id (GPEMU_RCCL_ID_BYTES bytes = ncclUniqueId); the caller carries it over
 *   gpemu_rccl_comm_create     every rank joins (ncclCommInitRank, collective); *comm_out is an opaque handle
 *   gpemu_rccl_comm_allgather  ncclAllGather of count doubles per rank on the communicator's own stream, host buffers
 *   gpemu_rccl_comm_destroy    the end of the communicator */
#define GPEMU_RCCL_ID_BYTES 128
int gpemu_rccl_unique_id(void *id_out, char *errbuf, size_t errlen);
int gpemu_rccl_comm_create(int device, int rank, int world, const void *id, void **comm_out, char *errbuf, size_t errlen);
int gpemu_rccl_comm_allgather(void *comm, const double *send, int count, double *recv, char *errbuf, size_t errlen);
void gpemu_rccl_comm_destroy(void *comm);

/* ---- model data (modelstruct.h:28-98: xmodel, training_vector) ------
 * Uploads the N x d design and the N training values to HBM and builds the
 * regression basis H (regression.c:9-67,100-112: nreg = 1 + order*d) there.
 * cov_fn_index is GPEMU_POWEREXP / MATERN32 / MATERN52. */
int gpemu_set_model(gpemu_ctx *ctx, int cov_fn_index, int regression_order,
                    int nmodel_points, int nparams,
                    const double *xmodel /* N*d host */, const double *training_vector /* N host */);
/* replace only the training vector (multi_modelstruct: same design, nr PCA columns) */
int gpemu_set_training(gpemu_ctx *ctx, const double *training_vector);

/* ---- a4: makeCovMatrix_fnptr (libEmu/emulator.c:636-653) -----------
 * Full N x N covariance matrix (both triangles) for the model's design at
 * the full theta vector, written to host memory c_out[N*N]. */
int gpemu_cov_matrix(gpemu_ctx *ctx, const double *thetas, int nthetas, double *c_out);

/* ---- a16: makeKVector_fnptr (libEmu/emulator.c:578-593) ------------
 * k[q*N + i] = cov(x_i, xq_q), entries < 1e-10 clamped to 0; M query rows. */
int gpemu_kvectors(gpemu_ctx *ctx, const double *thetas, int nthetas,
                   int npoints, const double *xq /* M*d host */, double *k_out /* M*N host */);

/* ---- a11: evalFnMulti / a9 estimateSigma / a10 getLogLikelyhood -----
 * (libEmu/maxmultimin.c:288-394, 215-273; libEmu/estimator-fns.c:38-103)
 * One likelihood evaluation at FULL thetas (the reference's evalFnMulti
 * passes theta[0] = 0; the drop-in wrapper does that).  Outputs (any may be
 * NULL):  neg_loglik = -logL with log det = 2*sum(log L_ii);  sigma2 =
 * y.Cinv.(y - H beta)/N;  beta[nreg];  logdet;  quad = r.Cinv.r.
 * *info = 0, or 1-based index of the first pivot <= 0 (then
 * GPEMU_ERR_NOT_PD is returned and the outputs are NaN). */
int gpemu_loglik(gpemu_ctx *ctx, const double *thetas, int nthetas,
                 double *neg_loglik, double *sigma2, double *beta,
                 double *logdet, double *quad, int *info);
/* as above, but only enqueues the device work (no host sync, no outputs):
 * used by the throughput bench to time back-to-back evaluations; follow the
 * last call with gpemu_loglik_collect. */
int gpemu_loglik_enqueue(gpemu_ctx *ctx, const double *thetas, int nthetas);
int gpemu_loglik_collect(gpemu_ctx *ctx, double *neg_loglik, double *sigma2, double *beta,
                         double *logdet, double *quad, int *info);

/* ---- a11 over a list of thetas: callEvalLhoodList (libRbind/rbind.c:626-724) and the
 * independent restarts of estimate_thetas_threaded (libEmu/estimate_threaded.c:101-113).
 * nb likelihood evaluations of the SAME model at nb theta vectors (thetas = nb rows of
 * nthetas), factored in lock-step on the device: every kernel handles all nb matrices, so
 * the latency-bound panel chain is paid once per batch.  Outputs are arrays of nb (beta:
 * nb*nreg), any may be NULL; status[b] is what gpemu_loglik would have returned for element b
 * (GPEMU_OK / GPEMU_ERR_NOT_PD / GPEMU_ERR_REGRESSION) and the function itself returns
 * GPEMU_OK when the batch ran.  Workspace: nb * (N+64) * N * 8 bytes of HBM. */
#define GPEMU_MAX_BATCH 64
int gpemu_loglik_batch(gpemu_ctx *ctx, int nb, const double *thetas, int nthetas,
                       double *neg_loglik, double *sigma2, double *beta, double *logdet,
                       double *quad, int *info, int *status);
int gpemu_loglik_batch_enqueue(gpemu_ctx *ctx, int nb, const double *thetas, int nthetas);
int gpemu_loglik_batch_collect(gpemu_ctx *ctx, int nb, double *neg_loglik, double *sigma2,
                               double *beta, double *logdet, double *quad, int *info, int *status);
/* the results of the last GPEMU_RESULT_RING enqueued batches stay readable (pinned ring): `back` = 0 is the newest
 * batch, 1 the one before, ...  Waits for that batch only, so a throughput caller collects EVERY batch while the
 * following ones are already running (what a restart pool consuming its results does). */
#define GPEMU_RESULT_RING 4
int gpemu_loglik_batch_collect_back(gpemu_ctx *ctx, int back, int nb, double *neg_loglik, double *sigma2,
                                    double *beta, double *logdet, double *quad, int *info, int *status);

/* ---- modes (SURVEY App. C2-C4 policy: literal by default, corrected forms behind flags) ----------------------
 * GPEMU_MODE_EXACT_GRAD: gpemu_grad / gpemu_loglik_grad[_batch] return the TRUE gradient of the value gpemu_loglik
 *   returns at theta[0] = 0, d(-logL)/dtheta_k = 1/2 tr(C^-1 dC_k) - 1/2 r^T C^-1 dC_k C^-1 r with r = y - H beta and
 *   the true dC/dtheta (pow-exp: full kernel value times D_k^2 e^{-2 theta_k}; Matern: analytic in log rho; nugget
 *   wherever the nugget rule adds it) instead of the reference's literal formulas (emulator.c:173-209 keeps one
 *   coordinate's factor; maxmultimin.c:514,532,594 scale by sigma^2 and use y).
 * GPEMU_MODE_MATERN_LOG: the Matern kernels take amplitude and nugget on the log scale (amp = e^theta0, nug =
 *   e^theta1) like the pow-exp kernel, instead of raw (emulator.c:355-356,448-449) -- with the raw form evalFnMulti's
 *   theta[0] = 0 (maxmultimin.c:311) makes C = theta1 * I and the reference cannot train a Matern model at all.
 *   Applies to fill, likelihood and prediction alike; a snapshot trained with it must be queried with it.
 * Matern gradients exist only with both flags.  Defaults come from the environment when the context is created
 * (GPEMU_EXACT_GRAD=1, GPEMU_MATERN_FIXED=1); both off = the reference's literal behaviour. */
#define GPEMU_MODE_EXACT_GRAD 1
#define GPEMU_MODE_MATERN_LOG 2
int gpemu_set_mode(gpemu_ctx *ctx, int flags);
int gpemu_get_mode(const gpemu_ctx *ctx);

/* ---- a12: gradFnMulti + getGradientCn (maxmultimin.c:416-550,571-608)
 * grad[nthetas-1] as the reference defines it (literal formulas, SURVEY
 * App. A.3): thetas are the FULL vector with theta[0] ignored (set to 0 for
 * the matrix, replaced by log sigma^2 for the amplitude factor). */
int gpemu_grad(gpemu_ctx *ctx, const double *thetas, int nthetas, double *grad, int *info);
/* a13: evalFnGradMulti (maxmultimin.c:615-618): value and gradient from ONE factorisation (the reference
 * fills and factors twice).  neg_loglik is the evalFnMulti value (theta[0] taken as 0). */
int gpemu_loglik_grad(gpemu_ctx *ctx, const double *thetas, int nthetas, double *neg_loglik, double *sigma2,
                      double *beta, double *grad, int *info);
/* a13 over a list of thetas: the value+gradient pairs the independent restarts of maxWithMultiMin
 * (libEmu/maxmultimin.c:82-119) ask for at the same time.  The nb factorisations (with their inverse
 * rows) run in lock-step; outputs as gpemu_loglik_grad per element (grad: nb rows of nthetas-1),
 * status[b] = GPEMU_OK / GPEMU_ERR_NOT_PD / GPEMU_ERR_REGRESSION.  Workspace nb*(2N+64)*N*8 bytes. */
int gpemu_loglik_grad_batch(gpemu_ctx *ctx, int nb, const double *thetas, int nthetas,
                            double *neg_loglik, double *sigma2, double *beta, double *grad,
                            int *info, int *status);
/* the same in two halves, like gpemu_loglik_batch_enqueue / _collect[_back]: enqueue puts the whole value+gradient
 * batch on the context's stream (staging, factorisation with inverse rows, C^-1 = U U^T, the gradient reductions, one
 * small copy into the pinned result ring) and returns without waiting -- no host synchronisation inside; collect waits
 * for THAT batch only and finishes on the host (the nreg x nreg solve and the reference's scalings, maxmultimin.c:
 * 503-535).  The ring is shared with the likelihood batches (GPEMU_RESULT_RING entries, `back` counts batches of
 * either kind; collecting a batch with the entry of the other kind is GPEMU_ERR_STATE).  A restart pool keeps two
 * contexts busy this way while its host threads do their BFGS arithmetic (estimate_threaded.c:172-188 keeps every
 * core busy; here: the device).  gpemu_loglik_grad_batch, gpemu_loglik_grad and gpemu_grad are enqueue + collect. */
int gpemu_loglik_grad_batch_enqueue(gpemu_ctx *ctx, int nb, const double *thetas, int nthetas);
int gpemu_loglik_grad_batch_collect(gpemu_ctx *ctx, int nb, double *neg_loglik, double *sigma2, double *beta,
                                    double *grad, int *info, int *status);
int gpemu_loglik_grad_batch_collect_back(gpemu_ctx *ctx, int back, int nb, double *neg_loglik, double *sigma2,
                                         double *beta, double *grad, int *info, int *status);

/* ---- a14/a15: chol_inverse_cov_matrix + alloc_emulator_struct -------
 * (libEmu/emulate-fns.c:275-299, emulator_struct.c:13-37)
 * Factorises C(thetas) once and keeps L^-1, C^-1 [y|H], beta and
 * (H^T C^-1 H)^-1 resident in HBM.  beta_out[nreg] optional. */
int gpemu_predict_setup(gpemu_ctx *ctx, const double *thetas, int nthetas, double *beta_out, int *info);
/* alloc_multi_emulator (multivar_support.c:30-52: alloc_emulator_struct for each of the nr PCA components of a multi-output
 * model) as ONE lock-step batch: ctxs[0 .. n-1] hold the same design, covariance function, regression order and modes on the
 * same device and a training vector of their own each (gpemu_set_model / gpemu_set_training); component c is factored at
 * thetas[c * nthetas ..] with its inverse rows beside the others (one launch sequence for all, in ctxs[0]'s workspace) and
 * ctxs[c] receives the prediction state gpemu_predict_setup(ctxs[c], ...) would have given it, bit for bit.  beta_out
 * (n x nreg), info (n, 1-based failed pivot or 0) and status (n, GPEMU_OK / GPEMU_ERR_NOT_PD / GPEMU_ERR_REGRESSION per
 * component) are optional; the return value is the first failure (contexts of failed components are left without a set-up). */
int gpemu_predict_setup_batch(gpemu_ctx *const *ctxs, int n, const double *thetas, int nthetas, double *beta_out, int *info,
                              int *status);
/* Pays what a process pays once -- the HIP runtime's start, the loading of this library's device code, its tables -- on a
 * throw-away context with a 64-point model, so that a caller can do it on a thread of its own while it is still reading its
 * input (interactive_mode: the snapshot).  Returns GPEMU_OK or the first error (GPEMU_ERR_NO_DEVICE ...). */
int gpemu_warm_start(int device);
/* optional: explicit C^-1 (N*N, both triangles) for emulator_struct.cinverse */
int gpemu_get_cinverse(gpemu_ctx *ctx, double *cinv_out);

/* ---- a19: emulate_point, batched (emulator_struct.c:124-143) --------
 * mean[q], var[q] for M query rows xq[M*d].  Host buffers. */
int gpemu_predict_batch(gpemu_ctx *ctx, int npoints, const double *xq, double *mean, double *var);
/* the same in two halves: enqueue stages the queries through pinned memory and returns at once (the device work
 * runs on the context's stream), collect waits and copies the M means/variances out.  One batch per context at a
 * time; different contexts -- the PCA components of a multi-output emulator (multivar_support.c:103-157) --
 * work on their batches concurrently. */
int gpemu_predict_batch_enqueue(gpemu_ctx *ctx, int npoints, const double *xq /* M*d host */);
int gpemu_predict_batch_collect(gpemu_ctx *ctx, int npoints, double *mean, double *var);
/* same with query / result buffers already resident in HBM (device pointers) */
int gpemu_predict_batch_dev(gpemu_ctx *ctx, int npoints, const double *xq_dev,
                            double *mean_dev, double *var_dev);

/* ---- low-level compatibility with the reference's host-matrix interface (what libRbind links against) ------
 * a14 chol_inverse_cov_matrix (libEmu/emulate-fns.c:275-299): the n x n matrix a (row stride lda, lower triangle
 * read) is replaced by its inverse (both triangles); *logdet = 2 sum log L_ii; *info as gpemu_loglik. */
int gpemu_chol_inverse(gpemu_ctx *ctx, int n, double *a_inout, int lda, double *logdet, int *info);
/* the C^-1-times-vector products of estimateBeta / getLogLikelyhood / makeEmulatedMean / makeEmulatedVariance
 * (libEmu/regression.c:120-176, estimator-fns.c:38-103, emulator.c:672-785) with C^-1 passed in host memory:
 * out[v*n + i] = sum_j a[i*lda + j] * v_rows[v*n + j] for nvec vectors stored as rows.  The matrix is uploaded when
 * its (pointer, size, 64-bit checksum over ALL its elements) differs from the copy the context holds: callers such as
 * the libRbind loops rewrite one cinverse buffer in place.  Cost per call with an unchanged matrix: one checksum pass
 * over its n*n doubles in host memory (0.1 s at n = 8192) -- gpemu_symm_pin(ctx, 1) is the caller's promise that the
 * buffer stays as it is until gpemu_symm_pin(ctx, 0) / gpemu_symm_invalidate, and removes that pass (the reference's
 * per-point loops over one cinverse, emulator.c:672-785, are such callers).  gpemu_symm_invalidate drops the cached
 * copy: the next call uploads without comparing. */
int gpemu_symm_apply(gpemu_ctx *ctx, int n, const double *a, int lda, int nvec, const double *v_rows, double *out_rows);
int gpemu_symm_invalidate(gpemu_ctx *ctx);
int gpemu_symm_pin(gpemu_ctx *ctx, int pinned);
/* a5 derivative_l_gauss (libEmu/emulator.c:173-209) written out: out[i*ldo + j] = exp(-0.5 e^{-2t} D^2 - 2t) D^2,
 * D = xcol[i] - xcol[j] (the ONE design coordinate the reference's formula looks at), t = theta_len */
int gpemu_derivative_gauss(gpemu_ctx *ctx, int n, const double *xcol, double theta_len, double *out, int ldo);
/* getGradientCn's trace(C^-1 dC/dtheta) (libEmu/maxmultimin.c:583-588): sum_ij a[i][j] b[j][i] of two host matrices,
 * one pass over both instead of the reference's N^3 dgemm */
int gpemu_trace_product(gpemu_ctx *ctx, int n, const double *a, int lda, const double *b, int ldb, double *trace);

/* ---- device memory helpers for callers that keep data resident ------ */
int gpemu_dev_alloc(gpemu_ctx *ctx, size_t bytes, void **dptr);
int gpemu_dev_free(gpemu_ctx *ctx, void *dptr);
int gpemu_dev_upload(gpemu_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int gpemu_dev_download(gpemu_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int gpemu_sync(gpemu_ctx *ctx);

/* ---- measurement: HIP-event timing of the kernels on ctx's stream ---
 * gpemu_prof_begin arms per-kernel-class event timing; every launch of that
 * class between begin and end is bracketed by hipEvents on the ctx stream.
 * gpemu_prof_end returns the number of launches and their summed duration. */
#define GPEMU_PROF_NONE    0
#define GPEMU_PROF_GEMM    1   /* f64 MFMA trailing-update / prediction GEMM */
#define GPEMU_PROF_FILL    2   /* covariance fill */
#define GPEMU_PROF_LEAF    3   /* diagonal-block factor + panel solve */
#define GPEMU_PROF_POTRF   4   /* whole factorisation (graph launch) */
#define GPEMU_PROF_GEMM_BIG 5  /* only the GEMM launches on the 128x128 8-wave kernel (the dominant kernel of a batch) */
#define GPEMU_PROF_GEMM_K512 6 /* only the GEMM launches with a contraction length >= 512 */
int gpemu_prof_begin(gpemu_ctx *ctx, int kernel_class);
int gpemu_prof_end(gpemu_ctx *ctx, int *nlaunches, double *total_ms, double *flops, double *bytes);

/* Diagnostics: with GPEMU_TRACE=1 in the environment at gpemu_ctx_create, every GEMM / leaf kernel of a
 * factorisation records (one workgroup in 16) its start and end on the device wall clock; this writes one line
 * per launch of the last factorisation:
 *   tag | start_ns end_ns sum_of_workgroup_ns workgroups sum_of_workgroup_shader_clocks stamp1 stamp2 stamp3
 * (stamps, in shader clocks since the workgroup started: GEMM prologue end / epilogue length / -; leaf factor:
 * block staged / first 16-column panel factored / first rank-16 update done)
 * Contexts of one GPU share the clock, so the files of concurrent contexts merge into one timeline
 * (tools/trace_timeline.py). */
int gpemu_trace_dump(gpemu_ctx *ctx, const char *path);

/* ---- low-level building blocks exported for parity tests ------------ */
/* C[m*n] = beta*C + alpha * A[m*K] * B[n*K]^T, host row-major buffers; beta is 0 or 1, and with beta = 1
 * alpha must be +1 or -1 (the accumulators start from C/alpha) */
int gpemu_test_gemm_nt(gpemu_ctx *ctx, int m, int n, int k, double alpha, int beta,
                       const double *a, const double *b, double *c);
/* micro-benchmark of one GEMM shape on device-resident random operands: cfg 0 = the automatic tile choice, 2 = 64x64
 * tiles (4 waves), 8 = 128x128 tiles (8 waves); tri = lower-trapezoid update as in the factorisation; HIP-event timed. */
int gpemu_test_gemm_bench(gpemu_ctx *ctx, int m, int n, int k, int ld, int cfg, int tri, int beta, int reps,
                          double *ms_avg, double *flops);
/* in-place lower Cholesky of a host n*n matrix (both triangles read as lower);
 * returns L in the lower triangle, zeros above. */
int gpemu_test_potrf(gpemu_ctx *ctx, int n, double *a, int *info);
/* the matrix a lock-step batch is factored FROM: stages nb matrices as gpemu_loglik_batch does (one launch of the batch
 * staging kernel, lower tiles only) and copies the N x N block of matrix b to out[N*N] without factorising; tiles
 * strictly above the diagonal are not written by that path. */
int gpemu_test_staged_matrix(gpemu_ctx *ctx, int nb, const double *thetas, int nthetas, int b, double *out);
/* the workgroup -> tile table of a GEMM launch with tiles_m x tiles_n tiles (tri = 1: lower triangle) and super-blocks
 * of sb x sb tiles: entry q * 8 + x is the q-th tile of XCD x, (tm << 16) | tn, or -1 (unused tail slot).  Host logic
 * only (no device).  Returns the table length, or -GPEMU_ERR_ARG; writes min(length, cap) entries to out. */
int gpemu_test_tile_table(int tiles_m, int tiles_n, int tri, int sb, int *out, int cap);
/* the table of the square product with row-start skipping (C^-1 = U U^T; tile (r, c <= r), tile row r contracting from
 * k = max(k0, floor16(r * bm - kstart_off)) to k1): whole tile rows per XCD, rows dealt longest-work-first to the least
 * loaded XCD, each XCD's rows by decreasing k-range.  Same entry format and return value as above. */
int gpemu_test_row_table(int tiles_m, int bm, int kstart_off, int k0, int k1, int *out, int cap);

#ifdef __cplusplus
}
#endif
#endif /* GPEMU_H */

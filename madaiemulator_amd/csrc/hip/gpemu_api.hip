// gpemu_api.hip -- C-ABI entry points (include/gpemu.h) and host-side
// orchestration of the gfx950 kernels: recursive tall-matrix Cholesky with the
// right-hand sides (and optionally the identity) riding along as extra rows,
// likelihood assembly, prediction set-up and the batched prediction sweep.
#include "gpemu_internal.hpp"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <climits>
#include <algorithm>

using namespace gpemu;

#define HIPCHK(ctx, call)                                                                       \
	do {                                                                                        \
		hipError_t e__ = (call);                                                                \
		if (e__ != hipSuccess) {                                                                \
			char buf__[512];                                                                    \
			snprintf(buf__, sizeof buf__, "%s:%d: %s -> %s", __FILE__, __LINE__, #call,         \
			         hipGetErrorString(e__));                                                   \
			(ctx)->err = buf__;                                                                 \
			return GPEMU_ERR_HIP;                                                               \
		}                                                                                       \
	} while (0)

constexpr int INFO_NONE = 0x7f7f7f7f;   // "no failed pivot": what hipMemsetAsync(.., 0x7f, ..) leaves in *info

static int fail(gpemu_ctx *ctx, int code, const char *msg)
{
	if (ctx) ctx->err = msg;
	return code;
}

// ---------------------------------------------------------------------------
// profiling helpers
// ---------------------------------------------------------------------------
static inline bool prof_on(gpemu_ctx *ctx, int cls) { return ctx->prof.cls == cls; }

static void prof_mark(gpemu_ctx *ctx)
{
	hipEvent_t e;
	hipEventCreate(&e);
	hipEventRecord(e, ctx->stream);
	ctx->prof.ev.push_back(e);
}

struct ProfScope {
	gpemu_ctx *ctx;
	bool on;
	ProfScope(gpemu_ctx *c, int cls, double flops, double bytes) : ctx(c), on(prof_on(c, cls))
	{
		if (on) {
			prof_mark(ctx);
			ctx->prof.flops += flops;
			ctx->prof.bytes += bytes;
			ctx->prof.n++;
		}
	}
	~ProfScope() { if (on) prof_mark(ctx); }
};

// GPEMU_TRACE: next {start,end} slot of the per-launch device timestamps (nullptr when tracing is off / full)
static unsigned long long *trace_slot(gpemu_ctx *ctx, const char *fmt, int a = 0, int b = 0, int c = 0)
{
	if (!ctx->dTrace || ctx->trace_next >= ctx->trace_cap) return nullptr;
	char buf[96];
	snprintf(buf, sizeof buf, fmt, a, b, c);
	ctx->trace_tag.push_back(buf);
	return ctx->dTrace + 8 * (size_t)ctx->trace_next++;
}

// algorithmic flops of one GEMM call: 2 * (k-range) summed over the output elements the call owns
// (lower trapezoid for tri; rows of an upper-triangular A start at k = row - kstart_off; rows of a
// lower-triangular B end at k = col - kend_off)
static double gemm_flops(const GemmArgs &a)
{
	double fl = 0.0;
	if (a.kend_mode) {
		for (int j = 0; j < a.n; j++) {
			int ke = std::min(a.k1, j + 1 - a.kend_off);
			if (ke > a.k0) fl += 2.0 * a.m * (double)(ke - a.k0);
		}
		return fl;
	}
	for (int i = 0; i < a.m; i++) {
		const int ncols = a.tri ? std::max(0, std::min(a.n, i + a.diag_off + 1)) : a.n;
		int kb = a.k0;
		if (a.kstart_mode) kb = std::max(kb, i - a.kstart_off);
		if (a.k1 > kb) fl += 2.0 * ncols * (double)(a.k1 - kb);
	}
	return fl;
}

static hipError_t gemm(gpemu_ctx *ctx, const GemmArgs &a_in)
{
	GemmArgs a = a_in;
	a.big_tiles = ctx->sched.gemm_big_tiles;
	a.table_sb = ctx->sched.gemm_table;
	a.keep_idle_waves = ctx->sched.idle_waves ? 0 : 1;
	a.stagger_ticks = ctx->sched.stagger_us * 100;
	a.row_table = ctx->sched.corner_row_table;
	a.no_neg_modifier = ctx->sched.neg_modifier ? 0 : 1;
	a.trace = trace_slot(ctx, "gemm m=%d n=%d k=%d", a.m, a.n, a.k1 - a.k0);
	// GPEMU_PROF_GEMM: every GEMM launch; GPEMU_PROF_GEMM_BIG: only the launches that run the 128x128 8-wave kernel
	// (gemm_nt_kernel<128,128,4,4,2,0>, the dominant kernel of a batched factorisation); GPEMU_PROF_GEMM_K512: only
	// those with a contraction length of 512 or more (the compute-bound updates)
	const int cls = (prof_on(ctx, GPEMU_PROF_GEMM_BIG) && gemm_uses_big_tiles(a)) ? GPEMU_PROF_GEMM_BIG :
	                (prof_on(ctx, GPEMU_PROF_GEMM_K512) && a.k1 - a.k0 >= 512) ? GPEMU_PROF_GEMM_K512 : GPEMU_PROF_GEMM;
	const double fl = prof_on(ctx, cls) ? gemm_flops(a) * (a.nbatch > 1 ? a.nbatch : 1) : 0.0;
	ProfScope ps(ctx, cls, fl, 0.0);
	if (ps.on) {
		char buf[96];
		snprintf(buf, sizeof buf, "gemm m=%d n=%d k=%d tri=%d flops=%.4g", a.m, a.n, a.k1 - a.k0, a.tri, fl);
		ctx->prof.tag.push_back(buf);
	}
	return launch_gemm(ctx->stream, a);
}

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
extern "C" const char *gpemu_version(void) { return "gpemu-mi355x 0.3 (gfx950, fp64 MFMA)"; }

extern "C" int gpemu_device_memory(int device, size_t *free_bytes, size_t *total_bytes)
{
	int n = 0, cur = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GPEMU_ERR_NO_DEVICE;
	if (device < 0 || device >= n) return GPEMU_ERR_ARG;
	size_t fr = 0, tot = 0;
	(void)hipGetDevice(&cur);
	const bool ok = hipSetDevice(device) == hipSuccess && hipMemGetInfo(&fr, &tot) == hipSuccess;
	if (!ok) (void)hipGetLastError();
	(void)hipSetDevice(cur);                   // the caller's current device, on the error path too
	if (!ok) return GPEMU_ERR_HIP;
	if (free_bytes) *free_bytes = fr;
	if (total_bytes) *total_bytes = tot;
	return GPEMU_OK;
}

extern "C" int gpemu_device_count(void)
{
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) return 0;
	return n;
}

// the schedule switches of a new context (gpemu::Sched): a variable that is absent or out of range gives the default
static Sched read_environment()
{
	auto geti = [](const char *name, int dflt) { const char *v = getenv(name); return v ? atoi(v) : dflt; };
	Sched sc;
	int v = geti("GPEMU_GEMM_BIG_TILES", 1024);
	sc.gemm_big_tiles = v > 0 ? v : 1024;
	v = geti("GPEMU_GEMM_TABLE", 8);
	sc.gemm_table = v >= 0 && v <= 64 ? v : 8;
	sc.fill_gram = geti("GPEMU_FILL_GRAM", 1) != 0;
	sc.kvec_gram = geti("GPEMU_KVEC_GRAM", 1) != 0;
	sc.gemv_point = geti("GPEMU_GEMV_POINT", 1) != 0;
	sc.idle_waves = geti("GPEMU_IDLE_WAVES", 1) != 0;
	sc.neg_modifier = geti("GPEMU_NEG_MODIFIER", 1) != 0;
	sc.grad_gram = geti("GPEMU_GRAD_GRAM", 1) != 0;
	sc.stagger_us = std::max(0, std::min(1000, geti("GPEMU_STAGGER_US", 20)));
	sc.factor_ahead = geti("GPEMU_FACTOR_AHEAD", 1) != 0;
	sc.diag_inv_ahead = geti("GPEMU_DIAG_INV_AHEAD", 1) != 0;
	sc.leaf_pair = geti("GPEMU_LEAF_PAIR", 1) != 0;
	sc.corner_row_table = geti("GPEMU_CORNER_ROW_TABLE", 1) != 0;
	v = geti("GPEMU_LEAF_STAGED", -1);
	sc.leaf_staged = v == 0 || v == 1 ? v : -1;
	v = geti("GPEMU_NB_TOP", 0);
	sc.nb_top = v >= LEAF ? (v / LEAF) * LEAF : 0;
	sc.split_rhs_rows = geti("GPEMU_SPLIT_RHS_ROWS", 1) != 0;
	return sc;
}

extern "C" int gpemu_ctx_create(gpemu_ctx **out, int device)
{
	if (!out) return GPEMU_ERR_ARG;
	*out = nullptr;
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GPEMU_ERR_NO_DEVICE;
	if (device < 0 || device >= n) return GPEMU_ERR_ARG;
	if (hipSetDevice(device) != hipSuccess) return GPEMU_ERR_HIP;
	gpemu_ctx *ctx = new gpemu_ctx();
	ctx->sched = read_environment();
	ctx->device = device;
	ctx->res_len = 64 * 64 + 8;
	// the per-matrix result slots (three device and three pinned allocations: milliseconds) are made by the first
	// factorisation (ensure_batch_slots): a context that only ever answers queries -- a component of a multi-output emulator
	// set up by gpemu_predict_setup_batch -- never needs them
	ctx->batch_cap = 0;
	int least = 0, greatest = 0;
	if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { least = 0; greatest = 0; }
	bool ok = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, greatest) == hipSuccess;
	for (int i = 0; ok && i < gpemu_ctx::RES_RING; i++)
		ok = hipEventCreateWithFlags(&ctx->res_ev[i], hipEventDisableTiming) == hipSuccess;
	const char *tr = getenv("GPEMU_TRACE");
	if (ok && tr && atoi(tr) > 0) {                 // in-kernel timestamps: 4096 launch slots of 8 x u64
		ctx->trace_cap = 4096;
		if (hipMalloc(&ctx->dTrace, (size_t)ctx->trace_cap * 64) != hipSuccess) { ctx->dTrace = nullptr; ctx->trace_cap = 0; }
	}
	const char *ng = getenv("GPEMU_NO_GRAPH");
	if (ng && ng[0] == '1') ctx->use_graph = false;
	{
		const char *eg = getenv("GPEMU_EXACT_GRAD"), *mf = getenv("GPEMU_MATERN_FIXED");
		if (eg && atoi(eg) > 0) ctx->mode |= GPEMU_MODE_EXACT_GRAD;
		if (mf && atoi(mf) > 0) ctx->mode |= GPEMU_MODE_MATERN_LOG;
	}
	if (!ok) {
		(void)hipGetLastError();
		gpemu_ctx_destroy(ctx);
		return GPEMU_ERR_HIP;
	}
	*out = ctx;
	return GPEMU_OK;
}

static void free_graphs(gpemu_ctx *ctx)
{
	for (auto &kv : ctx->graphs) hipGraphExecDestroy(kv.second);
	ctx->graphs.clear();
	ctx->warm.clear();
}

static void free_model(gpemu_ctx *ctx)
{
	free_graphs(ctx);
	double **ptrs[] = {&ctx->dX, &ctx->dXg, &ctx->dMid, &ctx->dY, &ctx->dRrows, &ctx->dT, &ctx->dGramPart, &ctx->dLinvAug, &ctx->dBetaQ,
	                   &ctx->dKq, &ctx->dV, &ctx->dXq, &ctx->dMean, &ctx->dS, &ctx->dGradPart, &ctx->dAlpha};
	for (auto p : ptrs) { if (*p) hipFree(*p); *p = nullptr; }
	ctx->dVar = nullptr;                      // (the second half of dMean's allocation)
	ctx->T_rows = 0; ctx->pred_ready = false; ctx->cinv_ready = false; ctx->pred_batch = 0; ctx->stage_cap = 0;
	ctx->pred_pending = 0;
	ctx->S_dim = 0; ctx->S_cap = 0; ctx->gradpart_len = 0; ctx->alpha_cap = 0;
}

extern "C" void gpemu_ctx_destroy(gpemu_ctx *ctx)
{
	if (!ctx) return;
	hipSetDevice(ctx->device);
	if (ctx->stream) hipStreamSynchronize(ctx->stream);
	free_model(ctx);
	for (auto e : ctx->prof.ev) hipEventDestroy(e);
	if (ctx->dInfo) hipFree(ctx->dInfo);
	if (ctx->dTrace) hipFree(ctx->dTrace);
	if (ctx->dParams) hipFree(ctx->dParams);
	if (ctx->hParams) hipHostFree(ctx->hParams);
	for (auto e : ctx->param_ev) if (e) hipEventDestroy(e);
	if (ctx->dSym) hipFree(ctx->dSym);
	if (ctx->dSymV) hipFree(ctx->dSymV);
	if (ctx->dSymOut) hipFree(ctx->dSymOut);
	if (ctx->dRes) hipFree(ctx->dRes);
	if (ctx->hResRing) hipHostFree(ctx->hResRing);
	if (ctx->hInfoRing) hipHostFree(ctx->hInfoRing);
	if (ctx->dGradSum) hipFree(ctx->dGradSum);
	if (ctx->hGradRing) hipHostFree(ctx->hGradRing);
	if (ctx->hGph) hipHostFree(ctx->hGph);
	for (auto e : ctx->res_ev) if (e) hipEventDestroy(e);
	if (ctx->hStage) hipHostFree(ctx->hStage);
	if (ctx->stream) hipStreamDestroy(ctx->stream);
	delete ctx;
}

extern "C" int gpemu_set_mode(gpemu_ctx *ctx, int flags)
{
	if (!ctx || (flags & ~(GPEMU_MODE_EXACT_GRAD | GPEMU_MODE_MATERN_LOG))) return GPEMU_ERR_ARG;
	if (flags != ctx->mode) { ctx->pred_ready = false; ctx->cinv_ready = false; }   // the kernel's meaning may have changed
	ctx->mode = flags;
	return GPEMU_OK;
}
extern "C" int gpemu_get_mode(const gpemu_ctx *ctx) { return ctx ? ctx->mode : 0; }

extern "C" const char *gpemu_last_error(const gpemu_ctx *ctx) { return ctx ? ctx->err.c_str() : "null ctx"; }

extern "C" int gpemu_sync(gpemu_ctx *ctx)
{
	if (!ctx) return GPEMU_ERR_ARG;
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	return GPEMU_OK;
}

extern "C" int gpemu_dev_alloc(gpemu_ctx *ctx, size_t bytes, void **dptr)
{
	if (!ctx || !dptr) return GPEMU_ERR_ARG;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	HIPCHK(ctx, hipMalloc(dptr, bytes));
	return GPEMU_OK;
}
extern "C" int gpemu_dev_free(gpemu_ctx *ctx, void *dptr)
{
	if (!ctx) return GPEMU_ERR_ARG;
	HIPCHK(ctx, hipFree(dptr));
	return GPEMU_OK;
}
extern "C" int gpemu_dev_upload(gpemu_ctx *ctx, void *dst, const void *src, size_t bytes)
{
	if (!ctx) return GPEMU_ERR_ARG;
	HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	return GPEMU_OK;
}
extern "C" int gpemu_dev_download(gpemu_ctx *ctx, void *dst, const void *src, size_t bytes)
{
	if (!ctx) return GPEMU_ERR_ARG;
	HIPCHK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	return GPEMU_OK;
}

// ---------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------
// workspace for nb tall matrices of rows_each rows, packed one after the other (stride rows_each * Np)
static int ensure_T(gpemu_ctx *ctx, size_t rows_each, int nb = 1)
{
	const size_t rows = rows_each * (size_t)nb;
	ctx->nb = nb;
	ctx->T_stride = rows_each * (size_t)ctx->Np;
	if (ctx->T_rows >= rows) return GPEMU_OK;
	free_graphs(ctx);
	if (ctx->dT) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); hipFree(ctx->dT); ctx->dT = nullptr; ctx->T_rows = 0; }
	if (hipMalloc(&ctx->dT, rows * (size_t)ctx->Np * sizeof(double)) != hipSuccess) {
		(void)hipGetLastError();
		ctx->dT = nullptr;
		return fail(ctx, GPEMU_ERR_HIP, "out of device memory for the factorisation workspace (smaller batch?)");
	}
	ctx->T_rows = rows;
	return GPEMU_OK;
}

// per-matrix result slots (info word, Gram partials, Gram + log det, pinned mirrors) for a batch of nb
static int ensure_batch_slots(gpemu_ctx *ctx, int nb)
{
	if (nb <= ctx->batch_cap && ctx->dGramPart && ctx->dInfo) return GPEMU_OK;
	if (nb < ctx->batch_cap) nb = ctx->batch_cap;
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	free_graphs(ctx);                      // captured launches hold the old pointers
	if (ctx->dInfo) hipFree(ctx->dInfo);
	if (ctx->dRes) hipFree(ctx->dRes);
	if (ctx->hResRing) hipHostFree(ctx->hResRing);
	if (ctx->hInfoRing) hipHostFree(ctx->hInfoRing);
	if (ctx->dGramPart) hipFree(ctx->dGramPart);
	if (ctx->dGradSum) hipFree(ctx->dGradSum);
	if (ctx->hGradRing) hipHostFree(ctx->hGradRing);
	ctx->dGradSum = nullptr; ctx->hGradRing = nullptr;
	ctx->dInfo = nullptr; ctx->dRes = nullptr; ctx->hRes = ctx->hResRing = nullptr; ctx->hInfo = ctx->hInfoRing = nullptr; ctx->dGramPart = nullptr;
	ctx->res_seq = 0;                      // results still in the old ring are gone with it
	HIPCHK(ctx, hipMalloc(&ctx->dInfo, (size_t)nb * sizeof(int)));
	HIPCHK(ctx, hipMalloc(&ctx->dRes, (size_t)nb * ctx->res_len * sizeof(double)));
	HIPCHK(ctx, hipHostMalloc((void **)&ctx->hResRing, (size_t)gpemu_ctx::RES_RING * nb * ctx->res_len * sizeof(double)));
	HIPCHK(ctx, hipHostMalloc((void **)&ctx->hInfoRing, (size_t)gpemu_ctx::RES_RING * nb * sizeof(int)));
	ctx->hRes = ctx->hResRing; ctx->hInfo = ctx->hInfoRing;
	HIPCHK(ctx, hipMalloc(&ctx->dGramPart, (size_t)nb * (ctx->Np / 64) * ctx->Rp * ctx->Rp * sizeof(double)));
	HIPCHK(ctx, hipMalloc(&ctx->dGradSum, (size_t)nb * gpemu_ctx::GRAD_NP_MAX * sizeof(double)));
	HIPCHK(ctx, hipHostMalloc((void **)&ctx->hGradRing, (size_t)gpemu_ctx::RES_RING * nb * gpemu_ctx::GRAD_NP_MAX * sizeof(double)));
	ctx->batch_cap = nb;
	return GPEMU_OK;
}

extern "C" int gpemu_set_model(gpemu_ctx *ctx, int kind, int order, int N, int d, const double *X, const double *y)
{
	if (!ctx || !X || !y) return GPEMU_ERR_ARG;
	if (kind < GPEMU_POWEREXP || kind > GPEMU_MATERN52) return fail(ctx, GPEMU_ERR_ARG, "bad cov_fn_index");
	if (order < 0 || order > 3) return fail(ctx, GPEMU_ERR_ARG, "regression_order must be 0..3");
	if (N < 1 || d < 1 || d > GPEMU_MAX_PARAMS) return fail(ctx, GPEMU_ERR_ARG, "bad N or nparams");
	const int nreg = 1 + order * d;
	if (nreg + 1 > 64) return fail(ctx, GPEMU_ERR_ARG, "1 + nregression_fns must be <= 64");
	HIPCHK(ctx, hipSetDevice(ctx->device));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	free_model(ctx);
	ctx->res_seq = 0;                       // batches of the previous model can no longer be collected (their sizes are gone)
	ctx->kind = kind; ctx->order = order; ctx->N = N; ctx->d = d; ctx->nreg = nreg; ctx->nrhs = nreg + 1;
	ctx->Np = round_up(N, LEAF);
	ctx->Rp = 64;
	ctx->hX.assign(X, X + (size_t)N * d);
	ctx->hY.assign(y, y + N);
	HIPCHK(ctx, hipMalloc(&ctx->dX, (size_t)N * d * sizeof(double)));
	HIPCHK(ctx, hipMalloc(&ctx->dY, (size_t)N * sizeof(double)));
	HIPCHK(ctx, hipMalloc(&ctx->dRrows, (size_t)ctx->Rp * ctx->Np * sizeof(double)));
	if (ctx->batch_cap > 0)
		HIPCHK(ctx, hipMalloc(&ctx->dGramPart, (size_t)ctx->batch_cap * (ctx->Np / 64) * ctx->Rp * ctx->Rp * sizeof(double)));
	HIPCHK(ctx, hipMemcpyAsync(ctx->dX, ctx->hX.data(), (size_t)N * d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	HIPCHK(ctx, hipMemcpyAsync(ctx->dY, ctx->hY.data(), (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	{
		// the design centred per dimension (operands of the Gram-form fill) and the half ranges that bound its error
		std::vector<double> lo(d, HUGE_VAL), hi(d, -HUGE_VAL), xg((size_t)N * d);
		bool finite = true;
		for (int i = 0; i < N; i++)
			for (int k = 0; k < d; k++) {
				const double v = X[(size_t)i * d + k];
				if (!(fabs(v) <= 1e300)) finite = false;
				if (v < lo[k]) lo[k] = v;
				if (v > hi[k]) hi[k] = v;
			}
		ctx->xhalf.assign(d, HUGE_VAL);
		if (finite) {
			for (int k = 0; k < d; k++) ctx->xhalf[k] = 0.5 * (hi[k] - lo[k]);
			for (int i = 0; i < N; i++)
				for (int k = 0; k < d; k++) xg[(size_t)i * d + k] = X[(size_t)i * d + k] - 0.5 * (hi[k] + lo[k]);
			HIPCHK(ctx, hipMalloc(&ctx->dXg, (size_t)N * d * sizeof(double)));
			// (on the context's own stream, never the legacy stream: a plain hipMemcpy fails with "operation would make the
			// legacy stream depend on a capturing blocking stream" while ANOTHER host thread records its launch graph)
			HIPCHK(ctx, hipMemcpyAsync(ctx->dXg, xg.data(), (size_t)N * d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
			std::vector<double> mid(d);
			for (int k = 0; k < d; k++) mid[k] = 0.5 * (hi[k] + lo[k]);
			HIPCHK(ctx, hipMalloc(&ctx->dMid, (size_t)d * sizeof(double)));
			HIPCHK(ctx, hipMemcpyAsync(ctx->dMid, mid.data(), (size_t)d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
			HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
		}
	}
	HIPCHK(ctx, launch_build_rrows(ctx->stream, ctx->dRrows, ctx->Np, ctx->Rp, ctx->dX, ctx->dY, N, d, order));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	// (the factorisation workspace is made by the first factorisation, stage_matrices: a context whose prediction state comes
	// from gpemu_predict_setup_batch never factors anything itself)
	return GPEMU_OK;
}

extern "C" int gpemu_set_training(gpemu_ctx *ctx, const double *y)
{
	if (!ctx || !y || !ctx->dX) return GPEMU_ERR_ARG;
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	ctx->hY.assign(y, y + ctx->N);
	HIPCHK(ctx, hipMemcpyAsync(ctx->dY, ctx->hY.data(), (size_t)ctx->N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	HIPCHK(ctx, launch_build_rrows(ctx->stream, ctx->dRrows, ctx->Np, ctx->Rp, ctx->dX, ctx->dY, ctx->N, ctx->d, ctx->order));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	ctx->pred_ready = false; ctx->cinv_ready = false;
	return GPEMU_OK;
}

static int nthetas_for(const gpemu_ctx *ctx) { return ctx->kind == GPEMU_POWEREXP ? ctx->d + 2 : 3; }

// theta layout: modelstruct.c:300-308.  pow-exp exponentiates everything
// (emulator.c:115-123); the Matern kernels take amp and nugget raw (:355-357).
static int make_cov_params(gpemu_ctx *ctx, const double *thetas, int nthetas, CovParams *p)
{
	if (!thetas) return fail(ctx, GPEMU_ERR_ARG, "thetas is NULL");
	if (nthetas < nthetas_for(ctx)) return fail(ctx, GPEMU_ERR_ARG, "nthetas too small for this covariance function");
	memset(p, 0, sizeof *p);
	p->kind = ctx->kind;
	p->d = ctx->d;
	if (ctx->kind == GPEMU_POWEREXP) {
		p->amp = exp(thetas[0]);
		p->nug = exp(thetas[1]);
		p->eps = 0.0000000001;
		double wmax = 0.0;
		for (int k = 0; k < ctx->d; k++) {
			double r = exp(thetas[k + 2]);
			p->w[k] = sqrt(0.5) / r;
			if (p->w[k] > wmax) wmax = p->w[k];
		}
		p->cand = 2.0 * ctx->d * (p->eps * wmax) * (p->eps * wmax) + 1e-300;
	} else {
		// literal (emulator.c:355-356, 448-449): amplitude and nugget are used as they come.  GPEMU_MODE_MATERN_LOG
		// (SURVEY App. C2 "fixed mode"): both on the log scale like the pow-exp kernel's, so that evalFnMulti's
		// theta[0] = 0 (maxmultimin.c:311) means amplitude 1 and the model can be trained.
		const bool logscale = (ctx->mode & GPEMU_MODE_MATERN_LOG) != 0;
		p->amp = logscale ? exp(thetas[0]) : thetas[0];
		p->nug = logscale ? exp(thetas[1]) : thetas[1];
		p->eps = 0.0000000000000001;
		p->w[0] = 1.0 / exp(thetas[2]);
		p->cand = 2.0 * ctx->d * (p->eps * p->w[0]) * (p->eps * p->w[0]) + 1e-300;
	}
	// Gram form of the training fill (kernels_cov.hip): only while the centred, scaled design stays small -- the
	// cancellation error of |x'|^2 + |y'|^2 - 2 x'.y' is a few ulp of 2 * norm2
	p->gram = 0; p->cand_g = 0.0; p->cand_w = p->cand;
	if (ctx->dXg && (int)ctx->xhalf.size() == ctx->d) {
		double norm2 = 0.0;
		for (int k = 0; k < ctx->d; k++) {
			const double t = ctx->xhalf[k] * p->w[ctx->kind == GPEMU_POWEREXP ? k : 0];
			norm2 += t * t;
		}
		// the candidates of the nugget rule in a Gram-form distance: below the difference form's bound plus the form's own
		// cancellation error (a few ulp of |x'|^2 + |y'|^2 <= 2 norm2, 64 ulp allowed), whatever the length scales
		p->cand_w = p->cand + 64.0 * 2.220446049250313e-16 * (2.0 * norm2 + 1.0);
		if (ctx->sched.fill_gram && norm2 <= 16.0) {
			p->gram = 1;
			p->cand_g = p->cand_w;
		}
	}
	return GPEMU_OK;
}

// ---------------------------------------------------------------------------
// recursive tall Cholesky
//   T rows [0,Np)            : C (lower), identity padded
//   T rows [Np,Np+Rp)        : right-hand sides as rows (y, H columns) -> Z^T = (L^-1 [y|H])^T
//   T rows [Np+Rp,Np+Rp+Np)  : identity -> U = L^-T (only with inv)
// potrf_rec(c0,n) factors the column panel [c0,c0+n) for every row below it.
// ---------------------------------------------------------------------------
// *fa_done: set when the update ran with the factor-ahead tile, i.e. the 64x64 diagonal block at c0+k is already
// factored when the update has finished and the next leaf must not factor it again
static hipError_t trailing_update(gpemu_ctx *ctx, int c0, int k, int ncols, int inv, bool *fa_done)
{
	// C[rows >= r0, cols r0 .. r0+ncols) -= P P^T with P = the factored panel columns [c0, c0+k) and r0 = c0 + k
	const long ld = ctx->Np;
	const int r0 = c0 + k;
	const int row_end = ctx->Np + ctx->Rp + (inv ? c0 + k : 0);   // identity rows < c0+k have fill-in in the panel
	GemmArgs g;
	memset(&g, 0, sizeof g);
	g.C = ctx->dT + (long)r0 * ld + r0;
	g.A = ctx->dT + (long)r0 * ld + c0;
	g.B = g.A;
	g.ldc = g.lda = g.ldb = ld;
	g.m = row_end - r0;
	g.n = ncols;
	g.k0 = 0; g.k1 = k;
	g.alpha = -1.0; g.beta = 1;
	g.tri = 1; g.diag_off = 0;
	g.nbatch = ctx->nb; g.bsC = g.bsA = g.bsB = (long)ctx->T_stride;
	g.big_tiles = ctx->sched.gemm_big_tiles;                      // (gemm() sets these too: needed here for the tile-shape question)
	g.fa = ctx->sched.factor_ahead ? 1 : 0;
	g.fa_c0 = r0;
	g.fa_info = ctx->dInfo;
	*fa_done = g.fa && gemm_factor_ahead_ok(g);
	if (!*fa_done) g.fa = 0;
	const int c_rows = ctx->Np - r0;                              // rows of the matrix proper under r0
	if (ctx->sched.split_rhs_rows && gemm_uses_big_tiles(g) && c_rows % GEMM_BM == 0 && c_rows >= ncols && g.m > c_rows) {
		// 128x128 tiles: the 64 right-hand-side rows between the matrix rows and the identity rows would shift every tile row
		// behind them by half a tile and leave the last one half empty (2-5 % of the tile slots of a big update: 48 of 1224
		// at the first update of a batch at N = 8192).  Three launches instead: the matrix rows (triangular part, full 128-row
		// tiles), the right-hand-side rows on 64x64 tiles, the identity rows (gradient / inverse only).  Same k-ordered chain
		// per element whatever the tile shape: the bits do not change.
		GemmArgs a = g;
		a.m = c_rows;
		hipError_t e = gemm(ctx, a);
		if (e != hipSuccess) return e;
		GemmArgs b = g;
		b.C = g.C + (long)c_rows * ld; b.A = g.A + (long)c_rows * ld;
		b.m = ctx->Rp; b.tri = 0; b.force_cfg = 2;
		e = gemm(ctx, b);
		if (e != hipSuccess) return e;
		const int i_rows = g.m - c_rows - ctx->Rp;
		if (i_rows > 0) {
			GemmArgs c = g;
			c.C = g.C + (long)(c_rows + ctx->Rp) * ld; c.A = g.A + (long)(c_rows + ctx->Rp) * ld;
			c.m = i_rows; c.tri = 0;
			e = gemm(ctx, c);
		}
		return e;
	}
	return gemm(ctx, g);
}

// diag_done: the 64x64 diagonal block at (c0,c0) is already factored (by the factor-ahead tile of the update before)
// defer_c0 >= 0: the leaf solve of this block also solves, in place, the 64 rows under the diagonal block at defer_c0 (the
// pair's first block, which leaf_pair_kernel leaves untouched there)
static hipError_t potrf_rec(gpemu_ctx *ctx, int c0, int n, int inv, bool diag_done = false, int defer_c0 = -1)
{
	const long ld = ctx->Np;
	const int base_end = ctx->Np + ctx->Rp;
	if (n <= LEAF) {
		const int row_end = base_end + (inv ? c0 + LEAF : 0);
		ProfScope ps(ctx, GPEMU_PROF_LEAF, 0.0, 0.0);
		unsigned long long *trf = trace_slot(ctx, "leaf_factor c0=%d", c0);
		unsigned long long *trs = trace_slot(ctx, "leaf_solve c0=%d m=%d", c0, row_end - (c0 + LEAF));
		return launch_leaf(ctx->stream, ctx->dT, ld, c0, row_end - (c0 + LEAF), ctx->dInfo, trf, trs, ctx->nb,
		                   (long)ctx->T_stride, diag_done, ctx->sched.leaf_staged, ctx->sched.diag_inv_ahead != 0, defer_c0);
	}
	if (n == 2 * LEAF && ctx->sched.leaf_pair && ctx->sched.diag_inv_ahead) {
		// a 128-column pair: [factor the first diagonal block,] leaf_pair_kernel (solve of the first block + K=64 update with
		// its factor-ahead tile), then the second block's leaf solve with the deferred 64 rows of the first
		const int row_end = base_end + (inv ? c0 + LEAF : 0);
		const int m_below = row_end - (c0 + LEAF);
		ProfScope ps(ctx, GPEMU_PROF_LEAF, 0.0, 0.0);
		if (!diag_done) {
			unsigned long long *trf = trace_slot(ctx, "leaf_factor c0=%d", c0);
			hipError_t e = launch_leaf(ctx->stream, ctx->dT, ld, c0, 0, ctx->dInfo, trf, nullptr, ctx->nb, (long)ctx->T_stride, false);
			if (e != hipSuccess) return e;
		}
		unsigned long long *trp = trace_slot(ctx, "leaf_pair c0=%d m=%d", c0, m_below);
		const bool fa = ctx->sched.factor_ahead != 0;
		hipError_t e = launch_leaf_pair(ctx->stream, ctx->dT, ld, c0, m_below, ctx->dInfo, trp, ctx->nb, (long)ctx->T_stride, fa);
		if (e != hipSuccess) return e;
		return potrf_rec(ctx, c0 + LEAF, LEAF, inv, fa, c0);
	}
	// automatic outer panel width: a batch has enough tiles per launch to afford the longer panel chain of a wider
	// panel and gains from the larger K of its trailing updates and the fewer read-modify-write passes over the
	// trailing matrix (measured at 2x16, N=8192: 3.77 ms per evaluation at 512, 3.55 at 1024 with the first GEMM
	// epilogue; 3.329 at 1024, 3.302 at 2048, 3.314 at 4096 now; 2048 also wins at N = 4096, 12288, 16384)
	// With the inverse rows under the matrix (gradient, explicit inverse) 1024 is better again: 10.40 against 10.77 ms
	// per value+gradient evaluation in batches of 16.
	// Round 5, measured at N = 4096 (profiles/r05_n4096_schedule_switches.txt): with the inverse rows 512 beats 1024 there
	// (value+gradient batches of 16 / 64: +2 %), without them 1024 .. 4096 are within 0.5 % of each other.
	const int nb_top = ctx->sched.nb_top > 0 ? ctx->sched.nb_top : (ctx->nb >= 2 ? (inv ? (ctx->Np <= 4096 ? 512 : 1024) : 2048) : 512);
	if (n > nb_top) {
		// right-looking over panels of nb_top columns: the trailing update touches the whole remaining
		// matrix (thousands of tiles, K = panel width), which fills the chip far better than the few huge-K
		// tiles a pure recursion would produce at the top levels
		bool next_done = diag_done;
		for (int c = c0; c < c0 + n; c += nb_top) {
			const int nb = std::min(nb_top, c0 + n - c);
			hipError_t e = potrf_rec(ctx, c, nb, inv, next_done);
			next_done = false;
			if (e != hipSuccess) return e;
			const int rest = c0 + n - (c + nb);
			if (rest <= 0) continue;
			e = trailing_update(ctx, c, nb, rest, inv, &next_done);
			if (e != hipSuccess) return e;
		}
		return hipSuccess;
	}
	const int n1 = ((n / LEAF + 1) / 2) * LEAF;
	hipError_t e = potrf_rec(ctx, c0, n1, inv, diag_done);
	if (e != hipSuccess) return e;
	bool right_done = false;
	e = trailing_update(ctx, c0, n1, n - n1, inv, &right_done);
	if (e != hipSuccess) return e;
	return potrf_rec(ctx, c0 + n1, n - n1, inv, right_done);
}

static int run_potrf(gpemu_ctx *ctx, int inv)
{
	const bool profiling = ctx->prof.cls == GPEMU_PROF_GEMM || ctx->prof.cls == GPEMU_PROF_LEAF || ctx->prof.cls == GPEMU_PROF_GEMM_BIG ||
	                       ctx->prof.cls == GPEMU_PROF_GEMM_K512;
	if (ctx->dTrace) {
		HIPCHK(ctx, hipMemsetAsync(ctx->dTrace, 0, (size_t)ctx->trace_cap * 64, ctx->stream));   // fresh slots
	}
	double fl = (double)ctx->nb * ctx->Np * ctx->Np * ctx->Np / 3.0;
	ProfScope ps(ctx, GPEMU_PROF_POTRF, fl, 0.0);
	if (!ctx->use_graph || profiling) {
		ctx->trace_next = 0; ctx->trace_tag.clear();
		HIPCHK(ctx, potrf_rec(ctx, 0, ctx->Np, inv));
		return GPEMU_OK;
	}
	gpemu_ctx::GraphKey key{ctx->Np, ctx->Rp, inv, ctx->nb};
	auto it = ctx->graphs.find(key);
	if (it == ctx->graphs.end() && ctx->warm.insert(key).second) {
		// first factorisation of this shape: plain launches (host-side tables of the GEMM tile order are built
		// on first use and cannot be allocated under stream capture); the next call records the graph
		ctx->trace_next = 0; ctx->trace_tag.clear();
		HIPCHK(ctx, potrf_rec(ctx, 0, ctx->Np, inv));
		return GPEMU_OK;
	}
	if (it == ctx->graphs.end()) {
		hipGraph_t graph = nullptr;
		ctx->trace_next = 0; ctx->trace_tag.clear();
		HIPCHK(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal));
		hipError_t e = potrf_rec(ctx, 0, ctx->Np, inv);
		hipError_t e2 = hipStreamEndCapture(ctx->stream, &graph);
		if (e != hipSuccess || e2 != hipSuccess) {
			if (graph) hipGraphDestroy(graph);
			ctx->err = "potrf graph capture failed";
			return GPEMU_ERR_HIP;
		}
		hipGraphExec_t exec = nullptr;
		HIPCHK(ctx, hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
		hipGraphDestroy(graph);
		it = ctx->graphs.emplace(key, exec).first;
	}
	HIPCHK(ctx, hipGraphLaunch(it->second, ctx->stream));
	return GPEMU_OK;
}

// fill C(theta_b) into matrix b of T (lower tiles only), load the RHS rows, reset the info words
constexpr unsigned PARAM_RING = 4;

// rrows / rstride: right-hand-side rows of the batch when they are not the context's own (rstride != 0: matrix b takes
// rrows + b * rstride -- the components of a multi-output model, gpemu_predict_setup_batch)
static int stage_matrices(gpemu_ctx *ctx, const CovParams *ps, int nb, int inv, const double *rrows = nullptr, long rstride = 0)
{
	const int Np = ctx->Np, Rp = ctx->Rp;
	int rc = ensure_batch_slots(ctx, nb);
	if (rc) return rc;
	rc = ensure_T(ctx, (size_t)Np + Rp + (inv ? Np : 0), nb);
	if (rc) return rc;
	// one upload of the nb hyper-parameter sets, one launch for the nb fills and R-row copies.  The upload goes through a
	// pinned ring of four entries (a pageable source makes hipMemcpyAsync wait for the stream: enqueued batches of small
	// models then take 6 us per evaluation instead of 2); an entry is reused once its own copy has executed.
	if (!ctx->dParams) {
		HIPCHK(ctx, hipMalloc(&ctx->dParams, (size_t)GPEMU_MAX_BATCH * sizeof(CovParams)));
		HIPCHK(ctx, hipHostMalloc((void **)&ctx->hParams, (size_t)PARAM_RING * GPEMU_MAX_BATCH * sizeof(CovParams)));
		HIPCHK(ctx, hipHostMalloc((void **)&ctx->hGph, (size_t)PARAM_RING * GPEMU_MAX_BATCH * GPEMU_MAX_PARAMS * sizeof(double)));
		for (unsigned i = 0; i < PARAM_RING; i++) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->param_ev[i], hipEventDisableTiming));
	}
	{
		const unsigned slot = ctx->param_next++ % PARAM_RING;
		ctx->param_slot = slot;
		HIPCHK(ctx, hipEventSynchronize(ctx->param_ev[slot]));        // (a never-recorded event is complete)
		CovParams *hp = ctx->hParams + (size_t)slot * GPEMU_MAX_BATCH;
		memcpy(hp, ps, (size_t)nb * sizeof(CovParams));
		HIPCHK(ctx, hipMemcpyAsync(ctx->dParams, hp, (size_t)nb * sizeof(CovParams), hipMemcpyHostToDevice, ctx->stream));
		HIPCHK(ctx, hipEventRecord(ctx->param_ev[slot], ctx->stream));
	}
	{
		const double nlow = 0.5 * (double)Np * Np * nb;
		ProfScope ps_(ctx, GPEMU_PROF_FILL, 0.0, 8.0 * nlow);
		bool all_gram = ctx->dXg != nullptr;
		for (int b = 0; b < nb; b++) all_gram = all_gram && ps[b].gram;
		HIPCHK(ctx, launch_cov_stage_batch(ctx->stream, ctx->dT, Np, (long)ctx->T_stride, nb, ctx->dX, ctx->N, Np, ctx->d,
		                                   ctx->dParams, FILL_LOWER | FILL_IDENT_PAD, rrows ? rrows : ctx->dRrows, Rp, ctx->dXg, all_gram,
		                                   ctx->kind, rrows ? rstride : 0));
	}
	if (inv)
		HIPCHK(ctx, launch_set_identity_rows(ctx->stream, ctx->dT + (size_t)(Np + Rp) * Np, Np, Np, nb, (long)ctx->T_stride));
	HIPCHK(ctx, hipMemsetAsync(ctx->dInfo, 0x7f, (size_t)nb * sizeof(int), ctx->stream));
	return GPEMU_OK;
}

static int stage_matrix(gpemu_ctx *ctx, const CovParams &p, int inv) { return stage_matrices(ctx, &p, 1, inv); }

static int enqueue_results(gpemu_ctx *ctx)
{
	const int Np = ctx->Np, Rp = ctx->Rp, nb = ctx->nb;
	HIPCHK(ctx, launch_gram_partials(ctx->stream, ctx->dT + (size_t)Np * Np, Np, Np, ctx->nrhs, Rp, ctx->dGramPart, nb,
	                                 (long)ctx->T_stride));
	HIPCHK(ctx, launch_finish(ctx->stream, ctx->dGramPart, Np / 64, Rp, ctx->nrhs, ctx->dT, Np, ctx->N, ctx->dRes, nb,
	                          (long)ctx->T_stride, (long)ctx->res_len));
	// the results land in the next slot of a pinned ring (RES_RING batches stay readable: a throughput caller collects
	// batch j while batches j+1 .. j+RES_RING-1 are in flight); hRes / hInfo point at the newest slot
	const int slot = (int)(ctx->res_seq % gpemu_ctx::RES_RING);
	ctx->res_seq++;
	ctx->hRes = ctx->hResRing + (size_t)slot * ctx->batch_cap * ctx->res_len;
	ctx->hInfo = ctx->hInfoRing + (size_t)slot * ctx->batch_cap;
	ctx->res_nb[slot] = nb;
	ctx->res_kind[slot] = 0;
	HIPCHK(ctx, hipMemcpyAsync(ctx->hRes, ctx->dRes, ((size_t)(nb - 1) * ctx->res_len + (size_t)Rp * Rp + 1) * sizeof(double),
	                           hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(ctx, hipMemcpyAsync(ctx->hInfo, ctx->dInfo, (size_t)nb * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(ctx, hipEventRecord(ctx->res_ev[slot], ctx->stream));
	return GPEMU_OK;
}

// small dense Cholesky solve on the host (nreg x nreg): beta and Q = (H^T C^-1 H)^-1
// (regression.c:120-176 estimateBeta restated on the Gram matrix)
static bool small_chol_inverse(std::vector<double> &A, int n)
{
	for (int j = 0; j < n; j++) {
		double dsum = A[j * n + j];
		for (int k = 0; k < j; k++) dsum -= A[j * n + k] * A[j * n + k];
		if (!(dsum > 0.0)) return false;
		const double l = sqrt(dsum);
		A[j * n + j] = l;
		for (int i = j + 1; i < n; i++) {
			double s = A[i * n + j];
			for (int k = 0; k < j; k++) s -= A[i * n + k] * A[j * n + k];
			A[i * n + j] = s / l;
		}
	}
	// invert L, then A^-1 = L^-T L^-1
	std::vector<double> Li((size_t)n * n, 0.0);
	for (int c = 0; c < n; c++)
		for (int i = c; i < n; i++) {
			double s = (i == c) ? 1.0 : 0.0;
			for (int k = c; k < i; k++) s -= A[i * n + k] * Li[k * n + c];
			Li[i * n + c] = s / A[i * n + i];
		}
	for (int i = 0; i < n; i++)
		for (int j = 0; j < n; j++) {
			double s = 0.0;
			for (int k = std::max(i, j); k < n; k++) s += Li[k * n + i] * Li[k * n + j];
			A[i * n + j] = s;
		}
	return true;
}

struct HostLik { double yy, quad, sigma2, logdet; std::vector<double> beta, Q, Hy; int status; };

static HostLik host_likelihood(gpemu_ctx *ctx, int b = 0)
{
	HostLik r;
	const int Rp = ctx->Rp, nreg = ctx->nreg;
	const double *G = ctx->hRes + (size_t)b * ctx->res_len;
	r.logdet = G[Rp * Rp];
	r.yy = G[0];
	r.Hy.resize(nreg);
	r.Q.assign((size_t)nreg * nreg, 0.0);
	std::vector<double> HH((size_t)nreg * nreg);
	for (int a = 0; a < nreg; a++) {
		r.Hy[a] = G[(1 + a) * Rp];
		for (int b = 0; b < nreg; b++) HH[a * nreg + b] = G[(1 + a) * Rp + 1 + b];
	}
	r.Q = HH;
	r.beta.assign(nreg, NAN);
	r.quad = r.sigma2 = NAN;
	if (!small_chol_inverse(r.Q, nreg)) { r.status = GPEMU_ERR_REGRESSION; return r; }
	for (int a = 0; a < nreg; a++) {
		double s = 0.0;
		for (int b = 0; b < nreg; b++) s += r.Q[a * nreg + b] * r.Hy[b];
		r.beta[a] = s;
	}
	double bHy = 0.0, bHHb = 0.0;
	for (int a = 0; a < nreg; a++) {
		bHy += r.beta[a] * r.Hy[a];
		double s = 0.0;
		for (int b = 0; b < nreg; b++) s += HH[a * nreg + b] * r.beta[b];
		bHHb += r.beta[a] * s;
	}
	r.sigma2 = (r.yy - bHy) / (double)ctx->N;           // y.Cinv.(y - H beta)/N   (maxmultimin.c:259-263)
	r.quad = r.yy - 2.0 * bHy + bHHb;                   // r.Cinv.r               (estimator-fns.c:87-88)
	r.status = GPEMU_OK;
	return r;
}

// A batch of nb likelihood evaluations of the same model at nb theta vectors, factored in lock-step: every
// kernel of the factorisation handles all nb matrices (grid.y), so the latency-bound panel chain is paid once per
// batch and the trailing updates are nb times larger launches.  This is the device form of the reference's
// callEvalLhoodList (libRbind/rbind.c:626) and of the optimiser's independent restarts.
extern "C" int gpemu_loglik_batch_enqueue(gpemu_ctx *ctx, int nb, const double *thetas, int nthetas)
{
	if (!ctx) return GPEMU_ERR_ARG;
	if (!ctx->dX) return fail(ctx, GPEMU_ERR_STATE, "model not set");
	if (nb < 1 || nb > GPEMU_MAX_BATCH) return fail(ctx, GPEMU_ERR_ARG, "batch size must be 1..GPEMU_MAX_BATCH");
	std::vector<CovParams> ps((size_t)nb);
	for (int b = 0; b < nb; b++) {
		int rc = make_cov_params(ctx, thetas ? thetas + (size_t)b * nthetas : nullptr, nthetas, &ps[b]);
		if (rc) return rc;
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	int rc = stage_matrices(ctx, ps.data(), nb, 0);
	if (rc) return rc;
	rc = run_potrf(ctx, 0);
	if (rc) return rc;
	ctx->pred_ready = false; ctx->cinv_ready = false;
	return enqueue_results(ctx);
}

// results of element b of the last enqueued batch (after the stream has been synchronised)
static int collect_one(gpemu_ctx *ctx, int b, double *neg_loglik, double *sigma2, double *beta, double *logdet,
                       double *quad, int *info)
{
	const int inf = (ctx->hInfo[b] >= INFO_NONE) ? 0 : ctx->hInfo[b];
	if (info) *info = inf;
	if (inf != 0) {
		if (neg_loglik) *neg_loglik = NAN;
		if (sigma2) *sigma2 = NAN;
		if (logdet) *logdet = NAN;
		if (quad) *quad = NAN;
		if (beta) for (int a = 0; a < ctx->nreg; a++) beta[a] = NAN;
		return fail(ctx, GPEMU_ERR_NOT_PD, "covariance matrix is not positive definite");
	}
	HostLik r = host_likelihood(ctx, b);
	if (beta) for (int a = 0; a < ctx->nreg; a++) beta[a] = r.beta[a];
	if (sigma2) *sigma2 = r.sigma2;
	if (logdet) *logdet = r.logdet;
	if (quad) *quad = r.quad;
	if (neg_loglik) {
		const double log_2_pi = 1.83788;                                     // estimator-fns.c:48 (literal)
		const double ll = -(1.0 / 2.0) * r.logdet - (ctx->N / 2.0) * log_2_pi + r.quad * (-1.0 / 2.0);
		*neg_loglik = -1 * ll;
	}
	if (r.status) return fail(ctx, r.status, "H^T C^-1 H is not positive definite");
	return GPEMU_OK;
}

extern "C" int gpemu_loglik_batch_collect(gpemu_ctx *ctx, int nb, double *neg_loglik, double *sigma2, double *beta,
                                          double *logdet, double *quad, int *info, int *status)
{
	if (!ctx) return GPEMU_ERR_ARG;
	if (nb < 1 || nb != ctx->nb) return fail(ctx, GPEMU_ERR_STATE, "batch size differs from the enqueued batch");
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	for (int b = 0; b < nb; b++) {
		const int rc = collect_one(ctx, b, neg_loglik ? neg_loglik + b : nullptr, sigma2 ? sigma2 + b : nullptr,
		                           beta ? beta + (size_t)b * ctx->nreg : nullptr, logdet ? logdet + b : nullptr,
		                           quad ? quad + b : nullptr, info ? info + b : nullptr);
		if (status) status[b] = rc;
	}
	return GPEMU_OK;
}

// results of the batch enqueued `back` batches before the newest one (0 = newest); waits for THAT batch only
extern "C" int gpemu_loglik_batch_collect_back(gpemu_ctx *ctx, int back, int nb, double *neg_loglik, double *sigma2,
                                               double *beta, double *logdet, double *quad, int *info, int *status)
{
	if (!ctx) return GPEMU_ERR_ARG;
	if (back < 0 || back >= gpemu_ctx::RES_RING || (unsigned long long)back >= ctx->res_seq)
		return fail(ctx, GPEMU_ERR_STATE, "no such batch in the result ring");
	const int slot = (int)((ctx->res_seq - 1 - (unsigned long long)back) % gpemu_ctx::RES_RING);
	if (nb < 1 || nb != ctx->res_nb[slot]) return fail(ctx, GPEMU_ERR_STATE, "batch size differs from the enqueued batch");
	if (ctx->res_kind[slot] != 0) return fail(ctx, GPEMU_ERR_STATE, "that batch is a value+gradient batch: use gpemu_loglik_grad_batch_collect_back");
	HIPCHK(ctx, hipEventSynchronize(ctx->res_ev[slot]));
	double *saveR = ctx->hRes;
	int *saveI = ctx->hInfo;
	ctx->hRes = ctx->hResRing + (size_t)slot * ctx->batch_cap * ctx->res_len;
	ctx->hInfo = ctx->hInfoRing + (size_t)slot * ctx->batch_cap;
	for (int b = 0; b < nb; b++) {
		const int rc = collect_one(ctx, b, neg_loglik ? neg_loglik + b : nullptr, sigma2 ? sigma2 + b : nullptr,
		                           beta ? beta + (size_t)b * ctx->nreg : nullptr, logdet ? logdet + b : nullptr,
		                           quad ? quad + b : nullptr, info ? info + b : nullptr);
		if (status) status[b] = rc;
	}
	ctx->hRes = saveR; ctx->hInfo = saveI;
	return GPEMU_OK;
}

extern "C" int gpemu_loglik_batch(gpemu_ctx *ctx, int nb, const double *thetas, int nthetas, double *neg_loglik,
                                  double *sigma2, double *beta, double *logdet, double *quad, int *info, int *status)
{
	int rc = gpemu_loglik_batch_enqueue(ctx, nb, thetas, nthetas);
	if (rc) return rc;
	return gpemu_loglik_batch_collect(ctx, nb, neg_loglik, sigma2, beta, logdet, quad, info, status);
}

extern "C" int gpemu_loglik_enqueue(gpemu_ctx *ctx, const double *thetas, int nthetas)
{
	if (!ctx) return GPEMU_ERR_ARG;
	return gpemu_loglik_batch_enqueue(ctx, 1, thetas, nthetas);
}

extern "C" int gpemu_loglik_collect(gpemu_ctx *ctx, double *neg_loglik, double *sigma2, double *beta, double *logdet,
                                    double *quad, int *info)
{
	if (!ctx) return GPEMU_ERR_ARG;
	if (ctx->nb != 1) return fail(ctx, GPEMU_ERR_STATE, "the enqueued work is a batch: use gpemu_loglik_batch_collect");
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	return collect_one(ctx, 0, neg_loglik, sigma2, beta, logdet, quad, info);
}

extern "C" int gpemu_loglik(gpemu_ctx *ctx, const double *thetas, int nthetas, double *neg_loglik, double *sigma2,
                            double *beta, double *logdet, double *quad, int *info)
{
	int rc = gpemu_loglik_enqueue(ctx, thetas, nthetas);
	if (rc) return rc;
	return gpemu_loglik_collect(ctx, neg_loglik, sigma2, beta, logdet, quad, info);
}

// ---------------------------------------------------------------------------
// covariance matrix / k vectors to host
// ---------------------------------------------------------------------------
extern "C" int gpemu_cov_matrix(gpemu_ctx *ctx, const double *thetas, int nthetas, double *c_out)
{
	if (!ctx || !c_out) return GPEMU_ERR_ARG;
	if (!ctx->dX) return fail(ctx, GPEMU_ERR_STATE, "model not set");
	CovParams p;
	int rc = make_cov_params(ctx, thetas, nthetas, &p);
	if (rc) return rc;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	const int Np = ctx->Np, N = ctx->N;
	double *buf = nullptr;
	HIPCHK(ctx, hipMalloc(&buf, (size_t)Np * Np * sizeof(double)));
	hipError_t e = launch_cov_fill(ctx->stream, buf, Np, ctx->dX, N, Np, ctx->dX, N, Np, ctx->d, p, 0);
	if (e == hipSuccess)
		e = hipMemcpy2DAsync(c_out, (size_t)N * sizeof(double), buf, (size_t)Np * sizeof(double),
		                     (size_t)N * sizeof(double), N, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	hipFree(buf);
	HIPCHK(ctx, e);
	return GPEMU_OK;
}

// k-vectors of M query rows (device) into out (Mp x Np, zero padded): Gram form when the hyper-parameters admit it
// (make_cov_params: p.gram) and the context was not told otherwise, else the difference form
static hipError_t fill_kvectors(gpemu_ctx *ctx, double *out, const double *xq_dev, int M, int Mp, const CovParams &p)
{
	if (p.gram && ctx->sched.kvec_gram && ctx->dXg && ctx->dMid)
		return launch_cov_kvec_gram(ctx->stream, out, ctx->Np, xq_dev, M, Mp, ctx->dX, ctx->dXg, ctx->dMid, ctx->N, ctx->Np, ctx->d, p);
	return launch_cov_fill(ctx->stream, out, ctx->Np, xq_dev, M, Mp, ctx->dX, ctx->N, ctx->Np, ctx->d, p, FILL_CLAMP);
}

extern "C" int gpemu_kvectors(gpemu_ctx *ctx, const double *thetas, int nthetas, int M, const double *xq, double *k_out)
{
	if (!ctx || !xq || !k_out || M < 1) return GPEMU_ERR_ARG;
	if (!ctx->dX) return fail(ctx, GPEMU_ERR_STATE, "model not set");
	CovParams p;
	int rc = make_cov_params(ctx, thetas, nthetas, &p);
	if (rc) return rc;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	const int Np = ctx->Np, N = ctx->N, Mp = round_up(M, 64);
	double *buf = nullptr, *dq = nullptr;
	HIPCHK(ctx, hipMalloc(&buf, (size_t)Mp * Np * sizeof(double)));
	hipError_t e = hipMalloc(&dq, (size_t)M * ctx->d * sizeof(double));
	if (e == hipSuccess) e = hipMemcpyAsync(dq, xq, (size_t)M * ctx->d * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) e = fill_kvectors(ctx, buf, dq, M, Mp, p);
	if (e == hipSuccess)
		e = hipMemcpy2DAsync(k_out, (size_t)N * sizeof(double), buf, (size_t)Np * sizeof(double),
		                     (size_t)N * sizeof(double), M, hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	hipFree(buf);
	if (dq) hipFree(dq);
	HIPCHK(ctx, e);
	return GPEMU_OK;
}

// ---------------------------------------------------------------------------
// prediction
// ---------------------------------------------------------------------------
static int factor_with_inverse(gpemu_ctx *ctx, const double *thetas, int nthetas, CovParams *p, int *info)
{
	int rc = make_cov_params(ctx, thetas, nthetas, p);
	if (rc) return rc;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	rc = stage_matrix(ctx, *p, 1);
	if (rc) return rc;
	rc = run_potrf(ctx, 1);
	if (rc) return rc;
	rc = enqueue_results(ctx);
	if (rc) return rc;
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	const int inf = (*ctx->hInfo >= INFO_NONE) ? 0 : *ctx->hInfo;
	if (info) *info = inf;
	if (inf) return fail(ctx, GPEMU_ERR_NOT_PD, "covariance matrix is not positive definite");
	return GPEMU_OK;
}

// The prediction state of `dst` from element b of the factorisation (with inverse rows) that sits in `src`'s workspace and
// whose results have been collected (src == dst, b == 0: the single call).  Everything runs on src's stream, which is
// synchronised before the function returns; dst's own stream has nothing in flight (checked by the callers).
static int build_prediction_state(gpemu_ctx *src, int b, gpemu_ctx *dst, const CovParams &p, const HostLik &r, const double *thetas,
                                  int nthetas)
{
	const int Np = src->Np, Rp = src->Rp, nreg = src->nreg, N = src->N;
	const size_t la_rows = (size_t)Np + Rp;
	if (!dst->dLinvAug) HIPCHK(src, hipMalloc(&dst->dLinvAug, la_rows * Np * sizeof(double)));
	if (!dst->dBetaQ) HIPCHK(src, hipMalloc(&dst->dBetaQ, (size_t)(nreg + nreg * nreg) * sizeof(double)));
	const double *Tb = src->dT + (size_t)b * src->T_stride;
	const double *Zt = Tb + (size_t)Np * Np;
	const double *U = Tb + (size_t)(Np + Rp) * Np;
	// rows [0,Np): L^-1 = U^T
	HIPCHK(src, launch_transpose(src->stream, dst->dLinvAug, Np, U, Np, Np));
	// (C^-1 [y|H])^T = Z^T U^T : rows Np.. of LinvAug used as scratch first
	GemmArgs g;
	memset(&g, 0, sizeof g);
	g.C = dst->dLinvAug + (size_t)Np * Np; g.ldc = Np;
	g.A = Zt; g.lda = Np;
	g.B = U; g.ldb = Np;
	g.m = Rp; g.n = Np; g.k0 = 0; g.k1 = Np; g.alpha = 1.0; g.beta = 0;
	HIPCHK(src, gemm(src, g));
	std::vector<double> cr((size_t)Rp * Np);
	HIPCHK(src, hipMemcpyAsync(cr.data(), dst->dLinvAug + (size_t)Np * Np, cr.size() * sizeof(double),
	                           hipMemcpyDeviceToHost, src->stream));
	HIPCHK(src, hipStreamSynchronize(src->stream));
	// row 0 <- gamma = C^-1 y - (C^-1 H) beta = C^-1 (y - H beta); rows 1.. keep W^T = (C^-1 H)^T
	for (int j = 0; j < Np; j++) {
		double s = cr[j];
		for (int a = 0; a < nreg; a++) s -= r.beta[a] * cr[(size_t)(1 + a) * Np + j];
		cr[j] = (j < N) ? s : 0.0;
	}
	for (int a = src->nrhs; a < Rp; a++)
		for (int j = 0; j < Np; j++) cr[(size_t)a * Np + j] = 0.0;
	HIPCHK(src, hipMemcpyAsync(dst->dLinvAug + (size_t)Np * Np, cr.data(), cr.size() * sizeof(double),
	                           hipMemcpyHostToDevice, src->stream));
	std::vector<double> bq(nreg + (size_t)nreg * nreg);
	for (int a = 0; a < nreg; a++) bq[a] = r.beta[a];
	for (int a = 0; a < nreg * nreg; a++) bq[nreg + a] = r.Q[a];
	HIPCHK(src, hipMemcpyAsync(dst->dBetaQ, bq.data(), bq.size() * sizeof(double), hipMemcpyHostToDevice, src->stream));
	HIPCHK(src, hipStreamSynchronize(src->stream));
	dst->h_beta = r.beta; dst->h_Q = r.Q;
	dst->pred_cov = p;
	dst->kappa = p.amp + p.nug;                       // cov(x*,x*): emulator_struct.c:135 (nugget included)
	dst->pred_ready = true;
	dst->cinv_ready = false;
	dst->last_thetas.assign(thetas, thetas + nthetas);
	return GPEMU_OK;
}

extern "C" int gpemu_predict_setup(gpemu_ctx *ctx, const double *thetas, int nthetas, double *beta_out, int *info)
{
	if (!ctx) return GPEMU_ERR_ARG;
	if (!ctx->dX) return fail(ctx, GPEMU_ERR_STATE, "model not set");
	ctx->pred_ready = false; ctx->cinv_ready = false;
	CovParams p;
	int rc = factor_with_inverse(ctx, thetas, nthetas, &p, info);
	if (rc) return rc;
	HostLik r = host_likelihood(ctx);
	if (r.status) return fail(ctx, r.status, "H^T C^-1 H is not positive definite");
	rc = build_prediction_state(ctx, 0, ctx, p, r, thetas, nthetas);
	if (rc) return rc;
	ctx->fact_in_T = true;
	if (beta_out) for (int a = 0; a < ctx->nreg; a++) beta_out[a] = r.beta[a];
	return GPEMU_OK;
}

// alloc_multi_emulator (multivar_support.c:30-52) loops alloc_emulator_struct over the nr PCA components of a multi-output
// model: same design, covariance function and regression order, a training vector and thetas of its own each.  Here the nr
// factorisations with their inverse rows run as ONE lock-step batch in the first context's workspace (blockIdx.y = component,
// every component under its own right-hand-side rows), and each context receives its own prediction state; afterwards the
// contexts answer queries on their own streams as if gpemu_predict_setup had been called on each -- with the same bits: an
// element of a lock-step batch is the evaluation done alone.
extern "C" int gpemu_predict_setup_batch(gpemu_ctx *const *ctxs, int n, const double *thetas, int nthetas, double *beta_out, int *info,
                                         int *status)
{
	if (!ctxs || n < 1 || !ctxs[0]) return GPEMU_ERR_ARG;
	gpemu_ctx *lead = ctxs[0];
	if (n > GPEMU_MAX_BATCH) return fail(lead, GPEMU_ERR_ARG, "at most GPEMU_MAX_BATCH components per call");
	if (!lead->dX) return fail(lead, GPEMU_ERR_STATE, "model not set");
	for (int c = 0; c < n; c++) {
		gpemu_ctx *x = ctxs[c];
		if (!x || !x->dX) return fail(lead, GPEMU_ERR_STATE, "model not set in every context");
		for (int e = 0; e < c; e++) if (ctxs[e] == x) return fail(lead, GPEMU_ERR_ARG, "the same context twice");
		if (x->device != lead->device || x->kind != lead->kind || x->order != lead->order || x->N != lead->N || x->d != lead->d ||
		    x->mode != lead->mode || x->hX != lead->hX)
			return fail(lead, GPEMU_ERR_ARG, "the contexts of a batched set-up share device, design, covariance function, regression order and modes");
		if (x->pred_pending) return fail(lead, GPEMU_ERR_STATE, "a prediction batch is enqueued in one of the contexts: collect it first");
		x->pred_ready = false; x->cinv_ready = false;
	}
	std::vector<CovParams> ps((size_t)n);
	for (int c = 0; c < n; c++) {
		int rc = make_cov_params(lead, thetas ? thetas + (size_t)c * nthetas : nullptr, nthetas, &ps[c]);
		if (rc) return rc;
	}
	HIPCHK(lead, hipSetDevice(lead->device));
	const int Np = lead->Np, Rp = lead->Rp, nreg = lead->nreg;
	// the components' right-hand-side rows [y_c | H]^T side by side (each context built its own at gpemu_set_model /
	// gpemu_set_training, synchronously)
	double *rr = nullptr;
	const size_t rlen = (size_t)Rp * Np;
	HIPCHK(lead, hipMalloc(&rr, (size_t)n * rlen * sizeof(double)));
	hipError_t e = hipSuccess;
	for (int c = 0; c < n && e == hipSuccess; c++)
		e = hipMemcpyAsync(rr + (size_t)c * rlen, ctxs[c]->dRrows, rlen * sizeof(double), hipMemcpyDeviceToDevice, lead->stream);
	int rc = GPEMU_OK;
	if (e != hipSuccess) { lead->err = std::string("right-hand-side rows: ") + hipGetErrorString(e); rc = GPEMU_ERR_HIP; }
	if (!rc) rc = stage_matrices(lead, ps.data(), n, 1, rr, (long)rlen);
	if (!rc) rc = run_potrf(lead, 1);
	if (!rc) rc = enqueue_results(lead);
	if (!rc && hipStreamSynchronize(lead->stream) != hipSuccess) { lead->err = "stream synchronisation failed"; rc = GPEMU_ERR_HIP; }
	hipFree(rr);
	if (rc) return rc;
	int worst = GPEMU_OK;
	for (int c = 0; c < n; c++) {
		const int inf = (lead->hInfo[c] >= INFO_NONE) ? 0 : lead->hInfo[c];
		if (info) info[c] = inf;
		int st = GPEMU_OK;
		if (inf) st = GPEMU_ERR_NOT_PD;
		else {
			HostLik r = host_likelihood(lead, c);
			if (r.status) st = r.status;
			else {
				st = build_prediction_state(lead, c, ctxs[c], ps[c], r, thetas + (size_t)c * nthetas, nthetas);
				ctxs[c]->fact_in_T = (c == 0);                // (element 0 of the batch sits where a single set-up leaves it)
				if (st == GPEMU_OK && beta_out) for (int a = 0; a < nreg; a++) beta_out[(size_t)c * nreg + a] = r.beta[a];
			}
		}
		if (status) status[c] = st;
		if (st != GPEMU_OK && worst == GPEMU_OK) worst = st;
	}
	if (worst == GPEMU_ERR_NOT_PD) return fail(lead, worst, "covariance matrix is not positive definite");
	if (worst == GPEMU_ERR_REGRESSION) return fail(lead, worst, "H^T C^-1 H is not positive definite");
	return worst;
}

// Everything a process pays once before its first result -- the HIP runtime, the device's code objects (loaded at the first
// launch from each of this library's translation units), the exp and tile tables -- on a throw-away context with a 64-point
// model: one evaluation, one prediction set-up, one prediction.  Meant for a thread of its own while the caller is still
// reading its input (csrc/host: gpemu_host_warm_start).
extern "C" int gpemu_warm_start(int device)
{
	gpemu_ctx *ctx = nullptr;
	int rc = gpemu_ctx_create(&ctx, device);
	if (rc) return rc;
	const int N = 64, d = 2;
	std::vector<double> X((size_t)N * d), y(N);
	for (int i = 0; i < N; i++) {
		X[(size_t)i * d] = (i % 8) / 8.0 + 0.01 * i;
		X[(size_t)i * d + 1] = (i / 8) / 8.0;
		y[i] = std::sin(0.3 * i);
	}
	{
		// (one covariance function is enough: the instantiations for the others sit in the same code objects)
		const double th[4] = {0.0, -3.0, -1.0, -1.0};
		rc = gpemu_set_model(ctx, GPEMU_POWEREXP, 1, N, d, X.data(), y.data());
		double v, s2, m, var;
		int info = 0;
		if (!rc) rc = gpemu_loglik(ctx, th, 4, &v, &s2, nullptr, nullptr, nullptr, &info);
		if (!rc) rc = gpemu_predict_setup(ctx, th, 4, nullptr, &info);
		if (!rc) rc = gpemu_predict_batch(ctx, 1, X.data(), &m, &var);
	}
	gpemu_ctx_destroy(ctx);
	return rc;
}

constexpr int PRED_SPLIT_MAX = 16;

static int ensure_pred_batch(gpemu_ctx *ctx, int mb)
{
	if (ctx->pred_batch >= mb) return GPEMU_OK;
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	if (ctx->dKq) hipFree(ctx->dKq);
	if (ctx->dV) hipFree(ctx->dV);
	ctx->dKq = ctx->dV = nullptr; ctx->pred_batch = 0;
	HIPCHK(ctx, hipMalloc(&ctx->dKq, (size_t)mb * ctx->Np * sizeof(double)));
	// V also holds the split-K partial products of small batches: PRED_SPLIT_MAX slices of up to 128 query rows
	HIPCHK(ctx, hipMalloc(&ctx->dV, (size_t)std::max(mb, 128 * PRED_SPLIT_MAX) * (ctx->Np + ctx->Rp) * sizeof(double)));
	ctx->pred_batch = mb;
	return GPEMU_OK;
}

constexpr int PRED_BATCH_MAX = 16384;

extern "C" int gpemu_predict_batch_dev(gpemu_ctx *ctx, int M, const double *xq_dev, double *mean_dev, double *var_dev)
{
	if (!ctx || M < 1 || !xq_dev || !mean_dev || !var_dev) return GPEMU_ERR_ARG;
	if (!ctx->pred_ready) return fail(ctx, GPEMU_ERR_STATE, "gpemu_predict_setup has not been called");
	HIPCHK(ctx, hipSetDevice(ctx->device));
	const int Np = ctx->Np, Rp = ctx->Rp, N = ctx->N, d = ctx->d;
	const int cap = std::min(PRED_BATCH_MAX, round_up(M, 64));
	int rc = ensure_pred_batch(ctx, cap);
	if (rc) return rc;
	for (int q0 = 0; q0 < M; q0 += cap) {
		const int mb = std::min(cap, M - q0);
		const int mbp = round_up(mb, 64);
		GemmArgs g;
		memset(&g, 0, sizeof g);
		g.C = ctx->dV; g.ldc = Np + Rp;
		g.A = ctx->dKq; g.lda = Np;
		g.B = ctx->dLinvAug; g.ldb = Np;
		g.m = mb; g.n = Np + Rp; g.k0 = 0; g.k1 = Np; g.alpha = 1.0; g.beta = 0;
		g.kend_mode = 1; g.kend_off = 0;
		g.big_tiles = ctx->sched.gemm_big_tiles;
		// a few queries (emulate_point: ONE) give one or two tile rows with K = N each: split K over the chip
		// (0.46 -> 0.1 ms per call at N=8192); the slices are summed in order by the finishing kernel
		int nslice = 1;
		{
			const long tiles = (long)(mbp / 64) * ((Np + Rp + 63) / 64);      // 64x64 tiles of the unsplit product
			if (tiles < 1024) nslice = (int)std::max(1L, std::min((long)std::min(PRED_SPLIT_MAX, Np / 512), 2048 / tiles));
			if ((long)nslice * mbp > 128L * PRED_SPLIT_MAX) nslice = 1;       // capacity of dV for the partial products
		}
		// up to 16 queries with a long contraction (emulate_point: ONE query): the few-queries path -- a one-thread-per-design-
		// point k-vector kernel, the skinny split-K product, an epilogue with the slice sums fused in (three launches; the batch
		// kernels would set up their tables and tiles for 63 padding rows: 16 + 93 + 6 + 20 us of kernels at N = 8192)
		const bool few = mb <= 16 && nslice > 1;
		if (!few) {
			ProfScope ps(ctx, GPEMU_PROF_FILL, 0.0, 8.0 * (double)mbp * Np);
			HIPCHK(ctx, fill_kvectors(ctx, ctx->dKq, xq_dev + (size_t)q0 * d, mb, mbp, ctx->pred_cov));
		} else {
			HIPCHK(ctx, launch_kvec_small(ctx->stream, ctx->dKq, Np, xq_dev + (size_t)q0 * d, mb, ctx->dX, N, Np, d, ctx->pred_cov));
		}
		if (nslice > 1) { g.ksplit = nslice; g.bsC = (long)mbp * (Np + Rp); }
		if (few) {
			// up to 16 queries (emulate_point: one): the skinny kernel, one 16-row query tile, instead of 64-row GEMM
			// tiles (97 vs 115 us at N=8192; from 17 queries on the split-K GEMM is as fast)
			const int klen = (((Np + nslice - 1) / nslice) + 15) & ~15;
			ProfScope ps(ctx, GPEMU_PROF_GEMM, gemm_flops(g), 0.0);
			if (mb == 1 && ctx->sched.gemv_point)
				// ONE query: a matrix-vector stream over whole rows of L^-1 instead of the matrix unit's 16-row reads
				HIPCHK(ctx, launch_gemv_tri(ctx->stream, ctx->dKq, Np, ctx->dLinvAug, Np, ctx->dV, Np + Rp, (long)mbp * (Np + Rp),
				                            1, Np + Rp, Np, Np, nslice, klen));
			else
			HIPCHK(ctx, launch_skinny_nt(ctx->stream, ctx->dKq, Np, ctx->dLinvAug, Np, ctx->dV, Np + Rp, (long)mbp * (Np + Rp),
			                             1, Np + Rp, Np, Np, nslice, klen));
			HIPCHK(ctx, launch_predict_finish_small(ctx->stream, ctx->dV, Np + Rp, (long)mbp * (Np + Rp), nslice, mb, Np, ctx->nreg, d,
			                                        xq_dev + (size_t)q0 * d, ctx->dBetaQ, ctx->kappa, mean_dev + q0, var_dev + q0));
			continue;
		} else if (nslice == 1 && ctx->sched.split_rhs_rows && gemm_uses_big_tiles(g)) {
			// 128x128 tiles: the 64 columns of gamma and W^T behind the Np triangular ones would make a 65th tile column that is
			// half empty at the full contraction length (1.5 % of the sweep's tile time): they go to a 64x64-tile launch of
			// their own, as the right-hand-side rows of the factorisation's updates do.  Same chain per element, same bits.
			GemmArgs a = g;
			a.n = Np;
			HIPCHK(ctx, gemm(ctx, a));
			GemmArgs b = g;
			b.C = ctx->dV + Np; b.B = ctx->dLinvAug + (size_t)Np * Np;
			b.n = Rp; b.kend_mode = 0; b.force_cfg = 2;
			HIPCHK(ctx, gemm(ctx, b));
		} else
		HIPCHK(ctx, gemm(ctx, g));
		HIPCHK(ctx, launch_predict_finish(ctx->stream, ctx->dV, Np + Rp, mb, Np, ctx->nreg, ctx->order, d,
		                                  xq_dev + (size_t)q0 * d, ctx->dBetaQ, ctx->kappa, mean_dev + q0, var_dev + q0,
		                                  nslice, (long)mbp * (Np + Rp)));
	}
	return GPEMU_OK;
}

// host-buffer entry, asynchronous form: the queries are staged through pinned memory, the batch runs on the context's
// stream and the results come back into pinned memory; nothing blocks until gpemu_predict_batch_collect.  Several
// contexts (the PCA components of a multi-output emulator) can so work on one query at the same time.
extern "C" int gpemu_predict_batch_enqueue(gpemu_ctx *ctx, int M, const double *xq)
{
	if (!ctx || M < 1 || !xq) return GPEMU_ERR_ARG;
	if (!ctx->pred_ready) return fail(ctx, GPEMU_ERR_STATE, "gpemu_predict_setup has not been called");
	if (ctx->pred_pending) return fail(ctx, GPEMU_ERR_STATE, "a prediction batch is already enqueued: collect it first");
	HIPCHK(ctx, hipSetDevice(ctx->device));
	const int d = ctx->d;
	if (ctx->stage_cap < M) {
		HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
		if (ctx->dXq) hipFree(ctx->dXq);
		if (ctx->dMean) hipFree(ctx->dMean);
		if (ctx->hStage) hipHostFree(ctx->hStage);
		ctx->dXq = ctx->dMean = ctx->dVar = nullptr; ctx->hStage = nullptr; ctx->stage_cap = 0;
		const int cap = std::max(M, 64);
		HIPCHK(ctx, hipMalloc(&ctx->dXq, (size_t)cap * d * sizeof(double)));
		HIPCHK(ctx, hipMalloc(&ctx->dMean, (size_t)2 * cap * sizeof(double)));    // means, then variances (one allocation: a small
		ctx->dVar = ctx->dMean + cap;                                             // batch comes back in ONE copy)
		HIPCHK(ctx, hipHostMalloc((void **)&ctx->hStage, (size_t)cap * (d + 2) * sizeof(double)));
		ctx->stage_cap = cap;
	}
	double *hx = ctx->hStage, *hm = ctx->hStage + (size_t)ctx->stage_cap * d, *hv = hm + ctx->stage_cap;
	memcpy(hx, xq, (size_t)M * d * sizeof(double));
	HIPCHK(ctx, hipMemcpyAsync(ctx->dXq, hx, (size_t)M * d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	int rc = gpemu_predict_batch_dev(ctx, M, ctx->dXq, ctx->dMean, ctx->dVar);
	if (rc) return rc;
	if (ctx->stage_cap <= 1024) {
		// (hm | hv on the host and dMean | dVar on the device have the same layout, stage_cap entries apart)
		HIPCHK(ctx, hipMemcpyAsync(hm, ctx->dMean, ((size_t)ctx->stage_cap + M) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	} else {
		HIPCHK(ctx, hipMemcpyAsync(hm, ctx->dMean, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
		HIPCHK(ctx, hipMemcpyAsync(hv, ctx->dVar, (size_t)M * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	}
	ctx->pred_pending = M;
	return GPEMU_OK;
}

extern "C" int gpemu_predict_batch_collect(gpemu_ctx *ctx, int M, double *mean, double *var)
{
	if (!ctx || !mean || !var) return GPEMU_ERR_ARG;
	if (!ctx->pred_pending || M != ctx->pred_pending) return fail(ctx, GPEMU_ERR_STATE, "no enqueued prediction batch of this size");
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	const double *hm = ctx->hStage + (size_t)ctx->stage_cap * ctx->d, *hv = hm + ctx->stage_cap;
	memcpy(mean, hm, (size_t)M * sizeof(double));
	memcpy(var, hv, (size_t)M * sizeof(double));
	ctx->pred_pending = 0;
	return GPEMU_OK;
}

extern "C" int gpemu_predict_batch(gpemu_ctx *ctx, int M, const double *xq, double *mean, double *var)
{
	if (!ctx || M < 1 || !xq || !mean || !var) return GPEMU_ERR_ARG;
	int rc = gpemu_predict_batch_enqueue(ctx, M, xq);
	if (rc) return rc;
	return gpemu_predict_batch_collect(ctx, M, mean, var);
}

// ---------------------------------------------------------------------------
// explicit inverse: S = Aug Aug^T with Aug = [Z^T ; U]  ->  S[Rp+i][Rp+j] = (C^-1)_ij,
// S[Rp+i][a] = (C^-1 [y|H])_ia   (lower triangle only)
// ---------------------------------------------------------------------------
// corners of the nbc batch elements b0 .. b0+nbc-1 (one batched product)
static int build_corner(gpemu_ctx *ctx, int b0 = 0, int nbc = 1)
{
	const int Np = ctx->Np, Rp = ctx->Rp;
	const size_t dim = (size_t)Np + Rp;
	if (ctx->S_dim < dim || ctx->S_cap < nbc) {
		if (ctx->dS) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); hipFree(ctx->dS); ctx->dS = nullptr; }
		ctx->S_dim = 0; ctx->S_cap = 0;
		HIPCHK(ctx, hipMalloc(&ctx->dS, (size_t)nbc * dim * dim * sizeof(double)));
		ctx->S_dim = dim; ctx->S_cap = nbc;
	}
	GemmArgs g;
	memset(&g, 0, sizeof g);
	g.C = ctx->dS; g.ldc = (long)ctx->S_dim;
	g.A = ctx->dT + (size_t)b0 * ctx->T_stride + (size_t)Np * Np; g.lda = Np;
	g.B = g.A; g.ldb = Np;
	g.m = (int)dim; g.n = (int)dim; g.k0 = 0; g.k1 = Np; g.alpha = 1.0; g.beta = 0;
	g.tri = 1; g.diag_off = 0;
	g.kstart_mode = 1; g.kstart_off = Rp;
	g.nbatch = nbc; g.bsC = (long)(ctx->S_dim * ctx->S_dim); g.bsA = g.bsB = (long)ctx->T_stride;
	HIPCHK(ctx, gemm(ctx, g));
	return GPEMU_OK;
}

extern "C" int gpemu_get_cinverse(gpemu_ctx *ctx, double *cinv_out)
{
	if (!ctx || !cinv_out) return GPEMU_ERR_ARG;
	if (!ctx->pred_ready) return fail(ctx, GPEMU_ERR_STATE, "gpemu_predict_setup has not been called");
	HIPCHK(ctx, hipSetDevice(ctx->device));
	if (!ctx->cinv_ready && !ctx->fact_in_T) {
		// set up by gpemu_predict_setup_batch as a later component: its factorisation ran in the first context's workspace
		// (round 5: reading this context's own, never allocated workspace here was a GPU memory fault).  The explicit inverse is
		// the slow path anyway (N x N doubles to the host): factor this component alone, same thetas, same bits.
		const std::vector<double> th = ctx->last_thetas;
		int info = 0;
		int rc = gpemu_predict_setup(ctx, th.data(), (int)th.size(), nullptr, &info);
		if (rc) return rc;
	}
	if (!ctx->cinv_ready) {
		int rc = build_corner(ctx);
		if (rc) return rc;
		ctx->cinv_ready = true;
	}
	const int N = ctx->N, Rp = ctx->Rp;
	const size_t dim = ctx->S_dim;
	HIPCHK(ctx, hipMemcpy2DAsync(cinv_out, (size_t)N * sizeof(double), ctx->dS + (size_t)Rp * dim + Rp,
	                             dim * sizeof(double), (size_t)N * sizeof(double), N, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	// mirror the lower triangle into the upper one, 64 x 64 blocks at a time (a plain column walk over 512 MB at N = 8192
	// misses the cache on every element: 0.3 s of alloc_emulator_struct's 0.4)
	for (int i0 = 0; i0 < N; i0 += 64)
		for (int j0 = i0; j0 < N; j0 += 64)
			for (int i = i0; i < std::min(i0 + 64, N); i++)
				for (int j = std::max(j0, i + 1); j < std::min(j0 + 64, N); j++) cinv_out[(size_t)i * N + j] = cinv_out[(size_t)j * N + i];
	return GPEMU_OK;
}

// ---------------------------------------------------------------------------
// gradient (gradFnMulti, maxmultimin.c:416-550) and value+gradient (evalFnGradMulti, :615-618)
//
// One implementation for one evaluation and for a lock-step batch, in two halves:
//   enqueue: stage nb matrices with their inverse rows, factor them in lock-step (U = L^-T falls out), Gram / log det
//            as for a likelihood batch; then per chunk of corners C^-1 = U U^T (one batched GEMM), alpha = C^-1 y (or
//            C^-1 (y - H beta) with beta solved on the device, exact mode), the tile reductions tr(C^-1 dC_k),
//            alpha^T dC_k alpha, their second-stage sums, and one small copy into the pinned result ring.  No host
//            synchronisation anywhere: the call returns while the device works.
//   collect: waits for THAT batch's event and finishes on the host (nreg x nreg solve, the reference's scalings).
// ---------------------------------------------------------------------------
// corners S kept in flight at a time by the gradient of a batch: as many as fit in about 10 GB
static int grad_chunk_size(const gpemu_ctx *ctx, int nb)
{
	const double dim = (double)ctx->Np + ctx->Rp;
	const int fit = (int)(10.0e9 / (dim * dim * 8.0));
	return std::max(1, std::min(nb, fit));
}

static int grad_check_args(gpemu_ctx *ctx, int nthetas)
{
	if (!ctx->dX) return fail(ctx, GPEMU_ERR_STATE, "model not set");
	const int trainable_matern = (ctx->mode & GPEMU_MODE_EXACT_GRAD) && (ctx->mode & GPEMU_MODE_MATERN_LOG);
	if (ctx->kind != GPEMU_POWEREXP && !trainable_matern)
		return fail(ctx, GPEMU_ERR_ARG,
		            "Matern gradient needs GPEMU_MODE_EXACT_GRAD | GPEMU_MODE_MATERN_LOG: the reference's literal Matern "
		            "derivative matrices (emulator.c:401-433, 497-532) carry an accumulator across elements and its training "
		            "path zeroes the raw amplitude (maxmultimin.c:311,495) -- there is no literal Matern gradient to reproduce");
	if (nthetas < nthetas_for(ctx)) return fail(ctx, GPEMU_ERR_ARG, "nthetas too small");
	if (nthetas > GPEMU_MAX_PARAMS + 2) return fail(ctx, GPEMU_ERR_ARG, "nthetas too large");
	return GPEMU_OK;
}

// gradient reductions of the batch elements b0 .. b0+nbc-1 of the factorisation in the workspace, into dGradSum
static int grad_enqueue_chunk(gpemu_ctx *ctx, int b0, int nbc, const double *th_all, int nthetas)
{
	int rc = build_corner(ctx, b0, nbc);
	if (rc) return rc;
	const int N = ctx->N, d = ctx->d, Rp = ctx->Rp;
	const size_t dim = ctx->S_dim, sstride = dim * dim;
	const bool exact = (ctx->mode & GPEMU_MODE_EXACT_GRAD) != 0;
	const int nlen = ctx->kind == GPEMU_POWEREXP ? d : 1;           // length-scale directions
	const size_t gslot = (size_t)ctx->Np + 2 * GPEMU_MAX_PARAMS;    // per corner: alpha scratch | length thetas | beta
	if (ctx->alpha_cap < nbc) {
		if (ctx->dAlpha) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); hipFree(ctx->dAlpha); ctx->dAlpha = nullptr; }
		ctx->alpha_cap = 0;
		HIPCHK(ctx, hipMalloc(&ctx->dAlpha, (size_t)nbc * gslot * sizeof(double)));
		ctx->alpha_cap = nbc;
	}
	const int nt = (N + 63) / 64, ntiles = nt * (nt + 1) / 2;
	const int np = 2 * d + 2;
	const size_t need = (size_t)ntiles * np;
	if (ctx->gradpart_len < need * nbc) {
		if (ctx->dGradPart) { HIPCHK(ctx, hipStreamSynchronize(ctx->stream)); hipFree(ctx->dGradPart); }
		ctx->dGradPart = nullptr; ctx->gradpart_len = 0;
		HIPCHK(ctx, hipMalloc(&ctx->dGradPart, need * nbc * sizeof(double)));
		ctx->gradpart_len = need * nbc;
	}
	// the length thetas of the chunk: from the pinned entry that belongs to this batch's hyper-parameter upload (reused
	// only after param_ev of the entry, re-recorded below behind this copy)
	double *gph = ctx->hGph + ((size_t)ctx->param_slot * GPEMU_MAX_BATCH + b0) * GPEMU_MAX_PARAMS;
	for (int i = 0; i < nbc; i++)
		for (int k = 0; k < GPEMU_MAX_PARAMS; k++)
			gph[(size_t)i * GPEMU_MAX_PARAMS + k] = k < nlen ? th_all[(size_t)(b0 + i) * nthetas + 2 + k] : 0.0;
	HIPCHK(ctx, hipMemcpy2DAsync(ctx->dAlpha + ctx->Np, gslot * sizeof(double), gph, GPEMU_MAX_PARAMS * sizeof(double),
	                             GPEMU_MAX_PARAMS * sizeof(double), nbc, hipMemcpyHostToDevice, ctx->stream));
	HIPCHK(ctx, hipEventRecord(ctx->param_ev[ctx->param_slot], ctx->stream));
	if (exact)
		HIPCHK(ctx, launch_beta_solve(ctx->stream, ctx->dRes + (size_t)b0 * ctx->res_len, (long)ctx->res_len, Rp, ctx->nreg, nbc,
		                              ctx->dAlpha, (long)gslot, ctx->Np));
	int nparts = 0;
	// literal form: exp(-1/2 e^{-2 theta_k} D_k^2) of every pair of design points -- when the largest such argument of the
	// chunk stays small (|D_k| <= the coordinate's range) the kernel's exp needs no lower clamp
	bool noclamp = !exact && (int)ctx->xhalf.size() == d;
	for (int i = 0; noclamp && i < nbc; i++)
		for (int k = 0; k < nlen; k++) {
			const double range = 2.0 * ctx->xhalf[k];
			if (!(0.5 * exp(-2.0 * th_all[(size_t)(b0 + i) * nthetas + 2 + k]) * range * range < 600.0)) noclamp = false;
		}
	HIPCHK(ctx, launch_grad_partials(ctx->stream, ctx->dS, (long)dim, Rp, (long)sstride, nbc, ctx->dX, N, d, ctx->dAlpha, ctx->Np,
	                                 (long)gslot, ctx->dGradPart, (long)need, &nparts, exact ? ctx->kind : 0, ctx->nreg,
	                                 ctx->dParams + b0, noclamp, ctx->sched.grad_gram ? ctx->dXg : nullptr));
	HIPCHK(ctx, launch_grad_reduce(ctx->stream, ctx->dGradPart, (long)need, nparts, np, nbc,
	                               ctx->dGradSum + (size_t)b0 * gpemu_ctx::GRAD_NP_MAX, (long)gpemu_ctx::GRAD_NP_MAX));
	return GPEMU_OK;
}

extern "C" int gpemu_loglik_grad_batch_enqueue(gpemu_ctx *ctx, int nb, const double *thetas, int nthetas)
{
	if (!ctx || !thetas) return GPEMU_ERR_ARG;
	int rc = grad_check_args(ctx, nthetas);
	if (rc) return rc;
	if (nb < 1 || nb > GPEMU_MAX_BATCH) return fail(ctx, GPEMU_ERR_ARG, "batch size must be 1..GPEMU_MAX_BATCH");
	std::vector<double> th((size_t)nb * nthetas);
	std::vector<CovParams> ps((size_t)nb);
	for (int b = 0; b < nb; b++) {
		for (int i = 0; i < nthetas; i++) th[(size_t)b * nthetas + i] = thetas[(size_t)b * nthetas + i];
		th[(size_t)b * nthetas] = 0.0;                // maxmultimin.c:441
		rc = make_cov_params(ctx, &th[(size_t)b * nthetas], nthetas, &ps[b]);
		if (rc) return rc;
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	ctx->pred_ready = false; ctx->cinv_ready = false;
	rc = stage_matrices(ctx, ps.data(), nb, 1);
	if (rc) return rc;
	rc = run_potrf(ctx, 1);
	if (rc) return rc;
	rc = enqueue_results(ctx);                       // Gram, log det, info words -> the next slot of the pinned ring
	if (rc) return rc;
	const int slot = (int)((ctx->res_seq - 1) % gpemu_ctx::RES_RING);
	// the ring entry becomes a collectable value+gradient batch only once EVERYTHING of it is on the stream: until then it
	// is marked empty (res_nb = 0), so that after a failure below a later collect of this slot is refused
	// (GPEMU_ERR_STATE) instead of returning a gradient built from whatever the pinned ring held
	ctx->res_nb[slot] = 0;
	ctx->res_kind[slot] = 1;
	ctx->res_th[slot] = th;
	ctx->res_nthetas[slot] = nthetas;
	ctx->res_mode[slot] = ctx->mode;
	const int chunk = grad_chunk_size(ctx, nb);
	for (int b0 = 0; b0 < nb; b0 += chunk) {
		rc = grad_enqueue_chunk(ctx, b0, std::min(chunk, nb - b0), th.data(), nthetas);
		if (rc) return rc;
	}
	HIPCHK(ctx, hipMemcpyAsync(ctx->hGradRing + (size_t)slot * ctx->batch_cap * gpemu_ctx::GRAD_NP_MAX, ctx->dGradSum,
	                           (size_t)nb * gpemu_ctx::GRAD_NP_MAX * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(ctx, hipEventRecord(ctx->res_ev[slot], ctx->stream));     // (re-recorded: now behind the gradient sums as well)
	ctx->res_nb[slot] = nb;
	return GPEMU_OK;
}

// host half of a value+gradient batch whose results sit in ring slot `slot` (its event has been waited for)
static int grad_collect_slot(gpemu_ctx *ctx, int slot, int nb, double *neg_loglik, double *sigma2, double *beta, double *grad,
                             int *info, int *status)
{
	const int nthetas = ctx->res_nthetas[slot], ng = nthetas - 1, d = ctx->d;
	const bool exact = (ctx->res_mode[slot] & GPEMU_MODE_EXACT_GRAD) != 0;       // as it was when the batch was enqueued
	const int nlen = ctx->kind == GPEMU_POWEREXP ? d : 1;
	const double *th_all = ctx->res_th[slot].data();
	double *saveR = ctx->hRes;
	int *saveI = ctx->hInfo;
	ctx->hRes = ctx->hResRing + (size_t)slot * ctx->batch_cap * ctx->res_len;
	ctx->hInfo = ctx->hInfoRing + (size_t)slot * ctx->batch_cap;
	for (int b = 0; b < nb; b++) {
		const int inf = (ctx->hInfo[b] >= INFO_NONE) ? 0 : ctx->hInfo[b];
		int st = GPEMU_OK;
		if (info) info[b] = inf;
		if (grad) for (int i = 0; i < ng; i++) grad[(size_t)b * ng + i] = NAN;
		if (beta) for (int a = 0; a < ctx->nreg; a++) beta[(size_t)b * ctx->nreg + a] = NAN;
		if (neg_loglik) neg_loglik[b] = NAN;
		if (sigma2) sigma2[b] = NAN;
		if (inf) {
			st = fail(ctx, GPEMU_ERR_NOT_PD, "covariance matrix is not positive definite");
		} else {
			const HostLik r = host_likelihood(ctx, b);
			if (r.status) {
				st = fail(ctx, r.status, "H^T C^-1 H is not positive definite");
			} else {
				const double log_2_pi = 1.83788;
				if (neg_loglik) neg_loglik[b] = -1 * (-(1.0 / 2.0) * r.logdet - (ctx->N / 2.0) * log_2_pi + r.quad * (-1.0 / 2.0));
				if (sigma2) sigma2[b] = r.sigma2;
				if (beta) for (int a = 0; a < ctx->nreg; a++) beta[(size_t)b * ctx->nreg + a] = r.beta[a];
				if (grad) {
					const double *sums = ctx->hGradRing + ((size_t)slot * ctx->batch_cap + b) * gpemu_ctx::GRAD_NP_MAX;
					double *g = grad + (size_t)b * ng;
					if (exact) {
						// d(-logL)/dtheta = 1/2 sum_ab (A_ab - alpha_a alpha_b) dC_ab: slot nlen = nugget direction, slots < nlen the lengths
						g[0] = 0.5 * sums[nlen];
						for (int k = 0; k < nlen; k++) g[k + 1] = 0.5 * sums[k];
					} else {
						const double aa = sums[2 * d + 1];
						const double amp = exp(log(r.sigma2));                          // maxmultimin.c:503,514
						const double nug = exp(th_all[(size_t)b * nthetas + 1]);         // :515
						// G(dC) = -1/2 tr(A dC) + 1/2 alpha^T dC alpha ;  grad = -G   (:527,535; getGradientCn :571-608)
						g[0] = -1.0 * (-0.5 * nug * sums[2 * d] + 0.5 * nug * aa);
						for (int k = 0; k < d; k++) g[k + 1] = -1.0 * (amp * (-0.5 * sums[2 * k] + 0.5 * sums[2 * k + 1]));
					}
				}
			}
		}
		if (status) status[b] = st;
	}
	ctx->hRes = saveR; ctx->hInfo = saveI;
	return GPEMU_OK;
}

// results of the value+gradient batch enqueued `back` batches (of either kind) before the newest one; waits for it only
extern "C" int gpemu_loglik_grad_batch_collect_back(gpemu_ctx *ctx, int back, int nb, double *neg_loglik, double *sigma2,
                                                    double *beta, double *grad, int *info, int *status)
{
	if (!ctx) return GPEMU_ERR_ARG;
	if (back < 0 || back >= gpemu_ctx::RES_RING || (unsigned long long)back >= ctx->res_seq)
		return fail(ctx, GPEMU_ERR_STATE, "no such batch in the result ring");
	const int slot = (int)((ctx->res_seq - 1 - (unsigned long long)back) % gpemu_ctx::RES_RING);
	if (nb < 1 || nb != ctx->res_nb[slot]) return fail(ctx, GPEMU_ERR_STATE, "batch size differs from the enqueued batch");
	if (ctx->res_kind[slot] != 1) return fail(ctx, GPEMU_ERR_STATE, "that batch is a likelihood batch: use gpemu_loglik_batch_collect_back");
	HIPCHK(ctx, hipEventSynchronize(ctx->res_ev[slot]));
	return grad_collect_slot(ctx, slot, nb, neg_loglik, sigma2, beta, grad, info, status);
}

extern "C" int gpemu_loglik_grad_batch_collect(gpemu_ctx *ctx, int nb, double *neg_loglik, double *sigma2, double *beta,
                                               double *grad, int *info, int *status)
{
	return gpemu_loglik_grad_batch_collect_back(ctx, 0, nb, neg_loglik, sigma2, beta, grad, info, status);
}

// evalFnGradMulti for a list of thetas (the line-search points of independent restarts): enqueue + collect
extern "C" int gpemu_loglik_grad_batch(gpemu_ctx *ctx, int nb, const double *thetas, int nthetas, double *neg_loglik,
                                       double *sigma2, double *beta, double *grad, int *info, int *status)
{
	if (!ctx || !grad || !thetas) return GPEMU_ERR_ARG;
	int rc = gpemu_loglik_grad_batch_enqueue(ctx, nb, thetas, nthetas);
	if (rc) return rc;
	return gpemu_loglik_grad_batch_collect(ctx, nb, neg_loglik, sigma2, beta, grad, info, status);
}

// evalFnGradMulti (maxmultimin.c:615-618) with ONE factorisation shared by value and gradient: a batch of one
extern "C" int gpemu_loglik_grad(gpemu_ctx *ctx, const double *thetas, int nthetas, double *neg_loglik, double *sigma2,
                                 double *beta, double *grad, int *info)
{
	if (!ctx || !grad) return GPEMU_ERR_ARG;
	int st = GPEMU_OK;
	int rc = gpemu_loglik_grad_batch(ctx, 1, thetas, nthetas, neg_loglik, sigma2, beta, grad, info, &st);
	return rc ? rc : st;
}

extern "C" int gpemu_grad(gpemu_ctx *ctx, const double *thetas, int nthetas, double *grad, int *info)
{
	return gpemu_loglik_grad(ctx, thetas, nthetas, nullptr, nullptr, nullptr, grad, info);
}

// ---------------------------------------------------------------------------
// profiling
// ---------------------------------------------------------------------------
extern "C" int gpemu_prof_begin(gpemu_ctx *ctx, int cls)
{
	if (!ctx) return GPEMU_ERR_ARG;
	for (auto e : ctx->prof.ev) hipEventDestroy(e);
	ctx->prof.ev.clear();
	ctx->prof.cls = cls; ctx->prof.flops = ctx->prof.bytes = 0; ctx->prof.n = 0;
	ctx->prof.tag.clear();
	return GPEMU_OK;
}

extern "C" int gpemu_prof_end(gpemu_ctx *ctx, int *nlaunches, double *total_ms, double *flops, double *bytes)
{
	if (!ctx) return GPEMU_ERR_ARG;
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	double ms = 0.0;
	const bool dump = getenv("GPEMU_PROF_DUMP") != nullptr;
	for (size_t i = 0; i + 1 < ctx->prof.ev.size(); i += 2) {
		float t = 0.f;
		hipEventElapsedTime(&t, ctx->prof.ev[i], ctx->prof.ev[i + 1]);
		ms += t;
		if (dump && i / 2 < ctx->prof.tag.size()) fprintf(stderr, "[gpemu prof] %s ms=%.4f\n", ctx->prof.tag[i / 2].c_str(), t);
	}
	ctx->prof.tag.clear();
	if (nlaunches) *nlaunches = ctx->prof.n;
	if (total_ms) *total_ms = ms;
	if (flops) *flops = ctx->prof.flops;
	if (bytes) *bytes = ctx->prof.bytes;
	for (auto e : ctx->prof.ev) hipEventDestroy(e);
	ctx->prof.ev.clear();
	ctx->prof.cls = GPEMU_PROF_NONE;
	return GPEMU_OK;
}

// GPEMU_TRACE=1: write "tag start_ns end_ns" per kernel launch of the last factorisation (device wall clock,
// 100 MHz ticks converted to ns; the same clock for every context of a GPU) to a text file
extern "C" int gpemu_trace_dump(gpemu_ctx *ctx, const char *path)
{
	if (!ctx || !path) return GPEMU_ERR_ARG;
	if (!ctx->dTrace) { ctx->err = "tracing is off (set GPEMU_TRACE=1 before gpemu_ctx_create)"; return GPEMU_ERR_STATE; }
	HIPCHK(ctx, hipSetDevice(ctx->device));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	std::vector<unsigned long long> h(8 * (size_t)ctx->trace_next);
	if (!h.empty()) {
		HIPCHK(ctx, hipMemcpyAsync(h.data(), ctx->dTrace, h.size() * 8, hipMemcpyDeviceToHost, ctx->stream));
		HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	}
	FILE *f = fopen(path, "w");
	if (!f) { ctx->err = "cannot open trace file"; return GPEMU_ERR_ARG; }
	for (int i = 0; i < ctx->trace_next; i++) {
		const unsigned long long *q = &h[8 * (size_t)i];
		if (q[3] == 0) continue;
		fprintf(f, "%s | %llu %llu %llu %llu %llu %llu %llu %llu\n", ctx->trace_tag[i].c_str(), ~q[0] * 10ull, q[1] * 10ull,
		        q[2] * 10ull, q[3], q[4], q[5], q[6], q[7]);
	}
	fclose(f);
	return GPEMU_OK;
}

// ---------------------------------------------------------------------------
// low-level compatibility entries: the reference's libRbind-era interface passes N x N matrices through host
// memory (emulate-fns.c:275-299, regression.c:120-176, emulator.c:672-785).  The O(N^2)/O(N^3) work still runs here.
// ---------------------------------------------------------------------------
// C -> C^-1 in place (both triangles), log det C = 2 sum log L_ii; *info = 1-based index of the first pivot <= 0
extern "C" int gpemu_chol_inverse(gpemu_ctx *ctx, int n, double *a, int lda, double *logdet, int *info)
{
	if (!ctx || n < 1 || !a || lda < n) return GPEMU_ERR_ARG;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	gpemu_ctx tmp;                       // scratch state on the caller's stream: sizes of this matrix, no model
	tmp.device = ctx->device; tmp.stream = ctx->stream; tmp.use_graph = false; tmp.sched = ctx->sched;
	tmp.Np = round_up(n, LEAF); tmp.Rp = 64; tmp.N = n; tmp.nrhs = 0; tmp.nb = 1;
	const int Np = tmp.Np, Rp = tmp.Rp;
	const size_t rows = (size_t)2 * Np + Rp, dim = (size_t)Np + Rp;
	std::vector<double> h((size_t)Np * Np, 0.0), diag((size_t)n);
	for (int i = 0; i < Np; i++)
		for (int j = 0; j <= i; j++)
			h[(size_t)i * Np + j] = (i < n) ? a[(size_t)i * lda + j] : (i == j ? 1.0 : 0.0);
	int big = INFO_NONE, inf = 0;
	hipError_t e = hipMalloc(&tmp.dT, rows * Np * sizeof(double));
	if (e == hipSuccess) e = hipMalloc(&tmp.dInfo, sizeof(int));
	if (e == hipSuccess) e = hipMalloc(&tmp.dS, dim * dim * sizeof(double));
	if (e == hipSuccess) { tmp.S_dim = dim; tmp.S_cap = 1; }
	if (e == hipSuccess) e = hipMemcpyAsync(tmp.dT, h.data(), h.size() * 8, hipMemcpyHostToDevice, tmp.stream);
	if (e == hipSuccess) e = hipMemsetAsync(tmp.dT + (size_t)Np * Np, 0, (size_t)Rp * Np * 8, tmp.stream);
	if (e == hipSuccess) e = launch_set_identity_rows(tmp.stream, tmp.dT + (size_t)(Np + Rp) * Np, Np, Np);
	if (e == hipSuccess) e = hipMemcpyAsync(tmp.dInfo, &big, sizeof(int), hipMemcpyHostToDevice, tmp.stream);
	if (e == hipSuccess) e = potrf_rec(&tmp, 0, Np, 1);
	if (e == hipSuccess) e = hipMemcpyAsync(&inf, tmp.dInfo, sizeof(int), hipMemcpyDeviceToHost, tmp.stream);
	if (e == hipSuccess)
		e = hipMemcpy2DAsync(diag.data(), sizeof(double), tmp.dT, ((size_t)Np + 1) * sizeof(double), sizeof(double), n,
		                     hipMemcpyDeviceToHost, tmp.stream);
	if (e == hipSuccess) e = hipStreamSynchronize(tmp.stream);
	const int bad = (inf >= INFO_NONE) ? 0 : inf;
	if (e == hipSuccess && !bad) {
		int rc = build_corner(&tmp);         // C^-1 = U U^T, lower triangle at (Rp, Rp) of the corner matrix
		if (rc) e = hipErrorUnknown;
		if (e == hipSuccess)
			e = hipMemcpy2DAsync(a, (size_t)lda * sizeof(double), tmp.dS + (size_t)Rp * dim + Rp, dim * sizeof(double),
			                     (size_t)n * sizeof(double), n, hipMemcpyDeviceToHost, tmp.stream);
		if (e == hipSuccess) e = hipStreamSynchronize(tmp.stream);
	}
	if (tmp.dT) hipFree(tmp.dT);
	if (tmp.dInfo) hipFree(tmp.dInfo);
	if (tmp.dS) hipFree(tmp.dS);
	tmp.dT = nullptr; tmp.dInfo = nullptr; tmp.dS = nullptr; tmp.stream = nullptr;
	HIPCHK(ctx, e);
	if (info) *info = bad;
	if (bad) return fail(ctx, GPEMU_ERR_NOT_PD, "matrix is not positive definite");
	double ld = 0.0;
	for (int i = 0; i < n; i++) ld += log(diag[i]);
	if (logdet) *logdet = 2.0 * ld;
	for (int i = 0; i < n; i++)
		for (int j = i + 1; j < n; j++) a[(size_t)i * lda + j] = a[(size_t)j * lda + i];
	return GPEMU_OK;
}

// 64-bit checksum of every element of a host matrix (four independent multiply-xor lanes, one pass at memory speed):
// decides whether the device copy of a caller's C^-1 is still current.  The callers (libRbind-style loops,
// libEmu/regression.c:120-176 and emulator.c:672-785 callers) reuse ONE cinverse buffer and rewrite it in place, so
// pointer and sizes alone say nothing; a sampled fingerprint could miss an interior change.
static uint64_t matrix_checksum(const double *a, int n, int lda)
{
	uint64_t h[4] = {0x9E3779B97F4A7C15ull, 0xBF58476D1CE4E5B9ull, 0x94D049BB133111EBull, 0xD6E8FEB86659FD93ull};
	const uint64_t K = 0xFF51AFD7ED558CCDull;
	for (int i = 0; i < n; i++) {
		const double *row = a + (size_t)i * lda;
		int j = 0;
		for (; j + 4 <= n; j += 4) {
			uint64_t w[4];
			memcpy(w, row + j, sizeof w);
			h[0] = (h[0] ^ w[0]) * K; h[1] = (h[1] ^ w[1]) * K; h[2] = (h[2] ^ w[2]) * K; h[3] = (h[3] ^ w[3]) * K;
		}
		for (; j < n; j++) {
			uint64_t w;
			memcpy(&w, row + j, sizeof w);
			h[j & 3] = (h[j & 3] ^ w) * K;
		}
		h[0] ^= h[0] >> 29;                                 // row boundary: position-dependent
	}
	uint64_t r = h[0];
	for (int k = 1; k < 4; k++) r = (r ^ (h[k] + (r << 6) + (r >> 2))) * K;
	return r ^ (r >> 32);
}

// forget the cached device copy of the host matrix: the next gpemu_symm_apply uploads without comparing checksums
// (not needed for correctness; saves a caller that knows it has rewritten the buffer one pass over it)
extern "C" int gpemu_symm_invalidate(gpemu_ctx *ctx)
{
	if (!ctx) return GPEMU_ERR_ARG;
	ctx->sym_key = nullptr;
	ctx->sym_pinned = false;
	return GPEMU_OK;
}

// pinned = 1: the caller promises not to modify the matrix it passes to gpemu_symm_apply / gpemu_trace_product until it
// unpins (or invalidates): calls with the same (pointer, n, lda) then skip the per-call checksum pass (one read of
// N x N doubles from host memory -- 0.1 s at N = 8192, far more than the device product it guards).  The reference's
// per-point loops (makeEmulatedMean / makeEmulatedVariance over one cinverse, emulator.c:672-785) are such callers.
extern "C" int gpemu_symm_pin(gpemu_ctx *ctx, int pinned)
{
	if (!ctx) return GPEMU_ERR_ARG;
	ctx->sym_pinned = pinned != 0;
	return GPEMU_OK;
}

// out[v][i] = sum_j A[i][j] V[v][j] for nvec vectors stored as rows; A symmetric, host-resident, N x N with row
// stride lda.  A is uploaded when (pointer, sizes) differ from the cached copy, or -- unless pinned -- when the checksum
// of all its elements does (the checksum pass runs only when pointer and sizes match: a new matrix is uploaded at once).
extern "C" int gpemu_symm_apply(gpemu_ctx *ctx, int n, const double *a, int lda, int nvec, const double *v, double *out)
{
	if (!ctx || n < 1 || !a || lda < n || nvec < 1 || !v || !out) return GPEMU_ERR_ARG;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	const int Npad = round_up(n, 64);
	const bool same_key = ctx->dSym && ctx->sym_key == a && ctx->sym_N == n && ctx->sym_lda == lda;
	bool current = same_key && ctx->sym_pinned;
	uint64_t fp = 0;
	if (same_key && !current) {
		fp = matrix_checksum(a, n, lda);
		current = fp == ctx->sym_fp;
	}
	if (!current) {
		HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
		if (ctx->sym_pad != Npad || !ctx->dSym) {
			if (ctx->dSym) hipFree(ctx->dSym);
			ctx->dSym = nullptr;
			HIPCHK(ctx, hipMalloc(&ctx->dSym, (size_t)Npad * Npad * sizeof(double)));
			ctx->sym_pad = Npad;
			ctx->sym_vcap = 0;
		}
		HIPCHK(ctx, hipMemsetAsync(ctx->dSym, 0, (size_t)Npad * Npad * sizeof(double), ctx->stream));
		HIPCHK(ctx, hipMemcpy2DAsync(ctx->dSym, (size_t)Npad * sizeof(double), a, (size_t)lda * sizeof(double),
		                             (size_t)n * sizeof(double), n, hipMemcpyHostToDevice, ctx->stream));
		if (!same_key) fp = matrix_checksum(a, n, lda);      // (a new matrix: its checksum for the calls that follow)
		ctx->sym_key = a; ctx->sym_N = n; ctx->sym_lda = lda; ctx->sym_fp = fp;
	}
	if (ctx->sym_vcap < nvec) {
		HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
		if (ctx->dSymV) hipFree(ctx->dSymV);
		if (ctx->dSymOut) hipFree(ctx->dSymOut);
		ctx->dSymV = ctx->dSymOut = nullptr; ctx->sym_vcap = 0;
		const int cap = round_up(nvec, 64);
		HIPCHK(ctx, hipMalloc(&ctx->dSymV, (size_t)cap * Npad * sizeof(double)));
		HIPCHK(ctx, hipMalloc(&ctx->dSymOut, (size_t)cap * Npad * sizeof(double)));
		ctx->sym_vcap = cap;
	}
	HIPCHK(ctx, hipMemsetAsync(ctx->dSymV, 0, (size_t)nvec * Npad * sizeof(double), ctx->stream));
	HIPCHK(ctx, hipMemcpy2DAsync(ctx->dSymV, (size_t)Npad * sizeof(double), v, (size_t)n * sizeof(double),
	                             (size_t)n * sizeof(double), nvec, hipMemcpyHostToDevice, ctx->stream));
	GemmArgs g;
	memset(&g, 0, sizeof g);
	g.C = ctx->dSymOut; g.ldc = Npad;
	g.A = ctx->dSymV; g.lda = Npad;
	g.B = ctx->dSym; g.ldb = Npad;
	g.m = nvec; g.n = Npad; g.k0 = 0; g.k1 = Npad; g.alpha = 1.0; g.beta = 0;
	HIPCHK(ctx, gemm(ctx, g));
	HIPCHK(ctx, hipMemcpy2DAsync(out, (size_t)n * sizeof(double), ctx->dSymOut, (size_t)Npad * sizeof(double),
	                             (size_t)n * sizeof(double), nvec, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	return GPEMU_OK;
}

// a5 derivative_l_gauss (libEmu/emulator.c:173-209) materialised into host memory: the N x N matrix of the literal
// one-coordinate formula for the design column xcol[n] and the log length scale theta_len
extern "C" int gpemu_derivative_gauss(gpemu_ctx *ctx, int n, const double *xcol, double theta_len, double *out, int ldo)
{
	if (!ctx || n < 1 || !xcol || !out || ldo < n) return GPEMU_ERR_ARG;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	double *dx = nullptr, *dout = nullptr;
	HIPCHK(ctx, hipMalloc(&dx, (size_t)n * sizeof(double)));
	hipError_t e = hipMalloc(&dout, (size_t)n * n * sizeof(double));
	if (e == hipSuccess) e = hipMemcpyAsync(dx, xcol, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) e = launch_deriv_gauss(ctx->stream, dout, n, dx, n, theta_len);
	if (e == hipSuccess)
		e = hipMemcpy2DAsync(out, (size_t)ldo * sizeof(double), dout, (size_t)n * sizeof(double), (size_t)n * sizeof(double), n,
		                     hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	hipFree(dx);
	if (dout) hipFree(dout);
	HIPCHK(ctx, e);
	return GPEMU_OK;
}

// trace(A B) = sum_ij A[i][j] B[j][i] of two host-resident n x n matrices (row strides lda, ldb): getGradientCn's
// trace(C^-1 dC/dtheta) (libEmu/maxmultimin.c:583-588) as one pass over the two matrices instead of an N^3 dgemm.
// A goes through the same cache as gpemu_symm_apply -- literally: the gpemu_symm_apply call below is what makes it
// resident, with that entry's (pointer, size, checksum, pin) rules; B is uploaded on every call.
extern "C" int gpemu_trace_product(gpemu_ctx *ctx, int n, const double *a, int lda, const double *b, int ldb, double *trace)
{
	if (!ctx || n < 1 || !a || !b || lda < n || ldb < n || !trace) return GPEMU_ERR_ARG;
	std::vector<double> one((size_t)n, 0.0), tmp((size_t)n);
	int rc = gpemu_symm_apply(ctx, n, a, lda, 1, one.data(), tmp.data());      // makes sure a is resident in dSym
	if (rc) return rc;
	const int Npad = ctx->sym_pad;
	double *dB = nullptr, *dPart = nullptr;
	HIPCHK(ctx, hipMalloc(&dB, (size_t)Npad * Npad * sizeof(double)));
	hipError_t e = hipMalloc(&dPart, (size_t)n * sizeof(double));
	if (e == hipSuccess)
		e = hipMemcpy2DAsync(dB, (size_t)Npad * sizeof(double), b, (size_t)ldb * sizeof(double), (size_t)n * sizeof(double), n,
		                     hipMemcpyHostToDevice, ctx->stream);
	if (e == hipSuccess) e = launch_trace_product(ctx->stream, ctx->dSym, dB, Npad, n, dPart);
	if (e == hipSuccess) e = hipMemcpyAsync(tmp.data(), dPart, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	hipFree(dB);
	if (dPart) hipFree(dPart);
	HIPCHK(ctx, e);
	double t = 0.0;
	for (int i = 0; i < n; i++) t += tmp[i];
	*trace = t;
	return GPEMU_OK;
}

// ---------------------------------------------------------------------------
// building blocks exported for the parity tests
// ---------------------------------------------------------------------------
extern "C" int gpemu_test_gemm_nt(gpemu_ctx *ctx, int m, int n, int k, double alpha, int beta, const double *a,
                                  const double *b, double *c)
{
	if (!ctx || m < 1 || n < 1 || k < 1 || (k % GEMM_BK) != 0) return GPEMU_ERR_ARG;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	double *da = nullptr, *db = nullptr, *dc = nullptr;
	HIPCHK(ctx, hipMalloc(&da, (size_t)m * k * 8));
	HIPCHK(ctx, hipMalloc(&db, (size_t)n * k * 8));
	HIPCHK(ctx, hipMalloc(&dc, (size_t)m * n * 8));
	HIPCHK(ctx, hipMemcpyAsync(da, a, (size_t)m * k * 8, hipMemcpyHostToDevice, ctx->stream));
	HIPCHK(ctx, hipMemcpyAsync(db, b, (size_t)n * k * 8, hipMemcpyHostToDevice, ctx->stream));
	HIPCHK(ctx, hipMemcpyAsync(dc, c, (size_t)m * n * 8, hipMemcpyHostToDevice, ctx->stream));
	GemmArgs g;
	memset(&g, 0, sizeof g);
	g.C = dc; g.A = da; g.B = db; g.ldc = n; g.lda = k; g.ldb = k; g.m = m; g.n = n; g.k0 = 0; g.k1 = k;
	g.alpha = alpha; g.beta = beta;
	HIPCHK(ctx, gemm(ctx, g));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	HIPCHK(ctx, hipMemcpyAsync(c, dc, (size_t)m * n * 8, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	hipFree(da); hipFree(db); hipFree(dc);
	return GPEMU_OK;
}

// The matrix a lock-step batch is factored from: stages nb matrices exactly as gpemu_loglik_batch does (ONE launch of
// cov_stage_batch_kernel: lower tiles only, every matrix its own hyper-parameters) and copies the N x N block of matrix b
// to the host without factorising.  Tiles strictly above the diagonal are not written by that path (out keeps what the
// workspace held there).
extern "C" int gpemu_test_staged_matrix(gpemu_ctx *ctx, int nb, const double *thetas, int nthetas, int b, double *out)
{
	if (!ctx || !thetas || !out || nb < 1 || nb > GPEMU_MAX_BATCH || b < 0 || b >= nb) return GPEMU_ERR_ARG;
	if (!ctx->dX) return fail(ctx, GPEMU_ERR_STATE, "model not set");
	std::vector<CovParams> ps((size_t)nb);
	for (int i = 0; i < nb; i++) {
		int rc = make_cov_params(ctx, thetas + (size_t)i * nthetas, nthetas, &ps[i]);
		if (rc) return rc;
	}
	HIPCHK(ctx, hipSetDevice(ctx->device));
	ctx->pred_ready = false; ctx->cinv_ready = false;
	int rc = stage_matrices(ctx, ps.data(), nb, 0);
	if (rc) return rc;
	const int N = ctx->N, Np = ctx->Np;
	HIPCHK(ctx, hipMemcpy2DAsync(out, (size_t)N * sizeof(double), ctx->dT + (size_t)b * ctx->T_stride, (size_t)Np * sizeof(double),
	                             (size_t)N * sizeof(double), N, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
	return GPEMU_OK;
}

// the GEMM tile order of a launch with tiles_m x tiles_n tiles (tri: lower triangle only) and super-blocks of side sb:
// pure host logic, no device needed.  Returns the table length; fills out[0 .. min(len, cap)).
extern "C" int gpemu_test_tile_table(int tiles_m, int tiles_n, int tri, int sb, int *out, int cap)
{
	if (tiles_m < 1 || tiles_n < 1 || tiles_m > 32767 || tiles_n > 32767 || sb < 1) return -GPEMU_ERR_ARG;
	const std::vector<int> t = gpemu::build_tile_table(tiles_m, tiles_n, tri, sb);
	if (out)
		for (int i = 0; i < (int)t.size() && i < cap; i++) out[i] = t[i];
	return (int)t.size();
}

/* the row table of the square product with row-start skipping (gpemu::build_row_table): same hook, same conventions */
extern "C" int gpemu_test_row_table(int tiles_m, int bm, int kstart_off, int k0, int k1, int *out, int cap)
{
	if (tiles_m < 1 || tiles_m > 32767 || bm < 16 || k1 <= k0) return -GPEMU_ERR_ARG;
	const std::vector<int> t = gpemu::build_row_table(tiles_m, bm, kstart_off, k0, k1);
	if (out)
		for (int i = 0; i < (int)t.size() && i < cap; i++) out[i] = t[i];
	return (int)t.size();
}

namespace gpemu { hipError_t launch_fill_random(hipStream_t s, double *p, size_t n, unsigned seed); }

// times `reps` launches of one GEMM shape with HIP events on the ctx stream (device-resident random operands)
extern "C" int gpemu_test_gemm_bench(gpemu_ctx *ctx, int m, int n, int k, int ld_in, int cfg, int tri, int beta,
                                     int reps, double *ms_avg, double *flops)
{
	if (!ctx || m < 1 || n < 1 || k < 16 || (k % GEMM_BK) != 0 || reps < 1 || (cfg != 0 && cfg != 2 && cfg != 8)) return GPEMU_ERR_ARG;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	const long ld = std::max(std::max(k, n), ld_in);
	double *da = nullptr, *dc = nullptr;
	const size_t rows = (size_t)std::max(m, n);
	HIPCHK(ctx, hipMalloc(&da, rows * ld * 8));
	HIPCHK(ctx, hipMalloc(&dc, (size_t)m * ld * 8));
	HIPCHK(ctx, launch_fill_random(ctx->stream, da, rows * ld, 1u));
	HIPCHK(ctx, launch_fill_random(ctx->stream, dc, (size_t)m * ld, 2u));
	if (getenv("GPEMU_BENCH_ZERO")) {      // zero operands: the clock-limited ceiling (no data-dependent switching power)
		HIPCHK(ctx, hipMemsetAsync(da, 0, rows * ld * 8, ctx->stream));
		HIPCHK(ctx, hipMemsetAsync(dc, 0, (size_t)m * ld * 8, ctx->stream));
	}
	GemmArgs g;
	memset(&g, 0, sizeof g);
	g.C = dc; g.A = da; g.B = da; g.ldc = ld; g.lda = ld; g.ldb = ld; g.m = m; g.n = n; g.k0 = 0; g.k1 = k;
	g.alpha = -1.0; g.beta = beta; g.tri = tri;
	if (getenv("GPEMU_BENCH_LD0")) { g.lda = 0; g.ldb = 0; }   // every operand row aliases row 0: staging loads always hit L1/L2
	if (ctx->dTrace) {
		ctx->trace_next = 0; ctx->trace_tag.clear();
		HIPCHK(ctx, hipMemsetAsync(ctx->dTrace, 0, (size_t)ctx->trace_cap * 64, ctx->stream));
		g.trace = trace_slot(ctx, "gemm_bench m=%d n=%d k=%d", m, n, k);
	}
	g.force_cfg = cfg;                                      // 2: 64x64 tiles, 8: 128x128 tiles, 0: the automatic choice
	g.big_tiles = ctx->sched.gemm_big_tiles; g.table_sb = ctx->sched.gemm_table;
	g.keep_idle_waves = ctx->sched.idle_waves ? 0 : 1;
	g.stagger_ticks = ctx->sched.stagger_us * 100;
	g.no_neg_modifier = ctx->sched.neg_modifier ? 0 : 1;
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	hipError_t e = launch_gemm(ctx->stream, g);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	hipEventRecord(e0, ctx->stream);
	for (int r = 0; r < reps && e == hipSuccess; r++) e = launch_gemm(ctx->stream, g);
	hipEventRecord(e1, ctx->stream);
	if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
	float ms = 0.f;
	hipEventElapsedTime(&ms, e0, e1);
	hipEventDestroy(e0); hipEventDestroy(e1);
	hipFree(da); hipFree(dc);
	HIPCHK(ctx, e);
	if (ms_avg) *ms_avg = ms / reps;
	if (flops) *flops = gemm_flops(g);
	return GPEMU_OK;
}

extern "C" int gpemu_test_potrf(gpemu_ctx *ctx, int n, double *a, int *info)
{
	if (!ctx || n < 1 || !a) return GPEMU_ERR_ARG;
	HIPCHK(ctx, hipSetDevice(ctx->device));
	// run through a scratch ctx-like state: temporarily adopt sizes
	gpemu_ctx tmp;
	tmp.device = ctx->device; tmp.stream = ctx->stream; tmp.use_graph = false; tmp.sched = ctx->sched;
	tmp.Np = round_up(n, LEAF); tmp.Rp = 64; tmp.N = n; tmp.nrhs = 0;
	const int Np = tmp.Np;
	std::vector<double> h((size_t)(Np + 64) * Np, 0.0);
	for (int i = 0; i < Np; i++)
		for (int j = 0; j <= i; j++)
			h[(size_t)i * Np + j] = (i < n) ? a[(size_t)i * n + j] : (i == j ? 1.0 : 0.0);
	hipError_t e = hipMalloc(&tmp.dT, h.size() * 8);
	if (e == hipSuccess) e = hipMalloc(&tmp.dInfo, sizeof(int));
	int big = INFO_NONE;
	if (e == hipSuccess) e = hipMemcpyAsync(tmp.dT, h.data(), h.size() * 8, hipMemcpyHostToDevice, tmp.stream);
	if (e == hipSuccess) e = hipMemcpyAsync(tmp.dInfo, &big, sizeof(int), hipMemcpyHostToDevice, tmp.stream);
	if (e == hipSuccess) e = potrf_rec(&tmp, 0, Np, 0);
	if (e == hipSuccess) e = hipStreamSynchronize(tmp.stream);
	if (e == hipSuccess) e = hipMemcpyAsync(h.data(), tmp.dT, h.size() * 8, hipMemcpyDeviceToHost, tmp.stream);
	int inf = 0;
	if (e == hipSuccess) e = hipMemcpyAsync(&inf, tmp.dInfo, sizeof(int), hipMemcpyDeviceToHost, tmp.stream);
	if (e == hipSuccess) e = hipStreamSynchronize(tmp.stream);
	if (tmp.dT) hipFree(tmp.dT);
	if (tmp.dInfo) hipFree(tmp.dInfo);
	tmp.dT = nullptr; tmp.dInfo = nullptr; tmp.stream = nullptr;
	HIPCHK(ctx, e);
	if (info) *info = (inf >= INFO_NONE) ? 0 : inf;
	for (int i = 0; i < n; i++)
		for (int j = 0; j < n; j++) a[(size_t)i * n + j] = (j <= i) ? h[(size_t)i * Np + j] : 0.0;
	return GPEMU_OK;
}

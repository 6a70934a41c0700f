// gpemu_internal.hpp -- shared declarations of the gfx950 device library.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>
#include <map>
#include <set>

#include "gpemu.h"

namespace gpemu {

constexpr int LEAF = 64;          // diagonal block / padding granule
constexpr int GEMM_BM = 128;
constexpr int GEMM_BN = 128;
constexpr int GEMM_BK = 16;

inline int round_up(int x, int m) { return ((x + m - 1) / m) * m; }

// Covariance parameters handed to kernels by value.
// pow-exp (emulator.c:101-152): amp = exp(t0), nug = exp(t1), w[k] = sqrt(0.5)/exp(t_{k+2}) (coordinate scale:
//   exponent = -sum ((x_k-y_k) w_k)^2), eps = 1e-10
// Matern  (emulator.c:344-386, 438-480): amp = t0, nug = t1, w[0] = 1/exp(t2) (all coordinates), eps = 1e-16
// cand: upper bound of sum ((x_k-y_k) w_k)^2 over pairs with every |x_k-y_k| < eps (plus rounding slack); only
//   those pairs run the exact per-coordinate "same point" test of the nugget rule.
struct CovParams {
	int kind;
	int d;
	double amp;
	double nug;
	double eps;
	double cand;
	int gram;            // 1: the square training fill may use the MFMA Gram form of the squared distances (kernels_cov.hip)
	int pad_;
	double cand_g;       // Gram form: squared scaled distances at or below this are recomputed from differences
	double cand_w;       // the exact gradient's Gram-form weights (any length scale): likewise, with that form's own error bound
	double w[GPEMU_MAX_PARAMS];
};

// C[m x n] = beta*C + alpha * A[m x K] * B[n x K]^T over k in [k0,k1)
struct GemmArgs {
	double *C;
	const double *A;
	const double *B;
	long ldc, lda, ldb;
	int m, n;
	int k0, k1;          // multiples of GEMM_BK
	double alpha;
	int beta;            // 0 or 1
	int tri;             // 1: skip tiles with tn*BN > tm*BM + BM-1 + diag_off
	int diag_off;
	int kstart_mode;     // 1: k starts at max(k0, floor_BK(tm*BM - kstart_off))   (upper-triangular A rows)
	int kstart_off;
	int kend_mode;       // 1: k ends at min(k1, ceil_BK(tn*BN + BN - kend_off))   (lower-triangular B rows)
	int kend_off;
	long bsC, bsA, bsB;  // element strides between the matrices of a batch (grid.y = nbatch)
	int nbatch;          // 0/1: single problem
	int ksplit;          // > 1: grid.y = k-slices of ONE problem; slice s takes k in [k0 + s*klen, k0 + (s+1)*klen) and
	                     // writes its partial product to C + s*bsC (beta = 0); the consumer sums the slices in order
	int order_mode;      // 2: dense enumeration of the lower-triangular tiles, 3: tile table (set by launch_gemm)
	const int *tile_table; // order_mode 3: entry blockIdx.x = (tm << 16) | tn, or -1 for none
	unsigned long long *trace;   // optional {first start, last end} device timestamps of this launch (GPEMU_TRACE)
	int fa;              // factor-ahead: the workgroup of tile (0,0) factors its updated 64x64 diagonal block (64x64 tiles only)
	int fa_c0;           // global column of that block (for the 1-based index of a failed pivot)
	int *fa_info;        // info words (one per matrix of the batch)
	// schedule switches of the calling context (Sched below), carried per call: nothing process-wide
	int big_tiles;       // 128x128 tiles once a lock-step launch has this many of them (0: 1024)
	int table_sb;        // XCD-blocked tile table for launches of >= 512 tiles: side of the super-blocks (0: off)
	int force_cfg;       // test/bench hook: 2 = 64x64 tiles, 8 = 128x128 tiles, 0 = automatic
	int keep_idle_waves; // 1: waves above the diagonal of a triangular update's diagonal tiles compute their (unread) output anyway (A/B switch)
	int no_neg_modifier; // 1: alpha = -1 by negating the accumulators on the way in and out, as rounds 1-3 did (A/B switch)
	int row_table;       // 1: the square product with row-start skipping (C^-1 = U U^T) gives whole tile rows to an XCD (build_row_table)
	int stagger_ticks;   // > 0: of the launch's first round, the workgroup in the odd slot of its CU starts this many 10 ns ticks late (kernel comment)
};

// Schedule switches of ONE context: read from the environment once, when the context is created (INTEGRATION.md lists
// the variables), and constant for its lifetime.  Two contexts of a process may differ; nothing process-wide is written
// after start-up, so contexts created and driven from several host threads (csrc/host/multi.c) cannot disturb each other,
// and a context's cached launch graphs always match its switches.
struct Sched {
	int gemm_big_tiles = 1024;   // GPEMU_GEMM_BIG_TILES
	int gemm_table = 8;          // GPEMU_GEMM_TABLE (0 .. 64)
	int factor_ahead = 1;        // GPEMU_FACTOR_AHEAD: the update's tile (0,0) factors the next diagonal block
	int fill_gram = 1;           // GPEMU_FILL_GRAM: MFMA Gram form of the training fill
	int kvec_gram = 1;           // GPEMU_KVEC_GRAM: MFMA Gram form of the prediction sweep's k-vectors
	int gemv_point = 1;          // GPEMU_GEMV_POINT: ONE query takes the matrix-vector kernel instead of the skinny MFMA product
	int idle_waves = 1;          // GPEMU_IDLE_WAVES: waves wholly above the diagonal of a diagonal tile issue no matrix instructions
	int nb_top = 0;              // GPEMU_NB_TOP: outer panel width; 0 = automatic (512 for one matrix, 2048 / 1024 for a batch)
	int split_rhs_rows = 1;      // GPEMU_SPLIT_RHS_ROWS: big-tile updates take the 64 right-hand-side rows in a launch of their own
	int neg_modifier = 1;        // GPEMU_NEG_MODIFIER: C - A B^T through the NEG bit of the fp64 matrix instruction
	int grad_gram = 1;           // GPEMU_GRAD_GRAM: the exact gradient's tile distances from the matrix unit (grad_exact_gram_kernel)
	int leaf_staged = -1;        // GPEMU_LEAF_STAGED: the leaf solve moves whole 512-byte row pieces through an LDS strip (1), element-wise (0), automatic by launch size (-1)
	int diag_inv_ahead = 1;      // GPEMU_DIAG_INV_AHEAD: the leaf solve takes the 16x16 diagonal inverses the factoring workgroup left in the block's upper part (0: every workgroup computes them)
	int leaf_pair = 1;           // GPEMU_LEAF_PAIR: first block of a 128-column pair in one launch (leaf solve + K=64 update, leaf_pair_kernel); 0: two launches
	int corner_row_table = 1;    // GPEMU_CORNER_ROW_TABLE: C^-1 = U U^T with whole tile rows per XCD (0: row-major enumeration, round-robin over the XCDs)
	int stagger_us = 20;         // GPEMU_STAGGER_US: first-round offset between the two workgroups of a CU in the 128x128 GEMM (0 = none)
};

struct ProfState {
	int cls = GPEMU_PROF_NONE;
	std::vector<hipEvent_t> ev;   // pairs
	double flops = 0, bytes = 0;
	int n = 0;
	std::vector<std::string> tag;  // one label per launch (GPEMU_PROF_DUMP=1 prints them with their times)
};

} // namespace gpemu

struct gpemu_ctx {
	int device = 0;
	hipStream_t stream = nullptr;      // the context's one stream: every launch and copy of the context is ordered on it
	gpemu::Sched sched;                // schedule switches, fixed at creation
	std::string err;

	// model
	int kind = 0, order = 0, N = 0, d = 0, nreg = 0, nrhs = 0;
	int Np = 0, Rp = 0;
	double *dX = nullptr;        // N x d
	double *dXg = nullptr;       // N x d, centred per dimension (x - mid_k): operands of the Gram-form fill
	double *dMid = nullptr;      // d: the centres mid_k (the Gram-form k-vectors centre their query rows with them)
	std::vector<double> xhalf;   // d: half range of each design coordinate
	double *dY = nullptr;        // N
	double *dRrows = nullptr;    // Rp x Np : row 0 = y, rows 1..nreg = H columns, zero padded
	std::vector<double> hX, hY;

	// factorisation workspace: tall matrix T (rows x Np)
	double *dT = nullptr;
	size_t T_rows = 0;           // allocated rows (all matrices of a batch together)
	int nb = 1;                  // matrices factored in lock-step by the current / last factorisation
	size_t T_stride = 0;         // elements between consecutive matrices of the batch
	int batch_cap = 0;           // allocated per-matrix result slots (dInfo, dGramPart, dRes, hRes, hInfo)
	int *dInfo = nullptr;
	double *dGramPart = nullptr; // [Np/128][Rp*Rp]
	double *dRes = nullptr;      // Rp*Rp gram + logdet + spare
	double *hRes = nullptr;      // pinned mirror: the newest slot of the ring below
	int *hInfo = nullptr;        // pinned, likewise
	static constexpr int RES_RING = 4;   // result slots: batch j stays readable while j+1 .. j+3 are enqueued
	double *hResRing = nullptr;  // RES_RING x batch_cap x res_len
	int *hInfoRing = nullptr;    // RES_RING x batch_cap
	hipEvent_t res_ev[RES_RING] = {nullptr, nullptr, nullptr, nullptr};   // recorded behind the copies into a slot
	int res_nb[RES_RING] = {0, 0, 0, 0};
	int res_kind[RES_RING] = {0, 0, 0, 0};          // 0: likelihood batch, 1: value+gradient batch
	std::vector<double> res_th[RES_RING];           // value+gradient batches: the thetas they were enqueued with (theta[0] = 0)
	int res_nthetas[RES_RING] = {0, 0, 0, 0};
	int res_mode[RES_RING] = {0, 0, 0, 0};          // the context's mode flags when the batch was enqueued
	static constexpr int GRAD_NP_MAX = 2 * GPEMU_MAX_PARAMS + 2;   // reduced gradient sums per batch element
	double *dGradSum = nullptr;  // batch_cap x GRAD_NP_MAX
	double *hGradRing = nullptr; // pinned, RES_RING x batch_cap x GRAD_NP_MAX
	double *hGph = nullptr;      // pinned upload ring of the length thetas: PARAM_RING x GPEMU_MAX_BATCH x GPEMU_MAX_PARAMS
	unsigned param_slot = 0;     // ring entry of the hyper-parameter upload of the batch being enqueued
	unsigned long long res_seq = 0;      // batches enqueued since the ring was (re)allocated
	size_t res_len = 0;

	// cached launch graphs for potrf, keyed by (Np, rows_total, with_inverse)
	struct GraphKey { int Np; int aug_fixed; int inv; int nb; bool operator<(const GraphKey &o) const {
		if (Np != o.Np) return Np < o.Np; if (aug_fixed != o.aug_fixed) return aug_fixed < o.aug_fixed;
		if (inv != o.inv) return inv < o.inv; return nb < o.nb; } };
	std::map<GraphKey, hipGraphExec_t> graphs;
	std::set<GraphKey> warm;     // shapes factored once with plain launches (the graph is recorded on the second call)
	bool use_graph = true;

	// prediction state
	bool pred_ready = false;
	double *dLinvAug = nullptr;  // (Np + Rp) x Np : rows [0,Np) = L^-1, row Np = gamma, rows Np+1.. = W^T
	double *dBetaQ = nullptr;    // beta (nreg) then Q (nreg*nreg)
	double kappa = 0;
	gpemu::CovParams pred_cov;
	std::vector<double> h_beta, h_Q;
	double *dKq = nullptr, *dV = nullptr; // batch buffers
	int pred_batch = 0;
	double *dXq = nullptr, *dMean = nullptr, *dVar = nullptr; // staging for host-buffer entry
	int stage_cap = 0;
	double *hStage = nullptr;    // pinned: stage_cap*d query coordinates, then stage_cap means, then stage_cap variances
	int pred_pending = 0;        // queries of an enqueued, not yet collected prediction batch
	bool cinv_ready = false;
	bool fact_in_T = false;      // the factorisation (with inverse rows) behind the prediction state sits in THIS context's workspace, element 0
	                             // (false after gpemu_predict_setup_batch for every context but the first: their factorisations ran in the first one's)
	double *dS = nullptr;        // S_cap corners of (Rp+Np)^2 for explicit inverse / gradient (one per batch element in flight)
	size_t S_dim = 0;
	int S_cap = 0;

	// gradient scratch
	double *dGradPart = nullptr;
	// gpemu_symm_apply: a host-resident symmetric matrix kept on the device between calls
	double *dSym = nullptr, *dSymV = nullptr, *dSymOut = nullptr;
	const double *sym_key = nullptr;
	int sym_N = 0, sym_lda = 0, sym_pad = 0, sym_vcap = 0;
	uint64_t sym_fp = 0;          // checksum of every element of the cached host matrix
	bool sym_pinned = false;      // gpemu_symm_pin: the caller vouches for the buffer, no per-call checksum
	gpemu::CovParams *dParams = nullptr;   // hyper-parameters of the batch elements (GPEMU_MAX_BATCH slots)
	gpemu::CovParams *hParams = nullptr;   // pinned upload ring: PARAM_RING x GPEMU_MAX_BATCH slots, one event per ring entry
	hipEvent_t param_ev[4] = {nullptr, nullptr, nullptr, nullptr};
	unsigned param_next = 0;
	double *dAlpha = nullptr;    // gradient, per corner in flight: Np doubles of alpha = C^-1 y, then GPEMU_MAX_PARAMS length thetas
	int alpha_cap = 0;
	size_t gradpart_len = 0;     // doubles of dGradPart (all corners)

	int mode = 0;                 // GPEMU_MODE_* flags (gpemu_set_mode; defaults from the environment)
	gpemu::ProfState prof;
	// GPEMU_TRACE=1: per-launch device timestamps (wall_clock64) written by the kernels themselves, so that the
	// concurrent timeline of several contexts can be read (rocprofv3 serialises kernels)
	unsigned long long *dTrace = nullptr;
	int trace_cap = 0, trace_next = 0;
	std::vector<std::string> trace_tag;
	std::vector<double> last_thetas;
};

namespace gpemu {

// ---- kernels_cov.hip
hipError_t launch_cov_fill(hipStream_t s, double *out, long ld, const double *Xr, int nr, int nr_pad,
                           const double *Xc, int nc, int nc_pad, int d, const CovParams &p, int mode);
constexpr int FILL_LOWER = 1, FILL_CLAMP = 2, FILL_IDENT_PAD = 4;
// k-vectors of M query rows (padded to Mp) against the design in Gram form (p.gram set; Xg = centred design, mid = its centre)
hipError_t launch_cov_kvec_gram(hipStream_t s, double *out, long ld, const double *Xq, int M, int Mp, const double *X, const double *Xg,
                                const double *mid, int N, int Np, int d, const CovParams &p);
hipError_t launch_build_rrows(hipStream_t s, double *R, int Np, int Rp, const double *X, const double *y,
                              int N, int d, int order);
hipError_t launch_set_identity_rows(hipStream_t s, double *T, long ld, int n, int nbatch = 1, long bstride = 0);
hipError_t launch_cov_stage_batch(hipStream_t s, double *T, long ld, long bstride, int nb, const double *X, int N, int Np, int d,
                                  const CovParams *pp_dev, int mode, const double *Rrows, int Rp, const double *Xg = nullptr,
                                  bool all_gram = false, int kind = GPEMU_POWEREXP, long rstride = 0);
hipError_t launch_transpose(hipStream_t s, double *dst, long ldd, const double *src, long lds, int n);
hipError_t launch_predict_finish(hipStream_t s, const double *V, long ldv, int M, int Np, int nreg, int order, int d,
                                 const double *Xq, const double *betaQ, double kappa, double *mean, double *var,
                                 int nslice = 1, long sstride = 0);
// the few-queries path of the prediction sweep (emulate_point): k-vectors of up to 16 queries, one thread per design point; the
// epilogue with the slice sums fused in, one workgroup per query
hipError_t launch_kvec_small(hipStream_t s, double *Kq, long ld, const double *Xq, int M, const double *X, int N, int Np, int d,
                             const CovParams &p);
hipError_t launch_predict_finish_small(hipStream_t s, const double *Vp, long ldv, long sstride, int nslice, int M, int Np, int nreg,
                                       int d, const double *Xq, const double *betaQ, double kappa, double *mean, double *var);
hipError_t launch_grad_partials(hipStream_t s, const double *S, long lds, int soff, long sstride, int nb, const double *X, int N,
                                int d, double *ag, int np_pad, long gstride, double *part, long pstride, int *nparts,
                                int exact_kind = 0, int nbeta = 0, const CovParams *pp_dev = nullptr, bool lit_noclamp = false,
                                const double *Xg = nullptr);

hipError_t launch_beta_solve(hipStream_t s, const double *res, long rstride, int Rp, int nreg, int nb, double *ag, long gstride,
                             int np_pad);
hipError_t launch_grad_reduce(hipStream_t s, const double *part, long pstride, int ntiles, int np, int nb, double *sums, long sstride);

hipError_t launch_deriv_gauss(hipStream_t s, double *out, long ld, const double *xcol, int n, double theta_len);
hipError_t launch_trace_product(hipStream_t s, const double *A, const double *B, long ld, int n, double *part);

// ---- kernels_linalg.hip
hipError_t launch_gemm(hipStream_t s, const GemmArgs &a);
hipError_t launch_skinny_nt(hipStream_t s, const double *Kq, long ldk, const double *L, long ldl, double *Vp, long ldv,
                            long sstride, int tq, int ntot, int K, int ntri, int nslice, int klen);
hipError_t launch_gemv_tri(hipStream_t s, const double *Kq, long ldk, const double *L, long ldl, double *Vp, long ldv,
                           long sstride, int mq, int ntot, int K, int ntri, int nslice, int klen);
std::vector<int> build_tile_table(int tiles_m, int tiles_n, int tri, int S, int bm = 128, int bn = 128);
std::vector<int> build_row_table(int tiles_m, int bm, int kstart_off, int k0, int k1);
hipError_t launch_leaf(hipStream_t s, double *T, long ld, int c0, int m_below, int *info,
                       unsigned long long *trace_factor = nullptr, unsigned long long *trace_solve = nullptr,
                       int nbatch = 1, long bstride = 0, bool skip_factor = false, int staged = -1, bool pre = true, int c0b = -1);
hipError_t launch_leaf_pair(hipStream_t s, double *T, long ld, int c0, int m_below, int *info, unsigned long long *trace, int nbatch,
                            long bstride, bool fa);
bool gemm_factor_ahead_ok(const GemmArgs &a);
bool gemm_uses_big_tiles(const GemmArgs &a);
hipError_t launch_gram_partials(hipStream_t s, const double *Z, long ld, int Np, int nrhs, int Rp, double *part,
                                int nbatch = 1, long zstride = 0);
hipError_t launch_finish(hipStream_t s, const double *part, int nparts, int Rp, int nrhs, const double *T, long ld,
                         int N, double *res, int nbatch = 1, long tstride = 0, long rstride = 0);

} // namespace gpemu

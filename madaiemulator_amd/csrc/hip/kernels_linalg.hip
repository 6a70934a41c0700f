// kernels_linalg.hip -- fp64 dense kernels for gfx950 (MI355X):
//   * gemm_nt_kernel : C = beta*C + alpha*A*B^T on v_mfma_f64_16x16x4_f64; 128x128 / 64x64 tiles, operand chunks by
//                      LDS-DMA into a swizzled double buffer, two fragment sets
//   * leaf_kernel    : 64x64 diagonal-block Cholesky in LDS + forward substitution of the panel rows below
//   * gram / finish  : Z Z^T Gram matrix of the solved right-hand sides and sum(log L_ii)
//
// These stand in for gsl_linalg_cholesky_decomp / _invert and the gsl_blas
// dgemm/dgemv/ddot calls of the reference's likelihood path
// (libEmu/maxmultimin.c:325,361; libEmu/regression.c:128-171;
// libEmu/estimator-fns.c:87-88) -- see DESIGN.md for the formulation.
#include "gpemu_internal.hpp"
#include <algorithm>
#include <mutex>
#include <tuple>

namespace gpemu {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

// ---------------------------------------------------------------------------
// GEMM  C[m x n] = beta*C + alpha * A[m x K] * B[n x K]^T   (row-major, k contiguous in A and B)
//
// 4 waves as 2x2 (64x64 tiles) or 8 waves as 4x2 (128x128 tiles); each wave owns a (BM/WGM)x(BN/WGN) sub-tile of
// 16x16 MFMA tiles.  MFMA f64 16x16x4 operand maps (cdna_hip_programming.md
// section 3): A lane l -> A[row l&15][k l>>4], B lane l -> B[k l>>4][col l&15],
// D reg r -> D[row (l>>4)+4r][col l&15].  A lane reads its fragment as one ds_read_b128 = the doubles k = 8t+2g, 8t+2g+1
// of its row and feeds them to MFMA k-steps 2t and 2t+1 (A and B use the same k permutation: contraction unchanged).
// Two tile shapes: the fp64 MFMA rate per CU is low (one 16x16x4 per 64 cycles per SIMD), so short or narrow updates
// need many small tiles to cover 256 CUs while the big trailing updates want 128x128 for L2 traffic; launch_gemm
// picks per call.
// ---------------------------------------------------------------------------
// (A persistent variant -- grid = resident workgroups, tile loop with the next tile's first chunk and the C
// tile prefetched under the epilogue -- was measured 5-10 % slower on the big updates (register pressure,
// imbalance of the static tile stride in triangular mode) and only ~10 % faster on the narrow ones; not kept.)
// GPEMU_TRACE slot = 8 x u64, zero-initialised: [0] max(~start) i.e. earliest workgroup start, [1] latest end,
// [2] sum of workgroup lifetimes, [3] workgroups (device wall clock, 100 MHz), [4] sum of workgroup lifetimes in
// shader clocks (s_memtime): [4]/[2] = the shader clock the kernel really ran at
struct TraceT0 { unsigned long long wall, clk; };
__device__ __forceinline__ TraceT0 trace_begin(unsigned long long *t)
{
	TraceT0 r = {0, 0};
	// one workgroup in 16 reports (same-address atomics cost ~12 ns each: every workgroup reporting inflated the
	// many-small-workgroup kernels several times over)
	if (!t || threadIdx.x != 0 || ((blockIdx.x + 5 * blockIdx.y) & 15) != 0) return r;
	r.wall = wall_clock64();
	r.clk = clock64();
	atomicMax(t, ~r.wall);
	return r;
}
__device__ __forceinline__ void trace_end(unsigned long long *t, TraceT0 t0)
{
	if (!t || threadIdx.x != 0 || t0.wall == 0) return;
	const unsigned long long now = wall_clock64();
	const unsigned long long clk = clock64();
	atomicMax(t + 1, now);
	atomicAdd(t + 2, now - t0.wall);
	atomicAdd(t + 3, 1ull);
	atomicAdd(t + 4, clk - t0.clk);
}

// ---- pieces of the 64x64 leaf factorisation used by the factor-ahead GEMM tile (defined further down)
constexpr int LP = LEAF + 2;
template <int P, int LD> __device__ __forceinline__ void panel_factor(double *A, int lane, int &bad, int bad_off);
template <int P, int LD> __device__ __forceinline__ void panel_update(double *A, int wave, int lane);
__device__ __forceinline__ void tri_inverse16(double *M, int o, double *tile, int lane);
__device__ __forceinline__ void diag_inverse_ahead(double *A, int blk, int lane, double *Dg, long ldg);

// FA = 1 (factor-ahead, 64x64 tiles of a triangular trailing update): the workgroup of tile (0,0) -- the diagonal block
// the NEXT leaf factorisation starts from -- does not store its updated tile: it keeps it in LDS, factors it there
// (the leaf_factor_kernel code) and writes L.  The 64 sequential pivots of that block then run beside the other tiles
// of the update instead of in a launch of their own behind it (one launch and ~10 us less on the critical chain per
// 64 columns); the arithmetic is the update's and the leaf's, so the bits do not change.
// Operand staging: the 16-deep operand chunks go global -> LDS directly (buffer_load ... lds, 1 KB per
// wave-instruction, no staging registers, no ds_write pass) into an unpadded image whose 16-byte slots are XOR-swizzled
// through the SOURCE address (the destination of an LDS-DMA is lane-linear); the registers this frees hold a second set
// of MFMA fragments, so the LDS reads of one half k-step are issued a whole block of 16 MFMAs before their use, and the
// one barrier of a k-step sits between the two blocks: after it the next chunk's DMA starts (a full k-step to land) and
// the first fragments of the next chunk are read under the second block.  Per accumulator the MFMA sequence is the
// k-ordered chain whatever the tile shape: 128x128 and 64x64 tiles give the same bits.
// (The register-staged loops of rounds 1-2, their L2-prefetch variant and the 128x64 / 256x128 / 128x256 shapes measured
// slower -- profiles/r02_gemm_*.txt -- and left the tree in round 3; `git log` has them.)
template <int BM, int BN, int MINW, int WGM, int WGN, int FA = 0, int NEG = 0>
__global__ __launch_bounds__(64 * WGM * WGN, MINW) void gemm_nt_kernel(GemmArgs g)
{
	constexpr int WM = BM / WGM, WN = BN / WGN;
	constexpr int TM = WM / 16, TN = WN / 16;
	constexpr int NW = WGM * WGN;                       // waves
	constexpr int NPA = BM / (8 * NW), NPB = BN / (8 * NW);      // LDS-DMA pieces (8 rows x 128 B) per wave and operand
	constexpr int SMEM_GEMM = 2 * (BM + BN) * GEMM_BK;        // two unpadded chunk images
	constexpr int SMEM = (FA && LEAF * LP > SMEM_GEMM) ? LEAF * LP : SMEM_GEMM;
	__shared__ double smem[SMEM];

	// batch of independent problems (lock-step factorisations): blockIdx.y selects the matrix
	g.C += (long)blockIdx.y * g.bsC;
	g.A += (long)blockIdx.y * g.bsA;
	g.B += (long)blockIdx.y * g.bsB;
	const int tiles_m = (g.m + BM - 1) / BM;
	// Tile order.  Launches of >= 512 tiles take it from the XCD-blocked table (order_mode 3, gemm_tile_table below);
	// smaller ones walk down tile columns (dense enumeration of the lower triangle, order_mode 2, or the plain 2-D
	// order): equally fast at these sizes -- the operands sit in the 256 MB Infinity Cache -- with more fabric traffic.
	int tm, tn;
	if (g.order_mode == 3) {
		// XCD-blocked order from a host-built table (gemm_tile_table below)
		const int e = g.tile_table[blockIdx.x];
		if (e < 0) return;
		tm = e >> 16;
		tn = e & 0xffff;
	} else if (g.order_mode == 2) {
		// dense enumeration of the lower-triangular tiles (square tiles, diag_off = 0, m >= n): block t -> column tn
		// with P(tn) <= t < P(tn+1), P(j) = j*tiles_m - j(j-1)/2 tiles in the columns before j.  The plain 2-D
		// enumeration launches the empty upper tiles too; they exit at once, but the dispatcher deals workgroups in
		// order and runs of hundreds of empty ones starve the CUs: 54 -> 67 TFLOP/s on a 16384^2 x 1024 update.
		const long t = blockIdx.x;
		if (g.kstart_mode && g.m == g.n) {
			// rows of the operand start at k = row: tile rows are ordered by decreasing work -> row-major enumeration
			// (t = r(r+1)/2 + c) puts the long-K tiles first and leaves the short ones for the tail of the launch
			int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
			while ((long)r * (r + 1) / 2 > t) r--;
			while ((long)(r + 1) * (r + 2) / 2 <= t) r++;
			tm = r;
			tn = (int)(t - (long)r * (r + 1) / 2);
		} else {
		const double b2 = 2.0 * tiles_m + 1.0;
		int j = (int)((b2 - sqrt(b2 * b2 - 8.0 * (double)t)) * 0.5);
		if (j < 0) j = 0;
		while (j > 0 && (long)j * tiles_m - (long)j * (j - 1) / 2 > t) j--;
		while ((long)(j + 1) * tiles_m - (long)(j + 1) * j / 2 <= t) j++;
		tn = j;
		tm = j + (int)(t - ((long)j * tiles_m - (long)j * (j - 1) / 2));
		}
	} else {
		tm = blockIdx.x % tiles_m;
		tn = blockIdx.x / tiles_m;
		// triangular B operand: the k-range grows with the tile column; start with the long tiles so that the tail
		// of the launch is made of short ones
		if (g.kend_mode) tn = (g.n + BN - 1) / BN - 1 - tn;
		// (the other way round -- neighbours share the query rows and walk down the rows of L^-1 -- was measured on the
		// prediction sweep: 0.70 against 1.01 M predictions/s, round 4)
	}
	if (g.tri && tn * BN > tm * BM + BM - 1 + g.diag_off) return;
	const TraceT0 tr0 = trace_begin(g.trace);
	// (prologue and epilogue at wave priority 3, the k-loop at 0: the prologue shrinks from 13.3 to 7.8 us -- its instructions
	// wait for issue slots behind the neighbours' 64-cycle matrix instructions -- and the launch gets 1-4 % SLOWER: those
	// slots were the neighbours' matrix instructions; the prologue's time is not idle matrix-pipe time.  Round 4.)
	if (BM == 128 && g.stagger_ticks > 0 && (long)blockIdx.y * gridDim.x + blockIdx.x < 512) {
		// The two workgroups that share a CU start together and their tiles take the same time: left alone they stay in
		// phase for the whole launch -- both store their tile, both load the next one and its first chunks in the same tens
		// of microseconds, during which the CU's matrix pipes have nothing to do (profiles/r04_pmc_sq.txt: busy 90 %).  Of the
		// launch's first round the workgroup in the odd slot of its CU (HW_ID.TG_ID, scratch/mb/hwid_pairs.hip) therefore
		// starts late by about the length of that phase; its successors inherit the offset, and from then on one
		// workgroup's memory phase runs under the other's matrix instructions.
		unsigned hw;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
		if ((hw >> 16) & 1) {
			const unsigned long long t0 = wall_clock64();
			while (wall_clock64() - t0 < (unsigned long long)g.stagger_ticks) __builtin_amdgcn_s_sleep(16);
		}
		// (spreading the whole first round over 0 .. 375 us, so that the 512 C-tile reads of a round do not meet either, made
		// the K=2048 launches 4-10 % slower: it is the pair on a CU that matters, not the chip-wide burst)
	}
	int kb = g.k0, ke = g.k1;
	if (g.kstart_mode) {
		int ks = (tm * BM - g.kstart_off) & ~(GEMM_BK - 1);
		if (ks > kb) kb = ks;
	}
	if (g.kend_mode) {
		int kx = (tn * BN + BN - g.kend_off + GEMM_BK - 1) & ~(GEMM_BK - 1);
		if (kx < ke) ke = kx;
	}
	if (g.ksplit > 1) {
		// split-K for updates with few tiles and a long K (a handful of prediction queries against L^-1): grid.y
		// enumerates k-slices, each writes its own partial tile (C was offset by blockIdx.y * bsC above)
		const int klen = (((g.k1 - g.k0) + g.ksplit - 1) / g.ksplit + GEMM_BK - 1) & ~(GEMM_BK - 1);
		const int lo = g.k0 + (int)blockIdx.y * klen;
		if (lo > kb) kb = lo;
		if (lo + klen < ke) ke = lo + klen;
	}

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = tid >> 6;
	const int wm = wave / WGN, wn = wave % WGN;

	// The accumulators start from the C tile itself (scaled by beta/alpha, alpha = +-1 when beta is set: exact), so
	// the read half of the read-modify-write overlaps the operand prologue and the epilogue is stores only.  (Read
	// in the epilogue, load and store of one element serialise: 11 us per 128x128 tile, 38 us under load.)
	const int row0 = tm * BM + wm * WM + (lane >> 4);
	const int col0 = tn * BN + wn * WN + (lane & 15);
	const bool full_tile = (tm * BM + BM <= g.m) && (tn * BN + BN <= g.n);
	// a wave whose whole sub-tile lies strictly above the diagonal of a triangular update (two of the eight waves of a
	// diagonal 128x128 tile, one of the four of a 64x64 one) computes output nobody reads: it keeps moving operand chunks
	// and meeting the barriers, but issues no matrix instruction, no fragment read, no C access -- its SIMD's matrix pipe
	// goes to the other workgroup of the CU.  (Not in the factor-ahead tile, whose LDS image is written by every wave.)
	// (through readfirstlane: a wave-uniform value the compiler can branch on -- under an EXEC mask the matrix instructions
	// would still be issued)
	const bool idle = __builtin_amdgcn_readfirstlane((int)(g.tri && !g.keep_idle_waves && !(FA && g.fa && tm == 0 && tn == 0) &&
	                                                     (tn * BN + wn * WN > tm * BM + wm * WM + WM - 1 + g.diag_off))) != 0;
	// C - A B^T (alpha = -1 on top of C: every update of the factorisation): the matrix instruction negates its A operand
	// (NEG bit of the fp64 MFMA, the builtin's last argument) instead of the accumulators being negated on the way in and on
	// the way out -- 64 vector instructions per lane and tile less, and the C tile's load no longer has to land before the
	// first operand chunk is requested.  fma(-a, b, c) = -fma(a, b, -c) exactly: same bits.
	// (NEG = 1 instantiations; launch_gemm picks them for beta != 0, alpha = -1.  Both loops in one kernel spilled 100 registers.)
	constexpr bool nega = NEG != 0;
	d4_t acc[TM][TN];
	if (idle) {
#pragma unroll
		for (int i = 0; i < TM; i++)
#pragma unroll
			for (int j = 0; j < TN; j++) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};
	} else if (g.beta) {
		if (full_tile) {
#pragma unroll
			for (int i = 0; i < TM; i++)
#pragma unroll
				for (int r = 0; r < 4; r++) {
					const double *crow = g.C + (long)(row0 + i * 16 + 4 * r) * g.ldc + col0;
#pragma unroll
					for (int j = 0; j < TN; j++) acc[i][j][r] = crow[j * 16];
				}
			// (the tile through buffer loads -- one lane offset, the row steps in scalar registers, the column steps in the
			// immediate field: ~40 vector instructions of address arithmetic less per lane -- measured equal to slightly slower,
			// round 4: the prologue's 13-15 us are not its vector instructions)
		} else {
#pragma unroll
			for (int i = 0; i < TM; i++)
#pragma unroll
				for (int r = 0; r < 4; r++) {
					const int row = row0 + i * 16 + 4 * r;
					const double *crow = g.C + (long)(row < g.m ? row : g.m - 1) * g.ldc;
#pragma unroll
					for (int j = 0; j < TN; j++) {
						const int col = col0 + j * 16;
						acc[i][j][r] = crow[col < g.n ? col : g.n - 1];
					}
				}
		}
		// C / alpha with alpha = +-1 (launch_gemm refuses anything else when beta is set): a sign flip, not 64 fp64
		// multiplications per lane (same bits; measured neutral)
		if (g.alpha < 0.0 && !nega) {
#pragma unroll
			for (int i = 0; i < TM; i++)
#pragma unroll
				for (int j = 0; j < TN; j++) acc[i][j] = -acc[i][j];
		}
	} else {
#pragma unroll
		for (int i = 0; i < TM; i++)
#pragma unroll
			for (int j = 0; j < TN; j++) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};
	}

	static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && NW % 2 == 0 && NPA + NPB <= 8 && GEMM_BK == 16 && WM % 16 == 0 && WN % 16 == 0,
	              "LDS-DMA loop: 16-deep chunks, 8-row pieces dealt to the waves in turn");
	if (kb < ke) {
		// LDS image (bytes): buffer b at b * BUFB; operand row R (A rows 0..BM-1, then the B rows) at R * 128; its 16-byte
		// segment s (doubles 2s, 2s+1 of the chunk) in slot s ^ f(R), f(R) = (r >> 1) ^ (4 <= r <= 11), r = R mod 16.
		// A ds_read_b128 is served in four groups of 16 non-contiguous lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}
		// and the same + 32 (MI355X_MICROARCH.md, LDS) -- i.e. with fragment lane = q + 16 g: rows q in {0-3, 12-15} at
		// segment 4t+g together with rows q in {4-11} at segment 4t+(g^1).  Rows alternate between the two 128-byte halves
		// of the 256-byte bank row (row stride 128 B), so the 8 even and the 8 odd rows of a group each need 8 different
		// slots: (4t+g) ^ f(q) for the outer rows and (4t+g) ^ 1 ^ f(q) for the middle ones are 8 different values because
		// f(q) ^ (4 <= q <= 11) = q >> 1.  (The first form of the swizzle, (R & 7) ^ ((R >> 3) & 1), assumed contiguous
		// 16-lane groups: SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE.)
		constexpr int BUFB = (BM + BN) * GEMM_BK * 8;
		char *lds = reinterpret_cast<char *>(smem);
		// DMA pieces: wave w moves rows 8 NW p + 8 w .. + 7 of the image for p = 0 .. NPA+NPB-1 (the first NPA: A rows),
		// lane l -> row + (l >> 3), slot l & 7 (8 NW is a multiple of 16: row mod 16, hence f, does not depend on p)
		const int uw = __builtin_amdgcn_readfirstlane(wave);
		const int prow = 8 * uw + (lane >> 3);
		const int prow16 = prow & 15;
		const int pseg = (lane & 7) ^ ((prow16 >> 1) ^ (((prow16 + 4) >> 3) & 1));
		int ra0 = tm * BM; if (ra0 > g.m - 1) ra0 = g.m - 1;
		int rb0 = tn * BN; if (rb0 > g.n - 1) rb0 = g.n - 1;
		const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(g.A + (long)ra0 * g.lda), 0, -1, 0x00020000);
		const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(g.B + (long)rb0 * g.ldb), 0, -1, 0x00020000);
		unsigned vo[8];                                   // (a size that depends on the template arguments loses the host stubs with this hipcc)
#pragma unroll
		for (int p = 0; p < NPA; p++) {
			int ar = tm * BM + 8 * NW * p + prow; if (ar > g.m - 1) ar = g.m - 1;
			vo[p] = (unsigned)((long)(ar - ra0) * g.lda * 8 + 16 * pseg);
		}
#pragma unroll
		for (int p = 0; p < NPB; p++) {
			int br = tn * BN + 8 * NW * p + prow; if (br > g.n - 1) br = g.n - 1;
			vo[NPA + p] = (unsigned)((long)(br - rb0) * g.ldb * 8 + 16 * pseg);
		}
		typedef __attribute__((address_space(3))) void *lds_ptr_t;
#define GEMM_DMA1(rs, d, p, kk) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)((d) + 1024 * NW * (p)), 16, vo[p], (kk) * 8, 0, 0)
#define GEMM_DMA(buf, kk)                                                                                         \
		do {                                                                                                      \
			char *d = lds + (buf) * BUFB + uw * 1024;                                                              \
			GEMM_DMA1(rsA, d, 0, kk);                                                                              \
			if constexpr (NPA > 1) GEMM_DMA1(rsA, d, 1, kk);                                                       \
			if constexpr (NPA > 2) GEMM_DMA1(rsA, d, 2, kk);                                                       \
			if constexpr (NPA > 3) GEMM_DMA1(rsA, d, 3, kk);                                                       \
			GEMM_DMA1(rsB, d, NPA, kk);                                                                            \
			if constexpr (NPB > 1) GEMM_DMA1(rsB, d, NPA + 1, kk);                                                 \
			if constexpr (NPB > 2) GEMM_DMA1(rsB, d, NPA + 2, kk);                                                 \
			if constexpr (NPB > 3) GEMM_DMA1(rsB, d, NPA + 3, kk);                                                 \
		} while (0)
		// fragment addresses: lane (q, gq) reads row q of its 16-row group, logical segment 4 t + gq
		const int q = lane & 15, gq = lane >> 4;
		const int fsw = (q >> 1) ^ (((q + 4) >> 3) & 1);
		const int fa0 = (wm * WM + q) * 128 + 16 * (gq ^ fsw);
		const int fb0 = (BM + wn * WN + q) * 128 + 16 * (gq ^ fsw);
#define GEMM_FRAGS(FA_, FB_, buf, t)                                                                              \
		do {                                                                                                      \
			const char *pa = lds + (buf) * BUFB + (fa0 ^ (64 * (t)));                                              \
			const char *pb = lds + (buf) * BUFB + (fb0 ^ (64 * (t)));                                              \
			_Pragma("unroll") for (int i = 0; i < TM; i++) FA_[i] = *reinterpret_cast<const d2_t *>(pa + i * 2048); \
			_Pragma("unroll") for (int j = 0; j < TN; j++) FB_[j] = *reinterpret_cast<const d2_t *>(pb + j * 2048); \
		} while (0)
#define GEMM_BLOCK(FA_, FB_, NEG_)                                                                                \
		do {                                                                                                      \
			_Pragma("unroll") for (int h = 0; h < 2; h++)                                                          \
				_Pragma("unroll") for (int i = 0; i < TM; i++)                                                     \
					_Pragma("unroll") for (int j = 0; j < TN; j++)                                                 \
						acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(FA_[i][h], FB_[j][h], acc[i][j], 0, 0, NEG_); \
		} while (0)
		d2_t xa[TM], xb[TN], ya[TM], yb[TN];
		// (every wave's buffer_load ... lds must have landed before the barrier that hands the chunk to the other waves: the
		// compiler places this wait itself today; it is spelled out so that correctness does not hang on that)
		// (chunk 1 requested together with chunk 0, waiting for chunk 0 alone -- vmcnt(NPA + NPB) -- was measured: the 13-14 us
		// of a tile's prologue did not move, round 4)
		GEMM_DMA(0, kb);
		unsigned long long clk_issue = 0;
		if (g.trace && tr0.wall) clk_issue = clock64();
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		if (g.trace && tr0.wall) atomicAdd(g.trace + 7, (unsigned long long)clock64() - clk_issue);     // C tile and chunk 0 landing
		if (kb + GEMM_BK < ke) GEMM_DMA(1, kb + GEMM_BK);
		if (!idle) GEMM_FRAGS(xa, xb, 0, 0);
		if (g.trace && tr0.wall) atomicAdd(g.trace + 5, (unsigned long long)clock64() - tr0.clk);   // prologue
		int cur = 0;
		for (int k = kb; k < ke; k += GEMM_BK) {
			if (!idle) {
				GEMM_FRAGS(ya, yb, cur, 1);
				__builtin_amdgcn_sched_barrier(0);
				GEMM_BLOCK(xa, xb, NEG);
			}
			__builtin_amdgcn_sched_barrier(0);
			// chunk k+1 (requested a k-step ago) has landed for every wave; every wave has read the last of chunk k
			asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
			__syncthreads();
			if (k + 2 * GEMM_BK < ke) GEMM_DMA(cur, k + 2 * GEMM_BK);
			if (!idle) {
				if (k + GEMM_BK < ke) GEMM_FRAGS(xa, xb, cur ^ 1, 0);
				__builtin_amdgcn_sched_barrier(0);
				GEMM_BLOCK(ya, yb, NEG);
			}
			__builtin_amdgcn_sched_barrier(0);
			cur ^= 1;
		}
#undef GEMM_DMA
#undef GEMM_DMA1
#undef GEMM_FRAGS
#undef GEMM_BLOCK
	}

	if (idle) { trace_end(g.trace, tr0); return; }
	// epilogue: alpha * accumulators (a sign flip or nothing for alpha = -+1, see above)
	unsigned long long clk_loop_end = 0;
	if (g.trace && tr0.wall) clk_loop_end = clock64();
	if (nega) {
		// nothing: the accumulators hold C - A B^T
	} else if (g.alpha == -1.0) {
#pragma unroll
		for (int i = 0; i < TM; i++)
#pragma unroll
			for (int j = 0; j < TN; j++) acc[i][j] = -acc[i][j];
	} else if (g.alpha != 1.0) {
#pragma unroll
		for (int i = 0; i < TM; i++)
#pragma unroll
			for (int j = 0; j < TN; j++) acc[i][j] = g.alpha * acc[i][j];
	}
	if (FA && g.fa && tm == 0 && tn == 0) {
		static_assert(!FA || (BM == LEAF && BN == LEAF && WGM == 2 && WGN == 2), "factor-ahead is for the 64x64 tiles, 4 waves");
		// the updated diagonal block goes to LDS instead of memory (every wave is past its last read of the operand
		// buffers: the k-loop ends with a barrier), is factored there and leaves as L
		double *A = smem;
#pragma unroll
		for (int i = 0; i < TM; i++)
#pragma unroll
			for (int r = 0; r < 4; r++)
#pragma unroll
				for (int j = 0; j < TN; j++)
					A[(wm * WM + i * 16 + (lane >> 4) + 4 * r) * LP + wn * WN + j * 16 + (lane & 15)] = acc[i][j][r];
		__syncthreads();
		int bad = 0;
		if (wave == 0) panel_factor<0, LP>(A, lane, bad, 0);
		__syncthreads();
		panel_update<0, LP>(A, wave, lane);
		__syncthreads();
		// (diag_inverse_ahead: the inverse of a finished 16x16 diagonal block goes to the block's upper part in HBM, where
		// the leaf solve finds it -- computed by a wave that would idle beside wave 0's next panel)
		if (wave == 0) panel_factor<1, LP>(A, lane, bad, 0);
		else if (wave == 1) diag_inverse_ahead(A, 0, lane, g.C, g.ldc);
		__syncthreads();
		panel_update<1, LP>(A, wave, lane);
		__syncthreads();
		if (wave == 0) panel_factor<2, LP>(A, lane, bad, 0);
		else if (wave == 2) diag_inverse_ahead(A, 1, lane, g.C, g.ldc);
		__syncthreads();
		panel_update<2, LP>(A, wave, lane);
		__syncthreads();
		if (wave == 0) panel_factor<3, LP>(A, lane, bad, 0);
		else if (wave == 3) diag_inverse_ahead(A, 2, lane, g.C, g.ldc);
		__syncthreads();
		if (wave == 1) diag_inverse_ahead(A, 3, lane, g.C, g.ldc);
		if (tid == 0 && bad) atomicMin(g.fa_info + blockIdx.y, g.fa_c0 + bad);
#pragma unroll
		for (int u = 0; u < 16; u++) {
			const int r = wave + 4 * u;
			if (lane <= r) g.C[(long)r * g.ldc + lane] = A[r * LP + lane];
		}
		trace_end(g.trace, tr0);
		return;
	}
	if (full_tile) {
#pragma unroll
		for (int i = 0; i < TM; i++)
#pragma unroll
			for (int r = 0; r < 4; r++) {
				double *crow = g.C + (long)(row0 + i * 16 + 4 * r) * g.ldc + col0;
#pragma unroll
				for (int j = 0; j < TN; j++) crow[j * 16] = acc[i][j][r];
			}
	} else {
#pragma unroll
		for (int i = 0; i < TM; i++)
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const int row = row0 + i * 16 + 4 * r;
				if (row >= g.m) continue;
				double *crow = g.C + (long)row * g.ldc;
#pragma unroll
				for (int j = 0; j < TN; j++) {
					const int col = col0 + j * 16;
					if (col >= g.n) continue;
					crow[col] = acc[i][j][r];
				}
			}
	}
	if (g.trace && tr0.wall) {
		__builtin_amdgcn_s_waitcnt(0);     // stores issued, loads returned
		atomicAdd(g.trace + 6, (unsigned long long)clock64() - clk_loop_end);                            // epilogue
	}
	trace_end(g.trace, tr0);
}

// ---------------------------------------------------------------------------
// Skinny product for a handful of prediction queries (emulate_point: ONE): Vp[s][q][n] = sum over the k-slice s of
// Kq[q][k] * L[n][k], q < 16*TQ.  A 64x64 GEMM tile would spend 63/64 of its MFMA work on padding rows; here a wave
// owns 16 columns n and TQ query tiles, streams its 16 rows of L once (32 contiguous bytes per lane and 16-k chunk,
// 128-byte segments per row) and keeps four independent accumulators per query tile so that the MFMAs pipeline.
// Rows n < ntri of L are lower triangular (L^-1): k beyond the block's last column is skipped.
// grid (ntot/64, nslice), 256 threads.
// ---------------------------------------------------------------------------
template <int TQ>
__global__ __launch_bounds__(256) void skinny_nt_kernel(const double *Kq, long ldk, const double *L, long ldl, double *Vp,
                                                        long ldv, long sstride, int K, int ntri, int klen)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int q = lane & 15, g = lane >> 4;
	const int n0 = blockIdx.x * 64 + 16 * wave;
	const int kb = blockIdx.y * klen;
	int ke = kb + klen < K ? kb + klen : K;
	if (n0 < ntri) {
		const int kx = (n0 + 16 + 15) & ~15;        // columns <= n0+15 of the triangular rows
		if (kx < ke) ke = kx;
	}
	d4_t acc[TQ][4];
#pragma unroll
	for (int t = 0; t < TQ; t++)
#pragma unroll
		for (int u = 0; u < 4; u++) acc[t][u] = (d4_t){0.0, 0.0, 0.0, 0.0};
	const double *lp = L + (long)(n0 + q) * ldl + 4 * g;
	const double *ap = Kq + (long)q * ldk + 4 * g;
	// memory-bound stream: four 16-k chunks (128 bytes per lane and operand) are requested before they are consumed
	int k = kb;
	for (; k + 64 <= ke; k += 64) {
		d4_t b[4], a[TQ][4];
#pragma unroll
		for (int c = 0; c < 4; c++) b[c] = *reinterpret_cast<const d4_t *>(lp + k + 16 * c);
#pragma unroll
		for (int t = 0; t < TQ; t++)
#pragma unroll
			for (int c = 0; c < 4; c++) a[t][c] = *reinterpret_cast<const d4_t *>(ap + (long)t * 16 * ldk + k + 16 * c);
#pragma unroll
		for (int c = 0; c < 4; c++)
#pragma unroll
			for (int t = 0; t < TQ; t++)
#pragma unroll
				for (int u = 0; u < 4; u++)
					acc[t][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t][c][u], b[c][u], acc[t][u], 0, 0, 0);
	}
	for (; k < ke; k += 16) {
		const d4_t b = *reinterpret_cast<const d4_t *>(lp + k);
#pragma unroll
		for (int t = 0; t < TQ; t++) {
			const d4_t a = *reinterpret_cast<const d4_t *>(ap + (long)t * 16 * ldk + k);
#pragma unroll
			for (int u = 0; u < 4; u++) acc[t][u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc[t][u], 0, 0, 0);
		}
	}
	double *vp = Vp + (long)blockIdx.y * sstride + n0 + q;
#pragma unroll
	for (int t = 0; t < TQ; t++)
#pragma unroll
		for (int r = 0; r < 4; r++)
			vp[(long)(16 * t + g + 4 * r) * ldv] = (acc[t][0][r] + acc[t][1][r]) + (acc[t][2][r] + acc[t][3][r]);
}

// ---------------------------------------------------------------------------
// ONE query (emulate_point, the call an MCMC driver makes per sample): the same product as a matrix-VECTOR stream.  The skinny kernel above feeds the matrix unit and therefore reads 16 rows of L per wave-instruction, 64 bytes of
// each (2.9 TB/s at N = 8192); here a wave walks its 16 rows one after the other and a wave-instruction reads 2 KB of ONE
// row (32 contiguous bytes per lane), the k-vector values of the lane's columns sit in registers, every lane keeps a
// partial sum per row and query, and a butterfly over the 64 lanes ends the slice.  Same grid and output layout as the
// skinny kernel (row block x k-slice; Vp[slice][query][n]); MQ = queries held per lane (1).  fp64 FMAs on the vector unit:
// 2 flops per 8 bytes streamed, far below its rate.
// ---------------------------------------------------------------------------
template <int MQ>
__global__ __launch_bounds__(256) void gemv_tri_kernel(const double *Kq, long ldk, const double *L, long ldl, double *Vp,
                                                       long ldv, long sstride, int K, int ntri, int klen)
{
	const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
	const int n0 = blockIdx.x * 64 + 16 * wave;
	const int kb = blockIdx.y * klen;
	int ke = kb + klen < K ? kb + klen : K;
	if (n0 < ntri) {
		const int kx = (n0 + 16 + 15) & ~15;        // columns <= n0+15 of the triangular rows
		if (kx < ke) ke = kx;
	}
	double acc[16][MQ];
#pragma unroll
	for (int r = 0; r < 16; r++)
#pragma unroll
		for (int m = 0; m < MQ; m++) acc[r][m] = 0.0;
	const double *lrow = L + (long)n0 * ldl;
	for (int k = kb + 4 * lane; k < ke; k += 256) {
		d4_t kv[MQ];
#pragma unroll
		for (int m = 0; m < MQ; m++) kv[m] = *reinterpret_cast<const d4_t *>(Kq + (long)m * ldk + k);
#pragma unroll
		for (int h = 0; h < 2; h++) {              // eight rows' loads in flight at a time
			d4_t l[8];
#pragma unroll
			for (int r = 0; r < 8; r++) l[r] = *reinterpret_cast<const d4_t *>(lrow + (long)(8 * h + r) * ldl + k);
#pragma unroll
			for (int r = 0; r < 8; r++)
#pragma unroll
				for (int m = 0; m < MQ; m++) {
					double a = acc[8 * h + r][m];
					a = fma(l[r][0], kv[m][0], a);
					a = fma(l[r][1], kv[m][1], a);
					a = fma(l[r][2], kv[m][2], a);
					a = fma(l[r][3], kv[m][3], a);
					acc[8 * h + r][m] = a;
				}
		}
	}
#pragma unroll
	for (int r = 0; r < 16; r++)
#pragma unroll
		for (int m = 0; m < MQ; m++) {
			double a = acc[r][m];
			a += __shfl_xor(a, 32);
			a += __shfl_xor(a, 16);
			a += __shfl_xor(a, 8);
			a += __shfl_xor(a, 4);
			a += __shfl_xor(a, 2);
			a += __shfl_xor(a, 1);
			acc[r][m] = a;
		}
	if (lane == 0) {
		double *vp = Vp + (long)blockIdx.y * sstride + n0;
#pragma unroll
		for (int m = 0; m < MQ; m++)
#pragma unroll
			for (int r = 0; r < 16; r++) vp[(long)m * ldv + r] = acc[r][m];
	}
}

// ONE query (row 0 of Kq); otherwise as launch_skinny_nt.  (Four queries per lane need 256 VGPRs -- one wave per SIMD --:
// from two queries on the skinny kernel stays.)
hipError_t launch_gemv_tri(hipStream_t s, const double *Kq, long ldk, const double *L, long ldl, double *Vp, long ldv,
                           long sstride, int mq, int ntot, int K, int ntri, int nslice, int klen)
{
	const dim3 grid(ntot / 64, nslice);
	if (mq != 1) return hipErrorInvalidValue;
	hipLaunchKernelGGL(gemv_tri_kernel<1>, grid, dim3(256), 0, s, Kq, ldk, L, ldl, Vp, ldv, sstride, K, ntri, klen);
	return hipGetLastError();
}

// Vp[s][q][n] for q < 16*tq (tq = 1..4), n < ntot (multiple of 64), slices of klen (multiple of 16) over [0, K)
hipError_t launch_skinny_nt(hipStream_t s, const double *Kq, long ldk, const double *L, long ldl, double *Vp, long ldv,
                            long sstride, int tq, int ntot, int K, int ntri, int nslice, int klen)
{
	const dim3 grid(ntot / 64, nslice);
	if (tq != 1) return hipErrorInvalidValue;       // more query tiles measured slower than the split-K GEMM
	hipLaunchKernelGGL(skinny_nt_kernel<1>, grid, dim3(256), 0, s, Kq, ldk, L, ldl, Vp, ldv, sstride, K, ntri, klen);
	return hipGetLastError();
}

// active-tile count for a tile shape (tri skips the tiles strictly above the diagonal)
static long count_tiles(const GemmArgs &a, int BM, int BN)
{
	const int tiles_m = (a.m + BM - 1) / BM, tiles_n = (a.n + BN - 1) / BN;
	if (!a.tri) return (long)tiles_m * tiles_n;
	long c = 0;
	for (int tn = 0; tn < tiles_n; tn++) {
		// smallest tm with tn*BN <= tm*BM + BM-1 + diag_off
		long lo = ((long)tn * BN - a.diag_off - (BM - 1) + BM - 1) / BM;
		if ((long)tn * BN - a.diag_off - (BM - 1) <= 0) lo = 0;
		if (lo < tiles_m) c += tiles_m - lo;
	}
	return c;
}

// Tile shape per call (measured on MI355X, profiles/r01_gemm_tile_sweep.txt, r02_gemm_*.txt): with thousands of 128x128
// tiles the big shape wins (less LDS/L2 traffic per flop); below that the 64x64 shape is never slower (more workgroups
// for 256 CUs, 4-5 resident per CU) and up to 3x faster on the narrow K<=256 updates of the factorisation.
// The switches travel in the GemmArgs of each call (filled from the calling context's Sched): nothing process-wide.
//   a.force_cfg  test/bench hook: 2 = 64x64, 8 = 128x128, 0 = automatic
//   a.big_tiles  128x128 tiles (8 waves) once a lock-step launch has this many of them (twice as many for one matrix)
//   a.table_sb   XCD-blocked tile order from a table for launches of >= 512 tiles: side of the super-blocks (0: off)
int choose_gemm_cfg(const GemmArgs &a)
{
	if (a.force_cfg == 2 || a.force_cfg == 8) return a.force_cfg;
	const int big_tiles = a.big_tiles > 0 ? a.big_tiles : 1024;
	// one matrix per launch: twice the threshold -- 1000-2000 tiles on the 512 resident workgroups of the chip are 2-4
	// rounds, and the partly filled last one costs more than the faster tile gains (6.3 against 6.2 ms per evaluation)
	const long thr = (a.nbatch > 1 || big_tiles < 64) ? big_tiles : 2L * big_tiles;   // (< 64: test settings, taken literally)
	// updates narrower than two 128-column tiles stay on 64x64 tiles however many rows they have (the tall matrices of the
	// gradient path reach any tile count at n = 64): half of a 128-wide tile would be idle there, and only the 64x64
	// tiles carry the factor-ahead epilogue (value+gradient batch 158.4 -> 155.4 ms)
	if (a.n < 256 && big_tiles >= 64) return 2;
	return count_tiles(a, 128, 128) * (a.nbatch > 1 ? a.nbatch : 1) >= thr ? 8 : 2;
}

bool gemm_uses_big_tiles(const GemmArgs &a) { return choose_gemm_cfg(a) != 2; }

// would launch_gemm run this update with the factor-ahead tile?  (the caller then skips the next leaf factorisation)
bool gemm_factor_ahead_ok(const GemmArgs &a)
{
	return a.fa && a.tri && a.diag_off == 0 && a.m >= LEAF && a.n >= LEAF && a.beta == 1 && a.alpha == -1.0 && !a.kstart_mode &&
	       !a.kend_mode && a.ksplit <= 1 && choose_gemm_cfg(a) == 2;
}


// ---------------------------------------------------------------------------
// Tile order.  Workgroups are dealt round-robin to the 8 XCDs (ids b and b+8 share an XCD and its 4 MB L2).  In the
// natural order the 64-128 tiles an XCD works on at a time lie in one tile column: one B panel, but a different A
// panel each -- every tile fetches its own A panel from the fabric (PMC: 3.4 GB per evaluation against 1.2 GB
// compulsory).  The table lists the valid tiles super-block by super-block (S x S tiles of 128x128, S = GemmArgs.table_sb) and gives
// XCD x the x-th eighth of that list, so that its concurrent tiles share S A and S B panels and all XCDs get equal
// shares (unequal shares leave runs of idle slots at the tail: measured -6 %).  DESIGN.md section 7 has the numbers.
// Tables are built on first use outside stream capture and cached per (device, shape).
// ---------------------------------------------------------------------------
struct TileTable { int *dptr; int len; };
static std::map<std::tuple<int, int, int, int, int, int, int>, TileTable> g_tile_tables;
static std::mutex g_tile_mutex;

// host-only part (no HIP call: the CPU tests check it through gpemu_test_tile_table): entry w = q * 8 + x of the table is
// the q-th tile of XCD x, (tm << 16) | tn, or -1 in the unused tail slots of the shorter shares
std::vector<int> build_tile_table(int tiles_m, int tiles_n, int tri, int S, int bm, int bn)
{
	if (S < 1) S = 1;
	// super-blocks of about (128 S)^2 elements whatever the tile shape
	const int Sm = std::max(1, S * 128 / std::max(bm, 1)), Sn = std::max(1, S * 128 / std::max(bn, 1));
	const int sbm = (tiles_m + Sm - 1) / Sm, sbn = (tiles_n + Sn - 1) / Sn;
	// all valid tiles, super-block after super-block (column-major over the blocks and inside each); a tile of a
	// triangular update is valid when its first column is not beyond its last row
	std::vector<int> seq;
	for (int bc = 0; bc < sbn; bc++)
		for (int br = 0; br < sbm; br++)
			for (int c = 0; c < Sn; c++)
				for (int r = 0; r < Sm; r++) {
					const int tm = br * Sm + r, tn = bc * Sn + c;
					if (tm >= tiles_m || tn >= tiles_n || (tri && (long)tn * bn > (long)tm * bm + bm - 1)) continue;
					seq.push_back((tm << 16) | tn);
				}
	// XCD x walks the x-th eighth of that sequence: equal shares (the tail slots of the shorter ones hold -1), and
	// the tiles it has in flight lie in one or two super-blocks
	const size_t L = seq.size(), maxlen = (L + 7) / 8;
	std::vector<int> table(8 * maxlen, -1);
	for (size_t x = 0; x < 8; x++) {
		const size_t lo = x * L / 8, hi = (x + 1) * L / 8;
		for (size_t q = lo; q < hi; q++) table[(q - lo) * 8 + x] = seq[q];
	}
	return table;
}

// The square product with row-start skipping (C^-1 = U U^T of the gradient path: tile (r, c <= r) contracts over k >= the
// first column of tile row r, so the tiles of ONE tile row have one k-range and run in step, sharing their A panel chunk by
// chunk; no two tiles of different rows are ever at the same k).  Dealt round-robin in row-major order, a row's tiles sit
// on all eight XCDs and every tile streams both its panels through the fabric: 21.6 GB per launch of 16 matrices at
// N = 4096 for 2.2 GB of operands, 3.5 TB/s at 0.75 of the matrix peak (profiles/r05_corner_product_tile_table_ab.txt).
// This table gives whole tile rows to an XCD -- rows dealt to the XCDs longest-work-first onto the least loaded one, each
// XCD's rows in order of decreasing k-range -- so that a row's A panel crosses the fabric once.  (Round 5's first attempt
// gave each XCD a CONTIGUOUS eighth of the rows by work: the XCD with the short rows finished last.)
// entry q * 8 + x = the q-th tile of XCD x, -1 beyond its share.
std::vector<int> build_row_table(int tiles_m, int bm, int kstart_off, int k0, int k1)
{
	struct Row { int r; long work; };
	std::vector<Row> rows;
	for (int r = 0; r < tiles_m; r++) {
		int ks = (r * bm - kstart_off) & ~(GEMM_BK - 1);
		if (ks < k0) ks = k0;
		const long K = std::max(0, k1 - ks);
		rows.push_back(Row{r, (long)(r + 1) * (K + 6 * GEMM_BK)});          // (a tile's fixed cost counted as six k-steps)
	}
	std::sort(rows.begin(), rows.end(), [](const Row &a, const Row &b) { return a.work != b.work ? a.work > b.work : a.r < b.r; });
	long load[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	std::vector<int> mine[8];
	for (const Row &w : rows) {
		int x = 0;
		for (int y = 1; y < 8; y++)
			if (load[y] < load[x]) x = y;
		load[x] += w.work;
		mine[x].push_back(w.r);
	}
	size_t maxlen = 0;
	std::vector<int> seq[8];
	for (int x = 0; x < 8; x++) {
		std::sort(mine[x].begin(), mine[x].end());                             // decreasing k-range = increasing row
		for (int r : mine[x])
			for (int c = 0; c <= r; c++) seq[x].push_back((r << 16) | c);
		maxlen = std::max(maxlen, seq[x].size());
	}
	std::vector<int> table(8 * maxlen, -1);
	for (int x = 0; x < 8; x++)
		for (size_t q = 0; q < seq[x].size(); q++) table[q * 8 + x] = seq[x][q];
	return table;
}

// rowtab: the row table above (tiles_n = kstart_off, tri = 2 + k0, sb = k1 in the cache key)
static TileTable gemm_tile_table(hipStream_t s, int tiles_m, int tiles_n, int tri, int bm, int bn, int sb, bool rowtab = false)
{
	int dev = 0;
	(void)hipGetDevice(&dev);
	const auto key = std::make_tuple(dev, tiles_m, tiles_n, tri, sb, bm, rowtab ? -1 : bn);
	std::lock_guard<std::mutex> lock(g_tile_mutex);
	auto it = g_tile_tables.find(key);
	if (it != g_tile_tables.end()) return it->second;
	hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
	if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return TileTable{nullptr, 0};
	const std::vector<int> table = rowtab ? build_row_table(tiles_m, bm, tiles_n, tri - 2, sb) : build_tile_table(tiles_m, tiles_n, tri, sb, bm, bn);
	TileTable tt{nullptr, (int)table.size()};
	if (hipMalloc(&tt.dptr, table.size() * sizeof(int)) != hipSuccess ||
	    hipMemcpyAsync(tt.dptr, table.data(), table.size() * sizeof(int), hipMemcpyHostToDevice, s) != hipSuccess ||
	    hipStreamSynchronize(s) != hipSuccess) {       // (the caller's stream, never the legacy one: another thread may be capturing)
		(void)hipGetLastError();
		return TileTable{nullptr, 0};
	}
	g_tile_tables[key] = tt;
	return tt;
}

hipError_t launch_gemm(hipStream_t s, const GemmArgs &a_in)
{
	GemmArgs a = a_in;
	if (a.m <= 0 || a.n <= 0) return hipSuccess;
	if (a.beta && a.alpha != 1.0 && a.alpha != -1.0) return hipErrorInvalidValue;   // accumulators start from C/alpha
	if (a.force_cfg != 0 && a.force_cfg != 2 && a.force_cfg != 8) return hipErrorInvalidValue;
	if (a.ksplit > 1) {
		if (a.beta || a.nbatch > 1) return hipErrorInvalidValue;      // slices write fresh partials of one problem
		a.nbatch = a.ksplit; a.bsA = 0; a.bsB = 0;                      // bsC = stride between the partial outputs
	}
	const int nbatch = a.nbatch > 1 ? a.nbatch : 1;
	const int cfg = choose_gemm_cfg(a);                                   // 8: 128x128 tiles, 8 waves; 2: 64x64 tiles, 4 waves
	if (cfg != 2) a.fa = 0;                                               // factor-ahead lives in the 64x64 tiles only
	const int bm = cfg == 8 ? 128 : 64, bn = bm;
	const int tiles_m = (a.m + bm - 1) / bm, tiles_n = (a.n + bn - 1) / bn;
	// lower-triangular updates enumerate only their non-empty tiles: by a closed form, or by table
	const bool tri_ok = a.tri && a.diag_off == 0 && a.m >= a.n;
	a.order_mode = tri_ok ? 2 : 0;
	int T = tri_ok ? (int)count_tiles(a, bm, bn) : tiles_m * tiles_n;
	// (not for the triangular-operand products of the prediction path: their K differs from tile column to tile column, so
	// equal shares of tiles are unequal shares of work -- measured 2x slower -- and the long-K-first order matters more)
	if (a.table_sb > 0 && count_tiles(a, bm, bn) >= 512 && (tri_ok || !a.tri) && !a.kstart_mode && !a.kend_mode) {
		if (tiles_m < 32768 && tiles_n < 32768) {
			const TileTable tt = gemm_tile_table(s, tiles_m, tiles_n, tri_ok ? 1 : 0, bm, bn, a.table_sb);
			if (tt.dptr) { a.order_mode = 3; a.tile_table = tt.dptr; T = tt.len; }
		}
	}
	if (a.row_table && a.kstart_mode && !a.kend_mode && tri_ok && a.m == a.n && bm == bn && tiles_m >= 16 && tiles_m < 32768 && a.ksplit <= 1) {
		const TileTable tt = gemm_tile_table(s, tiles_m, a.kstart_off, 2 + a.k0, bm, bn, a.k1, true);
		if (tt.dptr) { a.order_mode = 3; a.tile_table = tt.dptr; T = tt.len; }
	}
	// (the first-round offset pays from the second round on: a launch that fits two rounds of 512 resident workgroups or fewer
	// would only start its odd slots late)
	if ((long)T * nbatch < 1024 || a.ksplit > 1) a.stagger_ticks = 0;
	// C - A B^T on top of C (every update of the factorisation): the instantiations whose matrix instruction negates A
	const bool neg = a.beta && a.alpha == -1.0 && !a.no_neg_modifier;
	if (neg) {
		if (cfg == 8) hipLaunchKernelGGL((gemm_nt_kernel<128, 128, 4, 4, 2, 0, 1>), dim3(T, nbatch), dim3(512), 0, s, a);
		else if (a.fa) hipLaunchKernelGGL((gemm_nt_kernel<64, 64, 4, 2, 2, 1, 1>), dim3(T, nbatch), dim3(256), 0, s, a);
		else hipLaunchKernelGGL((gemm_nt_kernel<64, 64, 4, 2, 2, 0, 1>), dim3(T, nbatch), dim3(256), 0, s, a);
	} else {
		if (cfg == 8) hipLaunchKernelGGL((gemm_nt_kernel<128, 128, 4, 4, 2>), dim3(T, nbatch), dim3(512), 0, s, a);
		else if (a.fa) hipLaunchKernelGGL((gemm_nt_kernel<64, 64, 4, 2, 2, 1>), dim3(T, nbatch), dim3(256), 0, s, a);
		else hipLaunchKernelGGL((gemm_nt_kernel<64, 64, 4, 2, 2>), dim3(T, nbatch), dim3(256), 0, s, a);
	}
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Leaf = two kernels on the 64x64 diagonal block at (c0,c0) of T:
//
//  leaf_factor_kernel (1 workgroup, 4 waves): the block sits in LDS and is
//    factored in four 16-column panels.  A panel is factored by wave 0, one
//    matrix row per lane, 16 VGPR pairs, SGPR (readlane) broadcasts, no LDS in
//    the pivot chain; a DEPENDENT fp64 VALU instruction of a lone wave issues every
//    ~10 cycles on gfx950 (scratch/mb/fma_rate.hip: 4.8 with enough independent
//    work, which a pivot chain does not have), so everything outside the
//    panel -- the rank-16 trailing updates -- runs on the MFMA across all four
//    waves.  1/sqrt is the hardware estimate plus two Newton steps.  A pivot
//    <= 0 (or NaN) records its 1-based global index in *info (atomicMin) --
//    GSL_EDOM of gsl_linalg_cholesky_decomp (maxmultimin.c:325-350).
//  leaf_solve_kernel (4 waves, 16 panel rows each): X L^T = B in place on the
//    fp64 MFMA.  L is staged in LDS, each wave first inverts one 16x16 diagonal
//    block (nilpotent series on the MFMA, tri_inverse16), then per row tile  X_j^T = Linv_jj (B_j^T - sum_{i<j} L_ji X_i^T):
//    the D registers of one MFMA are exactly the B operand of the next (k slot
//    of lane group g in step r is g+4r in both maps), so the chain never leaves
//    registers.
// (One fused kernel with the old row-per-lane solve made hipcc spill ~2000 VGPRs.)
// ---------------------------------------------------------------------------
__device__ __forceinline__ double bcast_lane(double v, int srclane)
{
	// wave-wide broadcast of lane `srclane` (compile-time constant after unrolling) through SGPRs
	int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
	int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
	return __hiloint2double(hi, lo);
}

// one 16-column panel (columns 16P..16P+15) of the 64x64 block in LDS, rows 16P..63, row per lane (wave 0)
template <int P, int LD>
__device__ __forceinline__ void panel_factor(double *A, int lane, int &bad, int bad_off)
{
	double a[16];
#pragma unroll
	for (int c = 0; c < 16; c += 2) {
		d2_t v = *reinterpret_cast<const d2_t *>(&A[lane * LD + 16 * P + c]);
		a[c] = v[0]; a[c + 1] = v[1];
	}
	// 1/sqrt(pivot): hardware estimate + two Newton steps -- a chain of ~10 dependent fp64 operations.  The pivot of
	// column k+1 is final as soon as column k has updated it, so its chain is started right there and runs beside the
	// remaining independent updates of column k instead of in front of column k+1.
	auto refined_rsqrt = [](double p) {
		double rs = __builtin_amdgcn_rsq(p);
		double t = p * rs; double e = fma(-t, rs, 1.0); rs = fma(rs * 0.5, e, rs);
		t = p * rs; e = fma(-t, rs, 1.0); rs = fma(rs * 0.5, e, rs);
		return rs;
	};
	double p = bcast_lane(a[0], 16 * P);
	if (!(p > 0.0) && bad == 0) bad = bad_off + 16 * P + 1;
	double rs = refined_rsqrt(p);
#pragma unroll
	for (int k = 0; k < 16; k++) {
		const double lik = (lane == 16 * P + k) ? p * rs : a[k] * rs;
		a[k] = lik;
		double p_next = 1.0, rs_next = 1.0;
		if (k + 1 < 16) {
			a[k + 1] = fma(-lik, bcast_lane(lik, 16 * P + k + 1), a[k + 1]);
			p_next = bcast_lane(a[k + 1], 16 * P + k + 1);
			if (!(p_next > 0.0) && bad == 0) bad = bad_off + 16 * P + k + 2;
			rs_next = refined_rsqrt(p_next);
		}
#pragma unroll
		for (int c = k + 2; c < 16; c++)
			a[c] = fma(-lik, bcast_lane(lik, 16 * P + c), a[c]);
		p = p_next;
		rs = rs_next;
	}
	if (lane >= 16 * P) {
#pragma unroll
		for (int c = 0; c < 16; c += 2) {
			d2_t v = {a[c], a[c + 1]};
			*reinterpret_cast<d2_t *>(&A[lane * LD + 16 * P + c]) = v;
		}
	}
}
// trailing update after panel P: tiles (ti,tj), P < tj <= ti <= 3, C -= Pan_ti Pan_tj^T on the MFMA
template <int P, int LD>
__device__ __forceinline__ void panel_update(double *A, int wave, int lane)
{
	const int g = lane >> 4, q = lane & 15;
	int t = 0;
#pragma unroll
	for (int ti = P + 1; ti < 4; ti++)
#pragma unroll
		for (int tj = P + 1; tj <= ti; tj++) {
			if ((t & 3) == wave) {
				d4_t c;
#pragma unroll
				for (int r = 0; r < 4; r++) c[r] = A[(16 * ti + g + 4 * r) * LD + 16 * tj + q];
#pragma unroll
				for (int r = 0; r < 4; r++) {
					const double a = -A[(16 * ti + q) * LD + 16 * P + g + 4 * r];
					const double b = A[(16 * tj + q) * LD + 16 * P + g + 4 * r];
					c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
				}
#pragma unroll
				for (int r = 0; r < 4; r++) A[(16 * ti + g + 4 * r) * LD + 16 * tj + q] = c[r];
			}
			t++;
		}
}

__global__ __launch_bounds__(256) void leaf_factor_kernel(double *T, long ld, int c0, int *info, unsigned long long *trace, long bstride)
{
	T += (long)blockIdx.y * bstride;     // lock-step batch: one diagonal block per matrix
	info += blockIdx.y;
	__shared__ double A[LEAF * LP];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	double *D = T + (long)c0 * ld + c0;
	const TraceT0 tr0 = trace_begin(trace);
	{
		double v[16];
#pragma unroll
		for (int u = 0; u < 16; u++) v[u] = D[(long)(wave + 4 * u) * ld + lane];
#pragma unroll
		for (int u = 0; u < 16; u++) A[(wave + 4 * u) * LP + lane] = v[u];
	}
	__syncthreads();
	if (trace && tr0.wall) atomicAdd(trace + 5, (unsigned long long)clock64() - tr0.clk);          // block staged in LDS
	int bad = 0;
	if (wave == 0) panel_factor<0, LP>(A, lane, bad, 0);
	__syncthreads();
	if (trace && tr0.wall) atomicAdd(trace + 6, (unsigned long long)clock64() - tr0.clk);          // first panel factored
	panel_update<0, LP>(A, wave, lane);
	__syncthreads();
	if (trace && tr0.wall) atomicAdd(trace + 7, (unsigned long long)clock64() - tr0.clk);          // first update done
	if (wave == 0) panel_factor<1, LP>(A, lane, bad, 0);
	else if (wave == 1) diag_inverse_ahead(A, 0, lane, D, ld);
	__syncthreads();
	panel_update<1, LP>(A, wave, lane);
	__syncthreads();
	if (wave == 0) panel_factor<2, LP>(A, lane, bad, 0);
	else if (wave == 2) diag_inverse_ahead(A, 1, lane, D, ld);
	__syncthreads();
	panel_update<2, LP>(A, wave, lane);
	__syncthreads();
	if (wave == 0) panel_factor<3, LP>(A, lane, bad, 0);
	else if (wave == 3) diag_inverse_ahead(A, 2, lane, D, ld);
	__syncthreads();
	if (wave == 1) diag_inverse_ahead(A, 3, lane, D, ld);
	if (tid == 0 && bad) atomicMin(info, c0 + bad);
#pragma unroll
	for (int u = 0; u < 16; u++) {
		const int r = wave + 4 * u;
		if (lane <= r) D[(long)r * ld + lane] = A[r * LP + lane];
	}
	trace_end(trace, tr0);
}

// 1/x: hardware estimate + two Newton steps (an fp64 divide costs ~10 more dependent VALU operations of ~10 cycles each)
__device__ __forceinline__ double fast_rcp(double x)
{
	double y = __builtin_amdgcn_rcp(x);
	double e = fma(-x, y, 1.0);
	y = fma(y, e, y);
	e = fma(-x, y, 1.0);
	return fma(y, e, y);
}

// inverse of the 16x16 lower-triangular diagonal block `o` of M (LDS, stride LP), written back over it.
// L = D (I + N), N strictly lower => (I+N)^-1 = (I - N)(I + N^2)(I + N^4)(I + N^8) exactly (N^16 = 0):
// five 16x16x16 products on the MFMA; the D registers of a product are the B operand of the next one,
// the A operand goes through a private 16x17 LDS tile.
template <int LD, int TS = 17>
__device__ __forceinline__ void tri_inverse16_to(const double *M, int o, double *tile, int lane, double *dst, int dst_ld)
{
	const int g = lane >> 4, q = lane & 15;
	// B/D-layout element (row g+4r, col q); A-layout element (row q, k g+4r)
	double dinv_row[4], dinv_q;
	dinv_q = fast_rcp(M[(o + q) * LD + o + q]);
#pragma unroll
	for (int r = 0; r < 4; r++) dinv_row[r] = fast_rcp(M[(o + g + 4 * r) * LD + o + g + 4 * r]);
	d4_t nB;      // N in B layout: N[row][col] = L[row][col]/L[row][row], row > col
	double nA[4]; // N in A layout: N[q][g+4r]
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const int row = g + 4 * r;
		nB[r] = (row > q) ? M[(o + row) * LD + o + q] * dinv_row[r] : 0.0;
		const int k = g + 4 * r;
		nA[r] = (q > k) ? M[(o + q) * LD + o + k] * dinv_q : 0.0;
	}
	// S = N*N
	d4_t S = {0, 0, 0, 0};
#pragma unroll
	for (int r = 0; r < 4; r++) S = __builtin_amdgcn_mfma_f64_16x16x4f64(nA[r], nB[r], S, 0, 0, 0);
	// Q = (I - N)(I + S)
	d4_t B1, Q = {0, 0, 0, 0};
#pragma unroll
	for (int r = 0; r < 4; r++) B1[r] = S[r] + ((g + 4 * r == q) ? 1.0 : 0.0);
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const double a = ((q == g + 4 * r) ? 1.0 : 0.0) - nA[r];
		Q = __builtin_amdgcn_mfma_f64_16x16x4f64(a, B1[r], Q, 0, 0, 0);
	}
	// two more doublings: S <- S*S ; Q <- Q (I + S)
#pragma unroll
	for (int it = 0; it < 2; it++) {
		double sA[4], qA[4];
#pragma unroll
		for (int r = 0; r < 4; r++) tile[(g + 4 * r) * TS + q] = S[r];
#pragma unroll
		for (int r = 0; r < 4; r++) sA[r] = tile[q * TS + g + 4 * r];
		d4_t S2 = {0, 0, 0, 0};
#pragma unroll
		for (int r = 0; r < 4; r++) S2 = __builtin_amdgcn_mfma_f64_16x16x4f64(sA[r], S[r], S2, 0, 0, 0);
#pragma unroll
		for (int r = 0; r < 4; r++) tile[(g + 4 * r) * TS + q] = Q[r];
#pragma unroll
		for (int r = 0; r < 4; r++) qA[r] = tile[q * TS + g + 4 * r];
		d4_t B2, Q2 = {0, 0, 0, 0};
#pragma unroll
		for (int r = 0; r < 4; r++) B2[r] = S2[r] + ((g + 4 * r == q) ? 1.0 : 0.0);
#pragma unroll
		for (int r = 0; r < 4; r++) Q2 = __builtin_amdgcn_mfma_f64_16x16x4f64(qA[r], B2[r], Q2, 0, 0, 0);
		S = S2;
		Q = Q2;
	}
	// Linv = (I+N)^-1 D^-1 : column q scaled by 1/L[q][q]; D layout element (row g+4r, col q)
#pragma unroll
	for (int r = 0; r < 4; r++) dst[(g + 4 * r) * dst_ld + q] = Q[r] * dinv_q;
}


// in place over the diagonal block; the scratch tile is a 16x16 block of M itself (row stride LP) that the solve never
// reads: keeps the kernel at 33 KB of LDS so that it fits beside three resident GEMM workgroups
__device__ __forceinline__ void tri_inverse16(double *M, int o, double *tile, int lane)
{
	tri_inverse16_to<LP, LP>(M, o, tile, lane, M + o * LP + o, LP);
}

// Round 5: the four diagonal inverses of a factored 64x64 block are computed ONCE, by the workgroup that factors it (a
// wave that would otherwise idle beside wave 0's next 16-column panel; only the last block's inverse is behind the last
// panel), and parked in the strictly upper 16x16 blocks (0,1) (1,2) (2,3) (0,3) of the diagonal block in HBM -- nothing
// reads the upper part of a factored diagonal block.  The leaf solve stages the whole 64x64 block anyway and finds them
// there: in a lock-step batch every one of its workgroups used to repeat the inversion (two barriers and five dependent
// matrix products before the first solve step: 17 % of the kernel, scratch/mb/leaf_variants.hip).  Same arithmetic on the
// same inputs: same bits.  The LDS scratch is the same upper block of A, which the factorisation never touches.
__device__ __forceinline__ int diag_inverse_block(int blk) { return blk == 3 ? 3 : (blk + 1) + 4 * blk; }   // (row, col) = (v >> 2, v & 3): (0,1) (1,2) (2,3) (0,3)
__device__ __forceinline__ void diag_inverse_ahead(double *A, int blk, int lane, double *Dg, long ldg)
{
	const int v = diag_inverse_block(blk), sr = v >> 2, sc = v & 3;
	tri_inverse16_to<LP, LP>(A, 16 * blk, A + 16 * sr * LP + 16 * sc, lane, Dg + (long)(16 * sr) * ldg + 16 * sc, (int)ldg);
}

// STAGED = false: lane (q, g) reads and writes its 16 elements of panel row q one double at a time (16 wave-instructions of
//   16 rows x 32 bytes each way) -- the shortest path for ONE matrix, where the kernel is latency-bound (5.5 us).
// STAGED = true (round 5): the wave's 16 x 64 tile crosses HBM in whole 512-byte row pieces (16 bytes per lane, two rows
//   per wave-instruction) and changes into the matrix unit's D/B layout through a wave-private LDS strip, LEAF_SROWS rows at
//   a time.  In a lock-step batch the kernel is bound by the read-modify-write of the block column, and the access SHAPE
//   sets the rate: scratch/mb/ld_stride.hip measures 3.5 TB/s for the 32-byte pieces against 5.2 TB/s for whole row pieces
//   at N = 4096, B = 64 (3.0 against 3.9 at N = 8192, B = 16); the kernel itself ran at 3.6 TB/s.  Same arithmetic, same
//   bits.  LDS: 33 KB for L + 4 x 4.25 KB strips = 50 KB, three workgroups per CU.
constexpr int LEAF_SROWS = 8;                 // rows of the tile in the strip at a time
constexpr int LEAF_SPITCH = 512 + 32;         // bytes between strip rows: 8 rows x 4 lane groups x 8 bytes hit 64 different banks
// DBG (scratch/mb/leaf_variants.hip only; the product instantiates 0): bit 0 skips the diagonal inverses, bit 1 the chain
// PRE = true: the diagonal inverses come with the staged block (diag_inverse_ahead); false: every workgroup computes them (A/B switch)
// coalesced pieces (lane: doubles ccol, ccol + 1 of tile rows 2 u + crow) -> the matrix unit's B/D layout (lane (q, g):
// row q, doubles 16 j + g + 4 r) through the wave's strip, LEAF_SROWS rows at a time
__device__ __forceinline__ void strip_rows_to_tile(char *strip, const d2_t (&in)[8], int lane, d4_t (&R)[4])
{
	// (the strip is written as 16-byte pieces and read as doubles: the empty asm statements keep the compiler from reordering them)
	const int g = lane >> 4, q = lane & 15, crow = lane >> 5, ccol = 2 * (lane & 31);
#pragma unroll
	for (int h = 0; h < 16 / LEAF_SROWS; h++) {
#pragma unroll
		for (int u = 0; u < LEAF_SROWS / 2; u++)
			*reinterpret_cast<d2_t *>(strip + (2 * u + crow) * LEAF_SPITCH + 8 * ccol) = in[h * (LEAF_SROWS / 2) + u];
		asm volatile("" ::: "memory");
		if ((q / LEAF_SROWS) == h) {
			const char *sp = strip + (q % LEAF_SROWS) * LEAF_SPITCH + 8 * g;
#pragma unroll
			for (int j = 0; j < 4; j++)
#pragma unroll
				for (int r = 0; r < 4; r++) R[j][r] = *reinterpret_cast<const double *>(sp + 8 * (16 * j + 4 * r));
		}
		asm volatile("" ::: "memory");
	}
}
// c0b >= 0: the LAST workgroup of the launch has another job -- the 64 rows under the diagonal block at c0b (the sub-diagonal
// block of a pair that leaf_pair_kernel left unsolved in place, see there)
template <bool STAGED, bool PRE = true, int DBG = 0>
__global__ __launch_bounds__(256) void leaf_solve_kernel(double *T, long ld, int c0, int m_below, unsigned long long *trace, long bstride, int c0b)
{
	T += (long)blockIdx.y * bstride;
	int bx = blockIdx.x;
	if (c0b >= 0 && bx == (int)gridDim.x - 1) { bx = 0; c0 = c0b; m_below = LEAF; }
	__shared__ double M[LEAF * LP];        // L; diagonal 16x16 blocks replaced by their inverses
	__shared__ __attribute__((aligned(16))) char strip_all[STAGED ? 4 * LEAF_SROWS * LEAF_SPITCH : 16];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int g = lane >> 4, q = lane & 15;
	const TraceT0 tr0 = trace_begin(trace);
	// this wave's 16 panel rows: all 16 values per lane requested up front (one memory latency, overlapped
	// with staging L and inverting the diagonal blocks)
	const int prow0 = (bx * 4 + wave) * 16;
	int prow = prow0 + q;
	const bool valid = prow < m_below;
	if (!valid) prow = m_below - 1;
	double *bp = T + (long)(c0 + LEAF + prow) * ld + c0;
	d4_t R[4];
	// STAGED: lane l moves doubles 2 (l & 31), + 1 of tile rows 2 u + (l >> 5), u = 0 .. 7
	d2_t in[8];
	char *strip = strip_all + (STAGED ? wave * LEAF_SROWS * LEAF_SPITCH : 0);
	const int crow = lane >> 5, ccol = 2 * (lane & 31);
	if (STAGED) {
#pragma unroll
		for (int u = 0; u < 8; u++) {
			int r = prow0 + 2 * u + crow;
			if (r > m_below - 1) r = m_below - 1;
			in[u] = *reinterpret_cast<const d2_t *>(T + (long)(c0 + LEAF + r) * ld + c0 + ccol);
		}
	} else {
#pragma unroll
		for (int j = 0; j < 4; j++)
#pragma unroll
			for (int r = 0; r < 4; r++) R[j][r] = bp[16 * j + g + 4 * r];
	}
	{
		const double *D = T + (long)c0 * ld + c0;
		double v[16];
#pragma unroll
		for (int u = 0; u < 16; u++) v[u] = D[(long)(wave + 4 * u) * ld + lane];
#pragma unroll
		for (int u = 0; u < 16; u++) M[(wave + 4 * u) * LP + lane] = v[u];
	}
	if (!PRE) {
		__syncthreads();
		// wave w inverts diagonal block w; scratch = an upper block of M: (0,1) (1,2) (2,3) (0,3)
		const int sr = (wave == 3) ? 0 : wave, sc = (wave == 3) ? 3 : wave + 1;
		if (!(DBG & 1)) tri_inverse16(M, 16 * wave, M + 16 * sr * LP + 16 * sc, lane);
	}
	// rows 0 .. 7 of the tile through the strip into the lanes with q < 8, then rows 8 .. 15 into the others (the strip is
	// this wave's alone and a wave's LDS operations execute in order: no barrier)
	if (STAGED) strip_rows_to_tile(strip, in, lane, R);
	__syncthreads();
	if (prow0 >= m_below) return;
	d4_t X[4];
#pragma unroll
	for (int j = 0; j < 4; j++) {
		d4_t acc = R[j];
		d4_t xj = {0.0, 0.0, 0.0, 0.0};
		if (DBG & 2) xj = acc * M[q * LP + g];
		else {
#pragma unroll
			for (int i = 0; i < j; i++)
#pragma unroll
				for (int r = 0; r < 4; r++) {
					const double a = -M[(16 * j + q) * LP + 16 * i + g + 4 * r];
					acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[i][r], acc, 0, 0, 0);
				}
#pragma unroll
			for (int r = 0; r < 4; r++) {
				constexpr int dv = PRE ? 1 : 0;           // PRE: block j's inverse sits in the upper block diag_inverse_block(j)
				const int ir = dv ? (j == 3 ? 0 : j) : j, ic = dv ? (j == 3 ? 3 : j + 1) : j;
				const double a = M[(16 * ir + q) * LP + 16 * ic + g + 4 * r];
				xj = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[r], xj, 0, 0, 0);
			}
		}
		X[j] = xj;
		if (!STAGED && valid) {
#pragma unroll
			for (int r = 0; r < 4; r++) bp[16 * j + g + 4 * r] = xj[r];
		}
	}
	if (STAGED) {
#pragma unroll
		for (int h = 0; h < 16 / LEAF_SROWS; h++) {
			if ((q / LEAF_SROWS) == h) {
				char *sp = strip + (q % LEAF_SROWS) * LEAF_SPITCH + 8 * g;
#pragma unroll
				for (int j = 0; j < 4; j++)
#pragma unroll
					for (int r = 0; r < 4; r++) *reinterpret_cast<double *>(sp + 8 * (16 * j + 4 * r)) = X[j][r];
			}
			asm volatile("" ::: "memory");
#pragma unroll
			for (int u = 0; u < LEAF_SROWS / 2; u++) {
				const d2_t v = *reinterpret_cast<const d2_t *>(strip + (2 * u + crow) * LEAF_SPITCH + 8 * ccol);
				const int r = prow0 + h * LEAF_SROWS + 2 * u + crow;
				if (r < m_below) *reinterpret_cast<d2_t *>(T + (long)(c0 + LEAF + r) * ld + c0 + ccol) = v;
			}
			asm volatile("" ::: "memory");
		}
	}
	trace_end(trace, tr0);
}


// ---------------------------------------------------------------------------
// leaf_pair_kernel (round 5): the FIRST block of a 128-column pair in one launch -- the leaf solve of columns [c0, c0+64)
// and the K=64 update of columns [c0+64, c0+128) for the same rows (with its factor-ahead tile) -- so that the solved
// block X1 goes from the solve to the update in registers / LDS instead of through HBM: 4 passes over 64-column pieces
// of the block column (read B1, write X1, read and write C2) where the two launches made 5, and one launch less on the
// chain of ONE matrix.  Workgroup b takes rows 64 b .. 64 b + 63 under the diagonal block (wave w: 16 of them).
// The update needs L21 = X1 of the FIRST 64 rows as its B operand: every workgroup solves those rows again for itself
// (wave w rows 16 w .. 16 w + 15; a second chain beside its own), which is only right while nobody has overwritten them --
// so workgroup 0 does NOT store its X1 here; the next launch of the stream (the leaf solve of the pair's second block)
// carries one extra workgroup that solves and stores those 64 rows in place (leaf_solve_kernel, c0b).  Nothing between
// the two reads that block: every later update takes its operand rows from below the pair.
// Bits: the solve is leaf_solve_kernel's chain; the update feeds the matrix unit what gemm_nt_kernel feeds it -- per
// accumulator the chunks of 16 in order, inside a chunk k = 8 t + 2 g + h for (t, h) = (0,0) (0,1) (1,0) (1,1) in lane
// group g, A negated by the instruction -- so the result equals the two launches' bit for bit
// (test_schedule_switches_keep_parity, GPEMU_LEAF_PAIR=0).
// LDS: 33 KB (L with the diagonal inverses; then the image of L21; then, in workgroup 0, the tile being factored) + four
// 4.25 KB strips.
// ---------------------------------------------------------------------------
// the leaf solve's chain for one 16-row tile: X_j^T = Linv_jj (B_j^T - sum_{i<j} L_ji X_i^T); M holds L with the inverses of
// its diagonal 16x16 blocks in the upper blocks diag_inverse_block(j)
__device__ __forceinline__ void leaf_chain(const d4_t (&R)[4], const double *M, int lane, d4_t (&X)[4])
{
	const int g = lane >> 4, q = lane & 15;
#pragma unroll
	for (int j = 0; j < 4; j++) {
		d4_t acc = R[j];
#pragma unroll
		for (int i = 0; i < j; i++)
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const double a = -M[(16 * j + q) * LP + 16 * i + g + 4 * r];
				acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, X[i][r], acc, 0, 0, 0);
			}
		d4_t xj = {0.0, 0.0, 0.0, 0.0};
		const int ir = j == 3 ? 0 : j, ic = j == 3 ? 3 : j + 1;
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const double a = M[(16 * ir + q) * LP + 16 * ic + g + 4 * r];
			xj = __builtin_amdgcn_mfma_f64_16x16x4f64(a, acc[r], xj, 0, 0, 0);
		}
		X[j] = xj;
	}
}

__global__ __launch_bounds__(256, 3) void leaf_pair_kernel(double *T, long ld, int c0, int m_below, int *info, unsigned long long *trace,
                                                        long bstride, int fa)
{
	T += (long)blockIdx.y * bstride;
	__shared__ __attribute__((aligned(16))) double M[LEAF * LP];
	__shared__ __attribute__((aligned(16))) char strip_all[4 * LEAF_SROWS * LEAF_SPITCH];
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	const int g = lane >> 4, q = lane & 15, crow = lane >> 5, ccol = 2 * (lane & 31);
	const int b = blockIdx.x;
	const TraceT0 tr0 = trace_begin(trace);
		double *rows = T + (long)(c0 + LEAF) * ld + c0;          // first row under the diagonal block, column c0
	// (m_below is a multiple of 64 -- launch_leaf_pair refuses anything else -- so every row of every workgroup exists; the
	// wave index through readfirstlane: row bases in scalar registers, ONE lane offset per access shape instead of a 64-bit
	// address per row -- the kernel has to stay under 168 vector registers for three workgroups per CU)
	const int uw = __builtin_amdgcn_readfirstlane(wave);
	char *strip = strip_all + uw * LEAF_SROWS * LEAF_SPITCH;
	const int prow0 = (b * 4 + uw) * 16;
	double *wrow = rows + (long)prow0 * ld;                  // the wave's first row (scalar)
	const unsigned voff = (unsigned)(crow * ld + ccol);      // lane offset of the coalesced shape: row crow of a row pair, doubles ccol, ccol + 1
	// everything this workgroup reads from HBM before the update, requested up front: its wave's share of the first 64
	// rows, its rows of block 1, L
	d2_t in[8], in0[8];
	if (b != 0) {
		const double *w0 = rows + (long)(16 * uw) * ld;
#pragma unroll
		for (int u = 0; u < 8; u++) in0[u] = *reinterpret_cast<const d2_t *>(w0 + (long)(2 * u) * ld + voff);
	}
#pragma unroll
	for (int u = 0; u < 8; u++) in[u] = *reinterpret_cast<const d2_t *>(wrow + (long)(2 * u) * ld + voff);
	// the wave's rows of block 2: the update's accumulators start from them.  Requested up front as well: a workgroup's
	// life under load is its memory latency, and what it has in flight while it computes is what keeps HBM busy
	d4_t acc[4];
	const unsigned coff = (unsigned)(g * ld + LEAF + q);    // lane offset of the accumulator shape: row g of a row quad, column 64 + q
#pragma unroll
	for (int r = 0; r < 4; r++)
#pragma unroll
		for (int j = 0; j < 4; j++) acc[j][r] = (wrow + (long)(4 * r) * ld + coff)[16 * j];
	{
		const double *D = T + (long)c0 * ld + c0;
		double v[16];
#pragma unroll
		for (int u = 0; u < 16; u++) v[u] = D[(long)(uw + 4 * u) * ld + lane];
#pragma unroll
		for (int u = 0; u < 16; u++) M[(uw + 4 * u) * LP + lane] = v[u];
	}
	__syncthreads();
	d4_t X[4], X0[4];
	if (b != 0) {
		d4_t R0[4];
		strip_rows_to_tile(strip, in0, lane, R0);
		leaf_chain(R0, M, lane, X0);
	}
	{
		d4_t R[4];
		strip_rows_to_tile(strip, in, lane, R);
		leaf_chain(R, M, lane, X);
	}
	if (b == 0) {
#pragma unroll
		for (int j = 0; j < 4; j++) X0[j] = X[j];
	}
	// X1 of the wave's rows: through the strip to HBM in whole row pieces (not workgroup 0, see above), and from the strip
	// into the update's A fragments: af[2 c + t] = doubles 16 c + 8 t + 2 g, + 1 of row q (chunk c of 16, half t)
	d2_t af[8];
#pragma unroll
	for (int h = 0; h < 16 / LEAF_SROWS; h++) {
		if ((q / LEAF_SROWS) == h) {
			char *sp = strip + (q % LEAF_SROWS) * LEAF_SPITCH + 8 * g;
#pragma unroll
			for (int j = 0; j < 4; j++)
#pragma unroll
				for (int r = 0; r < 4; r++) *reinterpret_cast<double *>(sp + 8 * (16 * j + 4 * r)) = X[j][r];
		}
		asm volatile("" ::: "memory");
#pragma unroll
		for (int u = 0; u < LEAF_SROWS / 2; u++) {
			const d2_t v = *reinterpret_cast<const d2_t *>(strip + (2 * u + crow) * LEAF_SPITCH + 8 * ccol);
			if (b != 0) *reinterpret_cast<d2_t *>(wrow + (long)(h * LEAF_SROWS + 2 * u) * ld + voff) = v;
		}
		if ((q / LEAF_SROWS) == h) {
			const char *sp = strip + (q % LEAF_SROWS) * LEAF_SPITCH + 16 * g;
#pragma unroll
			for (int c = 0; c < 4; c++)
#pragma unroll
				for (int t = 0; t < 2; t++) af[2 * c + t] = *reinterpret_cast<const d2_t *>(sp + 8 * (16 * c + 8 * t));
		}
		asm volatile("" ::: "memory");
	}
	__syncthreads();                       // every wave is past its last read of L
	// the image of L21 = X1 of the first 64 rows, row-major over M
#pragma unroll
	for (int j = 0; j < 4; j++)
#pragma unroll
		for (int r = 0; r < 4; r++) M[(16 * uw + q) * LP + 16 * j + g + 4 * r] = X0[j][r];
	__syncthreads();
	// C2 - X1 L21^T: B fragment of lane (q, g) for the 16 columns j: doubles 16 c + 8 t + 2 g, + 1 of row 16 j + q of L21
#pragma unroll
	for (int c = 0; c < 4; c++)
#pragma unroll
		for (int t = 0; t < 2; t++) {
			d2_t bf[4];
#pragma unroll
			for (int j = 0; j < 4; j++) bf[j] = *reinterpret_cast<const d2_t *>(&M[(16 * j + q) * LP + 16 * c + 8 * t + 2 * g]);
#pragma unroll
			for (int h = 0; h < 2; h++)
#pragma unroll
				for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[2 * c + t][h], bf[j][h], acc[j], 0, 0, 1);
		}
	if (b == 0 && fa) {
		// factor-ahead, as in gemm_nt_kernel: the updated diagonal block is factored here and leaves as L with its diagonal
		// inverses; its 1-based failed pivot goes to the matrix's info word
		__syncthreads();                   // the last read of the L21 image
		double *A = M;
#pragma unroll
		for (int j = 0; j < 4; j++)
#pragma unroll
			for (int r = 0; r < 4; r++) A[(16 * uw + g + 4 * r) * LP + 16 * j + q] = acc[j][r];
		__syncthreads();
		double *Dg = T + (long)(c0 + LEAF) * ld + c0 + LEAF;
		int bad = 0;
		if (wave == 0) panel_factor<0, LP>(A, lane, bad, 0);
		__syncthreads();
		panel_update<0, LP>(A, wave, lane);
		__syncthreads();
		if (wave == 0) panel_factor<1, LP>(A, lane, bad, 0);
		else if (wave == 1) diag_inverse_ahead(A, 0, lane, Dg, ld);
		__syncthreads();
		panel_update<1, LP>(A, wave, lane);
		__syncthreads();
		if (wave == 0) panel_factor<2, LP>(A, lane, bad, 0);
		else if (wave == 2) diag_inverse_ahead(A, 1, lane, Dg, ld);
		__syncthreads();
		panel_update<2, LP>(A, wave, lane);
		__syncthreads();
		if (wave == 0) panel_factor<3, LP>(A, lane, bad, 0);
		else if (wave == 3) diag_inverse_ahead(A, 2, lane, Dg, ld);
		__syncthreads();
		if (wave == 1) diag_inverse_ahead(A, 3, lane, Dg, ld);
		if (tid == 0 && bad) atomicMin(info + blockIdx.y, c0 + LEAF + bad);
#pragma unroll
		for (int u = 0; u < 16; u++) {
			const int r = wave + 4 * u;
			if (lane <= r) Dg[(long)r * ld + lane] = A[r * LP + lane];
		}
		trace_end(trace, tr0);
		return;
	}
#pragma unroll
	for (int r = 0; r < 4; r++)
#pragma unroll
		for (int j = 0; j < 4; j++) (wrow + (long)(4 * r) * ld + coff)[16 * j] = acc[j][r];
	trace_end(trace, tr0);
}

hipError_t launch_leaf_pair(hipStream_t s, double *T, long ld, int c0, int m_below, int *info, unsigned long long *tr, int nbatch,
                            long bstride, bool fa)
{
	if (nbatch < 1) nbatch = 1;
	if (m_below < LEAF || m_below % LEAF) return hipErrorInvalidValue;
	hipLaunchKernelGGL(leaf_pair_kernel, dim3(m_below / LEAF, nbatch), dim3(256), 0, s, T, ld, c0, m_below, info, tr, bstride, fa ? 1 : 0);
	return hipGetLastError();
}

hipError_t launch_leaf(hipStream_t s, double *T, long ld, int c0, int m_below, int *info, unsigned long long *trf,
                       unsigned long long *trs, int nbatch, long bstride, bool skip_factor, int staged, bool pre, int c0b)
{
	if (nbatch < 1) nbatch = 1;
	if (!skip_factor)                    // (skipped: the diagonal block was factored by the update before, factor-ahead)
		hipLaunchKernelGGL(leaf_factor_kernel, dim3(1, nbatch), dim3(256), 0, s, T, ld, c0, info, trf, bstride);
	if (m_below > 0) {
		const dim3 grid((m_below + 63) / 64 + (c0b >= 0 ? 1 : 0), nbatch);
		// staged rows pay once the launch is bound by the block column's traffic (a lock-step batch: more workgroups than
		// the chip holds at once); one matrix is latency-bound and keeps the direct form (staged < 0: automatic)
		const bool st = staged < 0 ? (long)grid.x * grid.y >= 1024 : staged != 0;
		if (st && pre) hipLaunchKernelGGL((leaf_solve_kernel<true, true>), grid, dim3(256), 0, s, T, ld, c0, m_below, trs, bstride, c0b);
		else if (st) hipLaunchKernelGGL((leaf_solve_kernel<true, false>), grid, dim3(256), 0, s, T, ld, c0, m_below, trs, bstride, c0b);
		else if (pre) hipLaunchKernelGGL((leaf_solve_kernel<false, true>), grid, dim3(256), 0, s, T, ld, c0, m_below, trs, bstride, c0b);
		else hipLaunchKernelGGL((leaf_solve_kernel<false, false>), grid, dim3(256), 0, s, T, ld, c0, m_below, trs, bstride, c0b);
	}
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Gram partials: part[blk][a][b] = sum_{j in 64-column chunk blk} Z[a][j] Z[b][j]
// Z rows are the solved right-hand sides (rows Np.. of T): Z = L^-1 [y|H].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gram_part_kernel(const double *Z, long ld, int nrhs, int Rp, double *part,
                                                        long zstride, long pstride)
{
	Z += (long)blockIdx.y * zstride;
	part += (long)blockIdx.y * pstride;
	__shared__ double zs[64 * 65];
	const int tid = threadIdx.x;
	const int c0 = blockIdx.x * 64;
	for (int e = tid; e < nrhs * 64; e += 256) {
		int a = e >> 6, c = e & 63;
		zs[a * 65 + c] = Z[(long)a * ld + c0 + c];
	}
	__syncthreads();
	double *out = part + (long)blockIdx.x * Rp * Rp;
	for (int p = tid; p < nrhs * nrhs; p += 256) {
		int a = p / nrhs, b = p % nrhs;
		double sum = 0.0;
		for (int c = 0; c < 64; c++) sum += zs[a * 65 + c] * zs[b * 65 + c];
		out[a * Rp + b] = sum;
	}
}

hipError_t launch_gram_partials(hipStream_t s, const double *Z, long ld, int Np, int nrhs, int Rp, double *part,
                                int nbatch, long zstride)
{
	if (nbatch < 1) nbatch = 1;
	hipLaunchKernelGGL(gram_part_kernel, dim3(Np / 64, nbatch), dim3(256), 0, s, Z, ld, nrhs, Rp, part, zstride,
	                   (long)(Np / 64) * Rp * Rp);
	return hipGetLastError();
}

// finish: fixed-order reduction of the Gram partials and 2*sum(log L_ii)
// (replaces det = (prod L_ii)^2 of maxmultimin.c:355-358, which under/overflows)
__global__ __launch_bounds__(256) void finish_kernel(const double *part, int nparts, int Rp, int nrhs,
                                                     const double *T, long ld, int N, double *res, long tstride,
                                                     long rstride)
{
	part += (long)blockIdx.x * nparts * Rp * Rp;
	T += (long)blockIdx.x * tstride;
	res += (long)blockIdx.x * rstride;
	__shared__ double red[256];
	const int tid = threadIdx.x;
	for (int p = tid; p < nrhs * nrhs; p += 256) {
		int a = p / nrhs, b = p % nrhs;
		double sum = 0.0;
		for (int k = 0; k < nparts; k++) sum += part[(long)k * Rp * Rp + a * Rp + b];
		res[a * Rp + b] = sum;
	}
	double lsum = 0.0;
	for (int i = tid; i < N; i += 256) lsum += log(T[(long)i * ld + i]);
	red[tid] = lsum;
	__syncthreads();
	for (int st = 128; st > 0; st >>= 1) {
		if (tid < st) red[tid] += red[tid + st];
		__syncthreads();
	}
	if (tid == 0) res[Rp * Rp] = 2.0 * red[0];
}

hipError_t launch_finish(hipStream_t s, const double *part, int nparts, int Rp, int nrhs, const double *T, long ld,
                         int N, double *res, int nbatch, long tstride, long rstride)
{
	if (nbatch < 1) nbatch = 1;
	hipLaunchKernelGGL(finish_kernel, dim3(nbatch), dim3(256), 0, s, part, nparts, Rp, nrhs, T, ld, N, res, tstride,
	                   rstride);
	return hipGetLastError();
}

} // namespace gpemu

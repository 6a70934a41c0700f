// kernels_cov.hip -- covariance-function kernels for gfx950 (MI355X).
//
// cov_fill_kernel restates, per element, the three covariance functions of the
// reference (libEmu/emulator.c:101-152 pow-exp, :344-386 Matern 3/2,
// :438-480 Matern 5/2) and replaces the N^2 indirect calls of
// makeCovMatrix_fnptr (:636-653) and the N calls + clamp of makeKVector_fnptr
// (:578-593).  HBM-write-bound: 64x64 output tiles, every wave writes whole
// 512-byte row segments; design rows are staged once per tile in LDS.
#include "gpemu_internal.hpp"
#include <cmath>
#include <mutex>
#include <set>
#include <vector>

namespace gpemu {

constexpr int FT = 64;            // fill tile edge
constexpr int DCH = 8;            // dimensions handled per register chunk

// What bounds the fill is its VECTOR INSTRUCTION COUNT per element, all kinds together (fp64 arithmetic, the integer work of
// the exponent and table index, selects, address arithmetic), not the 8 bytes it writes: an fp64 VALU instruction issues
// every ~4.8 cycles per SIMD when enough independent work is in flight, ~10 when it depends on its predecessor
// (scratch/mb/fma_rate.hip, profiles/r04_fp64_fma_issue_rate.txt: 58-65 TFLOP/s of FMAs with >= 4 independent chains or
// waves per SIMD, 30 with one -- rounds 1-3 assumed 16 cycles throughout), and the Matern 5/2 element of round 3 compiled to
// 33 vector instructions of which 17 were fp64 arithmetic.  Everything below is about fewer instructions: coordinates are
// pre-scaled while staging, exp and sqrt are short table / Newton forms, the exact "same point" test of the nugget rule
// runs only for pairs whose distance makes it possible, and the common path carries no select for it at all.

// exp(x) for x <= 0: x = (64 m + j) ln2/64 + r, |r| <= ln2/128; exp = 2^m * 2^(j/64) * P5(r) (truncation 4e-17), ~1 ulp.
// 11 fp64 VALU operations: the rounding to an integer is the 1.5 * 2^52 addition (the integer
// sits in the low mantissa word: no rint, no convert), the scaling by 2^m an integer add on the exponent field (no
// ldexp).  Arguments below -700 are treated as -700 (1e-304: the result stays a normal number for the exponent add).
constexpr int EXP_TAB = 64;
__device__ __forceinline__ double fast_exp_neg(double x, const double *tab /* LDS: 2^(j/64), j < 64 */)
{
	x = fmax(x, -700.0);
	const double magic = 6755399441055744.0;                              // 1.5 * 2^52
	const double t = fma(x, 92.332482616893656768, magic);                // 64 / ln 2
	const int ki = __double2loint(t);
	const double kf = t - magic;
	double r = fma(kf, -1.08304246932675596327e-02, x);                   // ln2_hi / 64 (32 significant bits: exact product)
	r = fma(kf, -2.98158582698529346878e-12, r);                          // ln2_lo / 64
	double p = fma(r, 1.0 / 120.0, 1.0 / 24.0);
	p = fma(p, r, 1.0 / 6.0);
	p = fma(p, r, 0.5);
	p = fma(p, r, 1.0);
	p = fma(p, r, 1.0);
	const double v = tab[ki & (EXP_TAB - 1)] * p;
	return __hiloint2double(__double2hiint(v) + ((ki >> 6) << 20), __double2loint(v));
}

// The Gram-form fill's exp: a 1024-entry table 2^(j/1024) (8 KB, filled once per device from the host, copied to LDS per
// workgroup ALREADY MULTIPLIED by the amplitude) leaves |r| <= ln2/2048, so a cubic suffices (truncation r^4/24 < 6e-16):
// 8 fp64 operations instead of 11 + the amplitude product.  No clamp: the Gram form runs only while the squared scaled
// distance is <= 64 + slack (make_cov_params), so the argument is > -150 and every intermediate stays a normal number.
constexpr int EXP_TAB_G = 1024;
__device__ double g_exp2_tab[EXP_TAB_G];

__device__ __forceinline__ double fast_exp_neg_g(double x, const double *tab_amp /* LDS: amp * 2^(j/1024) */)
{
	const double magic = 6755399441055744.0;                              // 1.5 * 2^52
	const double t = fma(x, 92.332482616893656768 * 16.0, magic);         // 1024 / ln 2
	const int ki = __double2loint(t);
	const double kf = t - magic;
	// ONE reduction step with ln2/1024 rounded to double: the product is exact inside the fma, the constant's own error
	// (7.5e-20) times |kf| <= 1024/ln2 * 70 is below 8e-15 absolute in r -- relative in the result -- for the arguments the
	// Gram form admits (|x| <= 64 + slack); the hi/lo pair of the general exp would buy digits nobody checks (bar: 1e-13)
	const double r = fma(kf, -0.69314718055994530942 / 1024.0, x);
	double p = fma(r, 1.0 / 6.0, 0.5);
	p = fma(p, r, 1.0);
	p = fma(p, r, 1.0);
	const double v = tab_amp[ki & (EXP_TAB_G - 1)] * p;
	return __hiloint2double(__double2hiint(v) + ((ki >> 10) << 20), __double2loint(v));
}

// sqrt(a), a >= 0, for the Gram form: rsq estimate y (5e-8, measured: scratch/mb/rsq_prec.hip), s = a y, eps = 1 - s y =
// 1 - a y^2, and sqrt(a) = s (1 - eps)^(-1/2) = s + s eps (1/2 + 3/8 eps) + O(eps^3): one third-order correction, 5 fp64
// operations behind the estimate instead of the 6 of two Heron steps, error below 2e-16
__device__ __forceinline__ double fast_sqrt_g(double a)
{
	// (a clamp instead of an `a > 0 ? u : 0` select -- one instruction for three: sqrt(1e-300) = 1e-150 gives exactly the
	// value of a zero distance, amp * 1; the Gram form's cancellation can leave a tiny negative number here)
	a = fmax(a, 1e-300);
	const double y = __builtin_amdgcn_rsq(a);
	const double s = a * y;
	const double eps = fma(-s, y, 1.0);
	const double q = fma(eps, 0.375, 0.5);
	return fma(s * eps, q, s);
}

// sqrt(a), a >= 0: hardware rsqrt estimate + two Heron corrections
__device__ __forceinline__ double fast_sqrt(double a)
{
	const double y = __builtin_amdgcn_rsq(a);
	double s = a * y;
	const double h = 0.5 * y;
	s = fma(fma(-s, s, a), h, s);
	s = fma(fma(-s, s, a), h, s);
	return (a > 0.0) ? s : 0.0;
}

// out[r][c] = cov(Xr[r], Xc[c]);  rows/cols beyond nr/nc (padding up to the
// launch grid) get identity (square factorisation matrix) or zero.
// p.w[k] is the per-dimension scale applied while staging: pow-exp sqrt(0.5)/r_k (so the exponent is
// -sum d'^2), Matern 1/rho (so sqrt(sum d'^2) is distance/rho).  p.eps is the reference's per-coordinate
// "same point" threshold on the UNSCALED coordinates; p.cand bounds sum d'^2 for pairs that can pass it.
// tile t of the lower triangle, row-major: t = tr(tr+1)/2 + tc
__device__ __forceinline__ void lower_tile(long t, int &tr, int &tc)
{
	int r = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
	while ((long)r * (r + 1) / 2 > t) r--;
	while ((long)(r + 1) * (r + 2) / 2 <= t) r++;
	tr = r;
	tc = (int)(t - (long)r * (r + 1) / 2);
}

// one FT x FT tile (tr, tc) of the fill
__device__ __forceinline__ void cov_fill_tile(double *out, long ld, const double *Xr, int nr, const double *Xc, int nc, int d,
                                              const CovParams &p, int mode, int tr, int tc)
{
	__shared__ double xr_s[FT * (GPEMU_MAX_PARAMS + 1)];
	__shared__ double tab[EXP_TAB];
	const int tid = threadIdx.x;
	const int sd = d + 1;
	for (int e = tid; e < FT * d; e += 256) {
		int r = e / d, k = e % d;
		int gr = tr * FT + r;
		xr_s[r * sd + k] = (gr < nr) ? Xr[(long)gr * d + k] * p.w[(p.kind == GPEMU_POWEREXP) ? k : 0] : 0.0;
	}
	if (tid < EXP_TAB) tab[tid] = exp2((double)tid * (1.0 / EXP_TAB));
	__syncthreads();

	const int col = tc * FT + (tid & 63);
	const int rsub = tid >> 6;                 // rows rsub, rsub+4, ...
	const bool colv = col < nc;

	double acc[16];
#pragma unroll
	for (int t = 0; t < 16; t++) acc[t] = 0.0;

	for (int k0 = 0; k0 < d; k0 += DCH) {
		double xc[DCH];
#pragma unroll
		for (int k = 0; k < DCH; k++) {
			const bool kv = (k0 + k) < d;
			xc[k] = (kv && colv) ? Xc[(long)col * d + k0 + k] * p.w[(p.kind == GPEMU_POWEREXP) ? (k0 + k) : 0] : 0.0;
		}
#pragma unroll
		for (int t = 0; t < 16; t++) {
			const double *xr = &xr_s[(rsub + 4 * t) * sd + k0];
#pragma unroll
			for (int k = 0; k < DCH; k++) {
				if (k0 + k < d) {
					const double diff = xr[k] - xc[k];
					acc[t] = fma(diff, diff, acc[t]);
				}
			}
		}
	}

#pragma unroll
	for (int t = 0; t < 16; t++) {
		const int row = tr * FT + rsub + 4 * t;
		double v;
		if (row < nr && colv) {
			const double a = acc[t];
			if (p.kind == GPEMU_POWEREXP) {
				v = fast_exp_neg(-a, tab) * p.amp;                                  // emulator.c:133,141
			} else {
				const double sdist = fast_sqrt(a);                                  // distance / rho
				if (p.kind == GPEMU_MATERN32) {
					const double root3 = 1.732050808;                               // emulator.c:359 (literal)
					v = p.amp * (1 + root3 * sdist) * fast_exp_neg(-root3 * sdist, tab);
				} else {
					const double root5 = 2.236067978;                               // emulator.c:452 (literal)
					v = p.amp * (1 + root5 * sdist + (5.0 / 3.0) * sdist * sdist) * fast_exp_neg(-root5 * sdist, tab);
				}
			}
			if (a <= p.cand) {
				// rare: the two points may coincide in every coordinate -> exact test on the raw coordinates
				// (emulator.c:136-150 / :368-384 / :462-478: nugget wherever ALL |x_k - y_k| < eps)
				int cnt = 0;
				for (int k = 0; k < d; k++)
					cnt += (fabs(Xr[(long)row * d + k] - Xc[(long)col * d + k]) < p.eps) ? 1 : 0;
				if (cnt == d) v += p.nug;
			}
			if ((mode & FILL_CLAMP) && v < 1E-10) v = 0.0;                          // emulator.c:588-590
		} else {
			v = ((mode & FILL_IDENT_PAD) && row == col) ? 1.0 : 0.0;
		}
		out[(long)row * ld + col] = v;
	}
}

// ---------------------------------------------------------------------------
// Gram form of the same tile for the square training matrix (Xr = Xc = the design): the squared scaled distances of
// a 64 x 64 tile come from the fp64 MFMA as  |x'|^2 + |y'|^2 - 2 x'.y'  -- the 2d subtract/FMA wave-instructions per
// element of the difference form (16 of its ~45 at d = 8) become d/4 + 1 matrix
// instructions per 256 elements on the other pipe.  x' = (x - mid) w: Xg holds the design centred per dimension
// (host, set_model), so |x'| is half the scaled range.  The host enables this form (p.gram) only while
// sum_k (w_k halfrange_k)^2 <= 16: the cancellation error in the squared distance, a few ulp of |x'|^2 + |y'|^2, then
// stays below 2e-14 absolute -- 2e-14 RELATIVE in a pow-exp element, less in a Matern one -- against the 1e-13 parity
// bar; beyond that (extreme length scales) the difference form above is used.  Pairs whose squared distance comes out
// below p.cand_g (the nugget rule's candidates plus that error) recompute it from coordinate differences and run the
// exact "same point" test on the raw coordinates (emulator.c:136-150 / :368-384 / :462-478).
// MFMA maps (16x16x4 f64): A lane (q, g) = A[row q][k g], B lane = B[k g][col q], D reg r = D[row g + 4r][col q].
// Wave w owns tile rows 16w .. 16w+15 and all four 16-column blocks.
// ---------------------------------------------------------------------------
typedef double d4g_t __attribute__((ext_vector_type(4)));

// value of an element from u2 = (c * distance / rho)^2 (Matern; c the literal root) or the pow-exp exponent, Gram form:
// the scaled coordinates already carry c, so u = sqrt(u2) is the exponent of the Matern kernels and their polynomial is
// 1 + u + c2 u^2 with c2 = (5/3) / c^2 (Matern 5/2: the reference's (5.0/3.0) s^2, emulator.c:470, in terms of u = c s)
template <int KIND>
__device__ __forceinline__ double cov_from_u2_gram(double a, const double *tab_amp)
{
	if (KIND == GPEMU_POWEREXP) return fast_exp_neg_g(-a, tab_amp);                       // emulator.c:133,141 (amp in the table)
	const double u = fast_sqrt_g(a);
	const double e = fast_exp_neg_g(-u, tab_amp);
	if (KIND == GPEMU_MATERN32) return e * (1.0 + u);                                     // emulator.c:359-376
	const double c2 = (5.0 / 3.0) / (2.236067978 * 2.236067978);
	return e * fma(u, fma(u, c2, 1.0), 1.0);                                              // emulator.c:452-470
}

// the Matern kernels' literal root (emulator.c:359, 452) rides in the coordinate scale: the MFMA then delivers
// (c distance / rho)^2 and no element multiplies by c
__device__ __forceinline__ double gram_root(int kind)
{
	return kind == GPEMU_POWEREXP ? 1.0 : (kind == GPEMU_MATERN32 ? 1.732050808 : 2.236067978);
}

// per-workgroup tables of the Gram form (LDS): amp * 2^(j/1024) and the per-dimension coordinate scales (root included)
__device__ __forceinline__ void gram_tables(const CovParams &p, int d, double *tab, double *wsc)
{
	const int tid = threadIdx.x;
	const double croot = gram_root(p.kind);
	for (int e = tid; e < EXP_TAB_G; e += 256) tab[e] = g_exp2_tab[e] * p.amp;
	if (tid < GPEMU_MAX_PARAMS) wsc[tid] = (tid < d) ? p.w[(p.kind == GPEMU_POWEREXP) ? tid : 0] * croot : 0.0;
}

// One tile.  The element loop has no data-dependent branch on its common path: the squared distances of the lane's 16
// elements come out of the MFMA accumulators; only if some lane of the WAVE holds a nugget-rule candidate (a squared
// distance below cand_g: coinciding design points, the diagonal) does the wave enter the exact re-computation; only
// edge tiles (rows / columns beyond n) take the bounds-checked store.
// RECT = false: the square training matrix (rows and columns are the design: Xa = Xb = X, Xag = Xbg = the centred design).
// RECT = true: k-vectors (makeKVector_fnptr, emulator.c:578-593): rows are query points, centred on the fly with the
// design's per-dimension centre `mid`; a wave whose query rows lie so far outside the design that the Gram form's
// cancellation bound no longer holds (|x'|^2 > 16) recomputes its elements from differences (the candidate path below).
template <int KIND, bool RECT = false>
__device__ __forceinline__ void cov_fill_tile_gram_k(double *out, long ld, const double *Xa, const double *Xag, const double *mid,
                                                     int na_rows, const double *Xb, const double *Xbg, int nb_cols, int d,
                                                     const CovParams &p, int mode, int tr, int tc, const double *tab, const double *wsc)
{
	const int tid = threadIdx.x;
	const double croot = gram_root(KIND);
	double cand_g = p.cand_g * croot * croot;
	const int lane = tid & 63, wave = tid >> 6;
	const int q = lane & 15, g = lane >> 4;
	const int arow = tr * FT + 16 * wave + q;                 // A operand row of this lane
	const bool av = arow < na_rows;
	const double *ap = (RECT ? Xa : Xag) + (long)(av ? arow : 0) * d;
	const double *bp[4];
	bool bv[4];
#pragma unroll
	for (int j = 0; j < 4; j++) {
		const int bc = tc * FT + 16 * j + q;
		bv[j] = bc < nb_cols;
		bp[j] = Xbg + (long)(bv[j] ? bc : 0) * d;
	}
	d4g_t acc[4];
#pragma unroll
	for (int j = 0; j < 4; j++) acc[j] = (d4g_t){0.0, 0.0, 0.0, 0.0};
	double na = 0.0, nb[4] = {0.0, 0.0, 0.0, 0.0};
	for (int k0 = 0; k0 < d; k0 += 4) {
		const int k = k0 + g;
		const bool kv = k < d;
		const double wk = wsc[kv ? k : 0];
		const double xa = (kv && av) ? (RECT ? (ap[k] - mid[k]) * wk : ap[k] * wk) : 0.0;
		na = fma(xa, xa, na);
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const double xb = (kv && bv[j]) ? bp[j][k] * wk : 0.0;
			nb[j] = fma(xb, xb, nb[j]);
			acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, -2.0 * xb, acc[j], 0, 0, 0);
		}
	}
	// |x'|^2 of the lane's A row / B column: sum the four k-groups (lanes q, q+16, q+32, q+48)
	na += __shfl_xor(na, 16); na += __shfl_xor(na, 32);
#pragma unroll
	for (int j = 0; j < 4; j++) { nb[j] += __shfl_xor(nb[j], 16); nb[j] += __shfl_xor(nb[j], 32); }
	if (RECT && __any(na > 16.0 * croot * croot)) cand_g = HUGE_VAL;      // query rows far outside the design: exact distances
	// one more matrix step adds |x'_row|^2 + |y'_col|^2: k slot 0 = (|x'|^2, 1), k slot 1 = (1, |y'|^2)
	{
		const double ea = (g == 0) ? na : (g == 1 ? 1.0 : 0.0);
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const double eb = (g == 0) ? 1.0 : (g == 1 ? nb[j] : 0.0);
			acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ea, eb, acc[j], 0, 0, 0);
		}
	}
	const int row0 = tr * FT + 16 * wave + g, col0 = tc * FT + q;
	// nugget-rule candidates (rare): distance from differences, exact "same point" test on the raw coordinates
	// (emulator.c:136-150 / :368-384 / :462-478)
	unsigned same = 0;
	bool cand = false;
#pragma unroll
	for (int r = 0; r < 4; r++)
#pragma unroll
		for (int j = 0; j < 4; j++) cand = cand || (acc[j][r] <= cand_g);
	if (__any(cand)) {
		// (one copy of the slow code: the element index is a run-time value here, the accumulators go through LDS-free
		// register selects -- this runs for the diagonal tiles and for duplicated design points only)
#pragma unroll 1
		for (int e = 0; e < 16; e++) {
			const int r = e >> 2, j = e & 3;
			double ae = 0.0;
#pragma unroll
			for (int rr = 0; rr < 4; rr++)
#pragma unroll
				for (int jj = 0; jj < 4; jj++) ae = (rr == r && jj == j) ? acc[jj][rr] : ae;
			{
				const int row = row0 + 4 * r, col = col0 + 16 * j;
				if (ae <= cand_g && row < na_rows && col < nb_cols) {
					int cnt = 0;
					double a = 0.0;
					for (int k = 0; k < d; k++) {
						const double D = Xa[(long)row * d + k] - Xb[(long)col * d + k];
						const double t = D * wsc[k];
						a = fma(t, t, a);
						cnt += (fabs(D) < p.eps) ? 1 : 0;
					}
					// the table exp below has no clamp of its own (the Gram form proper keeps its argument above -70); a query
					// row far outside the design arrives HERE with a squared scaled distance of any size (round 5: coordinates
					// of 30 at d = 16 gave exp(-23 000) = NaN): hold the exponent at -700 (1e-304, which the k-vector clamp of
					// emulator.c:588-590 turns into the zero the reference computes)
					a = fmin(a, KIND == GPEMU_POWEREXP ? 700.0 : 490000.0);
#pragma unroll
					for (int rr = 0; rr < 4; rr++)
#pragma unroll
						for (int jj = 0; jj < 4; jj++) acc[jj][rr] = (rr == r && jj == j) ? a : acc[jj][rr];
					if (cnt == d) same |= 1u << e;
				}
			}
		}
	}
	const bool full = (tr * FT + FT <= na_rows) && (tc * FT + FT <= nb_cols);
	if (full && (RECT || !(mode & FILL_CLAMP)) && !__any(same != 0)) {
		// the common tile: no coinciding points anywhere in the wave -- no nugget select in the element loop (it cost 7 of
		// the 33 vector instructions per element: mask test, both variants of the value, two selects)
#pragma unroll
		for (int r = 0; r < 4; r++) {
			double *orow = out + (long)(row0 + 4 * r) * ld + col0;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				double v = cov_from_u2_gram<KIND>(acc[j][r], tab);
				if (RECT && v < 1E-10) v = 0.0;                                           // emulator.c:588-590 (k-vectors are always clamped)
				orow[16 * j] = v;
				if (j & 1) __builtin_amdgcn_sched_barrier(0);
			}
		}
		return;
	}
	if (full && (RECT || !(mode & FILL_CLAMP))) {
		const double nug = p.nug;
#pragma unroll
		for (int r = 0; r < 4; r++) {
			double *orow = out + (long)(row0 + 4 * r) * ld + col0;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				double v = cov_from_u2_gram<KIND>(acc[j][r], tab);
				if (same & (1u << (4 * r + j))) v += nug;
				if (RECT && v < 1E-10) v = 0.0;                                           // emulator.c:588-590 (k-vectors are always clamped)
				orow[16 * j] = v;
				// two elements at a time: a dependent fp64 operation can issue as soon as its predecessor has gone through the
				// pipe, so with six waves per SIMD interleaving more chains per wave buys nothing and costs registers -- sixteen
				// interleaved exp/sqrt chains need 256 VGPRs, two fit 80 and leave six waves per SIMD to cover the loads
				if (j & 1) __builtin_amdgcn_sched_barrier(0);
			}
		}
		return;
	}
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const int row = row0 + 4 * r;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const int col = col0 + 16 * j;
			double v;
			if (row < na_rows && col < nb_cols) {
				v = cov_from_u2_gram<KIND>(acc[j][r], tab);
				if (same & (1u << (4 * r + j))) v += p.nug;
				if ((RECT || (mode & FILL_CLAMP)) && v < 1E-10) v = 0.0;                  // emulator.c:588-590
			} else {
				v = ((mode & FILL_IDENT_PAD) && row == col) ? 1.0 : 0.0;
			}
			out[(long)row * ld + col] = v;
		}
	}
}

__global__ __launch_bounds__(256) void cov_fill_kernel(double *out, long ld, const double *Xr, int nr,
                                                       const double *Xc, int nc, int d, CovParams p, int mode)
{
	int tr = blockIdx.y, tc = blockIdx.x;
	// FILL_LOWER: 1-D grid over the lower-triangular tiles only: empty workgroups are not free, the dispatcher deals
	// them in order like any other
	if (mode & FILL_LOWER) lower_tile(blockIdx.x, tr, tc);
	cov_fill_tile(out, ld, Xr, nr, Xc, nc, d, p, mode, tr, tc);
}

// Staging of a lock-step batch in ONE launch: matrix blockIdx.y gets its lower-triangular fill with its own
// hyper-parameters pp[blockIdx.y] (workgroups 0 .. ntiles-1) and a copy of the shared R rows [y H]^T below it
// (workgroups ntiles ..: one FT x FT block each; with rstride != 0 matrix b takes ITS OWN rows Rrows + b * rstride: the
// components of a multi-output model share the design and differ in the training vector).  Per-matrix launches and copies cost ~12 us each on the host and the
// stream: 1 ms per batch of 64, which is all a batch of small models (N < 1000) takes.
template <int KIND>
__global__ __launch_bounds__(256) void cov_stage_batch_kernel(double *T, long ld, long bstride, const double *X, int N, int Np,
                                                              int d, const CovParams *pp, int mode, const double *Rrows, int Rp,
                                                              const double *Xg, long rstride)
{
	double *out = T + (long)blockIdx.y * bstride;
	Rrows += (long)blockIdx.y * rstride;              // rstride 0: every matrix gets the same right-hand sides
	const long nt = Np / FT, ntiles = nt * (nt + 1) / 2;
	if ((long)blockIdx.x < ntiles) {
		// this matrix's hyper-parameters through LDS (the per-dimension scales are indexed per lane)
		__shared__ CovParams ps;
		static_assert(sizeof(CovParams) % sizeof(double) == 0, "CovParams is copied as doubles");
		for (int e = threadIdx.x; e < (int)(sizeof(CovParams) / sizeof(double)); e += 256)
			reinterpret_cast<double *>(&ps)[e] = reinterpret_cast<const double *>(pp + blockIdx.y)[e];
		__syncthreads();
		int tr, tc;
		lower_tile(blockIdx.x, tr, tc);
		if (ps.gram && Xg) {
			__shared__ double tab[EXP_TAB_G];
			__shared__ double wsc[GPEMU_MAX_PARAMS];
			gram_tables(ps, d, tab, wsc);
			__syncthreads();
			cov_fill_tile_gram_k<KIND>(out, ld, X, Xg, nullptr, N, X, Xg, N, d, ps, mode, tr, tc, tab, wsc);
		} else cov_fill_tile(out, ld, X, N, X, N, d, ps, mode, tr, tc);
		return;
	}
	const long rb = blockIdx.x - ntiles;              // block (rb / nt, rb % nt) of the Rp x Np rows
	const int r0 = (int)(rb / nt) * FT, c0 = (int)(rb % nt) * FT;
	double *dst = out + (long)Np * ld;
	for (int e = threadIdx.x; e < FT * FT; e += 256) {
		const int r = r0 + (e >> 6), c = c0 + (e & 63);
		if (r < Rp) dst[(long)r * ld + c] = Rrows[(long)r * Np + c];
	}
}

// The same staging when EVERY matrix of the batch takes the Gram form (the usual case: make_cov_params): no difference-
// form code and none of its 33 KB LDS tile in the kernel, so five workgroups fit a CU instead of three, and a workgroup
// fills GRAM_TPW tiles with one set of tables.  Workgroups beyond the tiles copy the shared R rows as above.
// waves per SIMD the staging kernel is compiled for (pow-exp, Matern): the fill's phases -- operands, distances on the matrix
// unit, sixteen element chains, stores -- run one behind the other inside a wave, and what overlaps them is the other waves of
// the SIMD.  Left to itself the compiler took 152 / 116 registers (three / four waves); bounded to four / five (120 / 91
// registers, no spill) the fill went from 66.6 / 67.6 to 58.6 / 62.8 us per N=8192 matrix; five / six spill and lose
// (77 / 72 us).  profiles/r05_fill_what_bounds_it.txt
#ifndef GPEMU_FILL_WAVES_PE
#define GPEMU_FILL_WAVES_PE 4
#endif
#ifndef GPEMU_FILL_WAVES_MT
#define GPEMU_FILL_WAVES_MT 5
#endif
constexpr int GRAM_TPW = 4;
template <int KIND>
__global__ __launch_bounds__(256, KIND == GPEMU_POWEREXP ? GPEMU_FILL_WAVES_PE : GPEMU_FILL_WAVES_MT) void cov_stage_gram_kernel(double *T, long ld, long bstride, const double *X, int N, int Np,
                                                             int d, const CovParams *pp, int mode, const double *Rrows, int Rp,
                                                             const double *Xg, long rstride)
{
	double *out = T + (long)blockIdx.y * bstride;
	Rrows += (long)blockIdx.y * rstride;              // rstride 0: every matrix gets the same right-hand sides
	const long nt = Np / FT, ntiles = nt * (nt + 1) / 2, ngroups = (ntiles + GRAM_TPW - 1) / GRAM_TPW;
	if ((long)blockIdx.x < ngroups) {
		__shared__ CovParams ps;
		__shared__ double tab[EXP_TAB_G];
		__shared__ double wsc[GPEMU_MAX_PARAMS];
		for (int e = threadIdx.x; e < (int)(sizeof(CovParams) / sizeof(double)); e += 256)
			reinterpret_cast<double *>(&ps)[e] = reinterpret_cast<const double *>(pp + blockIdx.y)[e];
		__syncthreads();
		gram_tables(ps, d, tab, wsc);
		__syncthreads();
		// the groups walk the tile list from its END: the last tile rows (the longest ones) first
		for (int i = 0; i < GRAM_TPW; i++) {
			const long t = ntiles - 1 - ((long)blockIdx.x * GRAM_TPW + i);
			if (t < 0) break;
			int tr, tc;
			lower_tile(t, tr, tc);
			cov_fill_tile_gram_k<KIND>(out, ld, X, Xg, nullptr, N, X, Xg, N, d, ps, mode, tr, tc, tab, wsc);
		}
		return;
	}
	const long rb = blockIdx.x - ngroups;             // block (rb / nt, rb % nt) of the Rp x Np rows
	const int r0 = (int)(rb / nt) * FT, c0 = (int)(rb % nt) * FT;
	double *dst = out + (long)Np * ld;
	for (int e = threadIdx.x; e < FT * FT; e += 256) {
		const int r = r0 + (e >> 6), c = c0 + (e & 63);
		if (r < Rp) dst[(long)r * ld + c] = Rrows[(long)r * Np + c];
	}
}

// the 2^(j/1024) table of the Gram-form exp: written once per device from host-computed values (exp2 of glibc)
// (copied on the caller's stream: a legacy-stream copy fails while another host thread is recording a launch graph)
static hipError_t ensure_exp_table(hipStream_t s)
{
	static std::mutex mu;
	static std::set<int> done;
	int dev = 0;
	hipError_t e = hipGetDevice(&dev);
	if (e != hipSuccess) return e;
	std::lock_guard<std::mutex> lock(mu);
	if (done.count(dev)) return hipSuccess;
	std::vector<double> h(EXP_TAB_G);
	for (int j = 0; j < EXP_TAB_G; j++) h[j] = std::exp2((double)j / EXP_TAB_G);
	e = hipMemcpyToSymbolAsync(HIP_SYMBOL(g_exp2_tab), h.data(), sizeof(double) * EXP_TAB_G, 0, hipMemcpyHostToDevice, s);
	if (e == hipSuccess) e = hipStreamSynchronize(s);
	if (e == hipSuccess) done.insert(dev);
	return e;
}

// (every matrix of a batch has the model's covariance function: `kind` selects the instantiation)
hipError_t launch_cov_stage_batch(hipStream_t s, double *T, long ld, long bstride, int nb, const double *X, int N, int Np, int d,
                                  const CovParams *pp_dev, int mode, const double *Rrows, int Rp, const double *Xg, bool all_gram,
                                  int kind, long rstride)
{
	if (Np % FT || FT != 64) return hipErrorInvalidValue;
	if (kind < GPEMU_POWEREXP || kind > GPEMU_MATERN52) return hipErrorInvalidValue;
	if (Xg) {
		const hipError_t e = ensure_exp_table(s);
		if (e != hipSuccess) return e;
	}
	const long nt = Np / FT;
	if (Xg && all_gram) {
		const long ntiles = nt * (nt + 1) / 2;
		const long blocks = (ntiles + GRAM_TPW - 1) / GRAM_TPW + ((Rp + FT - 1) / FT) * nt;
		const dim3 grid((unsigned)blocks, nb);
		if (kind == GPEMU_POWEREXP)
			hipLaunchKernelGGL(cov_stage_gram_kernel<GPEMU_POWEREXP>, grid, dim3(256), 0, s, T, ld, bstride, X, N, Np, d, pp_dev, mode, Rrows, Rp, Xg, rstride);
		else if (kind == GPEMU_MATERN32)
			hipLaunchKernelGGL(cov_stage_gram_kernel<GPEMU_MATERN32>, grid, dim3(256), 0, s, T, ld, bstride, X, N, Np, d, pp_dev, mode, Rrows, Rp, Xg, rstride);
		else
			hipLaunchKernelGGL(cov_stage_gram_kernel<GPEMU_MATERN52>, grid, dim3(256), 0, s, T, ld, bstride, X, N, Np, d, pp_dev, mode, Rrows, Rp, Xg, rstride);
		return hipGetLastError();
	}
	const long blocks = nt * (nt + 1) / 2 + ((Rp + FT - 1) / FT) * nt;
	const dim3 grid((unsigned)blocks, nb);
	if (kind == GPEMU_POWEREXP)
		hipLaunchKernelGGL(cov_stage_batch_kernel<GPEMU_POWEREXP>, grid, dim3(256), 0, s, T, ld, bstride, X, N, Np, d, pp_dev, mode, Rrows, Rp, Xg, rstride);
	else if (kind == GPEMU_MATERN32)
		hipLaunchKernelGGL(cov_stage_batch_kernel<GPEMU_MATERN32>, grid, dim3(256), 0, s, T, ld, bstride, X, N, Np, d, pp_dev, mode, Rrows, Rp, Xg, rstride);
	else
		hipLaunchKernelGGL(cov_stage_batch_kernel<GPEMU_MATERN52>, grid, dim3(256), 0, s, T, ld, bstride, X, N, Np, d, pp_dev, mode, Rrows, Rp, Xg, rstride);
	return hipGetLastError();
}

hipError_t launch_cov_fill(hipStream_t s, double *out, long ld, const double *Xr, int nr, int nr_pad,
                           const double *Xc, int nc, int nc_pad, int d, const CovParams &p, int mode)
{
	dim3 grid(nc_pad / FT, nr_pad / FT);
	if (mode & FILL_LOWER) {
		if (nr_pad != nc_pad) return hipErrorInvalidValue;      // lower-triangle fill is for the square matrix
		const long nt = nr_pad / FT;
		grid = dim3((unsigned)(nt * (nt + 1) / 2), 1);
	}
	hipLaunchKernelGGL(cov_fill_kernel, grid, dim3(256), 0, s, out, ld, Xr, nr, Xc, nc, d, p, mode);
	return hipGetLastError();
}

// k-vectors of a block of query points in Gram form (makeKVector_fnptr, emulator.c:578-593, clamp included): out[q][i] =
// cov(xq_q, x_i), rows q >= M and columns i >= N of the padded block are zero.  One workgroup fills GRAM_TPW tiles of one
// tile row (the same 64 queries against consecutive 64-point blocks of the design) with one set of tables; 9 KB of LDS,
// d/4 + 1 matrix instructions per 256 elements instead of the difference form's 2 d subtract/FMA wave-instructions per element.
template <int KIND>
__global__ __launch_bounds__(256, 4) void cov_kvec_gram_kernel(double *out, long ld, const double *Xq, int M, int Mp, const double *X,
                                                            const double *Xg, const double *mid, int N, int Np, int d, CovParams p)
{
	__shared__ double tab[EXP_TAB_G];
	__shared__ double wsc[GPEMU_MAX_PARAMS];
	__shared__ double mid_s[GPEMU_MAX_PARAMS];
	gram_tables(p, d, tab, wsc);
	if (threadIdx.x < GPEMU_MAX_PARAMS) mid_s[threadIdx.x] = ((int)threadIdx.x < d) ? mid[threadIdx.x] : 0.0;
	__syncthreads();
	const int ntc = Np / FT, ngc = (ntc + GRAM_TPW - 1) / GRAM_TPW;
	const int tr = blockIdx.x / ngc, tc0 = (blockIdx.x % ngc) * GRAM_TPW;
	for (int i = 0; i < GRAM_TPW; i++) {
		const int tc = tc0 + i;
		if (tc >= ntc) break;
		cov_fill_tile_gram_k<KIND, true>(out, ld, Xq, nullptr, mid_s, M, X, Xg, N, d, p, FILL_CLAMP, tr, tc, tab, wsc);
	}
	(void)Mp;
}

hipError_t launch_cov_kvec_gram(hipStream_t s, double *out, long ld, const double *Xq, int M, int Mp, const double *X, const double *Xg,
                                const double *mid, int N, int Np, int d, const CovParams &p)
{
	if (Np % FT || Mp % FT || !p.gram || !Xg || !mid) return hipErrorInvalidValue;
	const hipError_t e = ensure_exp_table(s);
	if (e != hipSuccess) return e;
	const int ntc = Np / FT, ngc = (ntc + GRAM_TPW - 1) / GRAM_TPW;
	const dim3 grid((unsigned)((Mp / FT) * ngc));
	if (p.kind == GPEMU_POWEREXP)
		hipLaunchKernelGGL(cov_kvec_gram_kernel<GPEMU_POWEREXP>, grid, dim3(256), 0, s, out, ld, Xq, M, Mp, X, Xg, mid, N, Np, d, p);
	else if (p.kind == GPEMU_MATERN32)
		hipLaunchKernelGGL(cov_kvec_gram_kernel<GPEMU_MATERN32>, grid, dim3(256), 0, s, out, ld, Xq, M, Mp, X, Xg, mid, N, Np, d, p);
	else
		hipLaunchKernelGGL(cov_kvec_gram_kernel<GPEMU_MATERN52>, grid, dim3(256), 0, s, out, ld, Xq, M, Mp, X, Xg, mid, N, Np, d, p);
	return hipGetLastError();
}

// pseudo-random fill in [-1,1) for the GEMM micro-benchmark (operands must not be zeros: DVFS, rule 25)
__global__ void fill_random_kernel(double *p, size_t n, unsigned seed)
{
	size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (; i < n; i += stride) {
		unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
		z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
		z ^= z >> 31;
		p[i] = (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
	}
}

hipError_t launch_fill_random(hipStream_t s, double *p, size_t n, unsigned seed)
{
	hipLaunchKernelGGL(fill_random_kernel, dim3(4096), dim3(256), 0, s, p, n, seed);
	return hipGetLastError();
}

// regression basis h(x) (libEmu/regression.c:9-67): h = [1, x, x^2, x^3] per coordinate
__device__ __forceinline__ double hfun(int a, const double *x, int d)
{
	if (a == 0) return 1.0;
	const int q = (a - 1) / d, k = (a - 1) % d;
	const double v = x[k];
	return q == 0 ? v : (q == 1 ? v * v : v * v * v);
}

// R rows (Rp x Np): row 0 = y, row 1+a = column a of the H matrix
// (makeHMatrix_fnptr, regression.c:100-112), zero padded.
__global__ void build_rrows_kernel(double *R, int Np, int Rp, const double *X, const double *y, int N, int d, int nreg)
{
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	const int a = blockIdx.y;
	if (j >= Np) return;
	double v = 0.0;
	if (j < N) {
		if (a == 0) v = y[j];
		else if (a <= nreg) v = hfun(a - 1, X + (long)j * d, d);
	}
	R[(long)a * Np + j] = v;
}

hipError_t launch_build_rrows(hipStream_t s, double *R, int Np, int Rp, const double *X, const double *y,
                              int N, int d, int order)
{
	const int nreg = 1 + order * d;
	dim3 grid((Np + 255) / 256, Rp);
	hipLaunchKernelGGL(build_rrows_kernel, grid, dim3(256), 0, s, R, Np, Rp, X, y, N, d, nreg);
	return hipGetLastError();
}

__global__ void set_identity_rows_kernel(double *T, long ld, int n, long bstride)
{
	T += (long)blockIdx.z * bstride;
	const long i = blockIdx.y;
	const int j = blockIdx.x * blockDim.x + threadIdx.x;
	if (j < n) T[i * ld + j] = (i == j) ? 1.0 : 0.0;
}

hipError_t launch_set_identity_rows(hipStream_t s, double *T, long ld, int n, int nbatch, long bstride)
{
	dim3 grid((n + 255) / 256, n, nbatch);
	hipLaunchKernelGGL(set_identity_rows_kernel, grid, dim3(256), 0, s, T, ld, n, bstride);
	return hipGetLastError();
}

// dst[j][i] = src[i][j] for an n x n block (n multiple of 64)
__global__ __launch_bounds__(256) void transpose_kernel(double *dst, long ldd, const double *src, long lds_, int n)
{
	__shared__ double t[64 * 65];
	const int bi = blockIdx.y * 64, bj = blockIdx.x * 64;
	const int tid = threadIdx.x;
	for (int e = tid; e < 4096; e += 256) {
		int r = e >> 6, c = e & 63;
		t[r * 65 + c] = src[(long)(bi + r) * lds_ + bj + c];
	}
	__syncthreads();
	for (int e = tid; e < 4096; e += 256) {
		int r = e >> 6, c = e & 63;
		dst[(long)(bj + r) * ldd + bi + c] = t[c * 65 + r];
	}
	(void)n;
}

hipError_t launch_transpose(hipStream_t s, double *dst, long ldd, const double *src, long lds_, int n)
{
	dim3 grid(n / 64, n / 64);
	hipLaunchKernelGGL(transpose_kernel, grid, dim3(256), 0, s, dst, ldd, src, lds_, n);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Prediction epilogue (makeEmulatedMean / makeEmulatedVariance,
// libEmu/emulator.c:672-704, 720-785), one wave per query row q of
//   V[q] = [ L^-1 k*  (Np entries) | k*.gamma | k*.W (nreg entries) ]
// mean = h.beta + k*.gamma ;  var = kappa - |L^-1 k*|^2 + q^T Q q,  q = h - W^T k*
// betaQ = beta (nreg) followed by Q = (H^T C^-1 H)^-1 (nreg x nreg).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void predict_finish_kernel(const double *V, long ldv, int M, int Np, int nreg, int d,
                                                             const double *Xq, const double *betaQ, double kappa,
                                                             double *mean, double *var)
{
	const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
	const int q = blockIdx.x * 4 + wave;
	if (q >= M) return;
	const double *v = V + (long)q * ldv;
	double ss = 0.0;
	for (int i = lane * 2; i < Np; i += 128) {
		const double2 t = *reinterpret_cast<const double2 *>(v + i);
		ss += t.x * t.x;
		ss += t.y * t.y;
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off);
	if (lane == 0) {
		const double *x = Xq + (long)q * d;
		const double *beta = betaQ;
		const double *Q = betaQ + nreg;
		double m = v[Np];
		double reg = 0.0;
		for (int a = 0; a < nreg; a++) m += hfun(a, x, d) * beta[a];
		for (int a = 0; a < nreg; a++) {
			const double qa = hfun(a, x, d) - v[Np + 1 + a];
			double t = 0.0;
			for (int b = 0; b < nreg; b++) t += Q[a * nreg + b] * (hfun(b, x, d) - v[Np + 1 + b]);
			reg += qa * t;
		}
		mean[q] = m;
		var[q] = kappa - ss + reg;
	}
}

// ---------------------------------------------------------------------------
// A handful of queries (emulate_point: ONE -- the call an MCMC driver makes per sample, emulator_struct.c:124-143): the
// 64-row tiles of the batch path would spend a table set-up and 63 padding rows on it.  One thread per design point instead
// computes k_i = cov(x_i, x*_q) for the up to 16 queries (difference form, exact nugget test, clamp: makeKVector_fnptr,
// emulator.c:578-593) and writes the 16 rows the skinny product reads (rows beyond M: zeros).
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void kvec_small_kernel(double *Kq, long ld, const double *Xq, int M, const double *X, int N, int Np,
                                                         int d, CovParams p)
{
	__shared__ double xq_s[16 * GPEMU_MAX_PARAMS];
	__shared__ double tab[EXP_TAB];
	const int tid = threadIdx.x;
	for (int e = tid; e < M * d; e += 256) xq_s[e] = Xq[e];
	if (tid < EXP_TAB) tab[tid] = exp2((double)tid * (1.0 / EXP_TAB));
	__syncthreads();
	const int i = blockIdx.x * 256 + tid;
	if (i >= Np) return;
	for (int q = 0; q < 16; q++) {
		double v = 0.0;
		if (q < M && i < N) {
			double a = 0.0;
			int same = 0;
			for (int k = 0; k < d; k++) {
				const double D = X[(long)i * d + k] - xq_s[q * d + k];
				const double t = D * p.w[(p.kind == GPEMU_POWEREXP) ? k : 0];
				a = fma(t, t, a);
				same += (fabs(D) < p.eps) ? 1 : 0;
			}
			if (p.kind == GPEMU_POWEREXP) {
				v = fast_exp_neg(-a, tab) * p.amp;                                  // emulator.c:133,141
			} else {
				const double sdist = fast_sqrt(a);                                  // distance / rho
				if (p.kind == GPEMU_MATERN32) {
					const double root3 = 1.732050808;                               // emulator.c:359 (literal)
					v = p.amp * (1 + root3 * sdist) * fast_exp_neg(-root3 * sdist, tab);
				} else {
					const double root5 = 2.236067978;                               // emulator.c:452 (literal)
					v = p.amp * (1 + root5 * sdist + (5.0 / 3.0) * sdist * sdist) * fast_exp_neg(-root5 * sdist, tab);
				}
			}
			if (same == d) v += p.nug;                                              // emulator.c:136-150 / :368-384 / :462-478
			if (v < 1E-10) v = 0.0;                                                 // emulator.c:588-590
		}
		Kq[(long)q * ld + i] = v;
	}
}

hipError_t launch_kvec_small(hipStream_t s, double *Kq, long ld, const double *Xq, int M, const double *X, int N, int Np, int d,
                             const CovParams &p)
{
	if (M < 1 || M > 16) return hipErrorInvalidValue;
	hipLaunchKernelGGL(kvec_small_kernel, dim3((Np + 255) / 256), dim3(256), 0, s, Kq, ld, Xq, M, X, N, Np, d, p);
	return hipGetLastError();
}

__global__ void sum_slices_kernel(double *V, long n, int nslice, long sstride);

// the epilogue for those few queries, one workgroup per query: the split-K slices of V are summed in slice order while the
// squared norm of the first Np entries is taken (256 threads instead of one wave), the regression part (nreg x nreg) is
// spread over the threads instead of lane 0 walking it with an integer division per basis function.
// mean = h.beta + k*.gamma ;  var = kappa - |L^-1 k*|^2 + q^T Q q,  q = h - W^T k*   (emulator.c:672-704, 720-785)
__global__ __launch_bounds__(256) void predict_finish_small_kernel(const double *Vp, long ldv, long sstride, int nslice, int Np,
                                                                   int nreg, int d, const double *Xq, const double *betaQ, double kappa,
                                                                   double *mean, double *var)
{
	__shared__ double red[256];
	__shared__ double tail[64];          // V[q][Np .. Np+63]: k*.gamma, then (W^T k*)_a
	__shared__ double qv[64], tv[64], hs[64];
	const int q = blockIdx.x, tid = threadIdx.x;
	const double *v0 = Vp + (long)q * ldv;
	double ss = 0.0;
	for (int n = tid; n < Np + 64; n += 256) {
		double v = v0[n];
		for (int s_ = 1; s_ < nslice; s_++) v += v0[(long)s_ * sstride + n];
		if (n < Np) ss = fma(v, v, ss);
		else tail[n - Np] = v;
	}
	red[tid] = ss;
	__syncthreads();
	for (int st = 128; st > 0; st >>= 1) {
		if (tid < st) red[tid] += red[tid + st];
		__syncthreads();
	}
	const double *x = Xq + (long)q * d;
	const double *beta = betaQ, *Q = betaQ + nreg;
	if (tid < nreg) {
		hs[tid] = hfun(tid, x, d);
		qv[tid] = hs[tid] - tail[1 + tid];
	}
	__syncthreads();
	if (tid < nreg) {
		double t = 0.0;
		for (int b = 0; b < nreg; b++) t += Q[tid * nreg + b] * qv[b];
		tv[tid] = t;
	}
	__syncthreads();
	if (tid == 0) {
		double m = tail[0], reg = 0.0;
		for (int a = 0; a < nreg; a++) {
			m += hs[a] * beta[a];
			reg += qv[a] * tv[a];
		}
		mean[q] = m;
		var[q] = kappa - red[0] + reg;
	}
}

hipError_t launch_predict_finish_small(hipStream_t s, const double *Vp, long ldv, long sstride, int nslice, int M, int Np, int nreg,
                                       int d, const double *Xq, const double *betaQ, double kappa, double *mean, double *var)
{
	if (nreg > 63 || M < 1) return hipErrorInvalidValue;
	if (nslice > 1) {
		// the slices are summed by the whole chip first (one workgroup per query reading nslice x 66 KB alone measured 60 us
		// against 6 for this launch); the epilogue below then reads ONE row per query
		const long n = (long)M * ldv;
		hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, s, const_cast<double *>(Vp), n, nslice, sstride);
		nslice = 1;
	}
	hipLaunchKernelGGL(predict_finish_small_kernel, dim3(M), dim3(256), 0, s, Vp, ldv, sstride, nslice, Np, nreg, d, Xq, betaQ, kappa,
	                   mean, var);
	return hipGetLastError();
}

// split-K partial products -> V (slice 0), summed in slice order: V[e] = sum_s V[s*sstride + e]
__global__ __launch_bounds__(256) void sum_slices_kernel(double *V, long n, int nslice, long sstride)
{
	const long e = 2 * ((long)blockIdx.x * 256 + threadIdx.x);
	if (e >= n) return;
	double2 t = *reinterpret_cast<const double2 *>(V + e);
	for (int s = 1; s < nslice; s++) {
		const double2 u = *reinterpret_cast<const double2 *>(V + (long)s * sstride + e);
		t.x += u.x;
		t.y += u.y;
	}
	*reinterpret_cast<double2 *>(V + e) = t;
}

hipError_t launch_predict_finish(hipStream_t s, const double *V, long ldv, int M, int Np, int nreg, int order, int d,
                                 const double *Xq, const double *betaQ, double kappa, double *mean, double *var,
                                 int nslice, long sstride)
{
	(void)order;
	if (nslice > 1) {
		const long n = (long)M * ldv;          // rows are contiguous (ldv = row length), an even number of doubles
		hipLaunchKernelGGL(sum_slices_kernel, dim3((unsigned)((n / 2 + 255) / 256)), dim3(256), 0, s, const_cast<double *>(V), n,
		                   nslice, sstride);
	}
	hipLaunchKernelGGL(predict_finish_kernel, dim3((M + 3) / 4), dim3(256), 0, s, V, ldv, M, Np, nreg, d, Xq, betaQ,
	                   kappa, mean, var);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Gradient partial sums (getGradientCn, libEmu/maxmultimin.c:571-608, with the
// dC/dtheta matrices of derivative_l_gauss, emulator.c:173-209, generated on
// the fly instead of being stored and multiplied by an N^3 dgemm).
// For each length direction k (pow-exp: ONE coordinate only, as the reference):
//   dC_ab = exp(-0.5*e^{-2 t_k} D^2 - 2 t_k) * D^2,  D = x_ak - x_bk
//   tr_k  = sum_ab A_ab dC_ab          (A = C^-1, symmetric)
//   q_k   = sum_ab alpha_a alpha_b dC_ab   (alpha = A y)
// plus tr_A = sum_a A_aa for the nugget direction.  A is read from the lower
// triangle of the corner matrix S (rows/cols offset soff); off-diagonal
// elements are counted twice.  One 64x64 tile per workgroup (lower tiles only);
// part[tile][2*d+2]; slot 2d = tr A, slot 2d+1 = this tile's share of alpha^T alpha (diagonal tiles).
// ag = per element [alpha (np_pad doubles) | gp (length thetas t_k, k < d)], gathered by gather_alpha_kernel: alpha is
// column 0 of the rows of S under soff (the [I rows x R cols] block of the corner), 66 KB apart at N = 8192 -- read in
// place by every tile it cost 3 % of a gradient evaluation.
// blockIdx.y = element of a lock-step batch (strides sstride, gstride, pstride).
// ---------------------------------------------------------------------------
// nbeta > 0 (exact-gradient mode): alpha = C^-1 (y - H beta) = column 0 minus sum_a beta_a column 1+a of the same rows;
// beta sits behind the length thetas in the element's slot of ag
__global__ void gather_alpha_kernel(const double *S, long lds_, int soff, long sstride, int N, double *ag, long gstride,
                                    int np_pad, int nbeta)
{
	const int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= N) return;
	const double *row = S + (long)blockIdx.y * sstride + (long)(soff + i) * lds_;
	double a = row[0];
	if (nbeta > 0) {
		const double *beta = ag + (long)blockIdx.y * gstride + np_pad + GPEMU_MAX_PARAMS;
		for (int k = 0; k < nbeta; k++) a -= beta[k] * row[1 + k];
	}
	ag[(long)blockIdx.y * gstride + i] = a;
}

// Sum over the 256 threads of a workgroup of NV (1 or 16) per-thread partials, in a fixed order.  Inside a wave the sixteen
// values are summed by a butterfly that HALVES what a lane carries at every step: lanes l and l ^ 32 split the sixteen
// between them (each adds the partner's half to its own), l ^ 16 the remaining eight, then four, then two -- after four
// steps lane l holds the sum over sixteen lanes of value (l >> 2) & 15, two plain steps finish the wave: 17 exchanges of
// a double per lane.  The four waves' totals meet in 64 doubles of LDS.  Returns the total of value `tid` in threads
// tid < NV.  Two barriers per call.  (Rounds 3-5a exchanged all 256 x 16 partials through LDS: 35 KB per workgroup, which
// with the coordinate tiles and the exp table left TWO workgroups per CU at d = 16 -- the gradient kernels ran at a
// seventh of their instruction rate.)
template <int NV>
__device__ __forceinline__ double block_sum(const double (&acc)[NV], double *scratch)
{
	static_assert(NV == 1 || NV == 8 || NV == 16, "one value, eight or sixteen");
	const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
	double v;
	if (NV == 8) {
		// (the same butterfly from eight values: three halving steps, lane l then holds value (l >> 3) & 7 summed over eight lanes)
		double a4[4], a2[2];
		{
			const bool hi = (lane & 32) != 0;
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const double mine = hi ? acc[(4 + i) % NV] : acc[i % NV], send = hi ? acc[i % NV] : acc[(4 + i) % NV];
				a4[i] = mine + __shfl_xor(send, 32);
			}
		}
		{
			const bool hi = (lane & 16) != 0;
#pragma unroll
			for (int i = 0; i < 2; i++) {
				const double mine = hi ? a4[2 + i] : a4[i], send = hi ? a4[i] : a4[2 + i];
				a2[i] = mine + __shfl_xor(send, 16);
			}
		}
		{
			const bool hi = (lane & 8) != 0;
			v = (hi ? a2[1] : a2[0]) + __shfl_xor(hi ? a2[0] : a2[1], 8);
		}
		v += __shfl_xor(v, 4);
		v += __shfl_xor(v, 2);
		v += __shfl_xor(v, 1);
		if ((lane & 7) == 0) scratch[wave * 16 + (lane >> 3)] = v;
	} else if (NV == 16) {
		double a8[8], a4[4], a2[2];
		{
			const bool hi = (lane & 32) != 0;
#pragma unroll
			for (int i = 0; i < 8; i++) {
				const double mine = hi ? acc[(8 + i) % NV] : acc[i % NV], send = hi ? acc[i % NV] : acc[(8 + i) % NV];
				a8[i] = mine + __shfl_xor(send, 32);
			}
		}
		{
			const bool hi = (lane & 16) != 0;
#pragma unroll
			for (int i = 0; i < 4; i++) {
				const double mine = hi ? a8[4 + i] : a8[i], send = hi ? a8[i] : a8[4 + i];
				a4[i] = mine + __shfl_xor(send, 16);
			}
		}
		{
			const bool hi = (lane & 8) != 0;
#pragma unroll
			for (int i = 0; i < 2; i++) {
				const double mine = hi ? a4[2 + i] : a4[i], send = hi ? a4[i] : a4[2 + i];
				a2[i] = mine + __shfl_xor(send, 8);
			}
		}
		{
			const bool hi = (lane & 4) != 0;
			v = (hi ? a2[1] : a2[0]) + __shfl_xor(hi ? a2[0] : a2[1], 4);
		}
		v += __shfl_xor(v, 2);
		v += __shfl_xor(v, 1);
		if ((lane & 3) == 0) scratch[wave * 16 + (lane >> 2)] = v;
	} else {
		v = acc[0];
		v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8);
		v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
		if (lane == 0) scratch[wave * 16] = v;
	}
	__syncthreads();
	double sum = 0.0;
	if (tid < NV) sum = ((scratch[tid] + scratch[16 + tid]) + scratch[32 + tid]) + scratch[48 + tid];
	__syncthreads();
	return sum;
}

// dynamic LDS of the two gradient kernels (doubles): two 64 x (d + 1) coordinate tiles, alpha of the tile's rows and
// columns, the exp table, per-direction constants, the reduction scratch
constexpr int GRAD_CHUNK = 4;                      // literal form: directions per reduction (2 sums each: eight values, block_sum<8>)
__host__ __device__ inline size_t grad_lds_doubles(int d, int tab_len)
{
	return (size_t)128 * (d + 1) + 128 + tab_len + 2 * (size_t)((d + 1) & ~1) + 64;
}

// exp(x), x <= 0, from the 1024-entry table 2^(j/1024) in LDS (the Gram-form fill's exp with amplitude 1), arguments
// below -700 taken as -700 (e^-700 = 2^-1010 is still a normal number for the exponent-field add)
__device__ __forceinline__ double fast_exp_neg_t(double x, const double *tab)
{
	return fast_exp_neg_g(fmax(x, -700.0), tab);
}

// CLAMP = false: the host has checked that 1/2 e^{-2 theta_k} D^2 stays below 600 for every pair of design points and
// every direction of the batch (the usual case), so the exp argument needs no lower bound
template <bool CLAMP>
__global__ __launch_bounds__(256, 4) void grad_part_kernel(const double *S, long lds_, int soff, long sstride, const double *X,
                                                        int N, int d, const double *ag, int np_pad, long gstride, double *part,
                                                        long pstride)
{
	S += (long)blockIdx.y * sstride;
	const double *alpha = ag + (long)blockIdx.y * gstride;
	const double *gp = alpha + np_pad;
	part += (long)blockIdx.y * pstride;
	const int np = 2 * d + 2;
	// lower-triangular tile index -> (tr, tc)
	const int t = blockIdx.x;
	int tr, tc;
	lower_tile(t, tr, tc);

	// coordinate tiles TRANSPOSED, [direction][64 points]: the 16 row coordinates a thread needs per direction sit 32 bytes
	// apart (immediate offsets, no address arithmetic in the element loop) and the column coordinates of a wave are contiguous
	extern __shared__ double grad_sm[];
	const int dpad = (d + 1) & ~1;
	double *xr_t = grad_sm, *xc_t = xr_t + 64 * (d + 1), *ar_s = xc_t + 64 * (d + 1), *ac_s = ar_s + 64, *tab = ac_s + 64;
	double *hk = tab + EXP_TAB_G, *e2k = hk + dpad, *scratch = e2k + dpad;
	const int tid = threadIdx.x;
	for (int e = tid; e < 64 * d; e += 256) {
		int r = e / d, k = e % d;
		int gr = tr * 64 + r, gc = tc * 64 + r;
		xr_t[k * 64 + r] = (gr < N) ? X[(long)gr * d + k] : 0.0;
		xc_t[k * 64 + r] = (gc < N) ? X[(long)gc * d + k] : 0.0;
	}
	if (tid < 64) {
		int gr = tr * 64 + tid, gc = tc * 64 + tid;
		ar_s[tid] = (gr < N) ? alpha[gr] : 0.0;
		ac_s[tid] = (gc < N) ? alpha[gc] : 0.0;
	}
	for (int e = tid; e < EXP_TAB_G; e += 256) tab[e] = g_exp2_tab[e];
	// per direction, once per workgroup: e^{-2 theta_k} (the factor exp(-2 theta) of emulator.c:203 leaves the element
	// loop: exp(-1/2 e^{-2t} D^2 - 2t) = e^{-2t} exp(-1/2 e^{-2t} D^2)) and half of it
	if (tid < d) {
		const double e2 = exp(-2.0 * gp[tid]);
		e2k[tid] = e2;
		hk[tid] = 0.5 * e2;
	}
	__syncthreads();

	const int c = tid & 63, rsub = tid >> 6;
	const int gc = tc * 64 + c;
	// weight x A_ab of the thread's 16 elements; the weight x alpha_a alpha_b of the quadratic forms is NOT kept per element
	// (32 more registers, and the kernel is paced by how many waves a SIMD holds: 252 registers = two waves, 128 = four):
	// alpha_b is the thread's own column factor and leaves the sums, alpha_a comes from LDS (one address per wave: a
	// broadcast) where it is used, and the weight is 2 for every element of a tile below the diagonal blocks (`interior`);
	// the diagonal and edge tiles take it from two bit masks
	const bool interior = tr > tc && tr * 64 + 63 < N && tc * 64 + 63 < N;
	double wa[16];
	unsigned m1 = 0, m2 = 0;               // elements of weight 1 (the diagonal) and 2
	double tsum = 0.0;
#pragma unroll
	for (int u = 0; u < 16; u++) {
		const int r = rsub + 4 * u;
		const int gr = tr * 64 + r;
		const bool valid = gr < N && gc < N && gc <= gr;
		const double a = valid ? S[(long)(soff + gr) * lds_ + soff + gc] : 0.0;
		const double w = valid ? ((gc == gr) ? 1.0 : 2.0) : 0.0;
		wa[u] = w * a;
		if (valid) { if (gc == gr) m1 |= 1u << u; else m2 |= 1u << u; }
		if (gr == gc && gr < N) tsum += a;
	}
	const double acq = ac_s[c];
	const double *arw = ar_s + rsub;       // alpha of the thread's rows: arw[4 u]
	// trace of A; alpha^T alpha, rows of the diagonal tiles in index order
	{
		const double one[1] = {tsum};
		const double tot = block_sum<1>(one, scratch);
		if (tid == 0) {
			part[(long)t * np + 2 * d] = tot;
			double aa = 0.0;
			if (tr == tc)
				for (int r = 0; r < 64; r++) aa += ar_s[r] * ar_s[r];
			part[(long)t * np + 2 * d + 1] = aa;
		}
	}
	for (int k0 = 0; k0 < d; k0 += GRAD_CHUNK) {
		double acc[2 * GRAD_CHUNK];
#pragma unroll
		for (int v = 0; v < 2 * GRAD_CHUNK; v++) acc[v] = 0.0;
#pragma unroll
		for (int j = 0; j < GRAD_CHUNK; j++) {
			const int k = k0 + j;
			if (k < d) {
				const double nh = -hk[k];
				const double xck = xc_t[k * 64 + c];
				const double *xrk = xr_t + k * 64 + rsub;
				if (interior) {
#pragma unroll
					for (int u = 0; u < 16; u++) {
						const double D = xrk[4 * u] - xck;
						const double uu = D * D;
						const double x = nh * uu;
						const double z = uu * (CLAMP ? fast_exp_neg_t(x, tab) : fast_exp_neg_g(x, tab));   // emulator.c:203 without its e^{-2 theta}
						acc[2 * j] = fma(wa[u], z, acc[2 * j]);
						acc[2 * j + 1] = fma(arw[4 * u], z, acc[2 * j + 1]);
						if (u & 1) __builtin_amdgcn_sched_barrier(0);            // two chains in flight (four waves per SIMD cover the rest)
					}
				} else {
#pragma unroll
					for (int u = 0; u < 16; u++) {
						const double D = xrk[4 * u] - xck;
						const double uu = D * D;
						const double x = nh * uu;
						const double z = uu * (CLAMP ? fast_exp_neg_t(x, tab) : fast_exp_neg_g(x, tab));
						const double wsel = ((m2 >> u) & 1u) ? 2.0 : (((m1 >> u) & 1u) ? 1.0 : 0.0);
						acc[2 * j] = fma(wa[u], z, acc[2 * j]);
						acc[2 * j + 1] = fma(wsel * arw[4 * u], z, acc[2 * j + 1]);
						if (u & 1) __builtin_amdgcn_sched_barrier(0);
					}
				}
				acc[2 * j + 1] *= interior ? 2.0 * acq : acq;
			}
		}
		const double tot = block_sum<2 * GRAD_CHUNK>(acc, scratch);
		const int v = tid, k = k0 + (v >> 1);
		if (tid < 2 * GRAD_CHUNK && k < d) part[(long)t * np + 2 * k + (v & 1)] = e2k[k] * tot;
	}
}

// ---------------------------------------------------------------------------
// Exact gradient of the objective gpemu_loglik returns (GPEMU_MODE_EXACT_GRAD; SURVEY App. C3/C4 "corrected form"):
//   d(-logL)/dtheta_k = 1/2 sum_ab (A_ab - alpha_a alpha_b) dC_ab/dtheta_k,  A = C^-1, alpha = A (y - H beta)
// (beta is the GLS minimiser, so its own dependence on theta drops out), with the true derivative matrices
//   pow-exp  (emulator.c:101-152): dC/dtheta_{k+2} = E_ab * D_k^2 e^{-2 theta_{k+2}},  E = amp exp(-1/2 sum_j D_j^2 e^{-2 theta_j})
//            -- the FULL kernel value, where the reference's derivative_l_gauss (:173-209) keeps one coordinate's factor
//   Matern 3/2 (:344-386), s = r/rho, c = 1.732050808: dC/dlog(rho) = amp c^2 s^2 e^{-cs}
//   Matern 5/2 (:438-480), c = 2.236067978:            dC/dlog(rho) = amp (s^2 (c^2 - 10/3) + (5/3) c s^3) e^{-cs}
//   nugget: dC/dlog(nug) = nug wherever the nugget rule adds it (every pair of coinciding rows, not only i == j)
// One 64x64 lower tile per workgroup, as grad_part_kernel; part[tile][k] for k < nd (nd = d for pow-exp, 1 for
// Matern) and part[tile][nd] for the nugget direction, within the same [tile][2d+2] slots.
// ---------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void grad_exact_kernel(const double *S, long lds_, int soff, long sstride, const double *X,
                                                         int N, int d, const double *ag, long gstride, double *part,
                                                         long pstride, const CovParams *pp)
{
	S += (long)blockIdx.y * sstride;
	const double *alpha = ag + (long)blockIdx.y * gstride;
	part += (long)blockIdx.y * pstride;
	const int np = 2 * d + 2;
	const int nd = (KIND == GPEMU_POWEREXP) ? d : 1;
	int tr, tc;
	lower_tile(blockIdx.x, tr, tc);
	const long t = blockIdx.x;

	__shared__ CovParams ps;
	extern __shared__ double grad_sm[];
	const int sd = d + 1, dpad = (d + 1) & ~1;
	double *xr_s = grad_sm, *xc_s = xr_s + 64 * sd, *ar_s = xc_s + 64 * sd, *ac_s = ar_s + 64, *tab = ac_s + 64;
	double *scratch = tab + EXP_TAB + 2 * dpad;
	const int tid = threadIdx.x;
	for (int e = tid; e < (int)(sizeof(CovParams) / sizeof(double)); e += 256)
		reinterpret_cast<double *>(&ps)[e] = reinterpret_cast<const double *>(pp + blockIdx.y)[e];
	for (int e = tid; e < 64 * d; e += 256) {
		int r = e / d, k = e % d;
		int gr = tr * 64 + r, gc = tc * 64 + r;
		xr_s[r * sd + k] = (gr < N) ? X[(long)gr * d + k] : 0.0;
		xc_s[r * sd + k] = (gc < N) ? X[(long)gc * d + k] : 0.0;
	}
	if (tid < 64) {
		int gr = tr * 64 + tid, gc = tc * 64 + tid;
		ar_s[tid] = (gr < N) ? alpha[gr] : 0.0;
		ac_s[tid] = (gc < N) ? alpha[gc] : 0.0;
	}
	if (tid < EXP_TAB) tab[tid] = exp2((double)tid * (1.0 / EXP_TAB));
	__syncthreads();

	const int c = tid & 63, rsub = tid >> 6;
	const int gc = tc * 64 + c;
	double wk[16];                 // weight of the element times its kernel factor (pow-exp: E; Matern: dC/dlog rho)
	double s_nug = 0.0;
#pragma unroll
	for (int u = 0; u < 16; u++) {
		const int r = rsub + 4 * u;
		const int gr = tr * 64 + r;
		const bool valid = gr < N && gc < N && gc <= gr;
		double W = 0.0;
		if (valid) {
			const double a = S[(long)(soff + gr) * lds_ + soff + gc];
			W = ((gc == gr) ? 1.0 : 2.0) * (a - ar_s[r] * ac_s[c]);
		}
		double d2 = 0.0;
		int same = 0;
		for (int k = 0; k < d; k++) {
			const double D = xr_s[r * sd + k] - xc_s[c * sd + k];
			const double v = D * ps.w[(KIND == GPEMU_POWEREXP) ? k : 0];
			d2 = fma(v, v, d2);
			same += (fabs(D) < ps.eps) ? 1 : 0;
		}
		if (KIND == GPEMU_POWEREXP) {
			wk[u] = W * (fast_exp_neg(-d2, tab) * ps.amp);
		} else {
			const double sdist = fast_sqrt(d2);
			if (KIND == GPEMU_MATERN32) {
				const double c3 = 1.732050808;
				wk[u] = W * (ps.amp * (c3 * c3) * d2 * fast_exp_neg(-c3 * sdist, tab));
			} else {
				const double c5 = 2.236067978;
				wk[u] = W * (ps.amp * (d2 * (c5 * c5 - 10.0 / 3.0) + (5.0 / 3.0) * c5 * d2 * sdist) * fast_exp_neg(-c5 * sdist, tab));
			}
		}
		if (same == d) s_nug += W * ps.nug;
	}
	// the nd + 1 sums of the tile (directions 0 .. nd-1, then the nugget), sixteen per reduction
	for (int k0 = 0; k0 <= nd; k0 += 16) {
		double acc[16];
#pragma unroll
		for (int j = 0; j < 16; j++) {
			const int k = k0 + j;
			double sk = 0.0;
			if (k == nd) {
				sk = s_nug;
			} else if (k < nd) {
				if (KIND == GPEMU_POWEREXP) {
					const double wkk = ps.w[k], xck = xc_s[c * sd + k];
#pragma unroll
					for (int u = 0; u < 16; u++) {
						const double v = (xr_s[(rsub + 4 * u) * sd + k] - xck) * wkk;      // v^2 = 1/2 D^2 e^{-2 theta}
						sk = fma(wk[u], 2.0 * v * v, sk);
					}
				} else {
#pragma unroll
					for (int u = 0; u < 16; u++) sk += wk[u];
				}
			}
			acc[j] = sk;
		}
		const double tot = block_sum<16>(acc, scratch);
		const int k = k0 + tid;
		if (tid < 16 && k <= nd) part[t * np + k] = tot;
	}
}

// The same sums with the tile's squared scaled distances from the fp64 MFMA (round 5).  grad_exact_kernel above spends
// 5 d vector instructions per element on the distance and the "same point" test and 4-5 more per element and direction
// (164 at d = 16: 4 ms per batch of 16 at N = 4096, d = 16 -- a sixth of a value+gradient batch, profiles/
// r05_pca8_cli_training_kernel_trace.txt); here the distances of a 64 x 64 tile are d/4 + 1 matrix instructions per wave
// (|x'|^2 + |y'|^2 - 2 x'.y' on the design centred per dimension, as the Gram-form fill), the exp is the fill's 1024-entry
// table form, and a direction costs a subtraction, a product and an FMA per element.  The Gram form's cancellation error
// -- a few ulp of |x'|^2 + |y'|^2 in the squared distance, i.e. that much RELATIVE error in a weight -- is harmless
// here at any length scale: the weights enter plain sums (no factorisation amplifies them), and the bar is 1e-8 of the
// largest gradient component.  (The fill itself keeps its bound, make_cov_params: there the error meets cond(C).)
// Pairs whose squared distance comes out below p.cand_w -- the nugget rule's candidates plus the cancellation bound -- are
// recomputed from differences and take the exact "same point" test on the raw coordinates, as in the fill.
// Lane (q, g) of wave w holds the elements (row 16 w + g + 4 r, column 16 j + q), r, j < 4.
// LDS (doubles): coordinate tiles transposed [d][64] x 2, alpha x 2, table 1024, scales d, reduction scratch 256 x 17.
__host__ __device__ inline size_t grad_gram_lds_doubles(int d) { return (size_t)128 * d + 128 + EXP_TAB_G + ((d + 1) & ~1) + 64; }

template <int KIND>
__global__ __launch_bounds__(256, KIND == GPEMU_POWEREXP ? 4 : 1) void grad_exact_gram_kernel(const double *S, long lds_, int soff, long sstride, const double *X,
                                                              const double *Xg, int N, int d, const double *ag, long gstride,
                                                              double *part, long pstride, const CovParams *pp)
{
	S += (long)blockIdx.y * sstride;
	const double *alpha = ag + (long)blockIdx.y * gstride;
	part += (long)blockIdx.y * pstride;
	const int np = 2 * d + 2;
	const int nd = (KIND == GPEMU_POWEREXP) ? d : 1;
	int tr, tc;
	lower_tile(blockIdx.x, tr, tc);
	const long t = blockIdx.x;

	__shared__ CovParams ps;
	extern __shared__ double grad_sm[];
	const int dpad = (d + 1) & ~1;
	double *xr_t = grad_sm, *xc_t = xr_t + 64 * d, *ar_s = xc_t + 64 * d, *ac_s = ar_s + 64, *tab = ac_s + 64;
	double *wsc = tab + EXP_TAB_G, *scratch = wsc + dpad;
	const int tid = threadIdx.x;
	for (int e = tid; e < (int)(sizeof(CovParams) / sizeof(double)); e += 256)
		reinterpret_cast<double *>(&ps)[e] = reinterpret_cast<const double *>(pp + blockIdx.y)[e];
	for (int e = tid; e < 64 * d; e += 256) {
		const int r = e / d, k = e % d;
		const int gr = tr * 64 + r, gc = tc * 64 + r;
		xr_t[k * 64 + r] = (gr < N) ? Xg[(long)gr * d + k] : 0.0;
		xc_t[k * 64 + r] = (gc < N) ? Xg[(long)gc * d + k] : 0.0;
	}
	if (tid < 64) {
		const int gr = tr * 64 + tid, gc = tc * 64 + tid;
		ar_s[tid] = (gr < N) ? alpha[gr] : 0.0;
		ac_s[tid] = (gc < N) ? alpha[gc] : 0.0;
	}
	__syncthreads();                                                  // ps is read below by every thread
	const double croot = gram_root(KIND);
	for (int e = tid; e < EXP_TAB_G; e += 256) tab[e] = g_exp2_tab[e] * ps.amp;
	if (tid < d) wsc[tid] = ps.w[(KIND == GPEMU_POWEREXP) ? tid : 0] * croot;
	__syncthreads();

	const int lane = tid & 63, wave = tid >> 6;
	const int q = lane & 15, g = lane >> 4;
	// squared scaled distances of the lane's 16 elements from the matrix unit
	d4g_t acc[4];
#pragma unroll
	for (int j = 0; j < 4; j++) acc[j] = (d4g_t){0.0, 0.0, 0.0, 0.0};
	double na = 0.0, nb[4] = {0.0, 0.0, 0.0, 0.0};
	for (int k0 = 0; k0 < d; k0 += 4) {
		const int k = k0 + g;
		const bool kv = k < d;
		const double wk_ = wsc[kv ? k : 0];
		const double xa = kv ? xr_t[k * 64 + 16 * wave + q] * wk_ : 0.0;
		na = fma(xa, xa, na);
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const double xb = kv ? xc_t[k * 64 + 16 * j + q] * wk_ : 0.0;
			nb[j] = fma(xb, xb, nb[j]);
			acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, -2.0 * xb, acc[j], 0, 0, 0);
		}
	}
	na += __shfl_xor(na, 16); na += __shfl_xor(na, 32);
#pragma unroll
	for (int j = 0; j < 4; j++) { nb[j] += __shfl_xor(nb[j], 16); nb[j] += __shfl_xor(nb[j], 32); }
	{
		const double ea = (g == 0) ? na : (g == 1 ? 1.0 : 0.0);
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const double eb = (g == 0) ? 1.0 : (g == 1 ? nb[j] : 0.0);
			acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ea, eb, acc[j], 0, 0, 0);
		}
	}
	const int row0 = 16 * wave + g, col0 = q;                         // tile-local
	// nugget-rule candidates (the diagonal, duplicated design points): exact distance and "same point" test
	const double cand_w = ps.cand_w * croot * croot;
	unsigned same = 0;
	{
		bool cand = false;
#pragma unroll
		for (int r = 0; r < 4; r++)
#pragma unroll
			for (int j = 0; j < 4; j++) cand = cand || (acc[j][r] <= cand_w);
		if (__any(cand)) {
#pragma unroll 1
			for (int e = 0; e < 16; e++) {
				const int r = e >> 2, j = e & 3;
				double ae = 0.0;
#pragma unroll
				for (int rr = 0; rr < 4; rr++)
#pragma unroll
					for (int jj = 0; jj < 4; jj++) ae = (rr == r && jj == j) ? acc[jj][rr] : ae;
				const int gr = tr * 64 + row0 + 4 * r, gc = tc * 64 + col0 + 16 * j;
				if (ae <= cand_w && gr < N && gc < N) {
					int cnt = 0;
					double a = 0.0;
					for (int k = 0; k < d; k++) {
						const double D = X[(long)gr * d + k] - X[(long)gc * d + k];
						const double v = D * wsc[k];
						a = fma(v, v, a);
						cnt += (fabs(D) < ps.eps) ? 1 : 0;
					}
#pragma unroll
					for (int rr = 0; rr < 4; rr++)
#pragma unroll
						for (int jj = 0; jj < 4; jj++) acc[jj][rr] = (rr == r && jj == j) ? a : acc[jj][rr];
					if (cnt == d) same |= 1u << e;
				}
			}
		}
	}
	// weight of every element times its kernel factor (pow-exp: the kernel value E; Matern: dC/dlog rho)
	double wk[4][4];                                                  // [r][j]
	double s_nug = 0.0;
#pragma unroll
	for (int r = 0; r < 4; r++) {
		const int lr = row0 + 4 * r, gr = tr * 64 + lr;
		const double arow = ar_s[lr];
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const int lc = col0 + 16 * j, gc = tc * 64 + lc;
			const bool valid = gr < N && gc < N && gc <= gr;
			double W = 0.0;
			if (valid) {
				const double a = S[(long)(soff + gr) * lds_ + soff + gc];
				W = ((gc == gr) ? 1.0 : 2.0) * (a - arow * ac_s[lc]);
			}
			// (the exponent held at -600: weights beyond that are 1e-261 of the amplitude either way, and the table exp --
			// which carries the amplitude and has no clamp of its own -- stays inside the normal numbers for any amplitude
			// above 2^-150; a cancellation that leaves a tiny negative squared distance counts as zero)
			const double u2 = fmin(fmax(acc[j][r], 0.0), KIND == GPEMU_POWEREXP ? 600.0 : 360000.0);
			if (KIND == GPEMU_POWEREXP) {
				wk[r][j] = W * fast_exp_neg_g(-u2, tab);                                   // amp in the table
			} else {
				const double u = fast_sqrt_g(u2);                                           // c * distance / rho
				const double e = fast_exp_neg_g(-u, tab);
				if (KIND == GPEMU_MATERN32) {
					wk[r][j] = W * (u2 * e);                                                // amp c^2 s^2 e^{-cs}
				} else {
					const double c5 = 2.236067978, ic2 = 1.0 / (c5 * c5);
					wk[r][j] = W * (e * u2 * fma((5.0 / 3.0) * ic2, u, 1.0 - (10.0 / 3.0) * ic2));   // amp (s^2 (c^2 - 10/3) + (5/3) c s^3) e^{-cs}
				}
			}
			if (same & (1u << (4 * r + j))) s_nug += W * ps.nug;
		}
	}
	// the nd + 1 sums of the tile (directions 0 .. nd-1, then the nugget), eight per reduction (sixteen kept 32 more
	// registers alive: the kernel is paced by how many waves a SIMD holds -- 1 711 us per batch of 16 at N=4096, d=16 with
	// two waves per SIMD, 968 us with four)
	for (int k0 = 0; k0 <= nd; k0 += 8) {
		double out[8];
#pragma unroll
		for (int jd = 0; jd < 8; jd++) {
			const int k = k0 + jd;
			double sk = 0.0;
			if (k == nd) {
				sk = s_nug;
			} else if (k < nd) {
				if (KIND == GPEMU_POWEREXP) {
					double xr4[4], xc4[4];
#pragma unroll
					for (int r = 0; r < 4; r++) xr4[r] = xr_t[k * 64 + row0 + 4 * r];
#pragma unroll
					for (int j = 0; j < 4; j++) xc4[j] = xc_t[k * 64 + col0 + 16 * j];
#pragma unroll
					for (int r = 0; r < 4; r++)
#pragma unroll
						for (int j = 0; j < 4; j++) {
							const double D = xr4[r] - xc4[j];
							sk = fma(wk[r][j], D * D, sk);
						}
					const double w = ps.w[k];
					sk *= 2.0 * w * w;                                                      // D^2 e^{-2 theta_k} = 2 (D w_k)^2
				} else {
#pragma unroll
					for (int r = 0; r < 4; r++)
#pragma unroll
						for (int j = 0; j < 4; j++) sk += wk[r][j];
				}
			}
			out[jd] = sk;
		}
		const double tot = block_sum<8>(out, scratch);
		const int k = k0 + tid;
		if (tid < 8 && k <= nd) part[t * np + k] = tot;
	}
}

// beta = (H^T C^-1 H)^-1 (H^T C^-1 y) of one batch element from its Gram matrix G = Z^T Z (res: Rp x Rp row-major, G[0][0]
// = y.Cinv.y, G[1+a][0] = (H^T Cinv y)_a, G[1+a][1+b] = (H^T Cinv H)_ab -- what finish_kernel leaves in dRes), by a
// Cholesky solve in LDS; written behind the element's alpha scratch, where gather_alpha_kernel expects it.  This is the
// estimateBeta of the exact-gradient mode (regression.c:120-176 on the Gram matrix) done on the device so that a
// value+gradient batch needs no host round trip between its factorisation and its gradient reductions.  One wave per
// element; a matrix that is not positive definite gives NaNs (the host reports GPEMU_ERR_REGRESSION for that element).
__global__ __launch_bounds__(64) void beta_solve_kernel(const double *res, long rstride, int Rp, int nreg, double *ag,
                                                        long gstride, int np_pad)
{
	__shared__ double A[64 * 65];
	__shared__ double b[64];
	const double *G = res + (long)blockIdx.x * rstride;
	double *beta = ag + (long)blockIdx.x * gstride + np_pad + GPEMU_MAX_PARAMS;
	const int i = threadIdx.x;
	if (i < nreg) {
		for (int j = 0; j < nreg; j++) A[i * 65 + j] = G[(long)(1 + i) * Rp + 1 + j];
		b[i] = G[(long)(1 + i) * Rp];
	}
	__syncthreads();
	for (int j = 0; j < nreg; j++) {
		const double pv = A[j * 65 + j];
		const double l = pv > 0.0 ? sqrt(pv) : nan("");
		__syncthreads();
		if (i == j) A[j * 65 + j] = l;
		if (i > j && i < nreg) A[i * 65 + j] /= l;
		__syncthreads();
		if (i > j && i < nreg)
			for (int k = j + 1; k <= i; k++) A[i * 65 + k] -= A[i * 65 + j] * A[k * 65 + j];
		__syncthreads();
	}
	// L z = b, then L^T beta = z
	for (int j = 0; j < nreg; j++) {
		if (i == j) b[j] /= A[j * 65 + j];
		__syncthreads();
		if (i > j && i < nreg) b[i] -= A[i * 65 + j] * b[j];
		__syncthreads();
	}
	for (int j = nreg - 1; j >= 0; j--) {
		if (i == j) b[j] /= A[j * 65 + j];
		__syncthreads();
		if (i < j) b[i] -= A[j * 65 + i] * b[j];
		__syncthreads();
	}
	if (i < nreg) beta[i] = b[i];
}

hipError_t launch_beta_solve(hipStream_t s, const double *res, long rstride, int Rp, int nreg, int nb, double *ag, long gstride,
                             int np_pad)
{
	if (nreg < 1 || nreg > 63) return hipErrorInvalidValue;
	hipLaunchKernelGGL(beta_solve_kernel, dim3(nb), dim3(64), 0, s, res, rstride, Rp, nreg, ag, gstride, np_pad);
	return hipGetLastError();
}

// sums[b][k] = sum over the tiles t of part[b][t][k] in a fixed order (thread j takes tiles j, j + 256, ...; then a tree
// over the 256 partial sums): the second stage of the gradient reductions, so that a batch hands back 2d+2 numbers per
// element instead of 2d+2 per tile.  grid (np, nb)
__global__ __launch_bounds__(256) void grad_reduce_kernel(const double *part, long pstride, int ntiles, int np, double *sums,
                                                          long sstride)
{
	__shared__ double red[256];
	const double *p = part + (long)blockIdx.y * pstride + blockIdx.x;
	double acc = 0.0;
	for (int t = threadIdx.x; t < ntiles; t += 256) acc += p[(long)t * np];
	red[threadIdx.x] = acc;
	__syncthreads();
	for (int st = 128; st > 0; st >>= 1) {
		if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
		__syncthreads();
	}
	if (threadIdx.x == 0) sums[(long)blockIdx.y * sstride + blockIdx.x] = red[0];
}

hipError_t launch_grad_reduce(hipStream_t s, const double *part, long pstride, int ntiles, int np, int nb, double *sums, long sstride)
{
	hipLaunchKernelGGL(grad_reduce_kernel, dim3(np, nb), dim3(256), 0, s, part, pstride, ntiles, np, sums, sstride);
	return hipGetLastError();
}

// nb corners sstride apart; ag: nb slots of gstride doubles [alpha scratch (np_pad) | the length thetas]; part: nb blocks
// of pstride doubles
hipError_t launch_grad_partials(hipStream_t s, const double *S, long lds_, int soff, long sstride, int nb, const double *X, int N,
                                int d, double *ag, int np_pad, long gstride, double *part, long pstride, int *nparts,
                                int exact_kind, int nbeta, const CovParams *pp_dev, bool lit_noclamp, const double *Xg)
{
	const int nt = (N + 63) / 64;
	const int ntiles = nt * (nt + 1) / 2;
	*nparts = ntiles;
	hipLaunchKernelGGL(gather_alpha_kernel, dim3((N + 255) / 256, nb), dim3(256), 0, s, S, lds_, soff, sstride, N, ag, gstride,
	                   np_pad, exact_kind ? nbeta : 0);
	const dim3 grid(ntiles, nb);
	const size_t lds_exact = grad_lds_doubles(d, EXP_TAB) * sizeof(double), lds_lit = grad_lds_doubles(d, EXP_TAB_G) * sizeof(double);
	// (more than 64 KB of dynamic LDS -- d beyond 20 or so -- has to be allowed per kernel)
	auto allow = [](const void *fn, size_t bytes) {
		return bytes <= 65536 ? hipSuccess : hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
	};
	if (exact_kind && Xg) {
		// exact form with the tile's distances from the matrix unit (grad_exact_gram_kernel)
		hipError_t e = ensure_exp_table(s);
		if (e != hipSuccess) return e;
		const size_t lds_g = grad_gram_lds_doubles(d) * sizeof(double);
		const void *fn = exact_kind == GPEMU_POWEREXP ? (const void *)grad_exact_gram_kernel<GPEMU_POWEREXP>
		               : exact_kind == GPEMU_MATERN32 ? (const void *)grad_exact_gram_kernel<GPEMU_MATERN32>
		                                              : (const void *)grad_exact_gram_kernel<GPEMU_MATERN52>;
		e = allow(fn, lds_g);
		if (e != hipSuccess) return e;
		if (exact_kind == GPEMU_POWEREXP)
			hipLaunchKernelGGL(grad_exact_gram_kernel<GPEMU_POWEREXP>, grid, dim3(256), lds_g, s, S, lds_, soff, sstride, X, Xg, N, d, ag,
			                   gstride, part, pstride, pp_dev);
		else if (exact_kind == GPEMU_MATERN32)
			hipLaunchKernelGGL(grad_exact_gram_kernel<GPEMU_MATERN32>, grid, dim3(256), lds_g, s, S, lds_, soff, sstride, X, Xg, N, d, ag,
			                   gstride, part, pstride, pp_dev);
		else
			hipLaunchKernelGGL(grad_exact_gram_kernel<GPEMU_MATERN52>, grid, dim3(256), lds_g, s, S, lds_, soff, sstride, X, Xg, N, d, ag,
			                   gstride, part, pstride, pp_dev);
		return hipGetLastError();
	}
	hipError_t ea = hipSuccess;
	if (exact_kind == GPEMU_POWEREXP) ea = allow((const void *)grad_exact_kernel<GPEMU_POWEREXP>, lds_exact);
	else if (exact_kind == GPEMU_MATERN32) ea = allow((const void *)grad_exact_kernel<GPEMU_MATERN32>, lds_exact);
	else if (exact_kind == GPEMU_MATERN52) ea = allow((const void *)grad_exact_kernel<GPEMU_MATERN52>, lds_exact);
	else ea = lit_noclamp ? allow((const void *)grad_part_kernel<false>, lds_lit) : allow((const void *)grad_part_kernel<true>, lds_lit);
	if (ea != hipSuccess) return ea;
	if (exact_kind == GPEMU_POWEREXP)
		hipLaunchKernelGGL(grad_exact_kernel<GPEMU_POWEREXP>, grid, dim3(256), lds_exact, s, S, lds_, soff, sstride, X, N, d, ag, gstride,
		                   part, pstride, pp_dev);
	else if (exact_kind == GPEMU_MATERN32)
		hipLaunchKernelGGL(grad_exact_kernel<GPEMU_MATERN32>, grid, dim3(256), lds_exact, s, S, lds_, soff, sstride, X, N, d, ag, gstride,
		                   part, pstride, pp_dev);
	else if (exact_kind == GPEMU_MATERN52)
		hipLaunchKernelGGL(grad_exact_kernel<GPEMU_MATERN52>, grid, dim3(256), lds_exact, s, S, lds_, soff, sstride, X, N, d, ag, gstride,
		                   part, pstride, pp_dev);
	else {
		const hipError_t e = ensure_exp_table(s);           // the literal form's exp reads the fill's 2^(j/1024) table
		if (e != hipSuccess) return e;
		if (lit_noclamp)
			hipLaunchKernelGGL(grad_part_kernel<false>, grid, dim3(256), lds_lit, s, S, lds_, soff, sstride, X, N, d, ag, np_pad, gstride,
			                   part, pstride);
		else
			hipLaunchKernelGGL(grad_part_kernel<true>, grid, dim3(256), lds_lit, s, S, lds_, soff, sstride, X, N, d, ag, np_pad, gstride,
			                   part, pstride);
	}
	return hipGetLastError();
}



// sum_ij A[i][j] * B[j][i] over an n x n pair (row stride ld): per-workgroup partial sums in a fixed order,
// part[blockIdx.x]; getGradientCn's trace(C^-1 dC) (maxmultimin.c:583-588) without forming the product
__global__ __launch_bounds__(256) void trace_product_kernel(const double *A, const double *B, long ld, int n, double *part)
{
	__shared__ double red[256];
	const int i = blockIdx.x;                       // one row of A (column of B) per workgroup
	double s = 0.0;
	for (int j = threadIdx.x; j < n; j += 256) s += A[(long)i * ld + j] * B[(long)j * ld + i];
	red[threadIdx.x] = s;
	__syncthreads();
	for (int st = 128; st > 0; st >>= 1) {
		if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
		__syncthreads();
	}
	if (threadIdx.x == 0) part[i] = red[0];
}

hipError_t launch_trace_product(hipStream_t s, const double *A, const double *B, long ld, int n, double *part)
{
	hipLaunchKernelGGL(trace_product_kernel, dim3(n), dim3(256), 0, s, A, B, ld, n, part);
	return hipGetLastError();
}
// derivative_l_gauss materialised (libEmu/emulator.c:173-209): out[i][j] = exp(-0.5 e^{-2t} D^2 - 2t) D^2 with
// D = x_i - x_j in ONE coordinate (the column passed in) -- the literal formula, other coordinates ignored
__global__ __launch_bounds__(256) void deriv_gauss_kernel(double *out, long ld, const double *xcol, int n, double theta_len)
{
	const int j = blockIdx.x * 64 + (threadIdx.x & 63);
	const int i0 = blockIdx.y * 64 + (threadIdx.x >> 6);
	if (j >= n) return;
	const double xj = xcol[j], e2 = exp(-2.0 * theta_len);
	for (int t = 0; t < 16; t++) {
		const int i = i0 + 4 * t;
		if (i >= n) break;
		const double r = xcol[i] - xj;
		out[(long)i * ld + j] = exp(-0.5 * e2 * r * r - 2 * theta_len) * r * r;
	}
}

hipError_t launch_deriv_gauss(hipStream_t s, double *out, long ld, const double *xcol, int n, double theta_len)
{
	hipLaunchKernelGGL(deriv_gauss_kernel, dim3((n + 63) / 64, (n + 63) / 64), dim3(256), 0, s, out, ld, xcol, n, theta_len);
	return hipGetLastError();
}
} // namespace gpemu

// kernels_linalg.hip -- fp64 dense kernels for gfx950 (MI355X):
//   * gemm_nt_kernel : C = beta*C + alpha*A*B^T on v_mfma_f64_16x16x4_f64, 128x128 tiles, LDS double buffer
//   * leaf_kernel    : 64x64 diagonal-block Cholesky in LDS + forward substitution of the panel rows below
//   * gram / finish  : Z Z^T Gram matrix of the solved right-hand sides and sum(log L_ii)
//
// These stand in for gsl_linalg_cholesky_decomp / _invert and the gsl_blas
// dgemm/dgemv/ddot calls of the reference's likelihood path
// (libEmu/maxmultimin.c:325,361; libEmu/regression.c:128-171;
// libEmu/estimator-fns.c:87-88) -- see DESIGN.md for the formulation.
#include "gpemu_internal.hpp"

namespace gpemu {

typedef double d4_t __attribute__((ext_vector_type(4)));
typedef double d2_t __attribute__((ext_vector_type(2)));

constexpr int LDS_S = GEMM_BK + 2;   // row stride (doubles): conflict-free ds_read_b64 fragments

// ---------------------------------------------------------------------------
// GEMM  C[m x n] = beta*C + alpha * A[m x K] * B[n x K]^T   (row-major, k contiguous in A and B)
//
// 256 threads = 4 waves as 2x2; each wave owns a 64x64 sub-tile = 4x4 MFMA
// 16x16 tiles (128 accumulator VGPRs).  MFMA f64 16x16x4 operand maps
// (cdna_hip_programming.md section 3): A lane l -> A[row l&15][k l>>4],
// B lane l -> B[k l>>4][col l&15], D reg r -> D[row (l>>4)+4r][col l&15].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmArgs g)
{
	__shared__ double As[2][GEMM_BM * LDS_S];
	__shared__ double Bs[2][GEMM_BN * LDS_S];

	const int tiles_m = (g.m + GEMM_BM - 1) / GEMM_BM;
	const int tm = blockIdx.x % tiles_m;
	const int tn = blockIdx.x / tiles_m;
	if (g.tri && tn * GEMM_BN > tm * GEMM_BM + GEMM_BM - 1 + g.diag_off) return;

	int kb = g.k0, ke = g.k1;
	if (g.kstart_mode) {
		int ks = (tm * GEMM_BM - g.kstart_off) & ~(GEMM_BK - 1);
		if (ks > kb) kb = ks;
	}
	if (g.kend_mode) {
		int kx = (tn * GEMM_BN + GEMM_BN - g.kend_off + GEMM_BK - 1) & ~(GEMM_BK - 1);
		if (kx < ke) ke = kx;
	}

	const int tid = threadIdx.x;
	const int lane = tid & 63;
	const int wave = tid >> 6;
	const int wm = wave >> 1, wn = wave & 1;

	// staging map: 4 x (row, 16-byte segment) per operand per thread
	const double *ag[4];
	const double *bg[4];
	int lofs[4];
#pragma unroll
	for (int it = 0; it < 4; it++) {
		int idx = tid + 256 * it;
		int row = idx >> 3, seg = idx & 7;
		int ar = tm * GEMM_BM + row; if (ar > g.m - 1) ar = g.m - 1;
		int br = tn * GEMM_BN + row; if (br > g.n - 1) br = g.n - 1;
		ag[it] = g.A + (long)ar * g.lda + 2 * seg;
		bg[it] = g.B + (long)br * g.ldb + 2 * seg;
		lofs[it] = row * LDS_S + 2 * seg;
	}

	d4_t acc[4][4];
#pragma unroll
	for (int i = 0; i < 4; i++)
#pragma unroll
		for (int j = 0; j < 4; j++) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};

	const int a_base = (wm * 64 + (lane & 15)) * LDS_S + (lane >> 4);
	const int b_base = (wn * 64 + (lane & 15)) * LDS_S + (lane >> 4);

	if (kb < ke) {
		d2_t ra[4], rb[4];
#pragma unroll
		for (int it = 0; it < 4; it++) {
			ra[it] = *reinterpret_cast<const d2_t *>(ag[it] + kb);
			rb[it] = *reinterpret_cast<const d2_t *>(bg[it] + kb);
		}
#pragma unroll
		for (int it = 0; it < 4; it++) {
			*reinterpret_cast<d2_t *>(&As[0][lofs[it]]) = ra[it];
			*reinterpret_cast<d2_t *>(&Bs[0][lofs[it]]) = rb[it];
		}
		__syncthreads();

		int cur = 0;
		for (int k = kb; k < ke; k += GEMM_BK) {
			const bool more = (k + GEMM_BK) < ke;
			if (more) {
#pragma unroll
				for (int it = 0; it < 4; it++) {
					ra[it] = *reinterpret_cast<const d2_t *>(ag[it] + k + GEMM_BK);
					rb[it] = *reinterpret_cast<const d2_t *>(bg[it] + k + GEMM_BK);
				}
			}
			const double *as = As[cur];
			const double *bs = Bs[cur];
#pragma unroll
			for (int s = 0; s < GEMM_BK / 4; s++) {
				double a[4], b[4];
#pragma unroll
				for (int i = 0; i < 4; i++) a[i] = as[a_base + i * 16 * LDS_S + 4 * s];
#pragma unroll
				for (int j = 0; j < 4; j++) b[j] = bs[b_base + j * 16 * LDS_S + 4 * s];
#pragma unroll
				for (int i = 0; i < 4; i++)
#pragma unroll
					for (int j = 0; j < 4; j++)
						acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
			}
			if (more) {
#pragma unroll
				for (int it = 0; it < 4; it++) {
					*reinterpret_cast<d2_t *>(&As[cur ^ 1][lofs[it]]) = ra[it];
					*reinterpret_cast<d2_t *>(&Bs[cur ^ 1][lofs[it]]) = rb[it];
				}
			}
			__syncthreads();
			cur ^= 1;
		}
	}

	// epilogue
	const int row0 = tm * GEMM_BM + wm * 64 + (lane >> 4);
	const int col0 = tn * GEMM_BN + wn * 64 + (lane & 15);
#pragma unroll
	for (int i = 0; i < 4; i++)
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const int row = row0 + i * 16 + 4 * r;
			if (row >= g.m) continue;
			double *crow = g.C + (long)row * g.ldc;
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const int col = col0 + j * 16;
				if (col >= g.n) continue;
				double v = g.alpha * acc[i][j][r];
				if (g.beta) v += crow[col];
				crow[col] = v;
			}
		}
}

hipError_t launch_gemm(hipStream_t s, const GemmArgs &a)
{
	if (a.m <= 0 || a.n <= 0) return hipSuccess;
	const int tiles_m = (a.m + GEMM_BM - 1) / GEMM_BM;
	const int tiles_n = (a.n + GEMM_BN - 1) / GEMM_BN;
	hipLaunchKernelGGL(gemm_nt_kernel, dim3(tiles_m * tiles_n), dim3(256), 0, s, a);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Leaf: factor the 64x64 diagonal block at (c0,c0) of the tall matrix T and
// solve X * L^T = B for the m_below rows under it (in place).  Every
// workgroup factors the block redundantly in LDS (no separate launch, no
// inter-workgroup hand-off); workgroup 0 writes L back.  Each wave then owns
// 64 panel rows, one row per lane, held in registers.
// A pivot <= 0 (or NaN) records its 1-based global index in *info (atomicMin)
// -- GSL_EDOM of gsl_linalg_cholesky_decomp (maxmultimin.c:325-350).
// ---------------------------------------------------------------------------
constexpr int LP = LEAF + 1;

__global__ __launch_bounds__(256) void leaf_kernel(double *T, long ld, int c0, int m_below, int *info)
{
	__shared__ double Ls[LEAF * LP];
	__shared__ double dg[LEAF];
	__shared__ double invd[LEAF];

	const int tid = threadIdx.x;
	double *D = T + (long)c0 * ld + c0;
	for (int e = tid; e < LEAF * LEAF; e += 256) {
		int r = e >> 6, c = e & 63;
		Ls[r * LP + c] = D[(long)r * ld + c];
	}
	__syncthreads();

	for (int j = 0; j < LEAF; j++) {
		const double p = Ls[j * LP + j];
		const double s = sqrt(p);
		const double inv = 1.0 / s;
		if (tid < LEAF) {
			if (tid == j) {
				dg[j] = s;
				invd[j] = inv;
				if (!(p > 0.0) && blockIdx.x == 0) atomicMin(info, c0 + j + 1);
			} else if (tid > j) {
				Ls[tid * LP + j] *= inv;
			}
		}
		__syncthreads();
		{
			const int i = tid >> 2;
			if (i > j) {
				const double lij = Ls[i * LP + j];
				for (int c = j + 1 + (tid & 3); c <= i; c += 4)
					Ls[i * LP + c] -= lij * Ls[c * LP + j];
			}
		}
		__syncthreads();
	}

	if (blockIdx.x == 0) {
		for (int e = tid; e < LEAF * LEAF; e += 256) {
			int r = e >> 6, c = e & 63;
			if (c < r) D[(long)r * ld + c] = Ls[r * LP + c];
			else if (c == r) D[(long)r * ld + c] = dg[r];
		}
	}

	// panel rows: X L^T = B  ->  column-oriented forward substitution, row per lane
	const int wave = tid >> 6, lane = tid & 63;
	const int prow = (blockIdx.x * 4 + wave) * 64 + lane;
	if (prow < m_below) {
		double *bp = T + (long)(c0 + LEAF + prow) * ld + c0;
		double b[LEAF];
#pragma unroll
		for (int k = 0; k < LEAF; k += 2) {
			d2_t v = *reinterpret_cast<const d2_t *>(bp + k);
			b[k] = v[0];
			b[k + 1] = v[1];
		}
#pragma unroll
		for (int k = 0; k < LEAF; k++) {
			const double xk = b[k] * invd[k];
			b[k] = xk;
#pragma unroll
			for (int j = k + 1; j < LEAF; j++)
				b[j] -= xk * Ls[j * LP + k];
		}
#pragma unroll
		for (int k = 0; k < LEAF; k += 2) {
			d2_t v = {b[k], b[k + 1]};
			*reinterpret_cast<d2_t *>(bp + k) = v;
		}
	}
}

hipError_t launch_leaf(hipStream_t s, double *T, long ld, int c0, int m_below, int *info)
{
	int nblk = (m_below + 255) / 256;
	if (nblk < 1) nblk = 1;
	hipLaunchKernelGGL(leaf_kernel, dim3(nblk), dim3(256), 0, s, T, ld, c0, m_below, info);
	return hipGetLastError();
}

// ---------------------------------------------------------------------------
// Gram partials: part[blk][a][b] = sum_{j in 64-column chunk blk} Z[a][j] Z[b][j]
// Z rows are the solved right-hand sides (rows Np.. of T): Z = L^-1 [y|H].
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gram_part_kernel(const double *Z, long ld, int nrhs, int Rp, double *part)
{
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

hipError_t launch_gram_partials(hipStream_t s, const double *Z, long ld, int Np, int nrhs, int Rp, double *part)
{
	hipLaunchKernelGGL(gram_part_kernel, dim3(Np / 64), dim3(256), 0, s, Z, ld, nrhs, Rp, part);
	return hipGetLastError();
}

// finish: fixed-order reduction of the Gram partials and 2*sum(log L_ii)
// (replaces det = (prod L_ii)^2 of maxmultimin.c:355-358, which under/overflows)
__global__ __launch_bounds__(256) void finish_kernel(const double *part, int nparts, int Rp, int nrhs,
                                                     const double *T, long ld, int N, double *res)
{
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
                         int N, double *res)
{
	hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(256), 0, s, part, nparts, Rp, nrhs, T, ld, N, res);
	return hipGetLastError();
}

} // namespace gpemu

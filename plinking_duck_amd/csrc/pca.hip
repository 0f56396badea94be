// pca.hip -- plink_pca's normalisation tables and the tall-skinny dense helpers of its on-device
// orthonormalisation (gfx950).  The contractions with the packed matrix run on the int8 matrix cores
// (score_i8.hip; pca_i8.hip for the transposed matrix of Step A).
//
// Data layout: the genotype matrix is variant-major; row v holds ceil(N/4)
// bytes of packed 2-bit calls (00 hom-ref, 01 het, 10 hom-alt, 11 missing;
// sample s in bits 2*(s%4) of byte s/4) followed by zero bytes up to `pitch`
// (a multiple of 16, so every row can be streamed as whole 16-byte lanes and
// the pad decodes as hom-ref, which every kernel cancels against N).
#include "device_utils.hpp"
#include "kernels.hpp"

#include <algorithm>

namespace pgh {

namespace {

// ---------------------------------------------------------------------------
// plink_pca
// ---------------------------------------------------------------------------

// Normalised-genotype table of each effective variant (NormalizeGenotypes,
// src/plink_common.cpp:1535-1543): t[g] = (g - center) * inv_stdev, missing -> 0.
__global__ __launch_bounds__(256) void k_norm_tables(const double *__restrict__ center,
                                                     const double *__restrict__ inv_stdev, uint32_t n,
                                                     double *__restrict__ ts) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) {
		return;
	}
	const double c = center[i], is = inv_stdev[i];
	ts[4 * static_cast<uint64_t>(i) + 0] = (0.0 - c) * is;
	ts[4 * static_cast<uint64_t>(i) + 1] = (1.0 - c) * is;
	ts[4 * static_cast<uint64_t>(i) + 2] = (2.0 - c) * is;
	ts[4 * static_cast<uint64_t>(i) + 3] = 0.0;
}

// rows of excluded samples -> 0 (keeps a sample subset out of the power iteration)
__global__ __launch_bounds__(256) void k_mask_rows(double *__restrict__ m, uint32_t n_rows, uint32_t stride,
                                                   uint32_t ncols, const uint8_t *__restrict__ mask2) {
	const uint64_t idx = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
	if (idx >= static_cast<uint64_t>(n_rows) * ncols) {
		return;
	}
	const uint32_t s = static_cast<uint32_t>(idx / ncols), c = static_cast<uint32_t>(idx % ncols);
	if (!((mask2[s >> 2] >> (2 * (s & 3))) & 1u)) {
		m[static_cast<uint64_t>(s) * stride + c] = 0.0;
	}
}

__global__ __launch_bounds__(256) void k_scale(double *__restrict__ m, uint64_t n, double f) {
	const uint64_t idx = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
	if (idx < n) {
		m[idx] *= f;
	}
}

// ---------------------------------------------------------------------------
// tall-skinny dense helpers for plink_pca's orthonormalisation (FP64)
// ---------------------------------------------------------------------------

// C[i][j] += sum_r A[r][i] * B[r][j]   (A: m x na, B: m x nb, row-major; C zeroed by the caller)
// One wave per 16x16 tile of C and per chunk of rows, on v_mfma_f64_16x16x4_f64:
// K runs over rows, 4 at a time; both operands are 128-byte row segments.
__global__ __launch_bounds__(256) void k_tall_gram(const double *__restrict__ A, uint32_t lda, uint32_t na,
                                                   const double *__restrict__ B, uint32_t ldb, uint32_t nb,
                                                   uint64_t m, uint32_t rows_per_wave, double *__restrict__ C,
                                                   uint32_t ldc) {
	const uint32_t tiles_b = (nb + 15) / 16;
	const uint32_t ta = blockIdx.x / tiles_b, tb = blockIdx.x % tiles_b;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t li = lane & 15u, lk = lane >> 4;
	const uint64_t r_begin = (static_cast<uint64_t>(blockIdx.y) * 4u + wave) * rows_per_wave;
	const uint64_t r_end = min(r_begin + rows_per_wave, m);
	const uint32_t ca = ta * 16u + li, cb = tb * 16u + li;
	const bool a_ok = ca < na, b_ok = cb < nb;
	f64x4 acc = {0.0, 0.0, 0.0, 0.0};
	for (uint64_t r = r_begin; r < r_end; r += 4) {
		const uint64_t row = r + lk;
		const bool in = row < r_end;
		const double a = (in && a_ok) ? A[row * lda + ca] : 0.0;
		const double b = (in && b_ok) ? B[row * ldb + cb] : 0.0;
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
	}
	if (r_begin < r_end) {
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const uint32_t i = ta * 16u + lk + 4u * r, j = tb * 16u + li;
			if (i < na && j < nb) {
				unsafeAtomicAdd(C + static_cast<uint64_t>(i) * ldc + j, acc[r]);
			}
		}
	}
}

// out[r][j] = beta * B[r][j] + alpha * sum_i A[r][i] * C[i][j]   (j < nb; C: na x nb, small)
// 4 rows per workgroup staged in LDS; a lane produces 4 adjacent columns of one row.
__global__ __launch_bounds__(256) void k_tall_times_small(const double *__restrict__ A, uint32_t lda, uint32_t na,
                                                          const double *__restrict__ C, uint32_t ldc, uint32_t nb,
                                                          double alpha, double beta, const double *__restrict__ B,
                                                          uint32_t ldb, double *__restrict__ out, uint32_t ldo,
                                                          uint64_t m) {
	extern __shared__ double s_rows[]; // [4][na]
	const uint32_t rr = threadIdx.x >> 6, jg = threadIdx.x & 63u;
	const uint64_t r0 = static_cast<uint64_t>(blockIdx.x) * 4u;
	for (uint32_t e = threadIdx.x; e < 4u * na; e += 256u) {
		const uint64_t row = r0 + e / na;
		s_rows[e] = row < m ? A[row * lda + e % na] : 0.0;
	}
	__syncthreads();
	const uint64_t row = r0 + rr;
	if (row >= m) {
		return;
	}
	for (uint32_t j0 = jg * 4u; j0 < nb; j0 += 256u) {
		double acc[4] = {0.0, 0.0, 0.0, 0.0};
		const uint32_t w = min(4u, nb - j0);
		for (uint32_t i = 0; i < na; i++) {
			const double a = s_rows[rr * na + i];
			const double *c = C + static_cast<uint64_t>(i) * ldc + j0;
#pragma unroll
			for (uint32_t q = 0; q < 4; q++) {
				if (q < w) {
					acc[q] = fma(a, c[q], acc[q]);
				}
			}
		}
		for (uint32_t q = 0; q < w; q++) {
			const double prev = beta != 0.0 ? beta * B[row * ldb + j0 + q] : 0.0;
			out[row * ldo + j0 + q] = prev + alpha * acc[q];
		}
	}
}

// dst[r][j] = src[r][j] for j < n (strided 2-D copy)
__global__ __launch_bounds__(256) void k_copy_cols(const double *__restrict__ src, uint32_t lds_, double *__restrict__ dst,
                                                   uint32_t ldd, uint32_t n, uint64_t m) {
	const uint64_t idx = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
	if (idx >= m * n) {
		return;
	}
	const uint64_t r = idx / n;
	const uint32_t j = static_cast<uint32_t>(idx % n);
	dst[r * ldd + j] = src[r * lds_ + j];
}

} // namespace

// ---------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------

hipError_t LaunchNormTables(const double *center, const double *inv_stdev, uint32_t n, double *ts,
                            hipStream_t stream) {
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_norm_tables, dim3((n + 255) / 256), dim3(256), 0, stream, center, inv_stdev, n, ts);
	return hipGetLastError();
}

hipError_t LaunchMaskRows(double *m, uint32_t n_rows, uint32_t stride, uint32_t n_cols, const uint8_t *mask2,
                          hipStream_t stream) {
	const uint64_t total = static_cast<uint64_t>(n_rows) * n_cols;
	if (total == 0 || !mask2) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_mask_rows, dim3(static_cast<uint32_t>((total + 255) / 256)), dim3(256), 0, stream, m, n_rows,
	                   stride, n_cols, mask2);
	return hipGetLastError();
}

hipError_t LaunchScale(double *m, uint64_t n, double f, hipStream_t stream) {
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_scale, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, stream, m, n, f);
	return hipGetLastError();
}

hipError_t LaunchTallGram(const double *A, uint32_t lda, uint32_t na, const double *B, uint32_t ldb, uint32_t nb,
                          uint64_t m, double *C, uint32_t ldc, hipStream_t stream) {
	if (m == 0 || na == 0 || nb == 0) {
		return hipSuccess;
	}
	const uint32_t tiles = ((na + 15) / 16) * ((nb + 15) / 16);
	// ~1024 waves per tile pair at most, >= 256 rows per wave
	uint32_t rows_per_wave = 1024;
	uint64_t chunks = (m + 4ull * rows_per_wave - 1) / (4ull * rows_per_wave);
	while (chunks > 65535) {
		rows_per_wave *= 2;
		chunks = (m + 4ull * rows_per_wave - 1) / (4ull * rows_per_wave);
	}
	hipLaunchKernelGGL(k_tall_gram, dim3(tiles, static_cast<uint32_t>(chunks)), dim3(256), 0, stream, A, lda, na, B, ldb,
	                   nb, m, rows_per_wave, C, ldc);
	return hipGetLastError();
}

hipError_t LaunchTallTimesSmall(const double *A, uint32_t lda, uint32_t na, const double *C, uint32_t ldc, uint32_t nb,
                                double alpha, double beta, const double *B, uint32_t ldb, double *out, uint32_t ldo,
                                uint64_t m, hipStream_t stream) {
	if (m == 0 || nb == 0) {
		return hipSuccess;
	}
	const uint64_t blocks = (m + 3) / 4;
	hipLaunchKernelGGL(k_tall_times_small, dim3(static_cast<uint32_t>(blocks)), dim3(256), 4 * na * sizeof(double), stream,
	                   A, lda, na, C, ldc, nb, alpha, beta, B, ldb, out, ldo, m);
	return hipGetLastError();
}

hipError_t LaunchCopyCols(const double *src, uint32_t ld_src, double *dst, uint32_t ld_dst, uint32_t n, uint64_t m,
                          hipStream_t stream) {
	const uint64_t total = m * n;
	if (total == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_copy_cols, dim3(static_cast<uint32_t>((total + 255) / 256)), dim3(256), 0, stream, src, ld_src,
	                   dst, ld_dst, n, m);
	return hipGetLastError();
}

} // namespace pgh

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
// A lane owns one ROW and up to W of its output columns in registers; the workgroup's 256 rows of A pass through
// LDS sixteen columns at a time (coalesced 128-byte row segments in, conflict-free column reads out), and C's
// entries are wave-uniform operands.  (The first form gave a lane four adjacent columns of a row: with the 20 or
// 10 columns plink_pca asks for, five or three lanes of a wave worked -- 1 ms per call, 43 calls per plink_pca.)
constexpr uint32_t kTtsRows = 256, kTtsChunk = 16;
template <int W>
__global__ __launch_bounds__(256) void k_tall_times_small(const double *__restrict__ A, uint32_t lda, uint32_t na,
                                                          const double *__restrict__ C, uint32_t ldc, uint32_t nb,
                                                          double alpha, double beta, const double *B, uint32_t ldb,
                                                          double *out, uint32_t ldo, uint64_t m) {
	__shared__ double s_a[kTtsRows][kTtsChunk + 1];
	const uint64_t r0 = static_cast<uint64_t>(blockIdx.x) * kTtsRows;
	const uint64_t row = r0 + threadIdx.x;
	for (uint32_t j0 = 0; j0 < nb; j0 += W) {
		double acc[W];
#pragma unroll
		for (int q = 0; q < W; q++) {
			acc[q] = 0.0;
		}
		for (uint32_t i0 = 0; i0 < na; i0 += kTtsChunk) {
			__syncthreads();
#pragma unroll
			for (uint32_t k = 0; k < kTtsChunk; k++) {
				const uint32_t e = threadIdx.x + 256u * k;
				const uint32_t rr = e / kTtsChunk, cc = e % kTtsChunk;
				s_a[rr][cc] = (r0 + rr < m && i0 + cc < na) ? A[(r0 + rr) * lda + i0 + cc] : 0.0;
			}
			__syncthreads();
#pragma unroll 1
			for (uint32_t i = 0; i < kTtsChunk; i++) { // (unrolled, C's entries overflow the scalar registers)
				const double a = s_a[threadIdx.x][i];
				const double *c = C + static_cast<uint64_t>(min(i0 + i, na - 1u)) * ldc; // (past na: a is 0)
#pragma unroll
				for (int q = 0; q < W; q++) {
					const double cq = j0 + q < nb ? c[j0 + q] : 0.0;
					acc[q] = fma(a, cq, acc[q]);
				}
			}
		}
		if (row < m) {
#pragma unroll
			for (int q = 0; q < W; q++) {
				if (j0 + q < nb) {
					const double prev = beta != 0.0 ? beta * B[row * ldb + j0 + q] : 0.0;
					out[row * ldo + j0 + q] = prev + alpha * acc[q];
				}
			}
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
	const uint32_t blocks = static_cast<uint32_t>((m + kTtsRows - 1) / kTtsRows);
#define PGH_TTS(W)                                                                                                     \
	hipLaunchKernelGGL(k_tall_times_small<W>, dim3(blocks), dim3(256), 0, stream, A, lda, na, C, ldc, nb, alpha, beta, B,  \
	                   ldb, out, ldo, m)
	if (nb <= 8) {
		PGH_TTS(8);
	} else if (nb <= 12) {
		PGH_TTS(12);
	} else if (nb <= 16) {
		PGH_TTS(16);
	} else if (nb <= 20) {
		PGH_TTS(20);
	} else if (nb <= 24) {
		PGH_TTS(24);
	} else {
		PGH_TTS(32); // wider outputs: 32 columns at a time, A re-read per group
	}
#undef PGH_TTS
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

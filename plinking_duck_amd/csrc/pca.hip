// pca.hip -- plink_pca's Step A (variant-side reduction on FP64 MFMA) and the tall-skinny dense
// helpers of its on-device orthonormalisation (gfx950).
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

// out[i][c] = sum over samples of ts[i][g(i,s)] * G[s][c]   (Step A, src/plink_pca.cpp:632-645)
// One workgroup per tile of VT variants; a lane walks samples s = tid, tid+256, ...,
// loads its G row once per sample and feeds all VT variants of the tile.
template <int NCOLS, int VT>
__global__ __launch_bounds__(256) void k_variant_reduce(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                        uint32_t sample_ct, const uint32_t *__restrict__ vlist,
                                                        uint32_t n_var, const double *__restrict__ ts,
                                                        const double *__restrict__ G, uint32_t g_stride,
                                                        double *__restrict__ out, uint32_t out_stride) {
	__shared__ double red[4][VT * NCOLS];
	__shared__ double s_t[VT][4];
	__shared__ uint32_t s_v[VT];
	const uint32_t i0 = blockIdx.x * VT;
	const uint32_t nv = min(static_cast<uint32_t>(VT), n_var - i0);
	if (threadIdx.x < VT * 4) {
		const uint32_t k = threadIdx.x >> 2;
		s_t[k][threadIdx.x & 3] = k < nv ? ts[4 * static_cast<uint64_t>(i0 + k) + (threadIdx.x & 3)] : 0.0;
	}
	if (threadIdx.x < VT) {
		s_v[threadIdx.x] = threadIdx.x < nv ? vlist[i0 + threadIdx.x] : vlist[i0];
	}
	__syncthreads();
	double acc[VT][NCOLS];
#pragma unroll
	for (int k = 0; k < VT; k++) {
#pragma unroll
		for (int c = 0; c < NCOLS; c++) {
			acc[k][c] = 0.0;
		}
	}
	for (uint32_t s = threadIdx.x; s < sample_ct; s += 256u) {
		double g[NCOLS];
#pragma unroll
		for (int c = 0; c < NCOLS; c++) {
			g[c] = G[static_cast<uint64_t>(s) * g_stride + c];
		}
		const uint32_t shift = 2u * (s & 15u);
#pragma unroll
		for (int k = 0; k < VT; k++) {
			const uint32_t w = reinterpret_cast<const uint32_t *>(rows + static_cast<uint64_t>(s_v[k]) * pitch)[s >> 4];
			const double x = s_t[k][(w >> shift) & 3u];
#pragma unroll
			for (int c = 0; c < NCOLS; c++) {
				acc[k][c] = fma(x, g[c], acc[k][c]);
			}
		}
	}
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
	for (int k = 0; k < VT; k++) {
#pragma unroll
		for (int c = 0; c < NCOLS; c++) {
			double v = acc[k][c];
#pragma unroll
			for (int off = 32; off > 0; off >>= 1) {
				v += __shfl_xor(v, off, 64);
			}
			if (lane == 0) {
				red[wave][k * NCOLS + c] = v;
			}
		}
	}
	__syncthreads();
	if (threadIdx.x < VT * NCOLS) {
		const uint32_t k = threadIdx.x / NCOLS, c = threadIdx.x % NCOLS;
		if (k < nv) {
			// fixed order: deterministic
			out[static_cast<uint64_t>(i0 + k) * out_stride + c] =
			    ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
		}
	}
}

// MFMA form of Step A:  out[v][c] = sum_s T_v[g(v,s)] * G[s][c]
// on v_mfma_f64_16x16x4_f64 tiles: M = 16 variants, K = 4 samples, N = 16 columns.
//   A[i][k] = T_{v_i}[g(v_i, s_k)]   lane l: i = l & 15 (variant), k = l >> 4 (sample 4q + k)
//   B[k][j] = G[s_k][j]              lane l: k = l >> 4, j = l & 15        (from the LDS chunk)
// A workgroup owns 128 variants (4 waves x 2 tiles) and streams every sample in chunks of
// 128: the G chunk (128 x 32 doubles) is staged through LDS once per workgroup, prefetched
// into registers while the previous chunk is multiplied.  Each lane reads its variant's
// 32 bytes of the chunk (the 4 lane groups of a variant share the load) and peels sample
// 4q + k at step q with a per-lane constant shift.  No atomics: a variant's whole sum
// lives in one wave.
// NQ quarter tiles: as in k_accumulate_mfma, 4 more columns on v_mfma_f64_4x4x4_4b_f64 with the
// big tile's A operand (block b = variants 4b..4b+3 of the tile).
template <int NCT, int NQ>
__global__ __launch_bounds__(256) void k_variant_reduce_mfma(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                             uint32_t sample_ct, const uint32_t *__restrict__ vlist,
                                                             uint32_t n_var, const double *__restrict__ ts,
                                                             const double *__restrict__ G, uint32_t g_stride,
                                                             uint32_t n_cols, double *__restrict__ out,
                                                             uint32_t out_stride, uint32_t chunks_per_split) {
	constexpr uint32_t kChunk = 128;           // samples per LDS chunk
	constexpr uint32_t kCols = 16 * NCT + 4 * NQ;
	constexpr uint32_t kQ = NQ > 0 ? NQ : 1;
	constexpr uint32_t kGPerThread = kChunk * kCols / 256;
	constexpr uint32_t kVT = 2;                // variant tiles per wave
	__shared__ double s_g[kChunk][kCols];
	__shared__ double s_t[4 * kVT * 16][4];    // tables of the workgroup's 128 variants
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t li = lane & 15u, lk = lane >> 4;
	const uint32_t v_wg = blockIdx.x * (4u * kVT * 16u);

	// tables + row pointers of this lane's two variants
	for (uint32_t e = threadIdx.x; e < 4u * kVT * 16u * 4u; e += 256u) {
		const uint32_t v = v_wg + (e >> 2);
		s_t[e >> 2][e & 3] = v < n_var ? ts[4 * static_cast<uint64_t>(v) + (e & 3)] : 0.0;
	}
	const uint8_t *row_ptr[kVT];
	uint32_t t_base[kVT];
#pragma unroll
	for (uint32_t t = 0; t < kVT; t++) {
		const uint32_t v_local = (wave * kVT + t) * 16u + li;
		const uint32_t v = v_wg + v_local;
		row_ptr[t] = rows + static_cast<uint64_t>(vlist[v < n_var ? v : 0]) * pitch;
		t_base[t] = v_local;
	}
	f64x4 acc[kVT][NCT];
#pragma unroll
	for (uint32_t t = 0; t < kVT; t++) {
#pragma unroll
		for (int c = 0; c < NCT; c++) {
			acc[t][c] = f64x4 {0.0, 0.0, 0.0, 0.0};
		}
	}
	double accq[kVT][kQ];
#pragma unroll
	for (uint32_t t = 0; t < kVT; t++) {
#pragma unroll
		for (uint32_t qq = 0; qq < kQ; qq++) {
			accq[t][qq] = 0.0;
		}
	}
	double r_g[kGPerThread];
	auto fetch = [&](uint32_t s0) {
#pragma unroll
		for (uint32_t j = 0; j < kGPerThread; j++) {
			const uint32_t e = threadIdx.x + 256u * j;
			const uint32_t s = s0 + e / kCols, c = e % kCols;
			r_g[j] = (s < sample_ct && c < n_cols) ? G[static_cast<uint64_t>(s) * g_stride + c] : 0.0;
		}
	};
	auto commit = [&]() {
#pragma unroll
		for (uint32_t j = 0; j < kGPerThread; j++) {
			const uint32_t e = threadIdx.x + 256u * j;
			s_g[e / kCols][e % kCols] = r_g[j];
		}
	};
	// blockIdx.y picks a run of sample chunks; with more than one run the partial sums of a
	// variant meet in `out` (zeroed by the launcher) through FP64 atomics
	const uint32_t all_chunks = (sample_ct + kChunk - 1) / kChunk;
	const uint32_t ch_begin = blockIdx.y * chunks_per_split;
	const uint32_t n_chunks = min(all_chunks, ch_begin + chunks_per_split);
	const bool split = gridDim.y > 1;
	const uint32_t lane_shift = 2u * lk; // sample 4q + k sits at bit 2*(4*(q&3) + k) of word q >> 2
	if (ch_begin >= n_chunks) {
		return;
	}
	fetch(ch_begin * kChunk);
	for (uint32_t ch = ch_begin; ch < n_chunks; ch++) {
		__syncthreads(); // everyone is done reading the previous chunk
		commit();
		__syncthreads();
		if (ch + 1 < n_chunks) {
			fetch((ch + 1) * kChunk);
		}
		// this lane's 128 calls (32 bytes) of each of its variants
		uint32_t w[kVT][8];
#pragma unroll
		for (uint32_t t = 0; t < kVT; t++) {
			const uint4 *p = reinterpret_cast<const uint4 *>(row_ptr[t] + static_cast<uint64_t>(ch) * (kChunk / 4));
			const uint4 a = p[0], b = p[1];
			w[t][0] = a.x;
			w[t][1] = a.y;
			w[t][2] = a.z;
			w[t][3] = a.w;
			w[t][4] = b.x;
			w[t][5] = b.y;
			w[t][6] = b.z;
			w[t][7] = b.w;
		}
#pragma unroll
		for (uint32_t q = 0; q < kChunk / 4; q++) {
			double b[NCT], bq[kQ];
#pragma unroll
			for (int c = 0; c < NCT; c++) {
				b[c] = s_g[4u * q + lk][16 * c + li];
			}
#pragma unroll
			for (int qq = 0; qq < NQ; qq++) {
				bq[qq] = s_g[4u * q + lk][16 * NCT + 4 * qq + (lane & 3u)];
			}
#pragma unroll
			for (uint32_t t = 0; t < kVT; t++) {
				const uint32_t g = __builtin_amdgcn_ubfe(w[t][q >> 2], 8u * (q & 3u) + lane_shift, 2u);
				const double a = s_t[t_base[t]][g];
#pragma unroll
				for (int c = 0; c < NCT; c++) {
					acc[t][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[c], acc[t][c], 0, 0, 0);
				}
#pragma unroll
				for (int qq = 0; qq < NQ; qq++) {
					accq[t][qq] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bq[qq], accq[t][qq], 0, 0, 0);
				}
			}
		}
	}
#pragma unroll
	for (uint32_t t = 0; t < kVT; t++) {
#pragma unroll
		for (int c = 0; c < NCT; c++) {
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const uint32_t v = v_wg + (wave * kVT + t) * 16u + lk + 4u * r;
				const uint32_t col = 16u * c + li;
				if (v < n_var && col < n_cols) {
					double *dst = out + static_cast<uint64_t>(v) * out_stride + col;
					if (split) {
						unsafeAtomicAdd(dst, acc[t][c][r]);
					} else {
						*dst = acc[t][c][r];
					}
				}
			}
		}
#pragma unroll
		for (int qq = 0; qq < NQ; qq++) {
			// D of the 4-block form: lane = 16 i + 4 b + j -> variant 4b + i of the tile, column j
			const uint32_t v = v_wg + (wave * kVT + t) * 16u + 4u * ((lane >> 2) & 3u) + (lane >> 4);
			const uint32_t col = 16u * NCT + 4u * qq + (lane & 3u);
			if (v < n_var && col < n_cols) {
				double *dst = out + static_cast<uint64_t>(v) * out_stride + col;
				if (split) {
					unsafeAtomicAdd(dst, accq[t][qq]);
				} else {
					*dst = accq[t][qq];
				}
			}
		}
	}
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

template <int NCOLS>
static hipError_t LaunchVariantReduceN(const RowView &view, const uint32_t *vlist, uint32_t n_var, const double *ts,
                                       const double *G, uint32_t g_stride, double *out, uint32_t out_stride,
                                       hipStream_t stream) {
	constexpr int VT = 8;
	hipLaunchKernelGGL((k_variant_reduce<NCOLS, VT>), dim3((n_var + VT - 1) / VT), dim3(256), 0, stream, view.rows,
	                   view.pitch, view.sample_ct, vlist, n_var, ts, G, g_stride, out, out_stride);
	return hipGetLastError();
}

hipError_t LaunchVariantReduce(const RowView &view, const uint32_t *vlist, uint32_t n_var, const double *ts,
                               const double *G, uint32_t g_stride, uint32_t n_cols, double *out, uint32_t out_stride,
                               hipStream_t stream) {
	if (n_var == 0) {
		return hipSuccess;
	}
	// The MFMA form reads whole 32-byte (128-call) pieces of a row: the row pitch must cover
	// ceil(N/128) of them, which holds for 128-byte-aligned pitches (rows >= 512 bytes).
	const bool mfma_ok = n_cols >= 3 && view.pitch % 32 == 0 &&
	                     static_cast<uint64_t>((view.sample_ct + 127) / 128) * 32 <= view.pitch;
	uint32_t c0 = 0;
	hipError_t e = hipSuccess;
	while (c0 < n_cols && e == hipSuccess) {
		const uint32_t left = n_cols - c0;
		if (mfma_ok) {
			// One workgroup per 128 variants is too coarse for a few hundred thousand variants
			// (781 workgroups on 256 CUs leave a quarter of the matrix pipes idle at the end):
			// split the sample axis until there are >= ~12 workgroups per CU.
			const uint32_t blocks = (n_var + 127) / 128;
			const uint32_t all_chunks = (view.sample_ct + 127) / 128;
			uint32_t splits = blocks >= 3072 ? 1 : (3072 + blocks - 1) / blocks;
			splits = std::min(splits, std::max(1u, all_chunks / 64)); // keep >= 64 chunks (8192 samples) per run
			const uint32_t chunks_per_split = (all_chunks + splits - 1) / splits;
			splits = (all_chunks + chunks_per_split - 1) / chunks_per_split;
			const uint32_t width = left > 28 ? 32 : (left > 24 ? 28 : (left > 20 ? 24 : (left > 16 ? 20 : 16)));
			if (splits > 1) {
				e = hipMemset2DAsync(out + c0, sizeof(double) * out_stride, 0, sizeof(double) * std::min(left, width),
				                     n_var, stream);
				if (e != hipSuccess) {
					break;
				}
			}
#define PGH_VR(NCT, NQ, WIDTH)                                                                                         \
	hipLaunchKernelGGL((k_variant_reduce_mfma<NCT, NQ>), dim3(blocks, splits), dim3(256), 0, stream, view.rows,        \
	                   view.pitch, view.sample_ct, vlist, n_var, ts, G + c0, g_stride,                                 \
	                   left < (WIDTH) ? left : (WIDTH), out + c0, out_stride, chunks_per_split);                       \
	c0 += (WIDTH)
			if (left > 28) {
				PGH_VR(2, 0, 32);
			} else if (left > 24) {
				PGH_VR(1, 3, 28);
			} else if (left > 20) {
				PGH_VR(1, 2, 24);
			} else if (left > 16) {
				PGH_VR(1, 1, 20);
			} else {
				PGH_VR(1, 0, 16);
			}
#undef PGH_VR
			e = hipGetLastError();
		} else if (left >= 8) {
			e = LaunchVariantReduceN<8>(view, vlist, n_var, ts, G + c0, g_stride, out + c0, out_stride, stream);
			c0 += 8;
		} else if (left >= 4) {
			e = LaunchVariantReduceN<4>(view, vlist, n_var, ts, G + c0, g_stride, out + c0, out_stride, stream);
			c0 += 4;
		} else if (left >= 2) {
			e = LaunchVariantReduceN<2>(view, vlist, n_var, ts, G + c0, g_stride, out + c0, out_stride, stream);
			c0 += 2;
		} else {
			e = LaunchVariantReduceN<1>(view, vlist, n_var, ts, G + c0, g_stride, out + c0, out_stride, stream);
			c0 += 1;
		}
	}
	return e;
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

// pca_i8.hip -- what plink_pca needs around the int8 contraction (score_i8.hip) to run BOTH of its products on
// the int8 matrix cores.
//
//   Step B / phase 3:  G2 = X^T Y,  BB = X^T U    -- sums over VARIANTS, one row of the packed matrix per term:
//                      k_score_i8 as plink_score uses it, weights = the dense factor, tables = NormalizeGenotypes.
//   Step A:            Y = X G1                   -- sums over SAMPLES.  The same kernel walks the TRANSPOSED packed
//                      matrix (sample-major, built once per pgh_pca call by k_transpose_2bit), and because the
//                      normalisation (g - c_v) s_v belongs to the OUTPUT row here, the two integer planes are
//                      multiplied out separately -- A = C G1 (codes) and Mm = Miss G1 (missing indicators) -- and
//                      k_pca_combine applies   Y[v] = s_v (A[v] - 3 Mm[v]) - c_v s_v (colsum(G1) - Mm[v])
//                      (a missing call contributes 0, src/plink_common.cpp:1535-1543: code 3 is taken back out).
#include "device_utils.hpp"
#include "kernels.hpp"

namespace pgh {

namespace {

// 256 variants x 256 samples per workgroup: 256 rows x 64 B in, 256 sample rows x 64 B out, through LDS.
__global__ __launch_bounds__(256) void k_transpose_2bit(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                        uint32_t sample_ct, const uint32_t *__restrict__ vlist,
                                                        uint32_t n_var, uint8_t *__restrict__ out, uint64_t out_pitch) {
	__shared__ __attribute__((aligned(16))) uint8_t s_in[256][64 + 16]; // +16: rows of one byte column spread over banks
	const uint32_t v0 = blockIdx.y * 256u, s0 = blockIdx.x * 256u;
	const uint32_t t = threadIdx.x;
	const uint64_t in_col = static_cast<uint64_t>(s0 / 4u) + 16u * (t & 3u);
#pragma unroll
	for (uint32_t n = 0; n < 4; n++) {
		const uint32_t r = (t >> 2) + 64u * n;
		uint4 v = make_uint4(0, 0, 0, 0);
		if (v0 + r < n_var && in_col < pitch) {
			v = *reinterpret_cast<const uint4 *>(rows + static_cast<uint64_t>(vlist[v0 + r]) * pitch + in_col);
		}
		*reinterpret_cast<uint4 *>(&s_in[r][16u * (t & 3u)]) = v;
	}
	__syncthreads();
	const uint32_t s = s0 + t;
	if (s >= sample_ct) {
		return;
	}
	const uint32_t byte = t >> 2, shift = 2u * (t & 3u);
	uint32_t w[16];
#pragma unroll
	for (uint32_t k = 0; k < 16; k++) {
		uint32_t acc = 0;
#pragma unroll
		for (uint32_t j = 0; j < 16; j++) {
			acc |= ((static_cast<uint32_t>(s_in[16u * k + j][byte]) >> shift) & 3u) << (2u * j);
		}
		w[k] = acc;
	}
	uint4 *dst = reinterpret_cast<uint4 *>(out + static_cast<uint64_t>(s) * out_pitch + (v0 / 4u));
#pragma unroll
	for (uint32_t q = 0; q < 4; q++) {
		if (v0 / 4u + 16u * q < out_pitch) {
			dst[q] = make_uint4(w[4 * q], w[4 * q + 1], w[4 * q + 2], w[4 * q + 3]);
		}
	}
}

__global__ __launch_bounds__(256) void k_iota(uint32_t *__restrict__ p, uint32_t n) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) {
		p[i] = i;
	}
}

// out[c] += sum over rows of m[r][c]  (n_cols <= 64 per launch; out zeroed by the caller)
__global__ __launch_bounds__(256) void k_colsum(const double *__restrict__ m, uint64_t n_rows, uint32_t stride,
                                                uint32_t n_cols, double *__restrict__ out) {
	__shared__ double s_part[4][64];
	const uint32_t c = threadIdx.x & 63u, lane_row = threadIdx.x >> 6;
	double acc = 0.0;
	if (c < n_cols) {
		for (uint64_t r = static_cast<uint64_t>(blockIdx.x) * 4u + lane_row; r < n_rows; r += static_cast<uint64_t>(gridDim.x) * 4u) {
			acc += m[r * stride + c];
		}
	}
	s_part[lane_row][c] = acc;
	__syncthreads();
	if (threadIdx.x < n_cols) {
		unsafeAtomicAdd(out + threadIdx.x,
		                (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]));
	}
}

__global__ __launch_bounds__(256) void k_pca_combine(const double *__restrict__ a, const double *__restrict__ mm,
                                                     const double *__restrict__ colsum,
                                                     const double *__restrict__ center,
                                                     const double *__restrict__ inv_stdev, uint64_t n_var, uint32_t n_cols,
                                                     double *__restrict__ y, uint32_t y_stride) {
	const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
	if (i >= n_var * n_cols) {
		return;
	}
	const uint64_t v = i / n_cols;
	const uint32_t c = static_cast<uint32_t>(i % n_cols);
	const double s = inv_stdev[v], cs = center[v] * s;
	const double m = mm[i];
	y[v * y_stride + c] = s * (a[i] - 3.0 * m) - cs * (colsum[c] - m);
}

} // namespace

uint64_t TransposedPitch(uint32_t n_var) {
	return (static_cast<uint64_t>((n_var + 3u) / 4u) + 63u) / 64u * 64u; // whole 64-byte pieces, zero padded
}

hipError_t LaunchTranspose2bit(const RowView &view, const uint32_t *vlist, uint32_t n_var, uint8_t *out,
                               hipStream_t stream) {
	if (n_var == 0 || view.sample_ct == 0) {
		return hipSuccess;
	}
	const uint64_t out_pitch = TransposedPitch(n_var);
	hipError_t e = hipMemsetAsync(out, 0, out_pitch * view.sample_ct, stream);
	if (e != hipSuccess) {
		return e;
	}
	hipLaunchKernelGGL(k_transpose_2bit, dim3((view.sample_ct + 255) / 256, (n_var + 255) / 256), dim3(256), 0, stream,
	                   view.rows, view.pitch, view.sample_ct, vlist, n_var, out, out_pitch);
	return hipGetLastError();
}

hipError_t LaunchIota(uint32_t *p, uint32_t n, hipStream_t stream) {
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, stream, p, n);
	return hipGetLastError();
}

hipError_t LaunchColumnSums(const double *m, uint64_t n_rows, uint32_t stride, uint32_t n_cols, double *out,
                            hipStream_t stream) {
	hipError_t e = hipMemsetAsync(out, 0, sizeof(double) * n_cols, stream);
	if (e != hipSuccess || n_rows == 0) {
		return e;
	}
	const uint64_t want = (n_rows + 3) / 4;
	for (uint32_t c0 = 0; c0 < n_cols; c0 += 64) { // 64 columns per launch
		hipLaunchKernelGGL(k_colsum, dim3(static_cast<uint32_t>(want < 2048 ? want : 2048)), dim3(256), 0, stream, m + c0,
		                   n_rows, stride, n_cols - c0 < 64 ? n_cols - c0 : 64, out + c0);
	}
	return hipGetLastError();
}

hipError_t LaunchPcaCombine(const double *a, const double *mm, const double *colsum, const double *center,
                            const double *inv_stdev, uint64_t n_var, uint32_t n_cols, double *y, uint32_t y_stride,
                            hipStream_t stream) {
	const uint64_t items = n_var * n_cols;
	if (items == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_pca_combine, dim3(static_cast<uint32_t>((items + 255) / 256)), dim3(256), 0, stream, a, mm,
	                   colsum, center, inv_stdev, n_var, n_cols, y, y_stride);
	return hipGetLastError();
}

} // namespace pgh

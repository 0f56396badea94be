// reduce.hip -- element-wise sums of per-shard partials (per-sample scores, allele counts, plink_pca's G2 / BB):
// the combine step of a shard group held by one process (api_sharded.cpp).  HBM-bound, 16 bytes per lane.
#include "device_utils.hpp"
#include "kernels.hpp"

namespace pgh {

namespace {

__global__ __launch_bounds__(256) void k_add_f64(double *__restrict__ dst, const double *__restrict__ src, uint64_t n) {
	const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256u * 2u;
	for (uint64_t i = (static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x) * 2u; i < n; i += stride) {
		if (i + 1 < n) {
			const double2 a = *reinterpret_cast<const double2 *>(dst + i);
			const double2 b = *reinterpret_cast<const double2 *>(src + i);
			*reinterpret_cast<double2 *>(dst + i) = make_double2(a.x + b.x, a.y + b.y);
		} else {
			dst[i] += src[i];
		}
	}
}

__global__ __launch_bounds__(256) void k_add_u32(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src,
                                                 uint64_t n) {
	const uint64_t stride = static_cast<uint64_t>(gridDim.x) * 256u;
	for (uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x; i < n; i += stride) {
		dst[i] += src[i];
	}
}

uint32_t Blocks(uint64_t items) {
	const uint64_t b = (items + 255) / 256;
	return static_cast<uint32_t>(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

} // namespace

hipError_t LaunchAddF64(double *dst, const double *src, uint64_t n, hipStream_t stream) {
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_add_f64, dim3(Blocks((n + 1) / 2)), dim3(256), 0, stream, dst, src, n);
	return hipGetLastError();
}

hipError_t LaunchAddU32(uint32_t *dst, const uint32_t *src, uint64_t n, hipStream_t stream) {
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_add_u32, dim3(Blocks(n)), dim3(256), 0, stream, dst, src, n);
	return hipGetLastError();
}

} // namespace pgh

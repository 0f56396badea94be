// device_utils.hpp -- small device-side helpers shared by the kernel files.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pgh {

namespace {

constexpr uint32_t kLow = 0x55555555u; // low bit of every 2-bit slot

__device__ __forceinline__ uint32_t WaveSum(uint32_t x) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		x += __shfl_xor(x, off, 64);
	}
	return x;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint4 LoadStream(const uint4 *p) {
	// once-read stream: non-temporal so it does not evict the mask / tables from L2
	const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
	return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ void StoreStream(uint4 *p, const uint4 &o) {
	u32x4 v = {o.x, o.y, o.z, o.w};
	__builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
}

typedef double f64x4 __attribute__((ext_vector_type(4)));

} // namespace

} // namespace pgh

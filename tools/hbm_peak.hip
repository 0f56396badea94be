// Attainable HBM rates of this box: a kernel that only streams 16 B per lane and xors (read), one
// that only stores 16 B per lane (write), and the unpack shape (1 byte read : 4.5 bytes written).
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UNROLL>
__global__ __launch_bounds__(256) void stream(const u32x4 *__restrict__ p, size_t n_vec, unsigned *out) {
	unsigned acc = 0;
	const size_t stride = static_cast<size_t>(gridDim.x) * 256 * UNROLL;
	for (size_t i = static_cast<size_t>(blockIdx.x) * 256 * UNROLL + threadIdx.x; i + 256 * (UNROLL - 1) < n_vec; i += stride) {
		u32x4 v[UNROLL];
#pragma unroll
		for (int k = 0; k < UNROLL; k++) {
			v[k] = __builtin_nontemporal_load(p + i + 256 * k);
		}
#pragma unroll
		for (int k = 0; k < UNROLL; k++) {
			acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
		}
	}
	if (acc == 0x12345678u) {
		out[0] = acc;
	}
}

__global__ __launch_bounds__(256) void fill(u32x4 *__restrict__ p, size_t n_vec, unsigned seed) {
	const size_t stride = static_cast<size_t>(gridDim.x) * 256;
	const u32x4 v = {seed, seed + 1, seed + 2, seed + 3};
	for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n_vec; i += stride) {
		__builtin_nontemporal_store(v, p + i);
	}
}

// one 4-byte load -> four 16-byte stores + 2 bytes, the traffic shape of k_unpack
__global__ __launch_bounds__(256) void expand(const unsigned *__restrict__ src, u32x4 *__restrict__ dst, size_t n_words) {
	const size_t stride = static_cast<size_t>(gridDim.x) * 256;
	for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n_words; i += stride) {
		const unsigned w = __builtin_nontemporal_load(src + i);
		const u32x4 v = {w & 0x03030303u, (w >> 2) & 0x03030303u, (w >> 4) & 0x03030303u, (w >> 6) & 0x03030303u};
		__builtin_nontemporal_store(v, dst + i);
	}
}

template <int UNROLL>
void run(const u32x4 *d, size_t n_vec, unsigned *d_out, int blocks) {
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipLaunchKernelGGL(stream<UNROLL>, dim3(blocks), dim3(256), 0, 0, d, n_vec, d_out);
	hipEventRecord(e0);
	for (int r = 0; r < 3; r++) {
		hipLaunchKernelGGL(stream<UNROLL>, dim3(blocks), dim3(256), 0, 0, d, n_vec, d_out);
	}
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	printf("unroll %d, %6d workgroups: %.2f TB/s\n", UNROLL, blocks, 3.0 * n_vec * 16 / ms / 1e9);
}

int main() {
	const size_t bytes = 64ull << 30;
	u32x4 *d;
	unsigned *d_out;
	hipMalloc(&d, bytes);
	hipMalloc(&d_out, 4);
	hipMemset(d, 0x5a, bytes);
	const size_t n_vec = bytes / 16;
	for (int blocks : {2048, 8192, 65536}) {
		run<4>(d, n_vec, d_out, blocks);
		run<8>(d, n_vec, d_out, blocks);
	}
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	float ms = 0;
	for (int blocks : {8192, 65536}) {
		hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 0, 0, d, n_vec, 1u);
		hipEventRecord(e0);
		for (int r = 0; r < 3; r++) {
			hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 0, 0, d, n_vec, 2u + r);
		}
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		hipEventElapsedTime(&ms, e0, e1);
		printf("write only, %6d workgroups: %.2f TB/s\n", blocks, 3.0 * n_vec * 16 / ms / 1e9);
	}
	{
		// 8 GB of packed input expands into 32 GB of output
		const size_t n_words = (8ull << 30) / 4;
		const unsigned *src = reinterpret_cast<const unsigned *>(d);
		u32x4 *dst = d + (16ull << 30) / 16;
		hipLaunchKernelGGL(expand, dim3(65536), dim3(256), 0, 0, src, dst, n_words);
		hipEventRecord(e0);
		for (int r = 0; r < 3; r++) {
			hipLaunchKernelGGL(expand, dim3(65536), dim3(256), 0, 0, src, dst, n_words);
		}
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		hipEventElapsedTime(&ms, e0, e1);
		printf("1 B read : 4 B written: %.2f TB/s total\n", 3.0 * n_words * 20 / ms / 1e9);
	}
	return 0;
}

// Attainable HBM read rate of this box: a kernel that only streams 16 B per lane and xors.
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
	return 0;
}

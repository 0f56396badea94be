// Attainable FP64 matrix-pipe rate: waves that do nothing but v_mfma_f64_16x16x4_f64 on
// independent accumulators (no memory traffic).  Prints TFLOP/s for 1..4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int ACCS>
__global__ __launch_bounds__(256) void spin(double *out, int iters, double a0, double b0) {
	f64x4 acc[ACCS];
	for (int i = 0; i < ACCS; i++) {
		acc[i] = f64x4 {0.0, 0.0, 0.0, 0.0};
	}
	double a = a0 + threadIdx.x, b = b0;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < ACCS; i++) {
			acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
		}
	}
	double s = 0.0;
	for (int i = 0; i < ACCS; i++) {
		s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	}
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
	double *d_out;
	const int cus = 256;
	hipMalloc(&d_out, sizeof(double) * cus * 16 * 256);
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const int iters = 20000;
	for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu++) {
		const int blocks = cus * wgs_per_cu;
		hipLaunchKernelGGL(spin<4>, dim3(blocks), dim3(256), 0, 0, d_out, 100, 1.0, 1.0);
		hipEventRecord(e0);
		hipLaunchKernelGGL(spin<4>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0, 1.0);
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms = 0;
		hipEventElapsedTime(&ms, e0, e1);
		const double flops = 2.0 * 16 * 16 * 4 * 4.0 * iters * (blocks * 4.0);
		printf("%d wave(s)/SIMD: %.2f ms, %.1f TFLOP/s\n", wgs_per_cu, ms, flops / ms / 1e9);
	}
	return 0;
}

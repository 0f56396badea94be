// What costs k_accumulate_mfma its last 30 %?  The pure-MFMA loop of mfma_peak.hip with the
// kernel's other per-group work added one piece at a time:
//   mode 0: 4 MFMAs per group, constant operands
//   mode 1: + 4 data-dependent ds_read_b64 lookups feeding A (12 integer VALU ops to form the addresses)
//   mode 2: + one 16-byte global load per group feeding the lookups
//   mode 3: + a workgroup barrier every 16 groups
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 4) void spin(double *out, const uint4 *rows, int iters, double b0) {
	__shared__ double s_t[64][4];
	f64x4 acc[4];
	for (int i = 0; i < 4; i++) {
		acc[i] = f64x4 {0.0, 0.0, 0.0, 0.0};
	}
	if (threadIdx.x < 256) {
		s_t[threadIdx.x >> 2][threadIdx.x & 3] = 1.0 + threadIdx.x;
	}
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63, shift = 2 * (lane & 15), lk = lane >> 4;
	uint4 w = {0x12345678u + lane, 0x9abcdef0u ^ lane, 0x0f1e2d3cu + lane, 0x4b5a6978u ^ lane};
	const uint4 *p = rows + (blockIdx.x * 256 + threadIdx.x);
	double b = b0;
	for (int it = 0; it < iters; it++) {
		double a[4] = {1.0, 2.0, 3.0, 4.0};
		if (MODE >= 2) {
			w = p[(it & 63) * 65536];
		}
		if (MODE >= 1) {
			const uint32_t k = (it * 4 + lk) & 63;
			a[0] = s_t[k][(w.x >> shift) & 3u];
			a[1] = s_t[k][(w.y >> shift) & 3u];
			a[2] = s_t[k][(w.z >> shift) & 3u];
			a[3] = s_t[k][(w.w >> shift) & 3u];
			if (MODE == 1) {
				w.x = w.x * 5 + 1; // keep the lookups data dependent without a load
			}
		}
#pragma unroll
		for (int i = 0; i < 4; i++) {
			acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b, acc[i], 0, 0, 0);
		}
		if (MODE >= 3 && (it & 15) == 15) {
			__syncthreads();
		}
	}
	double s = 0.0;
	for (int i = 0; i < 4; i++) {
		s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
	}
	out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(double *d_out, const uint4 *d_rows) {
	const int blocks = 256 * 4, iters = 20000;
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	hipLaunchKernelGGL(spin<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, d_rows, 100, 1.0);
	hipEventRecord(e0);
	hipLaunchKernelGGL(spin<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, d_rows, iters, 1.0);
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double flops = 2.0 * 16 * 16 * 4 * 4.0 * iters * (blocks * 4.0);
	printf("mode %d: %.2f ms, %.1f TFLOP/s\n", MODE, ms, flops / ms / 1e9);
}

int main() {
	double *d_out;
	uint4 *d_rows;
	hipMalloc(&d_out, sizeof(double) * 1024 * 256);
	hipMalloc(&d_rows, sizeof(uint4) * (64ull * 65536 + 1024 * 256));
	hipMemset(d_rows, 0x5a, sizeof(uint4) * (64ull * 65536 + 1024 * 256));
	run<0>(d_out, d_rows);
	run<1>(d_out, d_rows);
	run<2>(d_out, d_rows);
	run<3>(d_out, d_rows);
	return 0;
}

// Lane layout probe for v_mfma_f64_4x4x4_4b_f64 (4 blocks of 4x4x4): for every pair of lanes
// (la, lb), A is one-hot at la and B one-hot at lb; prints which D lane (if any) receives the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void probe(double *out) {
	const int pair = blockIdx.x; // la * 64 + lb
	const int la = pair >> 6, lb = pair & 63;
	const int lane = threadIdx.x;
	const double a = lane == la ? 1.0 : 0.0;
	const double b = lane == lb ? 1.0 : 0.0;
	double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
	out[pair * 64 + lane] = d;
}

int main() {
	double *d_out;
	hipMalloc(&d_out, sizeof(double) * 4096 * 64);
	hipLaunchKernelGGL(probe, dim3(4096), dim3(64), 0, 0, d_out);
	std::vector<double> h(4096 * 64);
	hipMemcpy(h.data(), d_out, sizeof(double) * h.size(), hipMemcpyDeviceToHost);
	for (int la = 0; la < 64; la++) {
		printf("A lane %2d:", la);
		for (int lb = 0; lb < 64; lb++) {
			for (int l = 0; l < 64; l++) {
				if (h[(la * 64 + lb) * 64 + l] != 0.0) {
					printf(" B%d->D%d", lb, l);
				}
			}
		}
		printf("\n");
	}
	return 0;
}

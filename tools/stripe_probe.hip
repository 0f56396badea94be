// stripe_probe.hip -- HBM read rate of a column-stripe walk: a workgroup owns W bytes of every row and walks the
// rows of its slice, the access shape of every kernel that reduces over variants into per-sample registers
// (k_score_i8, k_class_cols*).  Rows are `pitch` bytes apart (the 500,000-sample record: 125,056 B).  Each thread
// keeps four 16-byte loads in flight; a trip of a 256-thread workgroup covers 16 KB = 16384 / W rows.
// Prints TB/s per stripe width, workgroup size and dispatch order (stripe index fastest or slice fastest).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int THREADS>
__global__ __launch_bounds__(THREADS) void walk(const unsigned char *__restrict__ rows, size_t pitch, unsigned n_rows,
                                                 unsigned w_bytes, unsigned rows_per_slice, int slice_fast,
                                                 unsigned n_stripes, unsigned *out) {
	const unsigned stripe = slice_fast ? blockIdx.x / ((n_rows + rows_per_slice - 1) / rows_per_slice) : blockIdx.x % n_stripes;
	const unsigned slice = slice_fast ? blockIdx.x % ((n_rows + rows_per_slice - 1) / rows_per_slice) : blockIdx.x / n_stripes;
	const unsigned chunks = w_bytes / 16;              // 16-byte chunks per row of the stripe
	const unsigned rows_per_pass = THREADS / chunks;   // rows one load instruction of the workgroup covers
	const unsigned r_in = threadIdx.x / chunks, c = threadIdx.x % chunks;
	const size_t col = static_cast<size_t>(stripe) * w_bytes + 16u * c;
	if (col + 16 > pitch) {
		return;
	}
	const unsigned r0 = slice * rows_per_slice, r1 = min(r0 + rows_per_slice, n_rows);
	unsigned acc = 0;
	for (unsigned r = r0 + r_in; r + 3 * rows_per_pass < r1; r += 4 * rows_per_pass) {
		u32x4 v[4];
#pragma unroll
		for (int k = 0; k < 4; k++) {
			v[k] = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(rows + static_cast<size_t>(r + k * rows_per_pass) * pitch + col));
		}
#pragma unroll
		for (int k = 0; k < 4; k++) {
			acc ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
		}
	}
	if (acc == 0x12345678u) {
		out[0] = acc;
	}
}

__global__ __launch_bounds__(256) void fill(u32x4 *__restrict__ p, size_t n_vec) {
	const size_t stride = static_cast<size_t>(gridDim.x) * 256;
	for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n_vec; i += stride) {
		const unsigned s = static_cast<unsigned>(i) * 2654435761u;
		const u32x4 v = {s, s + 1, s + 2, s + 3};
		__builtin_nontemporal_store(v, p + i);
	}
}

template <int THREADS>
static void Run(const unsigned char *d, size_t pitch, unsigned n_rows, unsigned w, unsigned rows_per_slice, int slice_fast,
                unsigned *d_out) {
	const unsigned n_stripes = static_cast<unsigned>((pitch + w - 1) / w);
	const unsigned n_slices = (n_rows + rows_per_slice - 1) / rows_per_slice;
	hipEvent_t e0, e1;
	hipEventCreate(&e0);
	hipEventCreate(&e1);
	const dim3 grid(n_stripes * n_slices);
	hipLaunchKernelGGL(walk<THREADS>, grid, dim3(THREADS), 0, 0, d, pitch, n_rows, w, rows_per_slice, slice_fast, n_stripes, d_out);
	hipEventRecord(e0);
	for (int r = 0; r < 2; r++) {
		hipLaunchKernelGGL(walk<THREADS>, grid, dim3(THREADS), 0, 0, d, pitch, n_rows, w, rows_per_slice, slice_fast, n_stripes, d_out);
	}
	hipEventRecord(e1);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	std::printf("stripe %5u B  threads %4d  rows/slice %6u  %-12s %6u workgroups: %.2f TB/s\n", w, THREADS, rows_per_slice,
	            slice_fast ? "slice-fast" : "stripe-fast", grid.x, 2.0 * n_rows * static_cast<double>(pitch) / ms / 1e9);
	hipEventDestroy(e0);
	hipEventDestroy(e1);
}

int main() {
	const size_t pitch = 125056;
	const unsigned n_rows = 262144; // 32.8 GB: far beyond the Infinity Cache
	unsigned char *d = nullptr;
	unsigned *d_out = nullptr;
	if (hipMalloc(reinterpret_cast<void **>(&d), pitch * n_rows) != hipSuccess || hipMalloc(reinterpret_cast<void **>(&d_out), 64) != hipSuccess) {
		std::fprintf(stderr, "hipMalloc failed\n");
		return 2;
	}
	hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, reinterpret_cast<u32x4 *>(d), pitch * n_rows / 16);
	hipDeviceSynchronize();
	for (unsigned w : {64u, 128u, 256u, 512u, 1024u, 2048u, 4096u}) {
		for (unsigned rps : {4096u, 16384u}) {
			for (int sf = 0; sf < 2; sf++) {
				Run<256>(d, pitch, n_rows, w, rps, sf, d_out);
			}
		}
	}
	for (unsigned w : {512u, 1024u, 4096u}) {
		Run<512>(d, pitch, n_rows, w, 16384, 0, d_out);
		Run<1024>(d, pitch, n_rows, w, 16384, 0, d_out);
	}
	hipFree(d);
	hipFree(d_out);
	return 0;
}

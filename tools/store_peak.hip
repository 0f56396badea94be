// store_peak.hip -- what HBM takes from a store-only kernel on this box, by store form.  k_unpack_wide writes 4.5
// bytes per byte it reads; this is the ceiling it is measured against.
//   forms: 16 B per lane non-temporal / plain / sc1 ("write-through"), 4 B and 8 B per lane plain;
//   shapes: grid-stride over one 32 GB buffer, 1 KiB per wave-instruction, 1-8 stores in flight per lane;
//   and the unpack's own traffic shape: 16 B read -> 64 B + 8 B written per lane.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <int FORM, int UNROLL>
__global__ __launch_bounds__(256) void store16(u32x4 *__restrict__ p, size_t n_vec, unsigned seed) {
	const size_t stride = static_cast<size_t>(gridDim.x) * 256 * UNROLL;
	const u32x4 v = {seed, seed ^ threadIdx.x, seed + 2, seed + 3};
	for (size_t i = static_cast<size_t>(blockIdx.x) * 256 * UNROLL + threadIdx.x; i + 256 * (UNROLL - 1) < n_vec; i += stride) {
#pragma unroll
		for (int k = 0; k < UNROLL; k++) {
			u32x4 *q = p + i + 256 * k;
			if (FORM == 0) {
				__builtin_nontemporal_store(v, q);
			} else if (FORM == 1) {
				*q = v;
			} else {
				asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(q), "v"(v) : "memory");
			}
		}
	}
}

template <class T, int UNROLL>
__global__ __launch_bounds__(256) void store_narrow(T *__restrict__ p, size_t n, unsigned seed) {
	const size_t stride = static_cast<size_t>(gridDim.x) * 256 * UNROLL;
	T v;
	__builtin_memset(&v, seed & 0xff, sizeof v);
	for (size_t i = static_cast<size_t>(blockIdx.x) * 256 * UNROLL + threadIdx.x; i + 256 * (UNROLL - 1) < n; i += stride) {
#pragma unroll
		for (int k = 0; k < UNROLL; k++) {
			p[i + 256 * k] = v;
		}
	}
}

// the unpack's shape: a lane reads 16 B and writes 4 x 16 B (1 KiB per wave-instruction) + 8 B
template <bool NT>
__global__ __launch_bounds__(256) void expand(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, unsigned long long *__restrict__ val,
                                               size_t n_vec) {
	const size_t stride = static_cast<size_t>(gridDim.x) * 256;
	for (size_t i = static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x; i < n_vec; i += stride) {
		const u32x4 w = __builtin_nontemporal_load(src + i);
		const size_t wave0 = (i & ~static_cast<size_t>(63)) * 4, lane = i & 63;
#pragma unroll
		for (int k = 0; k < 4; k++) {
			const u32x4 o = {w.x + k, w.y, w.z, w.w};
			if (NT) {
				__builtin_nontemporal_store(o, dst + wave0 + 64 * k + lane);
			} else {
				dst[wave0 + 64 * k + lane] = o;
			}
		}
		val[i] = w.x;
	}
}

template <class F>
static void Time(const char *what, double bytes, F launch) {
	hipEvent_t e0, e1;
	(void)hipEventCreate(&e0);
	(void)hipEventCreate(&e1);
	launch();
	(void)hipEventRecord(e0);
	for (int r = 0; r < 3; r++) {
		launch();
	}
	(void)hipEventRecord(e1);
	(void)hipEventSynchronize(e1);
	float ms = 0;
	(void)hipEventElapsedTime(&ms, e0, e1);
	std::printf("%-58s %.2f TB/s\n", what, 3.0 * bytes / ms / 1e9);
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
}

int main() {
	const size_t bytes = 32ull << 30;
	u32x4 *d = nullptr;
	if (hipMalloc(reinterpret_cast<void **>(&d), bytes + (8ull << 30)) != hipSuccess) {
		return 2;
	}
	const size_t n_vec = bytes / 16;
	for (int blocks : {4096, 16384, 65536}) {
		char name[96];
#define ROW(FORM, UNROLL, LABEL)                                                                                       \
	std::snprintf(name, sizeof name, "16 B/lane %-12s unroll %d, %6d workgroups", LABEL, UNROLL, blocks);              \
	Time(name, static_cast<double>(bytes), [&] { hipLaunchKernelGGL((store16<FORM, UNROLL>), dim3(blocks), dim3(256), 0, 0, d, n_vec, 7u); })
		ROW(0, 1, "non-temporal");
		ROW(0, 4, "non-temporal");
		ROW(0, 8, "non-temporal");
		ROW(1, 1, "plain");
		ROW(1, 4, "plain");
		ROW(2, 4, "sc1");
#undef ROW
	}
	Time("4 B/lane plain, unroll 4, 16384 workgroups", static_cast<double>(bytes), [&] {
		hipLaunchKernelGGL((store_narrow<unsigned, 4>), dim3(16384), dim3(256), 0, 0, reinterpret_cast<unsigned *>(d), bytes / 4, 7u);
	});
	Time("8 B/lane plain, unroll 4, 16384 workgroups", static_cast<double>(bytes), [&] {
		hipLaunchKernelGGL((store_narrow<u32x2, 4>), dim3(16384), dim3(256), 0, 0, reinterpret_cast<u32x2 *>(d), bytes / 8, 7u);
	});
	{
		// 6 GB of packed input -> 24 GB of calls + 3 GB of validity words (the unpack's 1 : 4 : 0.5)
		const size_t n_in = (6ull << 30) / 16;
		const u32x4 *src = d;
		u32x4 *dst = d + (8ull << 30) / 16;
		unsigned long long *val = reinterpret_cast<unsigned long long *>(d + (33ull << 30) / 16);
		const double total = n_in * (16.0 + 64.0 + 8.0);
		for (int blocks : {16384, 65536}) {
			char name[96];
			std::snprintf(name, sizeof name, "16 B read -> 64 B + 8 B written, non-temporal, %6d wg", blocks);
			Time(name, total, [&] { hipLaunchKernelGGL(expand<true>, dim3(blocks), dim3(256), 0, 0, src, dst, val, n_in); });
			std::snprintf(name, sizeof name, "16 B read -> 64 B + 8 B written, plain,        %6d wg", blocks);
			Time(name, total, [&] { hipLaunchKernelGGL(expand<false>, dim3(blocks), dim3(256), 0, 0, src, dst, val, n_in); });
		}
	}
	(void)hipFree(d);
	return 0;
}

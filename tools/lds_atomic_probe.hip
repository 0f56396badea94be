// lds_atomic_probe.hip -- what an LDS atomic add costs on gfx950, by type, address pattern and lane occupancy.
// k_score_dosage_fix scatters one FP64 term per explicit dosage into a 4096-sample tile in LDS (lane = 64-sample
// word, column = the sample's bit) and was LDS-bound; this is the price list it is
// redesigned against.  16 waves per workgroup, one workgroup per CU, all CUs; cycles are per wave-instruction per CU
// (LDS-pipe view: the 16 waves share it).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                                       \
	do {                                                                                                               \
		hipError_t e_ = (x);                                                                                           \
		if (e_ != hipSuccess) {                                                                                        \
			std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                             \
			std::exit(1);                                                                                              \
		}                                                                                                              \
	} while (0)

enum { kRowPad65, kColMajor, kLinear };

__device__ inline uint32_t Mix(uint32_t x) {
	x ^= x >> 16;
	x *= 0x7feb352dU;
	x ^= x >> 15;
	x *= 0x846ca68bU;
	x ^= x >> 16;
	return x;
}

// T = double / unsigned long long / unsigned / float.  Every lane adds to "its" row (lane) at a pseudo-random column b
// of a 64 x 64 tile; PATTERN picks where (row, column) lives.  HALF: a pseudo-random half of the lanes sits out.
template <class T, int PATTERN, bool HALF>
__global__ __launch_bounds__(1024) void probe(int iters, T *__restrict__ sink) {
	__shared__ T s_tile[64 * 65];
	for (uint32_t t = threadIdx.x; t < 64 * 65; t += 1024) {
		s_tile[t] = T(0);
	}
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t h = Mix(threadIdx.x * 2654435761u + blockIdx.x);
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int k = 0; k < 8; k++) {
			h = h * 1664525u + 1013904223u;
			const uint32_t b = h >> 26;
			const bool on = !HALF || ((h >> 13) & 1u);
			uint32_t at;
			if (PATTERN == kRowPad65) {
				at = lane * 65u + b;
			} else if (PATTERN == kColMajor) {
				at = b * 64u + lane;
			} else {
				at = ((it * 8 + k) & 63u) * 64u + lane;
			}
			if (on) {
				atomicAdd(&s_tile[at], T(1));
			}
		}
	}
	__syncthreads();
	if (s_tile[threadIdx.x] == T(123456789)) {
		sink[0] = T(1);
	}
}

// the non-atomic alternative when a tile is private to one wave: read, add, write
template <class T, int PATTERN>
__global__ __launch_bounds__(1024) void probe_rmw(int iters, T *__restrict__ sink) {
	__shared__ T s_tile[64 * 65];
	for (uint32_t t = threadIdx.x; t < 64 * 65; t += 1024) {
		s_tile[t] = T(0);
	}
	__syncthreads();
	const uint32_t lane = threadIdx.x & 63u;
	uint32_t h = Mix(threadIdx.x * 2654435761u + blockIdx.x);
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int k = 0; k < 8; k++) {
			h = h * 1664525u + 1013904223u;
			const uint32_t b = h >> 26;
			const uint32_t at = PATTERN == kRowPad65 ? lane * 65u + b : b * 64u + lane;
			s_tile[at] = s_tile[at] + T(1);   // (racy across waves: only the cost is of interest)
		}
	}
	__syncthreads();
	if (s_tile[threadIdx.x] == T(123456789)) {
		sink[0] = T(1);
	}
}

template <class K>
static double Time(K launch, int iters) {
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	launch(iters / 8);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(e0));
	launch(iters);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	// ns per wave-instruction per CU: a CU runs 16 waves x 8 x iters instructions
	return static_cast<double>(ms) * 1e6 / (16.0 * 8.0 * iters);
}

template <class T>
static void Type(const char *name, void *d_sink) {
	T *sink = static_cast<T *>(d_sink);
	const int iters = 20000;
	const double a = Time([&](int n) { probe<T, kRowPad65, false><<<256, 1024>>>(n, sink); }, iters);
	const double b = Time([&](int n) { probe<T, kColMajor, false><<<256, 1024>>>(n, sink); }, iters);
	const double c = Time([&](int n) { probe<T, kLinear, false><<<256, 1024>>>(n, sink); }, iters);
	const double ah = Time([&](int n) { probe<T, kRowPad65, true><<<256, 1024>>>(n, sink); }, iters);
	const double bh = Time([&](int n) { probe<T, kColMajor, true><<<256, 1024>>>(n, sink); }, iters);
	const double ra = Time([&](int n) { probe_rmw<T, kRowPad65><<<256, 1024>>>(n, sink); }, iters);
	const double rb = Time([&](int n) { probe_rmw<T, kColMajor><<<256, 1024>>>(n, sink); }, iters);
	std::printf("%-10s row*65+b %6.2f   b*64+row %6.2f   linear %6.2f   half lanes: %6.2f / %6.2f   read-add-write: %6.2f / %6.2f\n",
	            name, a, b, c, ah, bh, ra, rb);
}

int main() {
	void *d_sink;
	CHECK(hipMalloc(&d_sink, 64));
	std::printf("ns per wave-instruction per CU (16 waves on the CU; x ~2.1 GHz = LDS-pipe cycles)\n");
	Type<double>("add f64", d_sink);
	Type<unsigned long long>("add u64", d_sink);
	Type<float>("add f32", d_sink);
	Type<unsigned>("add u32", d_sink);
	return 0;
}

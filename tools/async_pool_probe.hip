// async_pool_probe.hip -- does data a kernel wrote into a hipMallocAsync block survive until the next kernel on the
// same stream reads it?  The library's pgh_missing_per_sample lost whole slices of such a block about once in thirty
// calls made next to table-function scan threads (profiles/r02_async_pool_ab.txt).  This is that call's shape with
// the library taken away: per call { hipMalloc out; hipMallocAsync scratch; memset out; k_write fills scratch slice
// by slice; k_sum adds the slices into out; hipFreeAsync scratch; copy out; sync; hipFree out }, on the calling
// thread's stream, with short-lived helper threads doing their own hipMalloc / kernel / copy / hipFree rounds in
// between (a scan pool coming and going).  First argument: "pool" (default), "malloc" for the control -- the same loop
// with the scratch from hipMalloc --, "keep" for the pool with its release threshold at the maximum.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHECK(x)                                                                                                       \
	do {                                                                                                               \
		hipError_t e_ = (x);                                                                                           \
		if (e_ != hipSuccess) {                                                                                        \
			std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                             \
			std::exit(1);                                                                                              \
		}                                                                                                              \
	} while (0)

constexpr unsigned kCols = 47, kPlanes = 32, kSlices = 417; // the failing call's numbers: 3,001 samples, 40,000 rows

__global__ __launch_bounds__(1024) void k_write(unsigned *__restrict__ scratch, unsigned spin) {
	const unsigned col = threadIdx.x;
	if (col >= kCols) {
		return;
	}
	unsigned v = blockIdx.x * 2654435761u + col;
	for (unsigned i = 0; i < spin; i++) { // some work per slice, as the tally has
		v = v * 1664525u + 1013904223u;
	}
	unsigned *dst = scratch + static_cast<size_t>(blockIdx.x) * kPlanes * kCols + col;
	for (unsigned p = 0; p < kPlanes; p++) {
		dst[p * kCols] = 1u + ((v >> p) & 0u); // every word 1: the sum is known
	}
}

__global__ __launch_bounds__(256) void k_sum(const unsigned *__restrict__ scratch, unsigned *__restrict__ out) {
	const unsigned col = threadIdx.x;
	if (col >= kCols) {
		return;
	}
	unsigned acc = 0;
	for (unsigned y = blockIdx.y * 16u; y < min(blockIdx.y * 16u + 16u, kSlices); y++) {
		const unsigned *sl = scratch + static_cast<size_t>(y) * kPlanes * kCols + col;
		for (unsigned p = 0; p < kPlanes; p++) {
			acc += __builtin_nontemporal_load(sl + p * kCols);
		}
	}
	atomicAdd(out + col, acc);
}

static void Helper(int rounds) {
	for (int r = 0; r < rounds; r++) {
		void *d = nullptr;
		CHECK(hipMalloc(&d, 32 << 10));
		CHECK(hipMemsetAsync(d, 0, 32 << 10, hipStreamPerThread));
		std::vector<unsigned> h(8192);
		CHECK(hipMemcpyAsync(h.data(), d, 32 << 10, hipMemcpyDeviceToHost, hipStreamPerThread));
		CHECK(hipStreamSynchronize(hipStreamPerThread));
		CHECK(hipFree(d));
	}
}

int main(int argc, char **argv) {
	const bool use_pool = !(argc > 1 && std::strcmp(argv[1], "malloc") == 0);
	if (argc > 1 && std::strcmp(argv[1], "keep") == 0) {
		// "keep": the pool never gives memory back at a synchronisation point (release threshold = max) -- does the
		// loss need a block that was released and mapped again?
		hipMemPool_t pool;
		CHECK(hipDeviceGetDefaultMemPool(&pool, 0));
		uint64_t keep = ~0ull;
		CHECK(hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &keep));
	}
	const int calls = argc > 2 ? std::atoi(argv[2]) : 600;
	const size_t scratch_bytes = static_cast<size_t>(kSlices) * kPlanes * kCols * 4u;
	const unsigned want = kSlices * kPlanes;
	int bad_calls = 0;
	hipStream_t st = hipStreamPerThread;
	for (int c = 0; c < calls; c++) {
		if (c % 6 == 0) { // a scan pool comes and goes
			std::vector<std::thread> pool;
			const int n = (c / 6) % 2 ? 6 : 1;
			for (int t = 0; t < n; t++) {
				pool.emplace_back(Helper, 20);
			}
			for (auto &t : pool) {
				t.join();
			}
		}
		unsigned *out = nullptr;
		void *scratch = nullptr;
		CHECK(hipMalloc(reinterpret_cast<void **>(&out), 12 << 10));
		if (use_pool) {
			CHECK(hipMallocAsync(&scratch, scratch_bytes, st));
		} else {
			CHECK(hipMalloc(&scratch, scratch_bytes));
		}
		CHECK(hipMemsetAsync(out, 0, kCols * 4u, st));
		k_write<<<kSlices, 1024, 0, st>>>(static_cast<unsigned *>(scratch), 2000u);
		k_sum<<<dim3(1, (kSlices + 15) / 16), 256, 0, st>>>(static_cast<unsigned *>(scratch), out);
		CHECK(hipGetLastError());
		if (use_pool) {
			CHECK(hipFreeAsync(scratch, st));
		}
		std::vector<unsigned> h(kCols);
		CHECK(hipMemcpyAsync(h.data(), out, kCols * 4u, hipMemcpyDeviceToHost, st));
		CHECK(hipStreamSynchronize(st));
		if (!use_pool) {
			CHECK(hipFree(scratch));
		}
		CHECK(hipFree(out));
		unsigned worst = want;
		for (unsigned v : h) {
			worst = v < worst ? v : worst;
		}
		if (worst != want) {
			bad_calls++;
			if (bad_calls <= 8) {
				std::printf("call %d: a column sums to %u of %u (%.1f %% short)\n", c, worst, want, 100.0 * (want - worst) / want);
			}
		}
	}
	std::printf("%s scratch: %d of %d calls wrong\n", use_pool ? "hipMallocAsync" : "hipMalloc", bad_calls, calls);
	return 0;
}

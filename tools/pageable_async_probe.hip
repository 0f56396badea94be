// pageable_async_probe.hip -- does hipMemcpyAsync(host-pageable -> device) read its source before it returns?
//
// Round 1 saw pgh_ld_pairs_dev's task list arrive on the device as zeros.  The call sequence was
//   hipMallocAsync(d, st); hipMemcpyAsync(d, vector.data(), H2D, st); kernel<<<st>>>(d); hipFreeAsync(d, st); return;
// with `vector` a frame-local std::vector (pageable) that died at `return`.  CUDA documents a pageable H2D
// async copy as staged before the call returns; HIP makes no such promise.  This probe settles it on the box:
// queue a long kernel on the stream, issue the async copy from a pageable buffer holding a pattern, overwrite
// the buffer with zeros the moment the call returns, and look at what reached the device.
//   "deferred" = the device received the overwritten bytes => the runtime read the source after returning,
//   so a source that dies with the caller's frame is a use-after-free in the caller (not a pool defect).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

#define CK(x)                                                                                                          \
	do {                                                                                                               \
		hipError_t e_ = (x);                                                                                           \
		if (e_ != hipSuccess) {                                                                                        \
			std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                                               \
			return 2;                                                                                                  \
		}                                                                                                              \
	} while (0)

__global__ void k_spin(long long cycles, int *sink) {
	const long long t0 = wall_clock64();
	while (wall_clock64() - t0 < cycles) {
	}
	if (sink && threadIdx.x == 12345) {
		*sink = 1;
	}
}

// consumer in the same stream, as k_ld_pairs was: what does a KERNEL queued right behind the copy see?
__global__ void k_consume(const uint32_t *src, uint32_t n, uint32_t *first_bad, uint32_t *zero_ct) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		const uint32_t v = src[i];
		if (v != 0xabcd0000u + (i & 0xffff)) {
			atomicMin(first_bad, i);
			if (v == 0) {
				atomicAdd(zero_ct, 1u);
			}
		}
	}
}
__global__ void k_touch(uint32_t *p, uint32_t n) {
	const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i < n) {
		p[i] += 1;
	}
}

// Part 4: round 1's call shape as closely as the evidence gives it.  A scan thread of an earlier query
// (its own hipStreamPerThread) takes pool blocks, zeroes one (plink_score's `miss`), runs kernels on them,
// frees them stream-ordered, synchronises and exits.  A NEW thread then does what pgh_ld_pairs_dev did:
// hipMallocAsync on ITS hipStreamPerThread, hipMemcpyAsync from a frame-local pageable vector, a kernel
// that reads the block, hipFreeAsync, and the vector dies.
#include <thread>
static int Part4(int iters, size_t bytes, int *failures, int *reused) {
	uint32_t *d_flags = nullptr;
	CK(hipMalloc(reinterpret_cast<void **>(&d_flags), 8));
	for (int it = 0; it < iters; it++) {
		void *earlier = nullptr;
		int rc1 = 0;
		std::thread t1([&] {
			hipStream_t st = hipStreamPerThread;
			void *scratch = nullptr, *miss = nullptr;
			if (hipMallocAsync(&scratch, 4096, st) != hipSuccess || hipMallocAsync(&miss, bytes, st) != hipSuccess) {
				rc1 = 1;
				return;
			}
			(void)hipMemsetAsync(miss, 0, bytes, st);
			hipLaunchKernelGGL(k_touch, dim3((bytes / 4 + 255) / 256), dim3(256), 0, st, static_cast<uint32_t *>(miss),
			                   static_cast<uint32_t>(bytes / 4));
			(void)hipMemsetAsync(miss, 0, bytes, st);
			(void)hipFreeAsync(scratch, st);
			(void)hipFreeAsync(miss, st);
			earlier = miss;
			if (it & 1) {
				(void)hipStreamSynchronize(st); // pgh_score_dev synchronised before it returned; try both
			}
		});
		t1.join();
		if (rc1) {
			return 2;
		}
		int rc2 = 0;
		std::thread t2([&] {
			hipStream_t st = hipStreamPerThread;
			uint32_t init[2] = {0xffffffffu, 0u};
			(void)hipMemcpy(d_flags, init, 8, hipMemcpyHostToDevice);
			void *d = nullptr;
			{
				std::vector<uint32_t> tasks(bytes / 4);
				for (size_t i = 0; i < tasks.size(); i++) {
					tasks[i] = 0xabcd0000u + static_cast<uint32_t>(i & 0xffff);
				}
				if (hipMallocAsync(&d, bytes, st) != hipSuccess) {
					rc2 = 1;
					return;
				}
				*reused += d == earlier;
				(void)hipMemcpyAsync(d, tasks.data(), bytes, hipMemcpyHostToDevice, st);
				hipLaunchKernelGGL(k_consume, dim3((bytes / 4 + 255) / 256), dim3(256), 0, st,
				                   static_cast<const uint32_t *>(d), static_cast<uint32_t>(bytes / 4), d_flags, d_flags + 1);
				(void)hipFreeAsync(d, st);
			} // the vector dies here, as it did at pgh_ld_pairs_dev's return
			std::vector<uint32_t> churn(bytes / 4, 0u); // and its storage is reused at once
			(void)hipStreamSynchronize(st);
			uint32_t got[2];
			(void)hipMemcpy(got, d_flags, 8, hipMemcpyDeviceToHost);
			if (got[0] != 0xffffffffu) {
				*failures += 1;
				std::printf("  iteration %d: kernel saw a wrong word at %u (%u zero words)\n", it, got[0], got[1]);
			}
			(void)churn;
		});
		t2.join();
		if (rc2) {
			return 2;
		}
	}
	CK(hipFree(d_flags));
	return 0;
}

int main() {
	const size_t sizes[] = {16, 4096, 65536, 1u << 20, 16u << 20, 64u << 20};
	hipStream_t st;
	CK(hipStreamCreate(&st));
	std::printf("%-10s %-6s %-6s %-10s\n", "bytes", "pool", "busy", "verdict");
	for (int pool = 0; pool < 2; pool++) {
		for (int busy = 0; busy < 2; busy++) {
			for (size_t bytes : sizes) {
				const size_t n = bytes / 4;
				void *d = nullptr;
				if (pool) {
					CK(hipMallocAsync(&d, bytes, st));
				} else {
					CK(hipMalloc(&d, bytes));
				}
				CK(hipMemsetAsync(d, 0xff, bytes, st));
				CK(hipStreamSynchronize(st));
				uint32_t *src = static_cast<uint32_t *>(std::malloc(bytes));
				for (size_t i = 0; i < n; i++) {
					src[i] = 0xabcd0000u + static_cast<uint32_t>(i & 0xffff);
				}
				if (busy) {
					hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, 5000000LL /* 50 ms at 100 MHz */, nullptr);
				}
				CK(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, st));
				std::memset(src, 0, bytes); // what a dying std::vector's storage may look like a moment later
				CK(hipStreamSynchronize(st));
				std::vector<uint32_t> back(n);
				CK(hipMemcpy(back.data(), d, bytes, hipMemcpyDeviceToHost));
				size_t zeros = 0, good = 0;
				for (size_t i = 0; i < n; i++) {
					zeros += back[i] == 0;
					good += back[i] == 0xabcd0000u + static_cast<uint32_t>(i & 0xffff);
				}
				const char *verdict = good == n ? "staged" : (zeros == n ? "DEFERRED(all)" : "DEFERRED(part)");
				std::printf("%-10zu %-6d %-6d %-10s  good=%zu zeros=%zu of %zu\n", bytes, pool, busy, verdict, good, zeros, n);
				std::free(src);
				if (pool) {
					CK(hipFreeAsync(d, st));
				} else {
					CK(hipFree(d));
				}
			}
		}
	}
	CK(hipStreamSynchronize(st));

	// Part 2: is the pageable copy executed in STREAM ORDER?  Queue a long kernel, then a memset of the
	// destination to zero, then the copy of a pattern -- all on one stream.  Stream order leaves the pattern.
	// Zeros mean the copy's write reached the device BEFORE the memset queued ahead of it ran: the runtime
	// performed the copy out of order.  (That is round 1's sequence: a pool block whose previous user's
	// hipMemsetAsync / kernels were still queued on the stream when the block was handed out again.)
	std::printf("\n%-10s %-6s %-12s %-10s\n", "bytes", "pool", "call_ms", "order");
	for (int pool = 0; pool < 2; pool++) {
		for (size_t bytes : sizes) {
			const size_t n = bytes / 4;
			void *d = nullptr;
			if (pool) {
				CK(hipMallocAsync(&d, bytes, st));
			} else {
				CK(hipMalloc(&d, bytes));
			}
			CK(hipStreamSynchronize(st));
			uint32_t *src = static_cast<uint32_t *>(std::malloc(bytes));
			for (size_t i = 0; i < n; i++) {
				src[i] = 0xabcd0000u + static_cast<uint32_t>(i & 0xffff);
			}
			hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, 5000000LL, nullptr);
			CK(hipMemsetAsync(d, 0, bytes, st));
			timespec t0, t1;
			clock_gettime(CLOCK_MONOTONIC, &t0);
			CK(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, st));
			clock_gettime(CLOCK_MONOTONIC, &t1);
			CK(hipStreamSynchronize(st));
			std::vector<uint32_t> back(n);
			CK(hipMemcpy(back.data(), d, bytes, hipMemcpyDeviceToHost));
			size_t zeros = 0, good = 0;
			for (size_t i = 0; i < n; i++) {
				zeros += back[i] == 0;
				good += back[i] == 0xabcd0000u + static_cast<uint32_t>(i & 0xffff);
			}
			const double ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_nsec - t0.tv_nsec) * 1e-6;
			std::printf("%-10zu %-6d %-12.3f %-10s  good=%zu zeros=%zu of %zu\n", bytes, pool, ms,
			            good == n ? "in-order" : "OUT-OF-ORDER", good, zeros, n);
			std::free(src);
			if (pool) {
				CK(hipFreeAsync(d, st));
			} else {
				CK(hipFree(d));
			}
		}
	}
	CK(hipStreamSynchronize(st));

	// Part 3: round 1's exact shape.  A pool block is used (memset + kernel queued), freed stream-ordered, and the
	// next hipMallocAsync on the same stream hands it out again while that work is still queued; the new owner
	// uploads a pattern with a pageable async copy.
	std::printf("\n%-10s %-8s %-10s\n", "bytes", "reused", "order");
	for (size_t bytes : sizes) {
		const size_t n = bytes / 4;
		void *d1 = nullptr, *d2 = nullptr;
		CK(hipMallocAsync(&d1, bytes, st));
		CK(hipStreamSynchronize(st));
		hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, 5000000LL, nullptr);
		CK(hipMemsetAsync(d1, 0, bytes, st));
		CK(hipFreeAsync(d1, st));
		CK(hipMallocAsync(&d2, bytes, st));
		uint32_t *src = static_cast<uint32_t *>(std::malloc(bytes));
		for (size_t i = 0; i < n; i++) {
			src[i] = 0xabcd0000u + static_cast<uint32_t>(i & 0xffff);
		}
		CK(hipMemcpyAsync(d2, src, bytes, hipMemcpyHostToDevice, st));
		CK(hipStreamSynchronize(st));
		std::vector<uint32_t> back(n);
		CK(hipMemcpy(back.data(), d2, bytes, hipMemcpyDeviceToHost));
		size_t zeros = 0, good = 0;
		for (size_t i = 0; i < n; i++) {
			zeros += back[i] == 0;
			good += back[i] == 0xabcd0000u + static_cast<uint32_t>(i & 0xffff);
		}
		std::printf("%-10zu %-8s %-10s  good=%zu zeros=%zu of %zu\n", bytes, d1 == d2 ? "yes" : "no",
		            good == n ? "in-order" : "OUT-OF-ORDER", good, zeros, n);
		std::free(src);
		CK(hipFreeAsync(d2, st));
		CK(hipStreamSynchronize(st));
	}

	// Part 5: parts 2 and 3 repeated, to tell a race from a one-off.  Same stream throughout:
	//   [spin 5 ms][memset(block, 0)] ... [copy(block <- pageable pattern)]   -- stream order leaves the pattern.
	//   A = the block stays allocated;  B = it is hipFreeAsync'ed and handed out again by hipMallocAsync in between.
	std::printf("\npart 5: 40 repetitions per row; 'bad' = runs in which zeros from the EARLIER memset survived the LATER copy\n");
	std::printf("%-10s %-22s %-6s\n", "bytes", "shape", "bad");
	for (size_t bytes : {size_t(65536), size_t(262144), size_t(1u << 20), size_t(4u << 20)}) {
		for (int shape = 0; shape < 3; shape++) { // 0: plain hipMalloc block, 1: pool block kept, 2: pool block freed + re-handed
			int bad = 0;
			const size_t n = bytes / 4;
			uint32_t *src = static_cast<uint32_t *>(std::malloc(bytes));
			std::vector<uint32_t> back(n);
			for (int it = 0; it < 40; it++) {
				void *d = nullptr;
				if (shape == 0) {
					CK(hipMalloc(&d, bytes));
				} else {
					CK(hipMallocAsync(&d, bytes, st));
				}
				CK(hipStreamSynchronize(st));
				for (size_t i = 0; i < n; i++) {
					src[i] = 0xabcd0000u + static_cast<uint32_t>(i & 0xffff);
				}
				hipLaunchKernelGGL(k_spin, dim3(1), dim3(64), 0, st, 500000LL, nullptr);
				CK(hipMemsetAsync(d, 0, bytes, st));
				if (shape == 2) {
					void *d2 = nullptr;
					CK(hipFreeAsync(d, st));
					CK(hipMallocAsync(&d2, bytes, st));
					d = d2;
				}
				CK(hipMemcpyAsync(d, src, bytes, hipMemcpyHostToDevice, st));
				CK(hipStreamSynchronize(st));
				CK(hipMemcpy(back.data(), d, bytes, hipMemcpyDeviceToHost));
				size_t good = 0;
				for (size_t i = 0; i < n; i++) {
					good += back[i] == 0xabcd0000u + static_cast<uint32_t>(i & 0xffff);
				}
				bad += good != n;
				if (shape == 0) {
					CK(hipFree(d));
				} else {
					CK(hipFreeAsync(d, st));
				}
			}
			std::free(src);
			std::printf("%-10zu %-22s %-6d\n", bytes,
			            shape == 0 ? "hipMalloc block" : (shape == 1 ? "pool block, kept" : "pool block, re-handed"), bad);
		}
	}
	CK(hipStreamSynchronize(st));

	std::printf("\npart 4: fresh threads, hipStreamPerThread, pool block, pageable vector, kernel consumer\n");
	for (size_t bytes : {size_t(48), size_t(4096), size_t(65536), size_t(1u << 20)}) {
		int failures = 0, reused = 0;
		const int iters = 200;
		if (Part4(iters, bytes, &failures, &reused) != 0) {
			return 2;
		}
		std::printf("%-10zu iterations=%d pool-block-reused=%d kernel-saw-wrong-data=%d\n", bytes, iters, reused, failures);
	}
	return 0;
}

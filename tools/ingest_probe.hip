// ingest_probe.hip -- where the ceiling of the disk -> HBM path is on this box.
//   1. host -> device copy rate out of pinned memory: linear 64 MB pieces, and the 2-D form pgh_open uses to
//      re-pitch plain records (125,000-byte rows into a 125,056-byte pitch);
//   2. page cache -> pinned memory: pread of a warm file by 1..32 threads.
// usage: ingest_probe <file of a few GB>
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double Now() {
	return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv) {
	const size_t piece = 64ull << 20, total = 2048ull << 20;
	uint8_t *h = nullptr, *d = nullptr;
	if (hipHostMalloc(reinterpret_cast<void **>(&h), total, hipHostMallocDefault) != hipSuccess ||
	    hipMalloc(reinterpret_cast<void **>(&d), total + (64u << 20)) != hipSuccess) {
		return 2;
	}
	for (size_t i = 0; i < total; i += 4096) {
		h[i] = static_cast<uint8_t>(i >> 12);
	}
	hipStream_t st;
	(void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
	for (int rep = 0; rep < 2; rep++) {
		double t0 = Now();
		for (size_t o = 0; o < total; o += piece) {
			(void)hipMemcpyAsync(d + o, h + o, piece, hipMemcpyHostToDevice, st);
		}
		(void)hipStreamSynchronize(st);
		double t1 = Now();
		std::printf("pinned -> device, linear 64 MB pieces:            %.1f GB/s\n", total / (t1 - t0) / 1e9);
		const size_t rb = 125000, pitch = 125056, rows = piece / pitch;
		t0 = Now();
		size_t moved = 0;
		for (size_t o = 0; o + rows * rb <= total; o += rows * rb) {
			(void)hipMemcpy2DAsync(d + (o / rb) * pitch, pitch, h + o, rb, rb, rows, hipMemcpyHostToDevice, st);
			moved += rows * rb;
		}
		(void)hipStreamSynchronize(st);
		t1 = Now();
		std::printf("pinned -> device, 2-D (125000 B rows -> pitch 125056): %.1f GB/s\n", moved / (t1 - t0) / 1e9);
	}
	if (argc > 1) {
		const int fd = open(argv[1], O_RDONLY);
		if (fd < 0) {
			std::perror("open");
			return 2;
		}
		const off_t size = lseek(fd, 0, SEEK_END);
		const size_t bytes = std::min<size_t>(total, static_cast<size_t>(size));
		for (unsigned threads : {1u, 4u, 8u, 16u, 32u}) {
			for (int rep = 0; rep < 2; rep++) {
				const double t0 = Now();
				std::vector<std::thread> pool;
				const size_t slice = (bytes + threads - 1) / threads;
				for (unsigned t = 0; t < threads; t++) {
					pool.emplace_back([&, t] {
						size_t lo = std::min(bytes, t * slice), hi = std::min(bytes, lo + slice);
						while (lo < hi) {
							const ssize_t got = pread(fd, h + lo, std::min<size_t>(hi - lo, 8u << 20), static_cast<off_t>(lo));
							if (got <= 0) {
								break;
							}
							lo += static_cast<size_t>(got);
						}
					});
				}
				for (auto &th : pool) {
					th.join();
				}
				const double t1 = Now();
				if (rep == 1) {
					std::printf("page cache -> pinned, %2u threads: %.1f GB/s\n", threads, bytes / (t1 - t0) / 1e9);
				}
			}
		}
		close(fd);
	}
	return 0;
}

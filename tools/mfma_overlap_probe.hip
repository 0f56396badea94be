// mfma_overlap_probe.hip -- does a SIMD of gfx950 run vector ALU work under its int8 matrix instructions?
// k_score_i8 spends ~450 cycles of operand building per 64-variant tile next to 576-1024 cycles of
// v_mfma_i32_16x16x64_i8, and its matrix pipe reads 55-71 % busy: the arithmetic of "no
// overlap at all".  This probe takes the kernel away and asks the hardware directly:
//   mfma    one wave per SIMD issuing independent matrix instructions back to back
//   valu    one wave per SIMD issuing v_perm_b32 / v_and_b32 back to back
//   pair    two waves per SIMD, one of each (roles dealt per SIMD from HW_ID): time = max -> overlap, sum -> none
//   mix R   one wave per SIMD, R vector ops placed behind every matrix instruction (sched_group_barrier)
//   mix2 R  two such waves per SIMD
// for the 16x16x64 and the 32x32x32 int8 shapes.  Times are per matrix instruction (or per vector op) per wave.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                                                       \
	do {                                                                                                               \
		hipError_t e_ = (x);                                                                                           \
		if (e_ != hipSuccess) {                                                                                        \
			std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                             \
			std::exit(1);                                                                                              \
		}                                                                                                              \
	} while (0)

constexpr int kMfmaPerIter = 16;

template <bool BIG>
struct Acc;
template <>
struct Acc<false> {
	v4i r[kMfmaPerIter];
	__device__ void zero() {
#pragma unroll
		for (int j = 0; j < kMfmaPerIter; j++) {
			r[j] = v4i {0, 0, 0, 0};
		}
	}
	__device__ void step(int j, v4i a, v4i b) {
		r[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, r[j], 0, 0, 0);
	}
	__device__ int fold() const {
		int s = 0;
#pragma unroll
		for (int j = 0; j < kMfmaPerIter; j++) {
			s += r[j].x + r[j].y + r[j].z + r[j].w;
		}
		return s;
	}
};
template <>
struct Acc<true> {
	v16i r[kMfmaPerIter / 2];   // 8 x 16 registers: the same 128 accumulator registers per two instructions' work
	__device__ void zero() {
#pragma unroll
		for (int j = 0; j < kMfmaPerIter / 2; j++) {
#pragma unroll
			for (int k = 0; k < 16; k++) {
				r[j][k] = 0;
			}
		}
	}
	__device__ void step(int j, v4i a, v4i b) {
		r[j & 7] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, r[j & 7], 0, 0, 0);
	}
	__device__ int fold() const {
		int s = 0;
#pragma unroll
		for (int j = 0; j < kMfmaPerIter / 2; j++) {
#pragma unroll
			for (int k = 0; k < 16; k++) {
				s += r[j][k];
			}
		}
		return s;
	}
};

// vector work of the kernel's kind: a byte gather and an AND, on eight independent chains
struct Valu {
	unsigned x[8];
	__device__ void init(unsigned seed) {
#pragma unroll
		for (int k = 0; k < 8; k++) {
			x[k] = seed * (k + 3);
		}
	}
	__device__ void step(int k, unsigned y) {
		x[k & 7] = __builtin_amdgcn_perm(x[k & 7], y, 0x02010007u) & 0x7f7f3f7fu;
	}
	__device__ unsigned fold() const {
		unsigned s = 0;
#pragma unroll
		for (int k = 0; k < 8; k++) {
			s ^= x[k];
		}
		return s;
	}
};

__device__ inline unsigned SimdId() {
	// HW_REG_HW_ID (4): SIMD_ID = bits 5:4
	return __builtin_amdgcn_s_getreg(4 | (4 << 6) | (1 << 11));
}

enum { kMfmaOnly, kValuOnly, kPair, kMix };

template <bool BIG, int MODE, int R>
__global__ __launch_bounds__(512) void probe(int iters, unsigned seed, int *__restrict__ sink, unsigned *__restrict__ roles) {
	extern __shared__ unsigned s_dyn[];   // sized by the host to keep one workgroup per CU
	__shared__ unsigned s_cnt[4];
	if (threadIdx.x < 4) {
		s_cnt[threadIdx.x] = 0;
	}
	__syncthreads();
	const unsigned lane = threadIdx.x & 63;
	bool do_mfma = MODE == kMfmaOnly || MODE == kMix, do_valu = MODE == kValuOnly;
	if (MODE == kPair) {
		const unsigned simd = SimdId();
		unsigned slot = 0;
		if (lane == 0) {
			slot = atomicAdd(&s_cnt[simd], 1u);
		}
		slot = __builtin_amdgcn_readfirstlane(slot);
		do_mfma = (slot & 1u) == 0;
		do_valu = !do_mfma;
		if (blockIdx.x == 0 && lane == 0) {
			roles[threadIdx.x >> 6] = simd | (slot << 8);
		}
	}
	__syncthreads();
	const v4i a = {static_cast<int>(seed + lane), static_cast<int>(seed ^ lane), 3, 4};
	const v4i b = {static_cast<int>(seed * 7 + lane), 1, static_cast<int>(lane), 2};
	if (MODE == kMix) {
		Acc<BIG> acc;
		Valu v;
		acc.zero();
		v.init(seed + threadIdx.x);
		for (int it = 0; it < iters; it++) {
#pragma unroll
			for (int j = 0; j < kMfmaPerIter; j++) {
				acc.step(j, a, b);
#pragma unroll
				for (int k = 0; k < R; k++) {
					v.step(j * R + k, seed);
				}
			}
#pragma unroll
			for (int j = 0; j < kMfmaPerIter; j++) {
				__builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       // one matrix instruction
				__builtin_amdgcn_sched_group_barrier(0x002, 2 * R, 0);   // R x (perm + and)
			}
		}
		if (acc.fold() + static_cast<int>(v.fold()) == 0x12345678) {
			sink[0] = 1;
		}
		return;
	}
	if (do_mfma) {
		Acc<BIG> acc;
		acc.zero();
		for (int it = 0; it < iters; it++) {
#pragma unroll
			for (int j = 0; j < kMfmaPerIter; j++) {
				acc.step(j, a, b);
			}
		}
		if (acc.fold() == 0x12345678) {
			sink[0] = 1;
		}
	} else if (do_valu) {
		Valu v;
		v.init(seed + threadIdx.x);
		for (int it = 0; it < iters; it++) {
#pragma unroll
			for (int k = 0; k < kMfmaPerIter * (R > 0 ? R : 1); k++) {
				v.step(k, seed);
			}
		}
		if (v.fold() == 0x12345678u) {
			sink[0] = 1;
		}
	}
}

template <bool BIG, int MODE, int R>
static double Run(int threads, int iters, int *d_sink, unsigned *d_roles) {
	const int grid = 256;
	const size_t lds = 96 * 1024;   // one workgroup per CU
	CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(probe<BIG, MODE, R>),
	                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	probe<BIG, MODE, R><<<grid, threads, lds>>>(iters / 8, 1u, d_sink, d_roles);
	CHECK(hipDeviceSynchronize());
	CHECK(hipEventRecord(e0));
	probe<BIG, MODE, R><<<grid, threads, lds>>>(iters, 1u, d_sink, d_roles);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	return static_cast<double>(ms) * 1e6 / (static_cast<double>(iters) * kMfmaPerIter);   // ns per slot of the loop
}

template <bool BIG>
static void Shape(const char *name, int *d_sink, unsigned *d_roles) {
	const int iters = 200000;
	std::printf("== %s ==   (ns per matrix instruction per wave; one workgroup per CU, 256 workgroups)\n", name);
	const double m1 = Run<BIG, kMfmaOnly, 0>(256, iters, d_sink, d_roles);
	std::printf("mfma   1 wave/SIMD                        %7.2f ns\n", m1);
	const double m2 = Run<BIG, kMfmaOnly, 0>(512, iters, d_sink, d_roles);
	std::printf("mfma   2 waves/SIMD                       %7.2f ns   (per wave; %.2f per SIMD)\n", m2, m2 / 2);
	const double v1 = Run<BIG, kValuOnly, 2>(256, iters, d_sink, d_roles);
	std::printf("valu   1 wave/SIMD, 2 x (perm+and) a slot  %7.2f ns   = %.2f ns per vector op\n", v1, v1 / 4);
	const double v3 = Run<BIG, kValuOnly, 3>(256, iters, d_sink, d_roles);
	std::printf("valu   1 wave/SIMD, 3 x (perm+and) a slot  %7.2f ns\n", v3);
	const double p2 = Run<BIG, kPair, 2>(512, iters, d_sink, d_roles);
	unsigned roles[8];
	CHECK(hipMemcpy(roles, d_roles, sizeof roles, hipMemcpyDeviceToHost));
	std::printf("pair   mfma wave + valu wave (2 pairs)     %7.2f ns   (max %.2f, sum %.2f)   roles:", p2,
	            m1 > v1 ? m1 : v1, m1 + v1);
	for (int w = 0; w < 8; w++) {
		std::printf(" s%u%c", roles[w] & 3u, ((roles[w] >> 8) & 1u) ? 'v' : 'm');
	}
	std::printf("\n");
	const double p3 = Run<BIG, kPair, 3>(512, iters, d_sink, d_roles);
	std::printf("pair   mfma wave + valu wave (3 pairs)     %7.2f ns   (max %.2f, sum %.2f)\n", p3, m1 > v3 ? m1 : v3,
	            m1 + v3);
	const double x1 = Run<BIG, kMix, 1>(256, iters, d_sink, d_roles);
	const double x2 = Run<BIG, kMix, 2>(256, iters, d_sink, d_roles);
	const double x3 = Run<BIG, kMix, 3>(256, iters, d_sink, d_roles);
	const double x4 = Run<BIG, kMix, 4>(256, iters, d_sink, d_roles);
	std::printf("mix    1 wave/SIMD, R pairs behind each    R=1 %6.2f  R=2 %6.2f  R=3 %6.2f  R=4 %6.2f ns\n", x1, x2, x3,
	            x4);
	const double y1 = Run<BIG, kMix, 1>(512, iters, d_sink, d_roles);
	const double y2 = Run<BIG, kMix, 2>(512, iters, d_sink, d_roles);
	const double y3 = Run<BIG, kMix, 3>(512, iters, d_sink, d_roles);
	std::printf("mix2   2 waves/SIMD (per wave)             R=1 %6.2f  R=2 %6.2f  R=3 %6.2f ns\n", y1, y2, y3);
}

int main() {
	int *d_sink;
	unsigned *d_roles;
	CHECK(hipMalloc(&d_sink, 64));
	CHECK(hipMalloc(&d_roles, 64));
	CHECK(hipMemset(d_roles, 0, 64));
	Shape<false>("v_mfma_i32_16x16x64_i8", d_sink, d_roles);
	Shape<true>("v_mfma_i32_32x32x32_i8", d_sink, d_roles);
	return 0;
}

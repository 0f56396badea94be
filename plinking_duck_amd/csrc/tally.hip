// tally.hip -- the HBM-bound reductions over the packed rows (gfx950): per-variant genotype-class
// tallies (PgrGetCounts), per-sample column tallies, the fused row+column pass, the
// plink_freq / plink_hardy epilogues and the synthetic generator.
//
// Data layout: the genotype matrix is variant-major; row v holds ceil(N/4)
// bytes of packed 2-bit calls (00 hom-ref, 01 het, 10 hom-alt, 11 missing;
// sample s in bits 2*(s%4) of byte s/4) followed by zero bytes up to `pitch`
// (a multiple of 16, so every row can be streamed as whole 16-byte lanes and
// the pad decodes as hom-ref, which every kernel cancels against N).
//
// 16 B per lane coalesced loads, v_bcnt_u32_b32 tallies with its free accumulate operand, wave
// reductions by shuffles, LDS only for the cross-wave step.
#include <algorithm>

#include "device_utils.hpp"
#include "kernels.hpp"

#include "hwe_core.hpp"
#include "synth.hpp"

namespace pgh {

namespace {

// ---------------------------------------------------------------------------
// synthetic generator
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_synth_fill(uint8_t *rows, uint64_t pitch, uint32_t sample_ct,
                                                    uint32_t variant_begin, uint32_t variant_ct, uint64_t seed,
                                                    uint32_t miss_threshold) {
	const uint32_t dwords = static_cast<uint32_t>(pitch / 4);
	const uint32_t d = blockIdx.x * 256u + threadIdx.x;
	if (d >= dwords) {
		return;
	}
	for (uint32_t r = blockIdx.y; r < variant_ct; r += gridDim.y) {
		const SynthVariant sv = SynthVariantParams(seed, variant_begin + r);
		uint32_t w = 0;
		const uint32_t s0 = d * 16u;
#pragma unroll 4
		for (uint32_t j = 0; j < 16; j++) {
			const uint32_t s = s0 + j;
			if (s < sample_ct) {
				w |= SynthGenotype(sv, s, miss_threshold) << (2 * j);
			}
		}
		reinterpret_cast<uint32_t *>(rows + static_cast<uint64_t>(r) * pitch)[d] = w;
	}
}

__global__ __launch_bounds__(256) void k_sanitize_tail(uint8_t *rows, uint64_t pitch, uint32_t sample_ct,
                                                       uint32_t variant_ct) {
	// one thread per (row, pad byte); rows are short on pad so this is tiny
	const uint32_t record_bytes = (sample_ct + 3) / 4;
	const uint32_t pad = static_cast<uint32_t>(pitch - record_bytes) + 1; // + the last data byte
	const uint64_t idx = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
	const uint64_t total = static_cast<uint64_t>(variant_ct) * pad;
	if (idx >= total) {
		return;
	}
	const uint32_t r = static_cast<uint32_t>(idx / pad);
	const uint32_t k = static_cast<uint32_t>(idx % pad);
	uint8_t *row = rows + static_cast<uint64_t>(r) * pitch;
	if (k == 0) {
		const uint32_t rem = sample_ct & 3;
		if (rem) {
			row[record_bytes - 1] &= static_cast<uint8_t>((1u << (2 * rem)) - 1);
		}
	} else {
		row[record_bytes - 1 + k] = 0;
	}
}

// ---------------------------------------------------------------------------
// genotype-class tally
// ---------------------------------------------------------------------------

struct Tally {
	uint32_t lo = 0;   // slots with the low bit set  (het + missing)
	uint32_t hi = 0;   // slots with the high bit set (hom-alt + missing)
	uint32_t both = 0; // missing
};

template <bool MASKED>
__device__ __forceinline__ void TallyWord(Tally &t, uint32_t w, uint32_t m) {
	const uint32_t sel = MASKED ? m : kLow;
	const uint32_t lo = w & sel;
	const uint32_t hi = (w >> 1) & sel;
	t.lo += __popc(lo);
	t.hi += __popc(hi);
	t.both += __popc(lo & hi);
}

template <bool MASKED>
__device__ __forceinline__ void TallyQuad(Tally &t, const uint4 &w, const uint4 &m) {
	TallyWord<MASKED>(t, w.x, m.x);
	TallyWord<MASKED>(t, w.y, m.y);
	TallyWord<MASKED>(t, w.z, m.z);
	TallyWord<MASKED>(t, w.w, m.w);
}

// One 256-thread workgroup per variant row: for long rows (>= 4 KiB).
template <bool MASKED>
__global__ __launch_bounds__(256) void k_counts_block(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                      uint32_t chunks, uint32_t v_first,
                                                      const uint32_t *__restrict__ vlist, uint32_t v_count,
                                                      const uint4 *__restrict__ mask2, uint32_t n_eff,
                                                      uint4 *__restrict__ out) {
	__shared__ uint32_t part[4][3];
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;
	for (uint32_t i = blockIdx.x; i < v_count; i += gridDim.x) {
		const uint32_t v = vlist ? vlist[i] : v_first + i;
		const uint4 *row = reinterpret_cast<const uint4 *>(rows + static_cast<uint64_t>(v) * pitch);
		Tally t;
		uint32_t c = threadIdx.x;
		// 4 independent 16-byte loads in flight per lane
		for (; c + 768u < chunks; c += 1024u) {
			const uint4 w0 = LoadStream(row + c);
			const uint4 w1 = LoadStream(row + c + 256u);
			const uint4 w2 = LoadStream(row + c + 512u);
			const uint4 w3 = LoadStream(row + c + 768u);
			uint4 m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
			if (MASKED) {
				m0 = mask2[c];
				m1 = mask2[c + 256u];
				m2 = mask2[c + 512u];
				m3 = mask2[c + 768u];
			}
			TallyQuad<MASKED>(t, w0, m0);
			TallyQuad<MASKED>(t, w1, m1);
			TallyQuad<MASKED>(t, w2, m2);
			TallyQuad<MASKED>(t, w3, m3);
		}
		for (; c < chunks; c += 256u) {
			const uint4 w = LoadStream(row + c);
			uint4 m = {0, 0, 0, 0};
			if (MASKED) {
				m = mask2[c];
			}
			TallyQuad<MASKED>(t, w, m);
		}
		const uint32_t lo = WaveSum(t.lo);
		const uint32_t hi = WaveSum(t.hi);
		const uint32_t both = WaveSum(t.both);
		if (lane == 0) {
			part[wave][0] = lo;
			part[wave][1] = hi;
			part[wave][2] = both;
		}
		__syncthreads();
		if (threadIdx.x == 0) {
			const uint32_t l = part[0][0] + part[1][0] + part[2][0] + part[3][0];
			const uint32_t h = part[0][1] + part[1][1] + part[2][1] + part[3][1];
			const uint32_t b = part[0][2] + part[1][2] + part[2][2] + part[3][2];
			uint4 r;
			r.y = l - b;                 // het
			r.z = h - b;                 // hom-alt
			r.w = b;                     // missing
			r.x = n_eff - r.y - r.z - b; // hom-ref (zero pad cancels here)
			out[i] = r;
		}
		__syncthreads();
	}
}

// One wave per variant row: short rows.  4 rows per 256-thread workgroup.
template <bool MASKED>
__global__ __launch_bounds__(256) void k_counts_wave(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                     uint32_t chunks, uint32_t v_first,
                                                     const uint32_t *__restrict__ vlist, uint32_t v_count,
                                                     const uint4 *__restrict__ mask2, uint32_t n_eff,
                                                     uint4 *__restrict__ out) {
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;
	for (uint32_t i = blockIdx.x * 4u + wave; i < v_count; i += gridDim.x * 4u) {
		const uint32_t v = vlist ? vlist[i] : v_first + i;
		const uint4 *row = reinterpret_cast<const uint4 *>(rows + static_cast<uint64_t>(v) * pitch);
		Tally t;
		for (uint32_t c = lane; c < chunks; c += 64u) {
			const uint4 w = LoadStream(row + c);
			uint4 m = {0, 0, 0, 0};
			if (MASKED) {
				m = mask2[c];
			}
			TallyQuad<MASKED>(t, w, m);
		}
		const uint32_t lo = WaveSum(t.lo);
		const uint32_t hi = WaveSum(t.hi);
		const uint32_t both = WaveSum(t.both);
		if (lane == 0) {
			uint4 r;
			r.y = lo - both;
			r.z = hi - both;
			r.w = both;
			r.x = n_eff - r.y - r.z - both;
			out[i] = r;
		}
	}
}

// ---------------------------------------------------------------------------
// per-sample missing tally (column sums of the missing indicator)
// ---------------------------------------------------------------------------
//
// A lane owns one 16-byte column (64 samples) and walks down a slice of rows.  The indicator
// m = w & (w>>1) & 0x5555.. has one bit per 2-bit slot; two of them pack into a full word of one bit per
// sample, and the words of successive rows are summed as a positional population count: bit-sliced
// counters (plane p holds bit p of every position's count) fed through a Harley-Seal carry-save tree.
// Each slice writes its planes to its own slab with plain stores; k_sum_cols1 adds the slices in the same
// bit-sliced form and turns bit positions into samples at the end.  (The first form widened SWAR fields
// 2 -> 4 -> 8 -> 32 bits into 64 registers per lane; it streamed at the same 6.25 TB/s -- the column walk,
// not the arithmetic, sets that -- with 3x the registers.)

// carry-save adder: two v_bitop3_b32 (majority 0xe8, parity 0x96; the C form a ^ b, (a & b) | (u & c), u ^ c
// compiled to three or four instructions)
__device__ __forceinline__ void Csa(uint32_t &h, uint32_t &l, uint32_t a, uint32_t b, uint32_t c) {
	h = __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);
	l = __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}

// bit-sliced counter of one 32-bit word of indicator bits, fed four rows at a time
template <int PLANES>
struct BitCounter {
	uint32_t p[PLANES]; // p[0] ones, p[1] twos, p[2] fours, p[3] eights, p[4] sixteens, ...
	uint32_t fours_a, eights_a; // carries waiting for their partner inside a 16-row trip

	__device__ __forceinline__ void Clear() {
#pragma unroll
		for (int k = 0; k < PLANES; k++) {
			p[k] = 0;
		}
		fours_a = eights_a = 0;
	}
	// rows 4g .. 4g+3 of a 16-row trip (g = 0..3, compile-time)
	template <int G>
	__device__ __forceinline__ void Add4(uint32_t x0, uint32_t x1, uint32_t x2, uint32_t x3) {
		uint32_t twos_a, twos_b, fours;
		Csa(twos_a, p[0], p[0], x0, x1);
		Csa(twos_b, p[0], p[0], x2, x3);
		Csa(fours, p[1], p[1], twos_a, twos_b);
		if (G == 0 || G == 2) {
			fours_a = fours;
			return;
		}
		uint32_t eights;
		Csa(eights, p[2], p[2], fours_a, fours);
		if (G == 1) {
			eights_a = eights;
			return;
		}
		uint32_t carry;
		Csa(carry, p[3], p[3], eights_a, eights);
#pragma unroll
		for (int k = 4; k < PLANES; k++) { // ripple the sixteens up
			const uint32_t t = p[k] & carry;
			p[k] ^= carry;
			carry = t;
		}
	}
};

// CLASS: which genotype code is tallied -- 3 missing (plink_missing, plink_score), 1 het,
// 2 hom-alt (read_pfile's sample-orient counts)
template <int CLASS>
__device__ __forceinline__ uint32_t ClassBits(uint32_t w) {
	if (CLASS == 3) {
		return w & (w >> 1) & 0x55555555u;
	}
	if (CLASS == 1) {
		return w & ~(w >> 1) & 0x55555555u;
	}
	return (w >> 1) & ~w & 0x55555555u;
}

// One genotype code per sample over a slice of rows, as a positional population count (the scheme of
// k_class_cols3 below with a single stream): a lane owns 16 bytes, its four indicator words (one bit per 2-bit
// slot) pack into two full words, each feeds a 16-plane bit-sliced counter through the carry-save tree.
// 1024 lanes per workgroup: a 16 KB stripe of every row.
constexpr uint32_t kCols1Threads = 1024;
constexpr int kCols1Planes = 16; // a slice holds at most 65280 rows
template <int CLASS>
__global__ __launch_bounds__(kCols1Threads) void k_class_cols1(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                              uint32_t chunks, uint32_t v_first,
                                                              const uint32_t *__restrict__ vlist, uint32_t v_count,
                                                              uint32_t slice_len,
                                                              const uint32_t *__restrict__ row_flags,
                                                              uint32_t *__restrict__ slabs, uint64_t slab_stride) {
	const uint32_t col = blockIdx.x * kCols1Threads + threadIdx.x;
	if (col >= chunks) {
		return;
	}
	const uint32_t i_begin = blockIdx.y * slice_len;
	const uint32_t i_end = min(i_begin + slice_len, v_count);
	BitCounter<kCols1Planes> ctr[2];
	ctr[0].Clear();
	ctr[1].Clear();
	// rows past the slice, and rows whose flag byte is zero (skipped scored variants), add nothing
	auto load = [&](uint32_t idx) {
		if (idx >= i_end || (row_flags && !(row_flags[idx] & 0xffu))) {
			return make_uint4(0, 0, 0, 0);
		}
		const uint32_t v = vlist ? vlist[idx] : v_first + idx;
		return LoadStream(reinterpret_cast<const uint4 *>(rows + static_cast<uint64_t>(v) * pitch) + col);
	};
#define PGH_COLS1_GROUP(G, R0, R1, R2, R3)                                                                             \
	ctr[0].Add4<G>(ClassBits<CLASS>(R0.x) | (ClassBits<CLASS>(R0.y) << 1), ClassBits<CLASS>(R1.x) | (ClassBits<CLASS>(R1.y) << 1), \
	               ClassBits<CLASS>(R2.x) | (ClassBits<CLASS>(R2.y) << 1), ClassBits<CLASS>(R3.x) | (ClassBits<CLASS>(R3.y) << 1)); \
	ctr[1].Add4<G>(ClassBits<CLASS>(R0.z) | (ClassBits<CLASS>(R0.w) << 1), ClassBits<CLASS>(R1.z) | (ClassBits<CLASS>(R1.w) << 1), \
	               ClassBits<CLASS>(R2.z) | (ClassBits<CLASS>(R2.w) << 1), ClassBits<CLASS>(R3.z) | (ClassBits<CLASS>(R3.w) << 1));
	for (uint32_t i = i_begin; i < i_end; i += 16u) {
		uint4 a0 = load(i), a1 = load(i + 1), a2 = load(i + 2), a3 = load(i + 3);
		uint4 b0 = load(i + 4), b1 = load(i + 5), b2 = load(i + 6), b3 = load(i + 7);
		PGH_COLS1_GROUP(0, a0, a1, a2, a3)
		a0 = load(i + 8), a1 = load(i + 9), a2 = load(i + 10), a3 = load(i + 11);
		PGH_COLS1_GROUP(1, b0, b1, b2, b3)
		b0 = load(i + 12), b1 = load(i + 13), b2 = load(i + 14), b3 = load(i + 15);
		PGH_COLS1_GROUP(2, a0, a1, a2, a3)
		PGH_COLS1_GROUP(3, b0, b1, b2, b3)
	}
#undef PGH_COLS1_GROUP
	// planes of word k: slabs[slice][k][plane][col]
	uint32_t *dst = slabs + static_cast<uint64_t>(blockIdx.y) * slab_stride + col;
#pragma unroll
	for (uint32_t k = 0; k < 2; k++) {
#pragma unroll
		for (int pl = 0; pl < kCols1Planes; pl++) {
			dst[(static_cast<uint64_t>(k) * kCols1Planes + pl) * chunks] = ctr[k].p[pl];
		}
	}
}

// out[s] += the slices' counts of sample s.  A lane owns one column (64 samples) and a run of slices, adds their
// 16-plane numbers in bit-sliced form and turns bit positions into samples at the end: packed word q holds
// sample 32q + k at bit 2k and sample 32q + 16 + k at bit 2k + 1.
constexpr int kCols1SumPlanes = 26;     // 1024 slices x 65280 rows < 2^26
constexpr uint32_t kCols1SumGroup = 16; // slices per lane
__global__ __launch_bounds__(256) void k_sum_cols1(const uint32_t *__restrict__ slabs, uint64_t slab_stride,
                                                   uint32_t chunks, uint32_t n_slices, uint32_t n,
                                                   uint32_t *__restrict__ out) {
	const uint32_t col = blockIdx.x * 256u + threadIdx.x;
	if (col >= chunks) {
		return;
	}
	uint32_t acc[2][kCols1SumPlanes];
#pragma unroll
	for (int k = 0; k < 2; k++) {
#pragma unroll
		for (int pl = 0; pl < kCols1SumPlanes; pl++) {
			acc[k][pl] = 0;
		}
	}
	const uint32_t y_begin = blockIdx.y * kCols1SumGroup;
	const uint32_t y_end = min(y_begin + kCols1SumGroup, n_slices);
	for (uint32_t y = y_begin; y < y_end; y++) {
		const uint32_t *sl = slabs + static_cast<uint64_t>(y) * slab_stride + col;
		uint32_t x[2 * kCols1Planes];
#pragma unroll
		for (int q = 0; q < 2 * kCols1Planes; q++) {
			x[q] = __builtin_nontemporal_load(sl + static_cast<uint64_t>(q) * chunks);
		}
#pragma unroll
		for (int k = 0; k < 2; k++) {
			uint32_t carry = 0;
#pragma unroll
			for (int pl = 0; pl < kCols1Planes; pl++) {
				Csa(carry, acc[k][pl], acc[k][pl], x[k * kCols1Planes + pl], carry);
			}
#pragma unroll
			for (int pl = kCols1Planes; pl < kCols1SumPlanes; pl++) {
				const uint32_t t = acc[k][pl] & carry;
				acc[k][pl] ^= carry;
				carry = t;
			}
		}
	}
#pragma unroll
	for (uint32_t t = 0; t < 64; t++) {
		const uint32_t s = 64u * col + t;
		const uint32_t q = t >> 5, bit = 2u * (t & 15u) + ((t >> 4) & 1u);
		uint32_t cnt = 0;
#pragma unroll
		for (int pl = 0; pl < kCols1SumPlanes; pl++) {
			cnt += ((acc[q][pl] >> bit) & 1u) << pl;
		}
		if (s < n && cnt) {
			atomicAdd(out + s, cnt);
		}
	}
}

// ---------------------------------------------------------------------------
// per-sample tallies of all three non-reference codes in one pass
// ---------------------------------------------------------------------------
//
// read_pfile's sample-orient counts need het, hom-alt and missing per sample; three passes of
// k_missing_cols read the matrix three times.  This is a positional population count: how often each
// BIT POSITION of a lane's 16 bytes is set over the rows.  A 2-bit word needs no unpacking for that --
//     n_lo[s]   = rows with the low bit of s set   = het + missing
//     n_hi[s]   = rows with the high bit of s set  = hom-alt + missing
//     n_both[s] = rows with both set               = missing
// so a lane's two raw dwords go straight into bit-sliced counters (plane p holds bit p of every position's
// count) through a Harley-Seal carry-save tree, 16 rows per trip: 15 carry-save adders (5 ops each) per
// word instead of per-row field arithmetic, and only the `both` stream costs extraction (lo & hi, two
// dwords packed into one).  A lane owns 8 bytes (32 samples): three counter words x 10 planes + carries
// keep it near 80 VGPRs, so enough waves are resident to hold sixteen rows in flight each.  Ten planes
// count to 1023: a workgroup takes a slice of 1008 rows and writes its planes raw (120 B per lane); k_sum_class_bits turns bit positions back into
// samples, adds the slices and takes the differences.  (The SWAR field-widening form this replaced spent
// ~70 VALU ops per 16 bytes and streamed at 0.62 of the HBM roofline; this one spends ~50.)
constexpr uint32_t kCols3Rows = 1008;  // rows per slice: a multiple of 16, <= 1023
constexpr uint32_t kCols3Super = 256;  // slices per launch (bounds the plane scratch to ~0.5 GB at N = 500k)
constexpr int kCols3Planes = 10;
constexpr uint32_t kCols3Words = 3;    // per lane: two raw dwords (lo / hi bits interleaved) + their packed `both` word

constexpr uint32_t kCols3Threads = 1024; // an 8 KB stripe of every row per workgroup
__global__ __launch_bounds__(kCols3Threads) void k_class_cols3(const uint8_t *__restrict__ rows, uint64_t pitch, uint32_t chunks,
                                                     uint32_t v_first, const uint32_t *__restrict__ vlist,
                                                     uint32_t v_count, uint32_t *__restrict__ slabs,
                                                     uint64_t slab_stride) {
	const uint32_t col = blockIdx.x * kCols3Threads + threadIdx.x; // 8 bytes = 32 samples per lane
	if (col >= chunks) {
		return;
	}
	const uint32_t i_begin = blockIdx.y * kCols3Rows;
	const uint32_t i_end = min(i_begin + kCols3Rows, v_count);
	BitCounter<kCols3Planes> ctr[kCols3Words];
#pragma unroll
	for (uint32_t k = 0; k < kCols3Words; k++) {
		ctr[k].Clear();
	}
	// rows past the slice read as all hom-ref: they add nothing to any counter
	auto load = [&](uint32_t idx) {
		if (idx >= i_end) {
			return make_uint2(0, 0);
		}
		const uint32_t v = vlist ? vlist[idx] : v_first + idx;
		const u32x2 w = __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(rows + static_cast<uint64_t>(v) * pitch) + col);
		return make_uint2(w.x, w.y);
	};
	// samples with both bits set: those of dword x at even positions, those of dword y at odd ones
	auto both = [](const uint2 &r) { return (r.x & (r.x >> 1) & kLow) | ((r.y & (r.y << 1)) & ~kLow); };
#define PGH_COLS3_GROUP(G, R0, R1, R2, R3)                                                                             \
	ctr[0].Add4<G>(R0.x, R1.x, R2.x, R3.x);                                                                            \
	ctr[1].Add4<G>(R0.y, R1.y, R2.y, R3.y);                                                                            \
	ctr[2].Add4<G>(both(R0), both(R1), both(R2), both(R3));
	for (uint32_t i = i_begin; i < i_end; i += 16u) {
		// sixteen rows per trip, all in flight before the first is used
		const uint2 r0 = load(i), r1 = load(i + 1), r2 = load(i + 2), r3 = load(i + 3);
		const uint2 r4 = load(i + 4), r5 = load(i + 5), r6 = load(i + 6), r7 = load(i + 7);
		const uint2 r8 = load(i + 8), r9 = load(i + 9), r10 = load(i + 10), r11 = load(i + 11);
		const uint2 r12 = load(i + 12), r13 = load(i + 13), r14 = load(i + 14), r15 = load(i + 15);
		PGH_COLS3_GROUP(0, r0, r1, r2, r3)
		PGH_COLS3_GROUP(1, r4, r5, r6, r7)
		PGH_COLS3_GROUP(2, r8, r9, r10, r11)
		PGH_COLS3_GROUP(3, r12, r13, r14, r15)
	}
#undef PGH_COLS3_GROUP
	// planes of word k: slabs[slice][k][plane][col]  (a wave writes 256 contiguous bytes per plane)
	uint32_t *dst = slabs + static_cast<uint64_t>(blockIdx.y) * slab_stride + col;
#pragma unroll
	for (uint32_t k = 0; k < kCols3Words; k++) {
#pragma unroll
		for (int pl = 0; pl < kCols3Planes; pl++) {
			dst[(static_cast<uint64_t>(k) * kCols3Planes + pl) * chunks] = ctr[k].p[pl];
		}
	}
}

// out[c][s] += class c of sample s summed over the slices: c = 0 het, 1 hom-alt, 2 missing.
// A lane owns one column of k_class_cols3 (32 samples) and a run of slices: it adds the slices' 10-plane
// numbers in bit-sliced form (a ripple-carry adder over planes, ~75 ops per word per slice, every plane word
// read once and coalesced) and only at the end turns bit positions into samples:
// raw dword j holds sample 16j + k's low bit at 2k and high bit at 2k + 1, the packed `both` word its
// both-bit at 2k + j.
constexpr int kCols3SumPlanes = 18;        // kCols3Super * kCols3Rows < 2^18
constexpr uint32_t kCols3SumGroup = 64;    // slices per lane: the rest of the parallelism comes from grid.y
__global__ __launch_bounds__(256) void k_sum_class_bits(const uint32_t *__restrict__ slabs, uint64_t slab_stride,
                                                        uint32_t chunks, uint32_t n_slices, uint32_t n,
                                                        uint32_t out_stride, uint32_t *__restrict__ out) {
	const uint32_t col = blockIdx.x * 256u + threadIdx.x;
	if (col >= chunks) {
		return;
	}
	uint32_t acc[kCols3Words][kCols3SumPlanes];
#pragma unroll
	for (uint32_t k = 0; k < kCols3Words; k++) {
#pragma unroll
		for (int pl = 0; pl < kCols3SumPlanes; pl++) {
			acc[k][pl] = 0;
		}
	}
	const uint32_t y_begin = blockIdx.y * kCols3SumGroup;
	const uint32_t y_end = min(y_begin + kCols3SumGroup, n_slices);
	for (uint32_t y = y_begin; y < y_end; y++) {
		const uint32_t *sl = slabs + static_cast<uint64_t>(y) * slab_stride + col;
		uint32_t x[kCols3Words * kCols3Planes]; // the slice's thirty plane words, all in flight together
#pragma unroll
		for (uint32_t q = 0; q < kCols3Words * kCols3Planes; q++) {
			x[q] = __builtin_nontemporal_load(sl + static_cast<uint64_t>(q) * chunks);
		}
#pragma unroll
		for (uint32_t k = 0; k < kCols3Words; k++) {
			uint32_t carry = 0;
#pragma unroll
			for (int pl = 0; pl < kCols3Planes; pl++) {
				Csa(carry, acc[k][pl], acc[k][pl], x[k * kCols3Planes + pl], carry);
			}
#pragma unroll
			for (int pl = kCols3Planes; pl < kCols3SumPlanes; pl++) {
				const uint32_t t = acc[k][pl] & carry;
				acc[k][pl] ^= carry;
				carry = t;
			}
		}
	}
#pragma unroll
	for (uint32_t t = 0; t < 32; t++) {
		const uint32_t s = 32u * col + t;
		const uint32_t j = t >> 4, k = t & 15u;
		uint32_t n_lo = 0, n_hi = 0, n_both = 0;
#pragma unroll
		for (int pl = 0; pl < kCols3SumPlanes; pl++) {
			n_lo += ((acc[j][pl] >> (2 * k)) & 1u) << pl;
			n_hi += ((acc[j][pl] >> (2 * k + 1)) & 1u) << pl;
			n_both += ((acc[2][pl] >> (2 * k + j)) & 1u) << pl;
		}
		if (s < n) {
			atomicAdd(out + s, n_lo - n_both);
			atomicAdd(out + static_cast<uint64_t>(out_stride) + s, n_hi - n_both);
			atomicAdd(out + 2ull * out_stride + s, n_both);
		}
	}
}

// ---------------------------------------------------------------------------
// fused pass: per-variant class tallies AND per-sample missing tallies
// ---------------------------------------------------------------------------
//
// plink_freq + plink_hardy + plink_missing (both modes) need the row sums and the column sums of the same
// matrix; this kernel reads every byte once for both.  Ownership is by column: a lane owns 16 bytes (64 samples)
// of every row of its slice.
//   * Column sums (missing calls per sample): k_class_cols1's positional population count -- the row's four
//     missing-indicator words pack into two words of one bit per sample, and those feed two 16-plane bit-sliced
//     counters through a Harley-Seal carry-save tree, sixteen rows per trip (a carry-save adder is two
//     v_bitop3_b32); the slice's planes go to its slab and k_sum_cols1 adds the slices.
//   * Row sums (class tallies per variant) cross lanes: each lane's per-row popcounts (lo | hi << 10 |
//     both << 20) go through an LDS tile [16 rows][256 lanes], sixteen lanes per row add 8 + 8 packed entries
//     (10-bit fields cannot overflow) and finish with a 16-lane shuffle tree, and the sixteenth-0 lanes add the
//     workgroup's partials to a [column block][variant] array that k_finish_tallies sums.
// Round 2's form kept SWAR fields for the columns (2 -> 4 -> 8 -> 16 bits, ~23 vector ops per row and lane) and
// drained the memory counter before counting: 83 vector ops per 16 bytes, 73 % of the kernel's cycles in vector
// issue, 0.63-0.69 of the HBM roofline.  This one spends ~7 on the columns (~50 in all), keeps eight rows' loads
// in flight under the arithmetic of the previous eight, and pays one barrier per sixteen rows.
constexpr uint32_t kFusedRows = 16;

__global__ __launch_bounds__(256) void k_fused_tally(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                     uint32_t chunks, uint32_t v_first, uint32_t v_count,
                                                     uint32_t slice_len, uint4 *__restrict__ partials,
                                                     uint32_t *__restrict__ slabs, uint64_t slab_stride) {
	__shared__ uint32_t s_p[2][kFusedRows][256];
	const uint32_t col = blockIdx.x * 256u + threadIdx.x;
	const bool live = col < chunks;
	const uint32_t i_begin = blockIdx.y * slice_len;
	const uint32_t i_end = min(i_begin + slice_len, v_count);
	// Lanes past the row's last 16-byte column (the last column block only) re-read that column -- no branch around
	// the loads, which would also keep the compiler from running them ahead of the arithmetic -- and their row
	// popcounts are masked off; their column counters are never written.
	const uint32_t col_c = live ? col : chunks - 1u;
	const uint32_t live_mask = live ? 0xffffffffu : 0u;
	BitCounter<kCols1Planes> ctr[2];
	ctr[0].Clear();
	ctr[1].Clear();
	// rows past the slice's end read row i_end - 1 again and count nothing (their words are zeroed)
	auto load_row = [&](uint32_t idx) {
		const uint32_t r = idx < i_end ? idx : i_end - 1u;
		const uint4 w = LoadStream(reinterpret_cast<const uint4 *>(rows + static_cast<uint64_t>(v_first + r) * pitch) + col_c);
		const uint32_t keep = idx < i_end ? 0xffffffffu : 0u;
		return make_uint4(w.x & keep, w.y & keep, w.z & keep, w.w & keep);
	};
	// the rows of a full slice interior need no clamp: chosen per tile (uniform)
	auto load_row_in = [&](uint32_t idx) {
		return LoadStream(reinterpret_cast<const uint4 *>(rows + static_cast<uint64_t>(v_first + idx) * pitch) + col_c);
	};
	// one row: packed class popcounts to the LDS tile; its two one-bit-per-sample missing words come back
	auto one_row = [&](uint32_t *slot, const uint4 &w, uint32_t &m01, uint32_t &m23) {
		const uint32_t b0 = w.x & (w.x >> 1) & kLow, b1 = w.y & (w.y >> 1) & kLow;
		const uint32_t b2 = w.z & (w.z >> 1) & kLow, b3 = w.w & (w.w >> 1) & kLow;
		m01 = b0 | (b1 << 1);
		m23 = b2 | (b3 << 1);
		const uint32_t both_ct = __popc(m01) + __popc(m23);
		const uint32_t all_ct = __popc(w.x) + __popc(w.y) + __popc(w.z) + __popc(w.w);
		// the low-bit planes of two words share a word: (x & m) | ((y << 1) & ~m)
		const uint32_t l01 = (w.x & kLow) | ((w.y << 1) & ~kLow);
		const uint32_t l23 = (w.z & kLow) | ((w.w << 1) & ~kLow);
		const uint32_t lo_ct = __popc(l01) + __popc(l23);
		*slot = (lo_ct | ((all_ct - lo_ct) << 10) | (both_ct << 20)) & live_mask;
	};
	// rows 4g .. 4g + 3 of a tile
#define PGH_FUSED_GROUP(G, TILE, R0, R1, R2, R3)                                                                       \
	{                                                                                                                  \
		uint32_t x0, x1, x2, x3, y0, y1, y2, y3;                                                                       \
		one_row(&TILE[4 * G + 0][threadIdx.x], R0, x0, y0);                                                            \
		one_row(&TILE[4 * G + 1][threadIdx.x], R1, x1, y1);                                                            \
		one_row(&TILE[4 * G + 2][threadIdx.x], R2, x2, y2);                                                            \
		one_row(&TILE[4 * G + 3][threadIdx.x], R3, x3, y3);                                                            \
		ctr[0].Add4<G>(x0, x1, x2, x3);                                                                                \
		ctr[1].Add4<G>(y0, y1, y2, y3);                                                                                \
	}
	// row sums of a finished tile: lane t -> (row t >> 4, sixteenth t & 15) sums 16 packed lanes (8 + 8), a
	// 16-lane shuffle tree over two words (lo | both << 16, hi), one atomic per sum
	auto reduce_tile = [&](uint32_t (*tile)[256], uint32_t i) {
		const uint32_t row = threadIdx.x >> 4, part = threadIdx.x & 15u;
		const uint4 *src = reinterpret_cast<const uint4 *>(&tile[row][part * 16u]);
		const uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
		const uint32_t p0 = q0.x + q0.y + q0.z + q0.w + q1.x + q1.y + q1.z + q1.w;
		const uint32_t p1 = q2.x + q2.y + q2.z + q2.w + q3.x + q3.y + q3.z + q3.w;
		uint32_t lb = (p0 & 0x3ffu) + (p1 & 0x3ffu) + (((p0 >> 20) + (p1 >> 20)) << 16);
		uint32_t hi = ((p0 >> 10) & 0x3ffu) + ((p1 >> 10) & 0x3ffu);
#pragma unroll
		for (int off = 8; off > 0; off >>= 1) {
			lb += __shfl_xor(lb, off, 64);
			hi += __shfl_xor(hi, off, 64);
		}
		if (part == 0 && i + row < i_end) {
			// this column block's share of the row, with a plain store: sixteen consecutive 16-byte entries per
			// tile.  (Device-scope atomics on the per-variant totals -- 93 million per pass at 500 k samples,
			// from 31 column blocks on eight XCDs -- were what held this kernel at 24 ms whatever its arithmetic
			// cost; k_finish_tallies adds the column blocks instead.)
			partials[static_cast<uint64_t>(blockIdx.x) * v_count + i + row] = make_uint4(0u, lb & 0xffffu, hi, lb >> 16);
		}
	};
	// Sixteen rows per trip in four groups over two register sets: while a group is counted the loads of the
	// group after next are in flight.  The LDS tile is double-buffered, so a tile costs ONE barrier: no wave can
	// reach tile k + 2's writes before every wave has passed tile k + 1's barrier, i.e. finished reading tile k.
	uint32_t buf = 0;
	if (i_begin < i_end) {
		uint4 a0 = load_row(i_begin), a1 = load_row(i_begin + 1), a2 = load_row(i_begin + 2), a3 = load_row(i_begin + 3);
		uint4 b0 = load_row(i_begin + 4), b1 = load_row(i_begin + 5), b2 = load_row(i_begin + 6), b3 = load_row(i_begin + 7);
		for (uint32_t i = i_begin; i < i_end; i += kFusedRows) {
			uint32_t (*tile)[256] = s_p[buf];
			if (i + 2 * kFusedRows <= i_end) {
				// every row this trip touches lies inside the slice: no clamps
				PGH_FUSED_GROUP(0, tile, a0, a1, a2, a3)
				a0 = load_row_in(i + 8), a1 = load_row_in(i + 9), a2 = load_row_in(i + 10), a3 = load_row_in(i + 11);
				PGH_FUSED_GROUP(1, tile, b0, b1, b2, b3)
				b0 = load_row_in(i + 12), b1 = load_row_in(i + 13), b2 = load_row_in(i + 14), b3 = load_row_in(i + 15);
				PGH_FUSED_GROUP(2, tile, a0, a1, a2, a3)
				a0 = load_row_in(i + 16), a1 = load_row_in(i + 17), a2 = load_row_in(i + 18), a3 = load_row_in(i + 19);
				PGH_FUSED_GROUP(3, tile, b0, b1, b2, b3)
				b0 = load_row_in(i + 20), b1 = load_row_in(i + 21), b2 = load_row_in(i + 22), b3 = load_row_in(i + 23);
			} else {
				PGH_FUSED_GROUP(0, tile, a0, a1, a2, a3)
				a0 = load_row(i + 8), a1 = load_row(i + 9), a2 = load_row(i + 10), a3 = load_row(i + 11);
				PGH_FUSED_GROUP(1, tile, b0, b1, b2, b3)
				b0 = load_row(i + 12), b1 = load_row(i + 13), b2 = load_row(i + 14), b3 = load_row(i + 15);
				PGH_FUSED_GROUP(2, tile, a0, a1, a2, a3)
				a0 = load_row(i + 16), a1 = load_row(i + 17), a2 = load_row(i + 18), a3 = load_row(i + 19);
				PGH_FUSED_GROUP(3, tile, b0, b1, b2, b3)
				b0 = load_row(i + 20), b1 = load_row(i + 21), b2 = load_row(i + 22), b3 = load_row(i + 23);
			}
			__syncthreads();
			reduce_tile(tile, i);
			buf ^= 1u;
		}
	}
#undef PGH_FUSED_GROUP
	if (live) {
		// planes of word k: slabs[slice][k][plane][col] (k_class_cols1's layout: k_sum_cols1 adds the slices)
		uint32_t *dst = slabs + static_cast<uint64_t>(blockIdx.y) * slab_stride + col;
#pragma unroll
		for (uint32_t k = 0; k < 2; k++) {
#pragma unroll
			for (int pl = 0; pl < kCols1Planes; pl++) {
				dst[(static_cast<uint64_t>(k) * kCols1Planes + pl) * chunks] = ctr[k].p[pl];
			}
		}
	}
}

// the column blocks' (., lo, hi, both) of a variant -> (hom_ref, het, hom_alt, missing)
__global__ __launch_bounds__(256) void k_finish_tallies(const uint4 *__restrict__ partials, uint32_t col_blocks,
                                                        uint4 *__restrict__ tallies, uint32_t n, uint32_t n_eff) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) {
		return;
	}
	uint32_t lo = 0, hi = 0, both = 0;
	for (uint32_t cb = 0; cb < col_blocks; cb++) {
		const uint4 t = partials[static_cast<uint64_t>(cb) * n + i];
		lo += t.y;
		hi += t.z;
		both += t.w;
	}
	uint4 r;
	r.y = lo - both;
	r.z = hi - both;
	r.w = both;
	r.x = n_eff - r.y - r.z - r.w;
	tallies[i] = r;
}

// ---------------------------------------------------------------------------
// plink_freq epilogue
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_freq_from_counts(const uint4 *__restrict__ counts, uint32_t n,
                                                          double *__restrict__ alt_freq, int32_t *__restrict__ obs_ct) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) {
		return;
	}
	const uint4 c = counts[i];
	const uint32_t obs = c.x + c.y + c.z;
	// src/plink_freq.cpp:541-543
	alt_freq[i] = obs ? (static_cast<double>(c.y) + 2.0 * static_cast<double>(c.z)) / (2.0 * static_cast<double>(obs))
	                  : __builtin_nan("");
	obs_ct[i] = static_cast<int32_t>(2u * obs);
}

// ---------------------------------------------------------------------------
// HWE
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(64) void k_hwe_batch(const uint32_t *__restrict__ counts, uint32_t n, uint32_t midp,
                                                  double *__restrict__ ln_p) {
	const uint32_t i = blockIdx.x * 64u + threadIdx.x;
	if (i >= n) {
		return;
	}
	ln_p[i] = HweLnP(static_cast<int32_t>(counts[4 * i + 1]), static_cast<int32_t>(counts[4 * i]),
	                 static_cast<int32_t>(counts[4 * i + 2]), midp);
}

// The exact test's walks are as long as the het-count distribution is wide: ~ sqrt(N) * 2pq steps either side of
// the mode, so a lane with a common variant runs four times as long as its neighbour with a rare one and the wave
// waits for its longest lane (a third of k_hwe_batch's lane-cycles at uniform allele frequencies).  Large batches
// are therefore tested in order of the minor-allele fraction -- a counting sort into 1,024 classes, longest walks
// first -- so that the 64 variants of a wave walk about equally far.  Results land at the variants' own positions;
// the order inside a class is whatever the atomics made it and changes nothing.
constexpr uint32_t kHweClasses = 1024;

__device__ __forceinline__ uint32_t HweClass(const uint32_t *__restrict__ counts, uint32_t i) {
	const uint64_t hom1 = counts[4 * i], hets = counts[4 * i + 1], hom2 = counts[4 * i + 2];
	const uint64_t n2 = 2 * (hom1 + hets + hom2);
	if (n2 == 0) {
		return kHweClasses - 1u; // nothing to walk: with the shortest
	}
	const uint64_t rare = 2 * (hom1 < hom2 ? hom1 : hom2) + hets; // <= n2 / 2
	const uint32_t c = static_cast<uint32_t>(rare * (2 * kHweClasses - 2) / n2); // 0 .. 1023, widest distribution last
	return kHweClasses - 1u - c;                                                   // ... first
}

__global__ __launch_bounds__(256) void k_hwe_classes(const uint32_t *__restrict__ counts, uint32_t n,
                                                     uint32_t *__restrict__ bins) {
	__shared__ uint32_t s_bins[kHweClasses];
	for (uint32_t k = threadIdx.x; k < kHweClasses; k += 256u) {
		s_bins[k] = 0;
	}
	__syncthreads();
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n; i += gridDim.x * 256u) {
		atomicAdd(&s_bins[HweClass(counts, i)], 1u);
	}
	__syncthreads();
	for (uint32_t k = threadIdx.x; k < kHweClasses; k += 256u) {
		if (s_bins[k]) {
			atomicAdd(&bins[k], s_bins[k]);
		}
	}
}

// bins -> first position of every class (one workgroup of kHweClasses lanes)
__global__ __launch_bounds__(1024) void k_hwe_class_starts(uint32_t *__restrict__ bins) {
	__shared__ uint32_t s_scan[kHweClasses];
	const uint32_t k = threadIdx.x;
	const uint32_t mine = bins[k];
	s_scan[k] = mine;
	__syncthreads();
	for (uint32_t d = 1; d < kHweClasses; d <<= 1) {
		const uint32_t add = k >= d ? s_scan[k - d] : 0u;
		__syncthreads();
		s_scan[k] += add;
		__syncthreads();
	}
	bins[k] = s_scan[k] - mine;
}

__global__ __launch_bounds__(256) void k_hwe_order(const uint32_t *__restrict__ counts, uint32_t n,
                                                   uint32_t *__restrict__ cursor, uint32_t *__restrict__ order) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n) {
		order[atomicAdd(&cursor[HweClass(counts, i)], 1u)] = i;
	}
}

__global__ __launch_bounds__(64) void k_hwe_batch_ordered(const uint32_t *__restrict__ counts,
                                                          const uint32_t *__restrict__ order, uint32_t n, uint32_t midp,
                                                          double *__restrict__ ln_p) {
	const uint32_t slot = blockIdx.x * 64u + threadIdx.x;
	if (slot >= n) {
		return;
	}
	const uint32_t i = order[slot];
	ln_p[i] = HweLnP(static_cast<int32_t>(counts[4 * i + 1]), static_cast<int32_t>(counts[4 * i]),
	                 static_cast<int32_t>(counts[4 * i + 2]), midp);
}

// chrX exact test, one workgroup per variant: lanes share out the table columns (male A-allele
// counts), each walks the female het distribution of its columns, and the three sums meet in a
// block reduction.  strata[i] = {female_hets, female_hom1, female_hom2, male1, male2}.
__global__ __launch_bounds__(256) void k_hwe_xchr_batch(const int32_t *__restrict__ strata, uint32_t n, uint32_t midp,
                                                        double *__restrict__ ln_p) {
	__shared__ double s_obs;
	__shared__ double s_part[4][3];
	const uint32_t i = blockIdx.x;
	const int32_t fh = strata[5 * i], f1 = strata[5 * i + 1], f2 = strata[5 * i + 2], m1 = strata[5 * i + 3],
	              m2 = strata[5 * i + 4];
	const XchrShape x = MakeXchrShape(fh, f1, f2, m1, m2);
	if (x.nf + x.nm <= 0) {
		if (threadIdx.x == 0) {
			ln_p[i] = 0.0;
		}
		return;
	}
	if (threadIdx.x == 0) {
		s_obs = XchrObserved(x, fh, m1);
	}
	__syncthreads();
	const double p_obs = s_obs;
	if (!(p_obs > 0.0)) {
		if (threadIdx.x == 0) {
			ln_p[i] = -INFINITY;
		}
		return;
	}
	const double hi = p_obs * (1.0 + kHweTieEps);
	const double lo = p_obs * (1.0 - kHweTieEps);
	double total = 0.0, tail = 0.0, ties = 0.0;
	for (int64_t m = x.m_lo + threadIdx.x; m <= x.m_hi; m += 256) {
		XchrColumn(x, m, lo, hi, total, tail, ties);
	}
	double v[3] = {total, tail, ties};
#pragma unroll
	for (int k = 0; k < 3; k++) {
		for (int d = 32; d > 0; d >>= 1) {
			v[k] += __shfl_xor(v[k], d, 64);
		}
	}
	if ((threadIdx.x & 63u) == 0) {
		s_part[threadIdx.x >> 6][0] = v[0];
		s_part[threadIdx.x >> 6][1] = v[1];
		s_part[threadIdx.x >> 6][2] = v[2];
	}
	__syncthreads();
	if (threadIdx.x == 0) {
		total = s_part[0][0] + s_part[1][0] + s_part[2][0] + s_part[3][0];
		tail = s_part[0][1] + s_part[1][1] + s_part[2][1] + s_part[3][1];
		ties = s_part[0][2] + s_part[1][2] + s_part[2][2] + s_part[3][2];
		if (midp) {
			tail -= 0.5 * ties;
		}
		double pv = tail / total;
		ln_p[i] = log(pv > 1.0 ? 1.0 : pv);
	}
}

} // namespace

// ---------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------

hipError_t LaunchSynthFill(uint8_t *rows, uint64_t pitch, uint32_t sample_ct, uint32_t variant_begin,
                           uint32_t variant_ct, uint64_t seed, uint32_t miss_threshold, hipStream_t stream) {
	if (variant_ct == 0) {
		return hipSuccess;
	}
	const uint32_t dwords = static_cast<uint32_t>(pitch / 4);
	dim3 grid((dwords + 255) / 256, variant_ct < 65535u ? variant_ct : 65535u);
	hipLaunchKernelGGL(k_synth_fill, grid, dim3(256), 0, stream, rows, pitch, sample_ct, variant_begin, variant_ct,
	                   seed, miss_threshold);
	return hipGetLastError();
}

hipError_t LaunchSanitizeTail(uint8_t *rows, uint64_t pitch, uint32_t sample_ct, uint32_t variant_ct,
                              hipStream_t stream) {
	if (variant_ct == 0) {
		return hipSuccess;
	}
	const uint32_t record_bytes = (sample_ct + 3) / 4;
	const uint64_t total = static_cast<uint64_t>(variant_ct) * (pitch - record_bytes + 1);
	const uint64_t blocks = (total + 255) / 256;
	hipLaunchKernelGGL(k_sanitize_tail, dim3(static_cast<uint32_t>(blocks)), dim3(256), 0, stream, rows, pitch,
	                   sample_ct, variant_ct);
	return hipGetLastError();
}

hipError_t LaunchCounts(const RowView &view, uint32_t v_first, const uint32_t *vlist, uint32_t v_count,
                        const uint8_t *mask2, uint32_t n_eff, uint32_t *out, hipStream_t stream) {
	if (v_count == 0) {
		return hipSuccess;
	}
	const uint32_t chunks = static_cast<uint32_t>((static_cast<uint64_t>(view.record_bytes) + 15) / 16);
	const uint4 *m = reinterpret_cast<const uint4 *>(mask2);
	uint4 *o = reinterpret_cast<uint4 *>(out);
	if (chunks >= 256) {
		// 256 CUs x 8 resident workgroups; beyond that the grid strides
		const uint32_t grid = v_count < (1u << 20) ? v_count : (1u << 20);
		if (mask2) {
			hipLaunchKernelGGL(k_counts_block<true>, dim3(grid), dim3(256), 0, stream, view.rows, view.pitch, chunks,
			                   v_first, vlist, v_count, m, n_eff, o);
		} else {
			hipLaunchKernelGGL(k_counts_block<false>, dim3(grid), dim3(256), 0, stream, view.rows, view.pitch, chunks,
			                   v_first, vlist, v_count, m, n_eff, o);
		}
	} else {
		const uint32_t blocks = (v_count + 3) / 4;
		const uint32_t grid = blocks < (1u << 20) ? blocks : (1u << 20);
		if (mask2) {
			hipLaunchKernelGGL(k_counts_wave<true>, dim3(grid), dim3(256), 0, stream, view.rows, view.pitch, chunks,
			                   v_first, vlist, v_count, m, n_eff, o);
		} else {
			hipLaunchKernelGGL(k_counts_wave<false>, dim3(grid), dim3(256), 0, stream, view.rows, view.pitch, chunks,
			                   v_first, vlist, v_count, m, n_eff, o);
		}
	}
	return hipGetLastError();
}

hipError_t LaunchFreqFromCounts(const uint32_t *counts, uint32_t n, double *alt_freq, int32_t *obs_ct,
                                hipStream_t stream) {
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_freq_from_counts, dim3((n + 255) / 256), dim3(256), 0, stream,
	                   reinterpret_cast<const uint4 *>(counts), n, alt_freq, obs_ct);
	return hipGetLastError();
}

void MissingPerSamplePlan(uint32_t record_bytes, uint32_t v_count, uint32_t *slice_len_out, uint32_t *slices_out) {
	const uint32_t chunks = (record_bytes + 15) / 16;
	const uint32_t col_blocks = (chunks + 255) / 256;
	// row slices for ~kWant workgroups (768 fit the chip at three per CU: several rounds of them, so that the last,
	// partly filled round is a small share of the launch), each a multiple of 16 rows
	static const uint32_t want_wgs = [] {
		const char *e = std::getenv("PGH_FUSED_WGS"); // tuning knob
		return e && std::atoi(e) > 0 ? static_cast<uint32_t>(std::atoi(e)) : 2048u;
	}();
	uint32_t want_slices = (want_wgs + col_blocks - 1) / col_blocks;
	if (want_slices > 1024) {
		want_slices = 1024;
	}
	uint32_t slice_len = (v_count + want_slices - 1) / want_slices;
	slice_len = ((slice_len + 15) / 16) * 16;
	if (slice_len < 96) {
		slice_len = 96;
	}
	if (slice_len > 65280u) {
		slice_len = 65280u; // 16 planes of column counters
	}
	*slice_len_out = slice_len;
	*slices_out = v_count ? (v_count + slice_len - 1) / slice_len : 0;
}

// slices of k_class_cols1: >= ~2048 workgroups where the rows allow it, multiples of 16 rows, <= 65280 rows
static void ClassCols1Plan(uint32_t record_bytes, uint32_t v_count, uint32_t *slice_len_out, uint32_t *slices_out) {
	const uint32_t chunks = (record_bytes + 15) / 16;
	const uint32_t col_blocks = (chunks + kCols1Threads - 1) / kCols1Threads;
	uint32_t want_slices = (2048 + col_blocks - 1) / col_blocks;
	if (want_slices > 1024) {
		want_slices = 1024;
	}
	uint32_t slice_len = (v_count + want_slices - 1) / want_slices;
	slice_len = ((slice_len + 15) / 16) * 16;
	if (slice_len < 96) {
		slice_len = 96;
	}
	if (slice_len > 65280u) {
		slice_len = 65280u;
	}
	*slice_len_out = slice_len;
	*slices_out = v_count ? (v_count + slice_len - 1) / slice_len : 0;
}

size_t MissingPerSampleScratchBytes(uint32_t record_bytes, uint32_t v_count) {
	// the fused kernel's uint32 partial rows, or k_class_cols1's plane words, whichever is larger
	uint32_t slice_len, slices;
	MissingPerSamplePlan(record_bytes, v_count, &slice_len, &slices);
	const uint64_t chunks = (record_bytes + 15) / 16;
	// the fused kernel: plane slabs as below (its slices are never more) + 16 bytes per column block and variant
	const uint64_t fused = static_cast<uint64_t>(slices) * chunks * 2 * kCols1Planes * sizeof(uint32_t) + 16 +
	                       ((chunks + 255) / 256) * static_cast<uint64_t>(v_count) * 16;
	ClassCols1Plan(record_bytes, v_count, &slice_len, &slices);
	const uint64_t planes = static_cast<uint64_t>(slices) * chunks * 2 * kCols1Planes * sizeof(uint32_t);
	return static_cast<size_t>(std::max(fused, planes));
}

hipError_t LaunchMissingPerSample(const RowView &view, uint32_t v_first, const uint32_t *vlist, uint32_t v_count,
                                  const uint32_t *row_flags, uint32_t *scratch, uint32_t *out, hipStream_t stream) {
	return LaunchClassPerSample(view, 3, v_first, vlist, v_count, row_flags, scratch, out, stream);
}

hipError_t LaunchClassPerSample(const RowView &view, int genotype_class, uint32_t v_first, const uint32_t *vlist,
                                uint32_t v_count, const uint32_t *row_flags, uint32_t *scratch, uint32_t *out,
                                hipStream_t stream) {
	hipError_t e = hipMemsetAsync(out, 0, sizeof(uint32_t) * view.sample_ct, stream);
	if (v_count == 0 || e != hipSuccess) {
		return e;
	}
	const uint32_t chunks = static_cast<uint32_t>((static_cast<uint64_t>(view.record_bytes) + 15) / 16);
	const uint32_t col_blocks = (chunks + kCols1Threads - 1) / kCols1Threads;
	uint32_t slice_len, slices;
	ClassCols1Plan(view.record_bytes, v_count, &slice_len, &slices);
	const uint64_t stride = static_cast<uint64_t>(chunks) * 2 * kCols1Planes; // dwords per slice
#define PGH_COLS(CLASS)                                                                                                \
	hipLaunchKernelGGL(k_class_cols1<CLASS>, dim3(col_blocks, slices), dim3(kCols1Threads), 0, stream, view.rows,      \
	                   view.pitch, chunks, v_first, vlist, v_count, slice_len, row_flags, scratch, stride)
	if (genotype_class == 1) {
		PGH_COLS(1);
	} else if (genotype_class == 2) {
		PGH_COLS(2);
	} else {
		PGH_COLS(3);
	}
#undef PGH_COLS
	e = hipGetLastError();
	if (e != hipSuccess) {
		return e;
	}
	hipLaunchKernelGGL(k_sum_cols1, dim3((chunks + 255) / 256, (slices + kCols1SumGroup - 1) / kCols1SumGroup), dim3(256), 0,
	                   stream, scratch, stride, chunks, slices, view.sample_ct, out);
	return hipGetLastError();
}

size_t ClassCounts3ScratchBytes(uint32_t record_bytes) {
	const uint64_t chunks = (static_cast<uint64_t>(record_bytes) + 7) / 8;
	return static_cast<size_t>(kCols3Super) * kCols3Words * kCols3Planes * chunks * 4u;
}

hipError_t LaunchClassCounts3(const RowView &view, uint32_t v_first, const uint32_t *vlist, uint32_t v_count,
                              uint8_t *scratch, uint32_t *out, uint32_t out_stride, hipStream_t stream) {
	if (v_count == 0) {
		return hipMemsetAsync(out, 0, sizeof(uint32_t) * 3ull * out_stride, stream);
	}
	const uint32_t chunks = static_cast<uint32_t>((static_cast<uint64_t>(view.record_bytes) + 7) / 8);
	const uint32_t col_blocks = (chunks + kCols3Threads - 1) / kCols3Threads;
	const uint32_t sum_blocks = (chunks + 255) / 256;
	const uint64_t slab_stride = static_cast<uint64_t>(chunks) * kCols3Words * kCols3Planes; // dwords per slice
	const uint32_t rows_per_launch = kCols3Super * kCols3Rows;
	uint32_t *slabs = reinterpret_cast<uint32_t *>(scratch);
	hipError_t e = hipMemsetAsync(out, 0, sizeof(uint32_t) * 3ull * out_stride, stream);
	if (e != hipSuccess) {
		return e;
	}
	for (uint32_t done = 0; done < v_count; done += rows_per_launch) {
		const uint32_t n_rows = std::min(rows_per_launch, v_count - done);
		const uint32_t slices = (n_rows + kCols3Rows - 1) / kCols3Rows;
		hipLaunchKernelGGL(k_class_cols3, dim3(col_blocks, slices), dim3(kCols3Threads), 0, stream, view.rows, view.pitch, chunks,
		                   v_first + done, vlist ? vlist + done : nullptr, n_rows, slabs, slab_stride);
		e = hipGetLastError();
		if (e != hipSuccess) {
			return e;
		}
		hipLaunchKernelGGL(k_sum_class_bits, dim3(sum_blocks, (slices + kCols3SumGroup - 1) / kCols3SumGroup), dim3(256), 0,
		                   stream, slabs, slab_stride, chunks, slices, view.sample_ct, out_stride, out);
		e = hipGetLastError();
		if (e != hipSuccess) {
			return e;
		}
	}
	return hipSuccess;
}

hipError_t LaunchFusedTally(const RowView &view, uint32_t v_first, uint32_t v_count, uint32_t *scratch,
                            uint32_t *counts, uint32_t *missing_per_sample, hipStream_t stream, bool accumulate) {
	hipError_t e = hipSuccess;
	if (!accumulate) {
		e = hipMemsetAsync(missing_per_sample, 0, sizeof(uint32_t) * view.sample_ct, stream);
	}
	if (v_count == 0 || e != hipSuccess) {
		return e;
	}
	const uint32_t chunks = static_cast<uint32_t>((static_cast<uint64_t>(view.record_bytes) + 15) / 16);
	const uint32_t col_blocks = (chunks + 255) / 256;
	uint32_t slice_len, slices;
	MissingPerSamplePlan(view.record_bytes, v_count, &slice_len, &slices);
	slice_len = (slice_len + kFusedRows - 1) / kFusedRows * kFusedRows; // <= 65280, so never more slices
	slices = (v_count + slice_len - 1) / slice_len;
	const uint64_t stride = static_cast<uint64_t>(chunks) * 2 * kCols1Planes; // dwords per slice (k_class_cols1's slabs)
	// scratch: the slices' planes, then the column blocks' row sums ([col_blocks][v_count] x 16 B)
	uint4 *partials = reinterpret_cast<uint4 *>(scratch + (static_cast<uint64_t>(slices) * stride + 3) / 4 * 4);
	hipLaunchKernelGGL(k_fused_tally, dim3(col_blocks, slices), dim3(256), 0, stream, view.rows, view.pitch, chunks,
	                   v_first, v_count, slice_len, partials, scratch, stride);
	hipLaunchKernelGGL(k_finish_tallies, dim3((v_count + 255) / 256), dim3(256), 0, stream, partials, col_blocks,
	                   reinterpret_cast<uint4 *>(counts), v_count, view.sample_ct);
	// the slices' planes -> per-sample counts, added onto missing_per_sample
	hipLaunchKernelGGL(k_sum_cols1, dim3((chunks + 255) / 256, (slices + kCols1SumGroup - 1) / kCols1SumGroup), dim3(256), 0,
	                   stream, scratch, stride, chunks, slices, view.sample_ct, missing_per_sample);
	return hipGetLastError();
}

size_t HweOrderScratchBytes(uint32_t n) {
	return n >= kHweOrderMin ? sizeof(uint32_t) * (static_cast<size_t>(n) + kHweClasses) : 0;
}

hipError_t LaunchHweBatch(const uint32_t *counts, uint32_t n, uint32_t midp, double *ln_p, hipStream_t stream,
                          void *order_scratch) {
	if (n == 0) {
		return hipSuccess;
	}
	if (order_scratch && n >= kHweOrderMin) {
		uint32_t *bins = static_cast<uint32_t *>(order_scratch), *order = bins + kHweClasses;
		hipError_t e = hipMemsetAsync(bins, 0, sizeof(uint32_t) * kHweClasses, stream);
		if (e != hipSuccess) {
			return e;
		}
		const uint32_t blocks = (n + 255u) / 256u;
		hipLaunchKernelGGL(k_hwe_classes, dim3(blocks < 1024u ? blocks : 1024u), dim3(256), 0, stream, counts, n, bins);
		hipLaunchKernelGGL(k_hwe_class_starts, dim3(1), dim3(kHweClasses), 0, stream, bins);
		hipLaunchKernelGGL(k_hwe_order, dim3(blocks), dim3(256), 0, stream, counts, n, bins, order);
		hipLaunchKernelGGL(k_hwe_batch_ordered, dim3((n + 63) / 64), dim3(64), 0, stream, counts, order, n, midp, ln_p);
		return hipGetLastError();
	}
	hipLaunchKernelGGL(k_hwe_batch, dim3((n + 63) / 64), dim3(64), 0, stream, counts, n, midp, ln_p);
	return hipGetLastError();
}

hipError_t LaunchHweXchrBatch(const int32_t *strata, uint32_t n, uint32_t midp, double *ln_p, hipStream_t stream) {
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_hwe_xchr_batch, dim3(n), dim3(256), 0, stream, strata, n, midp, ln_p);
	return hipGetLastError();
}

} // namespace pgh

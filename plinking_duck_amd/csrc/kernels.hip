// kernels.hip -- hand-written gfx950 (CDNA4) kernels of libpgenhip.
//
// Data layout: the genotype matrix is variant-major; row v holds ceil(N/4)
// bytes of packed 2-bit calls (00 hom-ref, 01 het, 10 hom-alt, 11 missing;
// sample s in bits 2*(s%4) of byte s/4) followed by zero bytes up to `pitch`
// (a multiple of 16, so every row can be streamed as whole 16-byte lanes and
// the pad decodes as hom-ref, which every kernel cancels against N).
//
// All of these are HBM-bound byte/bit kernels: 16 B per lane coalesced loads,
// v_bcnt_u32_b32 tallies with its free accumulate operand, wave reductions by
// DPP/ds_swizzle shuffles, LDS only for the cross-wave step and table staging.
#include "kernels.hpp"

#include <cstdlib>

#include "hwe_core.hpp"
#include "synth.hpp"

namespace pgh {

namespace {

constexpr uint32_t kLow = 0x55555555u; // low bit of every 2-bit slot

__device__ __forceinline__ uint32_t WaveSum(uint32_t x) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		x += __shfl_xor(x, off, 64);
	}
	return x;
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint4 LoadStream(const uint4 *p) {
	// once-read stream: non-temporal so it does not evict the mask / tables from L2
	const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p));
	return make_uint4(v.x, v.y, v.z, v.w);
}

__device__ __forceinline__ void StoreStream(uint4 *p, const uint4 &o) {
	u32x4 v = {o.x, o.y, o.z, o.w};
	__builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(p));
}

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
// A lane owns one 16-byte column (64 samples) and walks down a slice of rows.
// The indicator m = w & (w>>1) & 0x5555.. has one bit per 2-bit slot, so it is
// added SWAR-style: 2-bit fields (<=3 rows) -> 4-bit fields (<=15) -> 8-bit
// fields (<=255) -> 64 uint32 registers.  Each slice writes its totals to its
// own slab row with plain stores; k_sum_slabs adds the slices.  No atomics: a
// lane's 64 counters sit 256 B apart from its neighbour's, the worst shape for
// the memory-side atomic units.

struct MissAcc {
	uint32_t a4[8];
	uint32_t a8[16];
	uint32_t a32[64];
};

__device__ __forceinline__ uint32_t MissBits(uint32_t w) {
	return w & (w >> 1) & kLow;
}

__device__ __forceinline__ void Fold2To4(MissAcc &acc, const uint32_t a2[4]) {
#pragma unroll
	for (int j = 0; j < 4; j++) {
		acc.a4[2 * j] += a2[j] & 0x33333333u;
		acc.a4[2 * j + 1] += (a2[j] >> 2) & 0x33333333u;
	}
}

__device__ __forceinline__ void Fold4To8(MissAcc &acc) {
#pragma unroll
	for (int j = 0; j < 4; j++) {
		acc.a8[4 * j + 0] += acc.a4[2 * j] & 0x0f0f0f0fu;
		acc.a8[4 * j + 1] += (acc.a4[2 * j] >> 4) & 0x0f0f0f0fu;
		acc.a8[4 * j + 2] += acc.a4[2 * j + 1] & 0x0f0f0f0fu;
		acc.a8[4 * j + 3] += (acc.a4[2 * j + 1] >> 4) & 0x0f0f0f0fu;
		acc.a4[2 * j] = 0;
		acc.a4[2 * j + 1] = 0;
	}
}

__device__ __forceinline__ void Fold8To32(MissAcc &acc) {
	// a8[4j+q] byte b counts sample 16j + 4b + {0,2,1,3}[q]
#pragma unroll
	for (int j = 0; j < 4; j++) {
#pragma unroll
		for (int q = 0; q < 4; q++) {
			const uint32_t word = acc.a8[4 * j + q];
			acc.a8[4 * j + q] = 0;
			const int within = (q == 0) ? 0 : (q == 1 ? 2 : (q == 2 ? 1 : 3));
#pragma unroll
			for (int b = 0; b < 4; b++) {
				acc.a32[16 * j + 4 * b + within] += (word >> (8 * b)) & 0xffu;
			}
		}
	}
}

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

template <int CLASS>
__global__ __launch_bounds__(256) void k_missing_cols(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                      uint32_t chunks, uint32_t v_first,
                                                      const uint32_t *__restrict__ vlist, uint32_t v_count,
                                                      uint32_t slice_len, const uint32_t *__restrict__ row_flags,
                                                      uint32_t *__restrict__ slabs, uint32_t slab_stride) {
	// row_flags (optional): rows whose low byte is zero are not counted (skipped scored variants)
	const uint32_t col = blockIdx.x * 256u + threadIdx.x;
	if (col >= chunks) {
		return;
	}
	const uint32_t i_begin = blockIdx.y * slice_len;
	const uint32_t i_end = min(i_begin + slice_len, v_count);
	MissAcc acc;
#pragma unroll
	for (int j = 0; j < 8; j++) {
		acc.a4[j] = 0;
	}
#pragma unroll
	for (int j = 0; j < 16; j++) {
		acc.a8[j] = 0;
	}
#pragma unroll
	for (int j = 0; j < 64; j++) {
		acc.a32[j] = 0;
	}
	uint32_t n4 = 0, n8 = 0; // rows folded into the 4-bit / 8-bit fields so far
	uint32_t i = i_begin;
	auto row_ptr = [&](uint32_t idx) {
		const uint32_t v = vlist ? vlist[idx] : v_first + idx;
		return reinterpret_cast<const uint4 *>(rows + static_cast<uint64_t>(v) * pitch) + col;
	};
	auto fold = [&](const uint32_t a2[4], uint32_t take) {
		Fold2To4(acc, a2);
		n4 += take;
		if (n4 + 3 > 15) {
			Fold4To8(acc);
			n8 += n4;
			n4 = 0;
			if (n8 + 15 > 255) {
				Fold8To32(acc);
				n8 = 0;
			}
		}
	};
	// main loop: six independent 16-byte loads in flight per lane
	while (!row_flags && i + 6 <= i_end) {
		const uint4 w0 = LoadStream(row_ptr(i));
		const uint4 w1 = LoadStream(row_ptr(i + 1));
		const uint4 w2 = LoadStream(row_ptr(i + 2));
		const uint4 w3 = LoadStream(row_ptr(i + 3));
		const uint4 w4 = LoadStream(row_ptr(i + 4));
		const uint4 w5 = LoadStream(row_ptr(i + 5));
		uint32_t a[4], b[4];
		a[0] = ClassBits<CLASS>(w0.x) + ClassBits<CLASS>(w1.x) + ClassBits<CLASS>(w2.x);
		a[1] = ClassBits<CLASS>(w0.y) + ClassBits<CLASS>(w1.y) + ClassBits<CLASS>(w2.y);
		a[2] = ClassBits<CLASS>(w0.z) + ClassBits<CLASS>(w1.z) + ClassBits<CLASS>(w2.z);
		a[3] = ClassBits<CLASS>(w0.w) + ClassBits<CLASS>(w1.w) + ClassBits<CLASS>(w2.w);
		b[0] = ClassBits<CLASS>(w3.x) + ClassBits<CLASS>(w4.x) + ClassBits<CLASS>(w5.x);
		b[1] = ClassBits<CLASS>(w3.y) + ClassBits<CLASS>(w4.y) + ClassBits<CLASS>(w5.y);
		b[2] = ClassBits<CLASS>(w3.z) + ClassBits<CLASS>(w4.z) + ClassBits<CLASS>(w5.z);
		b[3] = ClassBits<CLASS>(w3.w) + ClassBits<CLASS>(w4.w) + ClassBits<CLASS>(w5.w);
		fold(a, 3);
		fold(b, 3);
		i += 6;
	}
	while (i < i_end) {
		if (!row_flags || (row_flags[i] & 0xffu)) { // wave-uniform
			const uint4 w0 = LoadStream(row_ptr(i));
			uint32_t a[4] = {ClassBits<CLASS>(w0.x), ClassBits<CLASS>(w0.y), ClassBits<CLASS>(w0.z), ClassBits<CLASS>(w0.w)};
			fold(a, 1);
		}
		i += 1;
	}
	Fold4To8(acc);
	Fold8To32(acc);
	uint4 *dst = reinterpret_cast<uint4 *>(slabs + static_cast<uint64_t>(blockIdx.y) * slab_stride + col * 64u);
#pragma unroll
	for (int k = 0; k < 16; k++) {
		dst[k] = make_uint4(acc.a32[4 * k], acc.a32[4 * k + 1], acc.a32[4 * k + 2], acc.a32[4 * k + 3]);
	}
}

// ---------------------------------------------------------------------------
// fused pass: per-variant class tallies AND per-sample missing tallies
// ---------------------------------------------------------------------------
//
// plink_freq + plink_hardy + plink_missing (both modes) need the row sums and the
// column sums of the same matrix; this kernel reads every byte once for both.
// Ownership is by column (as k_missing_cols): a lane keeps its 64 samples' missing
// counters in registers.  The row sums cross lanes: each lane's per-row popcounts
// go through an LDS tile [12 rows][256 lanes] (packed 10-bit fields), a 16-lane
// shuffle tree finishes the row, and 36 lanes add the workgroup's partials to the
// per-variant totals with one coalesced atomic instruction per 12 rows.
constexpr uint32_t kFusedRows = 12;

// column counters of the fused kernel: as MissAcc but the last level is 16-bit
// (<= 65535 rows per slice), which keeps the kernel under 168 VGPRs
struct MissAcc16 {
	uint32_t a4[8];
	uint32_t a8[16];
	uint32_t a16[32];
};

__device__ __forceinline__ void Fold8To16(MissAcc16 &acc) {
	// a16[2i + e] half h <- byte 2h + e of a8[i]
#pragma unroll
	for (int i = 0; i < 16; i++) {
		acc.a16[2 * i] += acc.a8[i] & 0x00ff00ffu;
		acc.a16[2 * i + 1] += (acc.a8[i] >> 8) & 0x00ff00ffu;
		acc.a8[i] = 0;
	}
}

__global__ __launch_bounds__(256) void k_fused_tally(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                     uint32_t chunks, uint32_t v_first, uint32_t v_count,
                                                     uint32_t slice_len, uint32_t *__restrict__ tallies,
                                                     uint32_t *__restrict__ slabs, uint32_t slab_stride) {
	__shared__ uint32_t s_p[kFusedRows][256];
	__shared__ uint32_t s_res[kFusedRows][3];
	const uint32_t col = blockIdx.x * 256u + threadIdx.x;
	const bool live = col < chunks;
	const uint32_t i_begin = blockIdx.y * slice_len;
	const uint32_t i_end = min(i_begin + slice_len, v_count);
	MissAcc16 acc;
#pragma unroll
	for (int j = 0; j < 8; j++) {
		acc.a4[j] = 0;
	}
#pragma unroll
	for (int j = 0; j < 16; j++) {
		acc.a8[j] = 0;
	}
#pragma unroll
	for (int j = 0; j < 32; j++) {
		acc.a16[j] = 0;
	}
	uint32_t n4 = 0, n8 = 0;
	auto fold = [&](const uint32_t a2[4], uint32_t take) {
#pragma unroll
		for (int j = 0; j < 4; j++) {
			acc.a4[2 * j] += a2[j] & 0x33333333u;
			acc.a4[2 * j + 1] += (a2[j] >> 2) & 0x33333333u;
		}
		n4 += take;
		if (n4 + 3 > 15) {
#pragma unroll
			for (int j = 0; j < 4; j++) {
				acc.a8[4 * j + 0] += acc.a4[2 * j] & 0x0f0f0f0fu;
				acc.a8[4 * j + 1] += (acc.a4[2 * j] >> 4) & 0x0f0f0f0fu;
				acc.a8[4 * j + 2] += acc.a4[2 * j + 1] & 0x0f0f0f0fu;
				acc.a8[4 * j + 3] += (acc.a4[2 * j + 1] >> 4) & 0x0f0f0f0fu;
				acc.a4[2 * j] = 0;
				acc.a4[2 * j + 1] = 0;
			}
			n8 += n4;
			n4 = 0;
			if (n8 + 15 > 255) {
				Fold8To16(acc);
				n8 = 0;
			}
		}
	};
	// one row: class popcounts packed as lo | hi << 10 | both << 20, missing bits into a2
	auto one_row = [&](const uint4 &w, uint32_t a2[4]) -> uint32_t {
		const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
		uint32_t lo_ct = 0, hi_ct = 0, both_ct = 0;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const uint32_t lo = ws[j] & kLow;
			const uint32_t hi = (ws[j] >> 1) & kLow;
			const uint32_t both = lo & hi;
			lo_ct += __popc(lo);
			hi_ct += __popc(hi);
			both_ct += __popc(both);
			a2[j] += both;
		}
		return lo_ct | (hi_ct << 10) | (both_ct << 20);
	};
	const uint4 zero4 = make_uint4(0, 0, 0, 0);
	auto load_row = [&](uint32_t idx) {
		return live ? LoadStream(reinterpret_cast<const uint4 *>(rows + static_cast<uint64_t>(v_first + idx) * pitch) +
		                         col)
		            : zero4;
	};
	for (uint32_t i = i_begin; i < i_end; i += kFusedRows) {
		const uint32_t nb = min(kFusedRows, i_end - i);
		if (nb == kFusedRows) {
#pragma unroll
			for (uint32_t h = 0; h < kFusedRows; h += 6) {
				const uint4 w0 = load_row(i + h), w1 = load_row(i + h + 1), w2 = load_row(i + h + 2);
				const uint4 w3 = load_row(i + h + 3), w4 = load_row(i + h + 4), w5 = load_row(i + h + 5);
				uint32_t a[4] = {0, 0, 0, 0};
				s_p[h + 0][threadIdx.x] = one_row(w0, a);
				s_p[h + 1][threadIdx.x] = one_row(w1, a);
				s_p[h + 2][threadIdx.x] = one_row(w2, a);
				fold(a, 3);
				uint32_t b[4] = {0, 0, 0, 0};
				s_p[h + 3][threadIdx.x] = one_row(w3, b);
				s_p[h + 4][threadIdx.x] = one_row(w4, b);
				s_p[h + 5][threadIdx.x] = one_row(w5, b);
				fold(b, 3);
			}
		} else {
			for (uint32_t r = 0; r < nb; r++) {
				const uint4 w = load_row(i + r);
				uint32_t a[4] = {0, 0, 0, 0};
				s_p[r][threadIdx.x] = one_row(w, a);
				fold(a, 1);
			}
		}
		__syncthreads();
		// row sums: lane t -> (row t >> 4, sixteenth t & 15) sums 16 packed lanes (8 + 8 so the
		// 10-bit fields cannot overflow), then a 16-lane shuffle tree
		{
			const uint32_t row = threadIdx.x >> 4, part = threadIdx.x & 15u;
			uint32_t lo = 0, hi = 0, both = 0;
			if (row < nb) {
				const uint4 *src = reinterpret_cast<const uint4 *>(&s_p[row][part * 16u]);
				const uint4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
				const uint32_t p0 = q0.x + q0.y + q0.z + q0.w + q1.x + q1.y + q1.z + q1.w;
				const uint32_t p1 = q2.x + q2.y + q2.z + q2.w + q3.x + q3.y + q3.z + q3.w;
				lo = (p0 & 0x3ffu) + (p1 & 0x3ffu);
				hi = ((p0 >> 10) & 0x3ffu) + ((p1 >> 10) & 0x3ffu);
				both = (p0 >> 20) + (p1 >> 20);
			}
#pragma unroll
			for (int off = 8; off > 0; off >>= 1) {
				lo += __shfl_xor(lo, off, 64);
				hi += __shfl_xor(hi, off, 64);
				both += __shfl_xor(both, off, 64);
			}
			if (part == 0 && row < nb) {
				s_res[row][0] = lo;
				s_res[row][1] = hi;
				s_res[row][2] = both;
			}
		}
		__syncthreads();
		if (threadIdx.x < nb * 3u) {
			const uint32_t r = threadIdx.x / 3u, f = threadIdx.x % 3u;
			atomicAdd(tallies + 4ull * (i + r) + 1u + f, s_res[r][f]);
		}
	}
	if (live) {
		// drain the narrow levels, then unscramble: a16[2(4j+q)+e] half h counts sample
		// 16j + 8h + 4e + {0,2,1,3}[q]
#pragma unroll
		for (int j = 0; j < 4; j++) {
			acc.a8[4 * j + 0] += acc.a4[2 * j] & 0x0f0f0f0fu;
			acc.a8[4 * j + 1] += (acc.a4[2 * j] >> 4) & 0x0f0f0f0fu;
			acc.a8[4 * j + 2] += acc.a4[2 * j + 1] & 0x0f0f0f0fu;
			acc.a8[4 * j + 3] += (acc.a4[2 * j + 1] >> 4) & 0x0f0f0f0fu;
		}
		Fold8To16(acc);
		uint32_t *dst = slabs + static_cast<uint64_t>(blockIdx.y) * slab_stride + col * 64u;
#pragma unroll
		for (int j = 0; j < 4; j++) {
#pragma unroll
			for (int h = 0; h < 2; h++) {
#pragma unroll
				for (int e = 0; e < 2; e++) {
					// samples 16j + 8h + 4e + {0,1,2,3}  <-  q = {0,2,1,3}
					uint4 o;
					o.x = (acc.a16[2 * (4 * j + 0) + e] >> (16 * h)) & 0xffffu;
					o.y = (acc.a16[2 * (4 * j + 2) + e] >> (16 * h)) & 0xffffu;
					o.z = (acc.a16[2 * (4 * j + 1) + e] >> (16 * h)) & 0xffffu;
					o.w = (acc.a16[2 * (4 * j + 3) + e] >> (16 * h)) & 0xffffu;
					reinterpret_cast<uint4 *>(dst)[4 * j + 2 * h + e] = o;
				}
			}
		}
	}
}

// (., lo, hi, both) -> (hom_ref, het, hom_alt, missing)
__global__ __launch_bounds__(256) void k_finish_tallies(uint4 *__restrict__ tallies, uint32_t n, uint32_t n_eff) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) {
		return;
	}
	const uint4 t = tallies[i];
	uint4 r;
	r.y = t.y - t.w;
	r.z = t.z - t.w;
	r.w = t.w;
	r.x = n_eff - r.y - r.z - r.w;
	tallies[i] = r;
}

// out[s] = sum over slices of slabs[slice][s]
__global__ __launch_bounds__(256) void k_sum_slabs(const uint32_t *__restrict__ slabs, uint32_t slab_stride,
                                                   uint32_t n_slabs, uint32_t n, uint32_t *__restrict__ out) {
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s >= n) {
		return;
	}
	uint32_t acc = 0;
	for (uint32_t k = 0; k < n_slabs; k++) {
		acc += slabs[static_cast<uint64_t>(k) * slab_stride + s];
	}
	out[s] = acc;
}

// ---------------------------------------------------------------------------
// 2-bit -> int8 unpack
// ---------------------------------------------------------------------------

// 8 bits (4 calls) -> 4 bytes, one call per byte
__device__ __forceinline__ uint32_t Spread4(uint32_t x) {
	uint32_t t = (x | (x << 12)) & 0x000f000fu;
	return (t | (t << 6)) & 0x03030303u;
}

__global__ __launch_bounds__(256) void k_unpack(const uint8_t *__restrict__ rows, uint64_t pitch, uint32_t sample_ct,
                                                uint32_t v_first, uint32_t v_count, int8_t *__restrict__ out,
                                                uint64_t out_pitch, uint64_t *__restrict__ validity,
                                                uint32_t fill4) {
	const uint32_t dwords = (sample_ct + 15) / 16;           // input dwords holding data
	const uint32_t val_words16 = ((sample_ct + 63) / 64) * 4; // uint16 slots per validity row
	const uint32_t d = blockIdx.x * 256u + threadIdx.x;
	if (d >= val_words16) {
		return;
	}
	for (uint32_t i = blockIdx.y; i < v_count; i += gridDim.y) {
		const uint8_t *row = rows + static_cast<uint64_t>(v_first + i) * pitch;
		uint32_t valid16 = 0;
		if (d < dwords) {
			const uint32_t w = __builtin_nontemporal_load(reinterpret_cast<const uint32_t *>(row) + d);
			uint4 o;
			uint32_t vbits = 0;
			uint32_t *op = &o.x;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t t = Spread4((w >> (8 * k)) & 0xffu);
				const uint32_t miss = t & (t >> 1) & 0x01010101u; // 1 in each missing byte
				const uint32_t mm = miss * 0xffu;                 // 0xff in each missing byte
				op[k] = (t & ~mm) | (fill4 & mm);
				vbits |= (((miss * 0x01020408u) >> 24) & 0xfu) << (4 * k);
			}
			valid16 = ~vbits & 0xffffu;
			const uint32_t left = sample_ct - d * 16u;
			if (left < 16u) {
				valid16 &= (1u << left) - 1u;
			}
			if (out) {
				StoreStream(reinterpret_cast<uint4 *>(out + static_cast<uint64_t>(i) * out_pitch) + d, o);
			}
		}
		if (validity) {
			uint16_t *vrow = reinterpret_cast<uint16_t *>(validity + static_cast<uint64_t>(i) * (val_words16 / 4));
			vrow[d] = static_cast<uint16_t>(valid16);
		}
	}
}

// Wide form for long rows: a lane takes 16 bytes (64 calls), expands them to 64 output
// bytes + one 64-bit validity word, and the wave's 4 KiB of output goes through LDS so
// that every global store instruction writes 1 KiB contiguous (lane-major -> piece-major).
template <bool NT_STORE>
__global__ __launch_bounds__(256) void k_unpack_wide(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                     uint32_t sample_ct, uint32_t v_first, uint32_t v_count,
                                                     int8_t *__restrict__ out, uint64_t out_pitch,
                                                     uint64_t *__restrict__ validity, uint32_t fill4) {
	__shared__ uint4 s_tile[4][256]; // per wave: 64 lanes x 4 pieces of 16 bytes
	const uint32_t chunks = (sample_ct + 63) / 64; // 16-byte input chunks == validity words per row
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t col = blockIdx.x * 256u + threadIdx.x;
	const uint32_t wave_col0 = blockIdx.x * 256u + wave * 64u;
	if (wave_col0 >= chunks) {
		return; // whole wave past the row (no barriers in this kernel)
	}
	for (uint32_t i = blockIdx.y; i < v_count; i += gridDim.y) {
		const uint8_t *row = rows + static_cast<uint64_t>(v_first + i) * pitch;
		uint4 w = make_uint4(0, 0, 0, 0);
		if (col < chunks) {
			w = LoadStream(reinterpret_cast<const uint4 *>(row) + col);
		}
		const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
		uint64_t vbits = 0;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			uint4 o;
			uint32_t *op = &o.x;
			uint32_t miss16 = 0;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t t = Spread4((ws[j] >> (8 * k)) & 0xffu);
				const uint32_t miss = t & (t >> 1) & 0x01010101u;
				const uint32_t mm = miss * 0xffu;
				op[k] = (t & ~mm) | (fill4 & mm);
				miss16 |= (((miss * 0x01020408u) >> 24) & 0xfu) << (4 * k);
			}
			vbits |= static_cast<uint64_t>(~miss16 & 0xffffu) << (16 * j);
			s_tile[wave][lane * 4u + j] = o;
		}
		// same-wave LDS exchange: LDS serves one wave's instructions in order, so a
		// wavefront-scope fence (ordering for the compiler, a waitcnt for the hardware) is
		// all the tile needs -- no workgroup barrier
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
		if (out) {
			uint8_t *orow = reinterpret_cast<uint8_t *>(out) + static_cast<uint64_t>(i) * out_pitch +
			                static_cast<uint64_t>(wave_col0) * 64u;
			const uint64_t row_left = out_pitch - static_cast<uint64_t>(wave_col0) * 64u;
#pragma unroll
			for (int k = 0; k < 4; k++) {
				const uint32_t piece = k * 64u + lane;
				const uint4 o = s_tile[wave][piece];
				if (static_cast<uint64_t>(piece) * 16u + 16u <= row_left) {
					if (NT_STORE) {
						StoreStream(reinterpret_cast<uint4 *>(orow) + piece, o);
					} else {
						reinterpret_cast<uint4 *>(orow)[piece] = o;
					}
				}
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // reads done before the next row's writes
		if (validity && col < chunks) {
			const uint32_t left = sample_ct - col * 64u;
			if (left < 64u) {
				vbits &= (1ull << left) - 1ull;
			}
			validity[static_cast<uint64_t>(i) * chunks + col] = vbits;
		}
	}
}

__global__ __launch_bounds__(256) void k_unpack_subset(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                       uint32_t v_first, uint32_t v_count,
                                                       const uint32_t *__restrict__ sel, uint32_t n_out,
                                                       int8_t *__restrict__ out, uint64_t out_pitch,
                                                       uint64_t *__restrict__ validity, int32_t fill) {
	// one lane per 16 output samples: gathers their 2-bit calls from the raw row
	const uint32_t val_words16 = ((n_out + 63) / 64) * 4;
	const uint32_t g = blockIdx.x * 256u + threadIdx.x;
	if (g >= val_words16) {
		return;
	}
	for (uint32_t i = blockIdx.y; i < v_count; i += gridDim.y) {
		const uint8_t *row = rows + static_cast<uint64_t>(v_first + i) * pitch;
		uint32_t valid16 = 0;
		const uint32_t k0 = g * 16u;
		if (k0 < n_out) {
			uint32_t o[4] = {0, 0, 0, 0};
#pragma unroll
			for (uint32_t j = 0; j < 16; j++) {
				const uint32_t k = k0 + j;
				uint32_t byte = 0;
				if (k < n_out) {
					const uint32_t s = sel[k];
					const uint32_t code = (row[s >> 2] >> (2 * (s & 3))) & 3u;
					if (code == 3u) {
						byte = static_cast<uint32_t>(fill) & 0xffu;
					} else {
						byte = code;
						valid16 |= 1u << j;
					}
				}
				o[j >> 2] |= byte << (8 * (j & 3));
			}
			if (out) {
				uint4 q = {o[0], o[1], o[2], o[3]};
				reinterpret_cast<uint4 *>(out + static_cast<uint64_t>(i) * out_pitch)[g] = q;
			}
		}
		if (validity) {
			uint16_t *vrow = reinterpret_cast<uint16_t *>(validity + static_cast<uint64_t>(i) * (val_words16 / 4));
			vrow[g] = static_cast<uint16_t>(valid16);
		}
	}
}

// ---------------------------------------------------------------------------
// plink_score
// ---------------------------------------------------------------------------

// Per scored variant: the value a sample of genotype class g contributes
// (src/plink_score.cpp:598-652), from the variant's class counts.
__global__ __launch_bounds__(256) void k_score_tables(const uint32_t *__restrict__ counts,
                                                      const uint8_t *__restrict__ flip, uint32_t n_scored, int mode,
                                                      double *__restrict__ ts, double *__restrict__ td,
                                                      uint32_t *__restrict__ ac) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_scored) {
		return;
	}
	const uint32_t het = counts[4 * i + 1];
	const uint32_t hom_alt = counts[4 * i + 2];
	const uint32_t non_missing = counts[4 * i] + het + hom_alt;
	double s[4] = {0.0, 0.0, 0.0, 0.0};
	double d[4] = {0.0, 0.0, 0.0, 0.0};
	uint32_t inc = 0;
	if (non_missing != 0) {
		const bool fl = flip && flip[i];
		const double sum_alt = static_cast<double>(het) + 2.0 * static_cast<double>(hom_alt);
		const double mean_alt = sum_alt / static_cast<double>(non_missing);
		if (mode == 2) { // center
			const double freq = mean_alt / 2.0;
			const double sd = sqrt(2.0 * freq * (1.0 - freq));
			if (sd != 0.0) {
				const double mean_scored = fl ? (2.0 - mean_alt) : mean_alt;
				for (int g = 0; g < 3; g++) {
					const double scored = fl ? (2.0 - static_cast<double>(g)) : static_cast<double>(g);
					s[g] = (scored - mean_scored) / sd;
				}
				inc = 2u;
			}
		} else {
			for (int g = 0; g < 3; g++) {
				const double scored = fl ? (2.0 - static_cast<double>(g)) : static_cast<double>(g);
				s[g] = scored;
				d[g] = scored;
			}
			inc = 2u;
			if (mode == 0) { // mean imputation
				const double scored = fl ? (2.0 - mean_alt) : mean_alt;
				s[3] = scored;
				d[3] = scored;
				inc = 2u | (2u << 8);
			}
		}
	}
	for (int g = 0; g < 4; g++) {
		ts[4 * static_cast<uint64_t>(i) + g] = s[g];
		td[4 * static_cast<uint64_t>(i) + g] = d[g];
	}
	ac[i] = inc;
}

// First (VALU) form of the accumulate: one lane per sample, a slice of the scored
// variants per workgroup row; tables staged in LDS, weights read wave-uniformly.
template <int NCOLS>
__global__ __launch_bounds__(256) void k_score_accumulate(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                          uint32_t sample_ct, const uint32_t *__restrict__ vlist,
                                                          uint32_t n_scored, uint32_t slice_len,
                                                          const double *__restrict__ weights, uint32_t w_stride,
                                                          uint32_t out_stride, const double *__restrict__ ts,
                                                          const double *__restrict__ td,
                                                          const uint32_t *__restrict__ ac, double *__restrict__ score,
                                                          double *__restrict__ dosage_sum,
                                                          uint32_t *__restrict__ allele_ct) {
	constexpr uint32_t kStage = 64; // variants staged in LDS at a time
	__shared__ double s_ts[kStage][4];
	__shared__ double s_td[kStage][4];
	__shared__ double s_w[kStage][NCOLS];
	__shared__ uint32_t s_ac[kStage];
	__shared__ uint32_t s_v[kStage];
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	const bool live = s < sample_ct;
	const uint32_t shift = 2u * (s & 15u);
	const uint32_t dword = s >> 4;
	const uint32_t i_begin = blockIdx.y * slice_len;
	const uint32_t i_end = min(i_begin + slice_len, n_scored);
	double acc[NCOLS];
#pragma unroll
	for (int c = 0; c < NCOLS; c++) {
		acc[c] = 0.0;
	}
	double dsum = 0.0;
	uint32_t act = 0;
	for (uint32_t base = i_begin; base < i_end; base += kStage) {
		const uint32_t cnt = min(kStage, i_end - base);
		__syncthreads();
		for (uint32_t k = threadIdx.x; k < cnt * 4u; k += 256u) {
			s_ts[k >> 2][k & 3] = ts[4 * static_cast<uint64_t>(base) + k];
			s_td[k >> 2][k & 3] = td ? td[4 * static_cast<uint64_t>(base) + k] : 0.0;
		}
		for (uint32_t k = threadIdx.x; k < cnt * NCOLS; k += 256u) {
			s_w[k / NCOLS][k % NCOLS] = weights[static_cast<uint64_t>(base + k / NCOLS) * w_stride + (k % NCOLS)];
		}
		for (uint32_t k = threadIdx.x; k < cnt; k += 256u) {
			s_ac[k] = ac ? ac[base + k] : 0u;
			s_v[k] = vlist[base + k];
		}
		__syncthreads();
		if (live) {
			for (uint32_t k = 0; k < cnt; k++) {
				const uint32_t w =
				    reinterpret_cast<const uint32_t *>(rows + static_cast<uint64_t>(s_v[k]) * pitch)[dword];
				const uint32_t g = (w >> shift) & 3u;
				const double x = s_ts[k][g];
				dsum += s_td[k][g];
				act += (s_ac[k] >> (g == 3u ? 8 : 0)) & 0xffu;
#pragma unroll
				for (int c = 0; c < NCOLS; c++) {
					acc[c] = fma(s_w[k][c], x, acc[c]);
				}
			}
		}
	}
	if (live) {
#pragma unroll
		for (int c = 0; c < NCOLS; c++) {
			unsafeAtomicAdd(score + static_cast<uint64_t>(s) * out_stride + c, acc[c]);
		}
		if (dosage_sum) {
			unsafeAtomicAdd(dosage_sum + s, dsum);
		}
		if (allele_ct) {
			atomicAdd(allele_ct + s, act);
		}
	}
}

// GEMV form (one weight column, the reference's SQL contract): 2 flop per call, so the
// contraction is not matrix-core work.  Table lookups instead ("four Russians"): the
// workgroup tabulates, per small group of scored variants, the possible (score, dosage)
// sums of one sample's calls in LDS, and a lane then needs one 16-byte LDS lookup + 2 FP64
// adds per group.  Variant slices combine by FP64 atomics.
struct alignas(16) ScorePair {
	double score;
	double dosage;
};

// Groups are PAIRS of variants: a 16-entry table of 16-byte entries is exactly one 256-byte
// LDS bank row, so two lanes on the same bank group hold the same entry (a broadcast) and
// ds_read_b128 lookups are conflict-free for ANY pattern.  (Groups of four -- 256-entry
// tables, half the lookups -- collided ~3.5 ways: SQ_LDS_BANK_CONFLICT 72 % of the LDS cycles,
// 85 ms per 1M x 500k against 44 ms here; replicating those tables per bank group cost more in
// stores than it saved.)  16 variants = 8 pair tables (2 KB) per barrier, double-buffered.
// (byte B of x) & mask in one VALU op (SDWA byte select); mask lives in a register
#define PGH_BYTE_AND(B)                                                                                                \
	__device__ __forceinline__ uint32_t ByteAnd##B(uint32_t x, uint32_t mask) {                                       \
		uint32_t r;                                                                                                    \
		asm("v_and_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_" #B " src1_sel:DWORD"        \
		    : "=v"(r)                                                                                                  \
		    : "v"(x), "v"(mask));                                                                                      \
		return r;                                                                                                      \
	}
PGH_BYTE_AND(0)
PGH_BYTE_AND(1)
PGH_BYTE_AND(2)
PGH_BYTE_AND(3)
#undef PGH_BYTE_AND

constexpr uint32_t kGemvPairs = 8; // pair tables per stage (16 variants)

// One stage: 8 pair tables at a compile-time LDS offset (the lookups then use the
// instruction's immediate offset), 16 words of this lane's 16 samples.
template <int BUF>
__device__ __forceinline__ void GemvPairsStage(const ScorePair (*tabs)[kGemvPairs][16], const uint32_t *w,
                                               double *acc_s, double *acc_d) {
	const uint32_t kF0 = 0xf0u;
#pragma unroll
	for (uint32_t pr = 0; pr < kGemvPairs; pr++) {
		const uint32_t w0 = w[2 * pr], w1 = w[2 * pr + 1];
		// nibble k of `even` / `odd` = 4-bit pattern (row0 | row1 << 2) of sample 2k / 2k+1
		const uint32_t kE = 0x33333333u;
		const uint32_t even = (w0 & kE) | ((w1 & kE) << 2);
		const uint32_t odd = ((w0 >> 2) & kE) | (w1 & ~kE);
		// byte offset of an entry = pattern * 16: the high nibble of a byte already is that,
		// the low nibbles come from the same words shifted up by 4
		const uint32_t even_lo = even << 4, odd_lo = odd << 4;
		const char *tab = reinterpret_cast<const char *>(tabs[BUF][pr]);
#define PGH_LOOKUP(B)                                                                                                  \
	{                                                                                                                  \
		const ScorePair e0 = *reinterpret_cast<const ScorePair *>(tab + ByteAnd##B(even_lo, kF0));                    \
		const ScorePair e1 = *reinterpret_cast<const ScorePair *>(tab + ByteAnd##B(odd_lo, kF0));                     \
		const ScorePair e2 = *reinterpret_cast<const ScorePair *>(tab + ByteAnd##B(even, kF0));                       \
		const ScorePair e3 = *reinterpret_cast<const ScorePair *>(tab + ByteAnd##B(odd, kF0));                        \
		acc_s[4 * B] += e0.score;                                                                                      \
		acc_d[4 * B] += e0.dosage;                                                                                     \
		acc_s[4 * B + 1] += e1.score;                                                                                  \
		acc_d[4 * B + 1] += e1.dosage;                                                                                 \
		acc_s[4 * B + 2] += e2.score;                                                                                  \
		acc_d[4 * B + 2] += e2.dosage;                                                                                 \
		acc_s[4 * B + 3] += e3.score;                                                                                  \
		acc_d[4 * B + 3] += e3.dosage;                                                                                 \
	}
		PGH_LOOKUP(0)
		PGH_LOOKUP(1)
		PGH_LOOKUP(2)
		PGH_LOOKUP(3)
#undef PGH_LOOKUP
	}
}

__global__ __launch_bounds__(256, 4) void k_score_gemv_pairs(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                          uint32_t sample_ct, const uint32_t *__restrict__ vlist,
                                                          uint32_t n_var, uint32_t slice_len,
                                                          const double *__restrict__ weights, uint32_t w_stride,
                                                          const double *__restrict__ ts,
                                                          const double *__restrict__ td, double *__restrict__ score,
                                                          uint32_t out_stride, double *__restrict__ dosage_sum) {
	constexpr uint32_t kPairs = kGemvPairs;
	constexpr uint32_t kStage = kPairs * 2;
	__shared__ ScorePair s_tab[2][kPairs][16];
	const uint32_t d = blockIdx.x * 256u + threadIdx.x; // this lane's 4-byte column: samples 16d .. 16d+15
	const uint32_t n_dwords = (sample_ct + 15) / 16;
	const bool live = d < n_dwords;
	const uint32_t i_begin = blockIdx.y * slice_len;
	const uint32_t i_end = min(i_begin + slice_len, n_var);
	double acc_s[16], acc_d[16];
#pragma unroll
	for (int j = 0; j < 16; j++) {
		acc_s[j] = 0.0;
		acc_d[j] = 0.0;
	}
	// threads 0..127: entry (t & 15) of pair table (t >> 4) for the stage starting at `base`
	auto build = [&](uint32_t base, uint32_t buf) {
		if (threadIdx.x < kPairs * 16) {
			const uint32_t pair = threadIdx.x >> 4, pat = threadIdx.x & 15u;
			double sc = 0.0, ds = 0.0;
#pragma unroll
			for (uint32_t q = 0; q < 2; q++) {
				const uint32_t i = base + pair * 2u + q;
				if (i < i_end) {
					const uint32_t g = (pat >> (2 * q)) & 3u;
					sc += weights[static_cast<uint64_t>(i) * w_stride] * ts[4 * static_cast<uint64_t>(i) + g];
					if (td) {
						ds += td[4 * static_cast<uint64_t>(i) + g];
					}
				}
			}
			s_tab[buf][pair][pat] = ScorePair {sc, ds};
		}
	};
	auto load_words = [&](uint32_t base, uint32_t w[kStage]) {
#pragma unroll
		for (uint32_t k = 0; k < kStage; k++) {
			const uint32_t i = base + k;
			w[k] = (live && i < i_end)
			           ? __builtin_nontemporal_load(
			                 reinterpret_cast<const uint32_t *>(rows + static_cast<uint64_t>(vlist[i]) * pitch) + d)
			           : 0u;
		}
	};
	uint32_t w_a[kStage], w_b[kStage];
	if (i_begin < i_end) {
		load_words(i_begin, w_a);
		build(i_begin, 0);
	}
	__syncthreads();
	// two stages per trip, so each half works on a compile-time table buffer
	for (uint32_t base = i_begin; base < i_end; base += 2 * kStage) {
		const bool more_b = base + kStage < i_end;
		if (more_b) {
			load_words(base + kStage, w_b);
			build(base + kStage, 1);
		}
		GemvPairsStage<0>(s_tab, w_a, acc_s, acc_d);
		__syncthreads();
		if (!more_b) {
			break;
		}
		if (base + 2 * kStage < i_end) {
			load_words(base + 2 * kStage, w_a);
			build(base + 2 * kStage, 0);
		}
		GemvPairsStage<1>(s_tab, w_b, acc_s, acc_d);
		__syncthreads();
	}
	if (live) {
#pragma unroll
		for (int j = 0; j < 16; j++) {
			const uint32_t s0 = d * 16u + j;
			if (s0 < sample_ct) {
				unsafeAtomicAdd(score + static_cast<uint64_t>(s0) * out_stride, acc_s[j]);
				if (dosage_sum) {
					unsafeAtomicAdd(dosage_sum + s0, acc_d[j]);
				}
			}
		}
	}
}

// MFMA form of the accumulate: a true dense contraction
//   out[s][c] += sum_v  T_v[g(v,s)] * W[v][c]
// on v_mfma_f64_16x16x4_f64 tiles: M = 16 samples, K = 4 variants, N = 16 columns.
//   A[i][k] = T_{v_k}[g(v_k, sample i)]   lane l: i = l & 15, k = l >> 4  (table lookup from LDS)
//   B[k][j] = W[v_k][j]                   lane l: k = l >> 4, j = l & 15
//   D[i][j]                               lane l, reg r: i = (l >> 4) + 4 r, j = l & 15
// A wave owns 64 consecutive samples (4 tiles) x NCT column tiles; one 16-byte
// load per lane (16 lanes share a row, 4 rows per wave) feeds all 4 tiles of a
// 4-variant group.  Tables / weights / row ids are staged through LDS 64 variants
// at a time; variant slices (blockIdx.y) are combined with FP64 atomics whose
// lanes cover 128-byte row segments.
typedef double f64x4 __attribute__((ext_vector_type(4)));

// NQ (0..3) extra QUARTER tiles of 4 columns ride on v_mfma_f64_4x4x4_4b_f64 (4 independent 4x4x4
// blocks, 16 cycles against the 16x16x4's 64): with block b = samples 4b..4b+3 its A operand has
// the very layout of the big tile (lane 16k + 4b + i = 16k + sample), so the looked-up a[t] feeds
// both; B is lane 16k + 4b + j -> W[v_k][col j] (the same 4 columns in every block) and D is lane
// 16i + 4b + j (layout measured with tools/mfma_probe.hip).  plink_pca's 2k = 20 columns are one
// tile + one quarter: 80 matrix-pipe cycles per group instead of the 128 of two padded tiles.
template <int NCT, int NQ, bool TRACK_DOSAGE>
__global__ __launch_bounds__(256, (NCT == 1 && NQ == 0) ? 4 : 2) void k_accumulate_mfma(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                         uint32_t sample_ct, const uint32_t *__restrict__ vlist,
                                                         uint32_t n_var, uint32_t slice_len,
                                                         const double *__restrict__ weights, uint32_t w_stride,
                                                         uint32_t n_cols, const double *__restrict__ ts,
                                                         double *__restrict__ out, uint32_t out_stride,
                                                         double *__restrict__ dosage_sum) {
	constexpr uint32_t kStage = 64;
	constexpr uint32_t kCols = 16 * NCT + 4 * NQ;
	constexpr uint32_t kQ = NQ > 0 ? NQ : 1; // array extents (unused when NQ == 0)
	constexpr uint32_t kWPerThread = kStage * kCols / 256; // weight doubles each thread stages
	// double-buffered stage: tables / weights / row offsets of 64 variants
	__shared__ double s_ts[2][kStage][4];
	__shared__ double s_w[2][kStage][kCols];
	__shared__ uint64_t s_off[2][kStage];
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;
	const uint32_t li = lane & 15u; // sample within tile (A), column within tile (B, D)
	const uint32_t lk = lane >> 4;  // variant within the group of 4 (A, B); row group (D)
	const uint32_t sample_base = (blockIdx.x * 4u + wave) * 64u;
	const bool wave_live = sample_base < sample_ct; // wave-uniform
	const uint8_t *col_ptr = rows + (sample_base >> 2);
	const uint32_t shift = 2u * li;
	const uint32_t i_begin = blockIdx.y * slice_len;
	const uint32_t i_end = min(i_begin + slice_len, n_var);

	f64x4 acc[4][NCT];
#pragma unroll
	for (int t = 0; t < 4; t++) {
#pragma unroll
		for (int c = 0; c < NCT; c++) {
			acc[t][c] = f64x4 {0.0, 0.0, 0.0, 0.0};
		}
	}
	double accq[4][kQ];
#pragma unroll
	for (int t = 0; t < 4; t++) {
#pragma unroll
		for (int q = 0; q < static_cast<int>(kQ); q++) {
			accq[t][q] = 0.0;
		}
	}
	double dsum[4] = {0.0, 0.0, 0.0, 0.0};

	// staging registers: the next stage is fetched from global memory while the
	// current one is being multiplied, then dropped into the other LDS buffer
	double r_ts = 0.0;
	double r_w[kWPerThread];
	uint64_t r_off = 0;
	auto fetch = [&](uint32_t base) {
		const uint32_t cnt = min(kStage, i_end - base);
		const uint32_t k = threadIdx.x; // kStage * 4 == 256
		r_ts = (k >> 2) < cnt ? ts[4 * static_cast<uint64_t>(base) + k] : 0.0;
#pragma unroll
		for (uint32_t j = 0; j < kWPerThread; j++) {
			const uint32_t e = threadIdx.x + 256u * j;
			const uint32_t v = e / kCols, c = e % kCols;
			r_w[j] = (v < cnt && c < n_cols) ? weights[static_cast<uint64_t>(base + v) * w_stride + c] : 0.0;
		}
		if (threadIdx.x < kStage) {
			r_off = static_cast<uint64_t>(vlist[base + (threadIdx.x < cnt ? threadIdx.x : 0)]) * pitch;
		}
	};
	auto commit = [&](uint32_t buf) {
		s_ts[buf][threadIdx.x >> 2][threadIdx.x & 3] = r_ts;
#pragma unroll
		for (uint32_t j = 0; j < kWPerThread; j++) {
			const uint32_t e = threadIdx.x + 256u * j;
			s_w[buf][e / kCols][e % kCols] = r_w[j];
		}
		if (threadIdx.x < kStage) {
			s_off[buf][threadIdx.x] = r_off;
		}
	};

	if (i_begin < i_end) {
		fetch(i_begin);
		commit(0);
	}
	__syncthreads();
	uint32_t buf = 0;
	for (uint32_t base = i_begin; base < i_end; base += kStage, buf ^= 1u) {
		const uint32_t cnt = min(kStage, i_end - base);
		const bool more = base + kStage < i_end;
		if (more) {
			fetch(base + kStage);
		}
		if (wave_live) {
			const uint32_t groups = (cnt + 3) / 4;
			// two-deep software pipeline: while group g is on the matrix pipe, group g+1's
			// operands are being looked up in LDS and the row loads of g+2..g+4 are in flight
			auto operands = [&](uint32_t k, const uint4 &w, double a[4], double b[NCT + kQ]) {
#pragma unroll
				for (int c = 0; c < NCT; c++) {
					b[c] = s_w[buf][k][16 * c + li];
				}
#pragma unroll
				for (int q = 0; q < NQ; q++) {
					b[NCT + q] = s_w[buf][k][16 * NCT + 4 * q + (lane & 3u)];
				}
				const uint32_t wt[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
				for (int t = 0; t < 4; t++) {
					a[t] = s_ts[buf][k][(wt[t] >> shift) & 3u];
				}
			};
			auto row_load = [&](uint32_t g) {
				return *reinterpret_cast<const uint4 *>(col_ptr + s_off[buf][g * 4u + lk]);
			};
			auto multiply = [&](const double a[4], const double b[NCT + kQ]) {
#pragma unroll
				for (int t = 0; t < 4; t++) {
					if (TRACK_DOSAGE) {
						dsum[t] += a[t];
					}
#pragma unroll
					for (int c = 0; c < NCT; c++) {
						acc[t][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t], b[c], acc[t][c], 0, 0, 0);
					}
#pragma unroll
					for (int q = 0; q < NQ; q++) {
						accq[t][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t], b[NCT + q], accq[t][q], 0, 0, 0);
					}
				}
			};
			if (groups == kStage / 4) {
				// full stage, fully unrolled so the ring of 4 row loads and the operand
				// double buffer are plain register renaming (no moves, no early waits)
				constexpr uint32_t kRing = 3; // row loads in flight (3 keeps NCT = 1 at 128 VGPRs = 4 waves/SIMD)
				uint4 w[kRing];
#pragma unroll
				for (uint32_t j = 0; j < kRing; j++) {
					w[j] = row_load(j);
				}
				double a[2][4], b[2][NCT + kQ];
				operands(lk, w[0], a[0], b[0]);
#pragma unroll
				for (uint32_t g4 = 0; g4 < kStage / 4; g4++) {
					if (g4 + kRing < kStage / 4) {
						w[g4 % kRing] = row_load(g4 + kRing); // slot of group g4, already turned into operands
					}
					if (g4 + 1 < kStage / 4) {
						operands((g4 + 1u) * 4u + lk, w[(g4 + 1) % kRing], a[(g4 + 1) % 2], b[(g4 + 1) % 2]);
					}
					multiply(a[g4 % 2], b[g4 % 2]);
				}
			} else {
				// ragged last stage of a slice
				for (uint32_t g4 = 0; g4 < groups; g4++) {
					double a[4], b[NCT + kQ];
					operands(g4 * 4u + lk, row_load(g4), a, b);
					multiply(a, b);
				}
			}
		}
		if (more) {
			// buffer buf^1 was last read during the previous stage; every wave has
			// passed the barrier that ended it
			commit(buf ^ 1u);
		}
		__syncthreads();
	}
	if (!wave_live) {
		return;
	}
#pragma unroll
	for (int t = 0; t < 4; t++) {
#pragma unroll
		for (int c = 0; c < NCT; c++) {
			const uint32_t col = 16u * c + li;
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const uint32_t s = sample_base + 16u * t + lk + 4u * r;
				if (s < sample_ct && col < n_cols) {
					unsafeAtomicAdd(out + static_cast<uint64_t>(s) * out_stride + col, acc[t][c][r]);
				}
			}
		}
#pragma unroll
		for (int q = 0; q < NQ; q++) {
			// D of the 4-block form: lane = 16 i + 4 b + j -> sample 4b + i, column j
			const uint32_t s = sample_base + 16u * t + 4u * ((lane >> 2) & 3u) + (lane >> 4);
			const uint32_t col = 16u * NCT + 4u * q + (lane & 3u);
			if (s < sample_ct && col < n_cols) {
				unsafeAtomicAdd(out + static_cast<uint64_t>(s) * out_stride + col, accq[t][q]);
			}
		}
		if (TRACK_DOSAGE) {
			// this lane saw the variants == lk (mod 4) of sample 16t + li
			double d = dsum[t];
			d += __shfl_xor(d, 16, 64);
			d += __shfl_xor(d, 32, 64);
			const uint32_t s = sample_base + 16u * t + li;
			if (lk == 0 && s < sample_ct) {
				unsafeAtomicAdd(dosage_sum + s, d);
			}
		}
	}
}

// allele_ct[s] = total - 2 * (scored, non-skipped variants at which s is missing)
// total = sum of the per-variant increments (ac[i] & 0xff); miss == NULL: mean imputation,
// every sample gets the full total.
__global__ __launch_bounds__(256) void k_allele_ct(const uint32_t *__restrict__ ac, uint32_t n_scored,
                                                   const uint32_t *__restrict__ miss, uint32_t sample_ct,
                                                   uint32_t *__restrict__ allele_ct) {
	__shared__ uint32_t part[4];
	uint32_t t = 0;
	for (uint32_t i = threadIdx.x; i < n_scored; i += 256u) {
		t += ac[i] & 0xffu;
	}
	t = WaveSum(t);
	if ((threadIdx.x & 63u) == 0) {
		part[threadIdx.x >> 6] = t;
	}
	__syncthreads();
	const uint32_t total = part[0] + part[1] + part[2] + part[3];
	for (uint32_t s = blockIdx.x * 256u + threadIdx.x; s < sample_ct; s += gridDim.x * 256u) {
		allele_ct[s] = total - (miss ? 2u * miss[s] : 0u);
	}
}

// ---------------------------------------------------------------------------
// plink_pca
// ---------------------------------------------------------------------------

// Normalised-genotype table of each effective variant (NormalizeGenotypes,
// src/plink_common.cpp:1535-1543): t[g] = (g - center) * inv_stdev, missing -> 0.
__global__ __launch_bounds__(256) void k_norm_tables(const double *__restrict__ center,
                                                     const double *__restrict__ inv_stdev, uint32_t n,
                                                     double *__restrict__ ts) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n) {
		return;
	}
	const double c = center[i], is = inv_stdev[i];
	ts[4 * static_cast<uint64_t>(i) + 0] = (0.0 - c) * is;
	ts[4 * static_cast<uint64_t>(i) + 1] = (1.0 - c) * is;
	ts[4 * static_cast<uint64_t>(i) + 2] = (2.0 - c) * is;
	ts[4 * static_cast<uint64_t>(i) + 3] = 0.0;
}

// out[i][c] = sum over samples of ts[i][g(i,s)] * G[s][c]   (Step A, src/plink_pca.cpp:632-645)
// One workgroup per tile of VT variants; a lane walks samples s = tid, tid+256, ...,
// loads its G row once per sample and feeds all VT variants of the tile.
template <int NCOLS, int VT>
__global__ __launch_bounds__(256) void k_variant_reduce(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                        uint32_t sample_ct, const uint32_t *__restrict__ vlist,
                                                        uint32_t n_var, const double *__restrict__ ts,
                                                        const double *__restrict__ G, uint32_t g_stride,
                                                        double *__restrict__ out, uint32_t out_stride) {
	__shared__ double red[4][VT * NCOLS];
	__shared__ double s_t[VT][4];
	__shared__ uint32_t s_v[VT];
	const uint32_t i0 = blockIdx.x * VT;
	const uint32_t nv = min(static_cast<uint32_t>(VT), n_var - i0);
	if (threadIdx.x < VT * 4) {
		const uint32_t k = threadIdx.x >> 2;
		s_t[k][threadIdx.x & 3] = k < nv ? ts[4 * static_cast<uint64_t>(i0 + k) + (threadIdx.x & 3)] : 0.0;
	}
	if (threadIdx.x < VT) {
		s_v[threadIdx.x] = threadIdx.x < nv ? vlist[i0 + threadIdx.x] : vlist[i0];
	}
	__syncthreads();
	double acc[VT][NCOLS];
#pragma unroll
	for (int k = 0; k < VT; k++) {
#pragma unroll
		for (int c = 0; c < NCOLS; c++) {
			acc[k][c] = 0.0;
		}
	}
	for (uint32_t s = threadIdx.x; s < sample_ct; s += 256u) {
		double g[NCOLS];
#pragma unroll
		for (int c = 0; c < NCOLS; c++) {
			g[c] = G[static_cast<uint64_t>(s) * g_stride + c];
		}
		const uint32_t shift = 2u * (s & 15u);
#pragma unroll
		for (int k = 0; k < VT; k++) {
			const uint32_t w = reinterpret_cast<const uint32_t *>(rows + static_cast<uint64_t>(s_v[k]) * pitch)[s >> 4];
			const double x = s_t[k][(w >> shift) & 3u];
#pragma unroll
			for (int c = 0; c < NCOLS; c++) {
				acc[k][c] = fma(x, g[c], acc[k][c]);
			}
		}
	}
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
#pragma unroll
	for (int k = 0; k < VT; k++) {
#pragma unroll
		for (int c = 0; c < NCOLS; c++) {
			double v = acc[k][c];
#pragma unroll
			for (int off = 32; off > 0; off >>= 1) {
				v += __shfl_xor(v, off, 64);
			}
			if (lane == 0) {
				red[wave][k * NCOLS + c] = v;
			}
		}
	}
	__syncthreads();
	if (threadIdx.x < VT * NCOLS) {
		const uint32_t k = threadIdx.x / NCOLS, c = threadIdx.x % NCOLS;
		if (k < nv) {
			// fixed order: deterministic
			out[static_cast<uint64_t>(i0 + k) * out_stride + c] =
			    ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
		}
	}
}

// MFMA form of Step A:  out[v][c] = sum_s T_v[g(v,s)] * G[s][c]
// on v_mfma_f64_16x16x4_f64 tiles: M = 16 variants, K = 4 samples, N = 16 columns.
//   A[i][k] = T_{v_i}[g(v_i, s_k)]   lane l: i = l & 15 (variant), k = l >> 4 (sample 4q + k)
//   B[k][j] = G[s_k][j]              lane l: k = l >> 4, j = l & 15        (from the LDS chunk)
// A workgroup owns 128 variants (4 waves x 2 tiles) and streams every sample in chunks of
// 128: the G chunk (128 x 32 doubles) is staged through LDS once per workgroup, prefetched
// into registers while the previous chunk is multiplied.  Each lane reads its variant's
// 32 bytes of the chunk (the 4 lane groups of a variant share the load) and peels sample
// 4q + k at step q with a per-lane constant shift.  No atomics: a variant's whole sum
// lives in one wave.
// NQ quarter tiles: as in k_accumulate_mfma, 4 more columns on v_mfma_f64_4x4x4_4b_f64 with the
// big tile's A operand (block b = variants 4b..4b+3 of the tile).
template <int NCT, int NQ>
__global__ __launch_bounds__(256) void k_variant_reduce_mfma(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                             uint32_t sample_ct, const uint32_t *__restrict__ vlist,
                                                             uint32_t n_var, const double *__restrict__ ts,
                                                             const double *__restrict__ G, uint32_t g_stride,
                                                             uint32_t n_cols, double *__restrict__ out,
                                                             uint32_t out_stride, uint32_t chunks_per_split) {
	constexpr uint32_t kChunk = 128;           // samples per LDS chunk
	constexpr uint32_t kCols = 16 * NCT + 4 * NQ;
	constexpr uint32_t kQ = NQ > 0 ? NQ : 1;
	constexpr uint32_t kGPerThread = kChunk * kCols / 256;
	constexpr uint32_t kVT = 2;                // variant tiles per wave
	__shared__ double s_g[kChunk][kCols];
	__shared__ double s_t[4 * kVT * 16][4];    // tables of the workgroup's 128 variants
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t li = lane & 15u, lk = lane >> 4;
	const uint32_t v_wg = blockIdx.x * (4u * kVT * 16u);

	// tables + row pointers of this lane's two variants
	for (uint32_t e = threadIdx.x; e < 4u * kVT * 16u * 4u; e += 256u) {
		const uint32_t v = v_wg + (e >> 2);
		s_t[e >> 2][e & 3] = v < n_var ? ts[4 * static_cast<uint64_t>(v) + (e & 3)] : 0.0;
	}
	const uint8_t *row_ptr[kVT];
	uint32_t t_base[kVT];
#pragma unroll
	for (uint32_t t = 0; t < kVT; t++) {
		const uint32_t v_local = (wave * kVT + t) * 16u + li;
		const uint32_t v = v_wg + v_local;
		row_ptr[t] = rows + static_cast<uint64_t>(vlist[v < n_var ? v : 0]) * pitch;
		t_base[t] = v_local;
	}
	f64x4 acc[kVT][NCT];
#pragma unroll
	for (uint32_t t = 0; t < kVT; t++) {
#pragma unroll
		for (int c = 0; c < NCT; c++) {
			acc[t][c] = f64x4 {0.0, 0.0, 0.0, 0.0};
		}
	}
	double accq[kVT][kQ];
#pragma unroll
	for (uint32_t t = 0; t < kVT; t++) {
#pragma unroll
		for (uint32_t qq = 0; qq < kQ; qq++) {
			accq[t][qq] = 0.0;
		}
	}
	double r_g[kGPerThread];
	auto fetch = [&](uint32_t s0) {
#pragma unroll
		for (uint32_t j = 0; j < kGPerThread; j++) {
			const uint32_t e = threadIdx.x + 256u * j;
			const uint32_t s = s0 + e / kCols, c = e % kCols;
			r_g[j] = (s < sample_ct && c < n_cols) ? G[static_cast<uint64_t>(s) * g_stride + c] : 0.0;
		}
	};
	auto commit = [&]() {
#pragma unroll
		for (uint32_t j = 0; j < kGPerThread; j++) {
			const uint32_t e = threadIdx.x + 256u * j;
			s_g[e / kCols][e % kCols] = r_g[j];
		}
	};
	// blockIdx.y picks a run of sample chunks; with more than one run the partial sums of a
	// variant meet in `out` (zeroed by the launcher) through FP64 atomics
	const uint32_t all_chunks = (sample_ct + kChunk - 1) / kChunk;
	const uint32_t ch_begin = blockIdx.y * chunks_per_split;
	const uint32_t n_chunks = min(all_chunks, ch_begin + chunks_per_split);
	const bool split = gridDim.y > 1;
	const uint32_t lane_shift = 2u * lk; // sample 4q + k sits at bit 2*(4*(q&3) + k) of word q >> 2
	if (ch_begin >= n_chunks) {
		return;
	}
	fetch(ch_begin * kChunk);
	for (uint32_t ch = ch_begin; ch < n_chunks; ch++) {
		__syncthreads(); // everyone is done reading the previous chunk
		commit();
		__syncthreads();
		if (ch + 1 < n_chunks) {
			fetch((ch + 1) * kChunk);
		}
		// this lane's 128 calls (32 bytes) of each of its variants
		uint32_t w[kVT][8];
#pragma unroll
		for (uint32_t t = 0; t < kVT; t++) {
			const uint4 *p = reinterpret_cast<const uint4 *>(row_ptr[t] + static_cast<uint64_t>(ch) * (kChunk / 4));
			const uint4 a = p[0], b = p[1];
			w[t][0] = a.x;
			w[t][1] = a.y;
			w[t][2] = a.z;
			w[t][3] = a.w;
			w[t][4] = b.x;
			w[t][5] = b.y;
			w[t][6] = b.z;
			w[t][7] = b.w;
		}
#pragma unroll
		for (uint32_t q = 0; q < kChunk / 4; q++) {
			double b[NCT], bq[kQ];
#pragma unroll
			for (int c = 0; c < NCT; c++) {
				b[c] = s_g[4u * q + lk][16 * c + li];
			}
#pragma unroll
			for (int qq = 0; qq < NQ; qq++) {
				bq[qq] = s_g[4u * q + lk][16 * NCT + 4 * qq + (lane & 3u)];
			}
#pragma unroll
			for (uint32_t t = 0; t < kVT; t++) {
				const uint32_t g = (w[t][q >> 2] >> (8u * (q & 3u) + lane_shift)) & 3u;
				const double a = s_t[t_base[t]][g];
#pragma unroll
				for (int c = 0; c < NCT; c++) {
					acc[t][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[c], acc[t][c], 0, 0, 0);
				}
#pragma unroll
				for (int qq = 0; qq < NQ; qq++) {
					accq[t][qq] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bq[qq], accq[t][qq], 0, 0, 0);
				}
			}
		}
	}
#pragma unroll
	for (uint32_t t = 0; t < kVT; t++) {
#pragma unroll
		for (int c = 0; c < NCT; c++) {
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const uint32_t v = v_wg + (wave * kVT + t) * 16u + lk + 4u * r;
				const uint32_t col = 16u * c + li;
				if (v < n_var && col < n_cols) {
					double *dst = out + static_cast<uint64_t>(v) * out_stride + col;
					if (split) {
						unsafeAtomicAdd(dst, acc[t][c][r]);
					} else {
						*dst = acc[t][c][r];
					}
				}
			}
		}
#pragma unroll
		for (int qq = 0; qq < NQ; qq++) {
			// D of the 4-block form: lane = 16 i + 4 b + j -> variant 4b + i of the tile, column j
			const uint32_t v = v_wg + (wave * kVT + t) * 16u + 4u * ((lane >> 2) & 3u) + (lane >> 4);
			const uint32_t col = 16u * NCT + 4u * qq + (lane & 3u);
			if (v < n_var && col < n_cols) {
				double *dst = out + static_cast<uint64_t>(v) * out_stride + col;
				if (split) {
					unsafeAtomicAdd(dst, accq[t][qq]);
				} else {
					*dst = accq[t][qq];
				}
			}
		}
	}
}

// rows of excluded samples -> 0 (keeps a sample subset out of the power iteration)
__global__ __launch_bounds__(256) void k_mask_rows(double *__restrict__ m, uint32_t n_rows, uint32_t stride,
                                                   uint32_t ncols, const uint8_t *__restrict__ mask2) {
	const uint64_t idx = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
	if (idx >= static_cast<uint64_t>(n_rows) * ncols) {
		return;
	}
	const uint32_t s = static_cast<uint32_t>(idx / ncols), c = static_cast<uint32_t>(idx % ncols);
	if (!((mask2[s >> 2] >> (2 * (s & 3))) & 1u)) {
		m[static_cast<uint64_t>(s) * stride + c] = 0.0;
	}
}

__global__ __launch_bounds__(256) void k_scale(double *__restrict__ m, uint64_t n, double f) {
	const uint64_t idx = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
	if (idx < n) {
		m[idx] *= f;
	}
}

// ---------------------------------------------------------------------------
// tall-skinny dense helpers for plink_pca's orthonormalisation (FP64)
// ---------------------------------------------------------------------------

// C[i][j] += sum_r A[r][i] * B[r][j]   (A: m x na, B: m x nb, row-major; C zeroed by the caller)
// One wave per 16x16 tile of C and per chunk of rows, on v_mfma_f64_16x16x4_f64:
// K runs over rows, 4 at a time; both operands are 128-byte row segments.
__global__ __launch_bounds__(256) void k_tall_gram(const double *__restrict__ A, uint32_t lda, uint32_t na,
                                                   const double *__restrict__ B, uint32_t ldb, uint32_t nb,
                                                   uint64_t m, uint32_t rows_per_wave, double *__restrict__ C,
                                                   uint32_t ldc) {
	const uint32_t tiles_b = (nb + 15) / 16;
	const uint32_t ta = blockIdx.x / tiles_b, tb = blockIdx.x % tiles_b;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t li = lane & 15u, lk = lane >> 4;
	const uint64_t r_begin = (static_cast<uint64_t>(blockIdx.y) * 4u + wave) * rows_per_wave;
	const uint64_t r_end = min(r_begin + rows_per_wave, m);
	const uint32_t ca = ta * 16u + li, cb = tb * 16u + li;
	const bool a_ok = ca < na, b_ok = cb < nb;
	f64x4 acc = {0.0, 0.0, 0.0, 0.0};
	for (uint64_t r = r_begin; r < r_end; r += 4) {
		const uint64_t row = r + lk;
		const bool in = row < r_end;
		const double a = (in && a_ok) ? A[row * lda + ca] : 0.0;
		const double b = (in && b_ok) ? B[row * ldb + cb] : 0.0;
		acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
	}
	if (r_begin < r_end) {
#pragma unroll
		for (int r = 0; r < 4; r++) {
			const uint32_t i = ta * 16u + lk + 4u * r, j = tb * 16u + li;
			if (i < na && j < nb) {
				unsafeAtomicAdd(C + static_cast<uint64_t>(i) * ldc + j, acc[r]);
			}
		}
	}
}

// out[r][j] = beta * B[r][j] + alpha * sum_i A[r][i] * C[i][j]   (j < nb; C: na x nb, small)
// 4 rows per workgroup staged in LDS; a lane produces 4 adjacent columns of one row.
__global__ __launch_bounds__(256) void k_tall_times_small(const double *__restrict__ A, uint32_t lda, uint32_t na,
                                                          const double *__restrict__ C, uint32_t ldc, uint32_t nb,
                                                          double alpha, double beta, const double *__restrict__ B,
                                                          uint32_t ldb, double *__restrict__ out, uint32_t ldo,
                                                          uint64_t m) {
	extern __shared__ double s_rows[]; // [4][na]
	const uint32_t rr = threadIdx.x >> 6, jg = threadIdx.x & 63u;
	const uint64_t r0 = static_cast<uint64_t>(blockIdx.x) * 4u;
	for (uint32_t e = threadIdx.x; e < 4u * na; e += 256u) {
		const uint64_t row = r0 + e / na;
		s_rows[e] = row < m ? A[row * lda + e % na] : 0.0;
	}
	__syncthreads();
	const uint64_t row = r0 + rr;
	if (row >= m) {
		return;
	}
	for (uint32_t j0 = jg * 4u; j0 < nb; j0 += 256u) {
		double acc[4] = {0.0, 0.0, 0.0, 0.0};
		const uint32_t w = min(4u, nb - j0);
		for (uint32_t i = 0; i < na; i++) {
			const double a = s_rows[rr * na + i];
			const double *c = C + static_cast<uint64_t>(i) * ldc + j0;
#pragma unroll
			for (uint32_t q = 0; q < 4; q++) {
				if (q < w) {
					acc[q] = fma(a, c[q], acc[q]);
				}
			}
		}
		for (uint32_t q = 0; q < w; q++) {
			const double prev = beta != 0.0 ? beta * B[row * ldb + j0 + q] : 0.0;
			out[row * ldo + j0 + q] = prev + alpha * acc[q];
		}
	}
}

// dst[r][j] = src[r][j] for j < n (strided 2-D copy)
__global__ __launch_bounds__(256) void k_copy_cols(const double *__restrict__ src, uint32_t lds_, double *__restrict__ dst,
                                                   uint32_t ldd, uint32_t n, uint64_t m) {
	const uint64_t idx = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
	if (idx >= m * n) {
		return;
	}
	const uint64_t r = idx / n;
	const uint32_t j = static_cast<uint32_t>(idx % n);
	dst[r * ldd + j] = src[r * lds_ + j];
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
	// enough row slices for >= ~2048 workgroups (256 CUs x 4 resident x 2), each a multiple of 6 rows
	uint32_t want_slices = (2048 + col_blocks - 1) / col_blocks;
	if (want_slices > 1024) {
		want_slices = 1024;
	}
	uint32_t slice_len = (v_count + want_slices - 1) / want_slices;
	slice_len = ((slice_len + 5) / 6) * 6;
	if (slice_len < 96) {
		slice_len = 96;
	}
	if (slice_len > 65280u) {
		slice_len = 65280u; // the fused kernel keeps 16-bit column counters; 65280 = 12 * 5440
	}
	*slice_len_out = slice_len;
	*slices_out = v_count ? (v_count + slice_len - 1) / slice_len : 0;
}

size_t MissingPerSampleScratchBytes(uint32_t record_bytes, uint32_t v_count) {
	uint32_t slice_len, slices;
	MissingPerSamplePlan(record_bytes, v_count, &slice_len, &slices);
	const uint64_t stride = static_cast<uint64_t>((record_bytes + 15) / 16) * 64;
	return static_cast<size_t>(slices) * stride * sizeof(uint32_t);
}

hipError_t LaunchMissingPerSample(const RowView &view, uint32_t v_first, const uint32_t *vlist, uint32_t v_count,
                                  const uint32_t *row_flags, uint32_t *scratch, uint32_t *out, hipStream_t stream) {
	return LaunchClassPerSample(view, 3, v_first, vlist, v_count, row_flags, scratch, out, stream);
}

hipError_t LaunchClassPerSample(const RowView &view, int genotype_class, uint32_t v_first, const uint32_t *vlist,
                                uint32_t v_count, const uint32_t *row_flags, uint32_t *scratch, uint32_t *out,
                                hipStream_t stream) {
	if (v_count == 0) {
		return hipMemsetAsync(out, 0, sizeof(uint32_t) * view.sample_ct, stream);
	}
	const uint32_t chunks = static_cast<uint32_t>((static_cast<uint64_t>(view.record_bytes) + 15) / 16);
	const uint32_t col_blocks = (chunks + 255) / 256;
	uint32_t slice_len, slices;
	MissingPerSamplePlan(view.record_bytes, v_count, &slice_len, &slices);
	const uint32_t stride = chunks * 64u;
#define PGH_COLS(CLASS)                                                                                                \
	hipLaunchKernelGGL(k_missing_cols<CLASS>, dim3(col_blocks, slices), dim3(256), 0, stream, view.rows, view.pitch,   \
	                   chunks, v_first, vlist, v_count, slice_len, row_flags, scratch, stride)
	if (genotype_class == 1) {
		PGH_COLS(1);
	} else if (genotype_class == 2) {
		PGH_COLS(2);
	} else {
		PGH_COLS(3);
	}
#undef PGH_COLS
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		return e;
	}
	hipLaunchKernelGGL(k_sum_slabs, dim3((view.sample_ct + 255) / 256), dim3(256), 0, stream, scratch, stride, slices,
	                   view.sample_ct, out);
	return hipGetLastError();
}

hipError_t LaunchFusedTally(const RowView &view, uint32_t v_first, uint32_t v_count, uint32_t *scratch,
                            uint32_t *counts, uint32_t *missing_per_sample, hipStream_t stream) {
	if (v_count == 0) {
		return hipMemsetAsync(missing_per_sample, 0, sizeof(uint32_t) * view.sample_ct, stream);
	}
	const uint32_t chunks = static_cast<uint32_t>((static_cast<uint64_t>(view.record_bytes) + 15) / 16);
	const uint32_t col_blocks = (chunks + 255) / 256;
	uint32_t slice_len, slices;
	MissingPerSamplePlan(view.record_bytes, v_count, &slice_len, &slices);
	slice_len = (slice_len + kFusedRows - 1) / kFusedRows * kFusedRows; // <= 65280, so never more slices
	slices = (v_count + slice_len - 1) / slice_len;
	const uint32_t stride = chunks * 64u;
	hipError_t e = hipMemsetAsync(counts, 0, 16ull * v_count, stream);
	if (e != hipSuccess) {
		return e;
	}
	hipLaunchKernelGGL(k_fused_tally, dim3(col_blocks, slices), dim3(256), 0, stream, view.rows, view.pitch, chunks,
	                   v_first, v_count, slice_len, counts, scratch, stride);
	hipLaunchKernelGGL(k_finish_tallies, dim3((v_count + 255) / 256), dim3(256), 0, stream,
	                   reinterpret_cast<uint4 *>(counts), v_count, view.sample_ct);
	hipLaunchKernelGGL(k_sum_slabs, dim3((view.sample_ct + 255) / 256), dim3(256), 0, stream, scratch, stride, slices,
	                   view.sample_ct, missing_per_sample);
	return hipGetLastError();
}

hipError_t LaunchUnpack(const RowView &view, uint32_t v_first, uint32_t v_count, int8_t *out, uint64_t out_pitch,
                        uint64_t *validity, int8_t fill, hipStream_t stream) {
	if (v_count == 0) {
		return hipSuccess;
	}
	const uint32_t val_words16 = ((view.sample_ct + 63) / 64) * 4;
	const uint32_t f = static_cast<uint8_t>(fill);
	const uint32_t fill4 = f * 0x01010101u;
	if (view.sample_ct >= 4096) {
		const uint32_t chunks = (view.sample_ct + 63) / 64;
		dim3 grid_w((chunks + 255) / 256, v_count < 65535u ? v_count : 65535u);
		static const int variant = [] {
			const char *e = getenv("PGH_UNPACK_VARIANT"); // tuning knob: 0 = non-temporal stores, 1 = plain stores
			return e ? atoi(e) : 0;
		}();
		if (variant == 1) {
			hipLaunchKernelGGL(k_unpack_wide<false>, grid_w, dim3(256), 0, stream, view.rows, view.pitch,
			                   view.sample_ct, v_first, v_count, out, out_pitch, validity, fill4);
		} else if (variant == 2) {
			dim3 grid((val_words16 + 255) / 256, v_count < 65535u ? v_count : 65535u);
			hipLaunchKernelGGL(k_unpack, grid, dim3(256), 0, stream, view.rows, view.pitch, view.sample_ct, v_first,
			                   v_count, out, out_pitch, validity, fill4);
		} else {
			hipLaunchKernelGGL(k_unpack_wide<true>, grid_w, dim3(256), 0, stream, view.rows, view.pitch,
			                   view.sample_ct, v_first, v_count, out, out_pitch, validity, fill4);
		}
		return hipGetLastError();
	}
	dim3 grid((val_words16 + 255) / 256, v_count < 65535u ? v_count : 65535u);
	hipLaunchKernelGGL(k_unpack, grid, dim3(256), 0, stream, view.rows, view.pitch, view.sample_ct, v_first, v_count,
	                   out, out_pitch, validity, fill4);
	return hipGetLastError();
}

hipError_t LaunchUnpackSubset(const RowView &view, uint32_t v_first, uint32_t v_count, const uint32_t *sel,
                              uint32_t n_out, int8_t *out, uint64_t out_pitch, uint64_t *validity, int8_t fill,
                              hipStream_t stream) {
	if (v_count == 0 || n_out == 0) {
		return hipSuccess;
	}
	const uint32_t val_words16 = ((n_out + 63) / 64) * 4;
	dim3 grid((val_words16 + 255) / 256, v_count < 65535u ? v_count : 65535u);
	hipLaunchKernelGGL(k_unpack_subset, grid, dim3(256), 0, stream, view.rows, view.pitch, v_first, v_count, sel,
	                   n_out, out, out_pitch, validity, static_cast<int32_t>(fill));
	return hipGetLastError();
}

hipError_t LaunchScoreTables(const uint32_t *counts, const uint8_t *flip, uint32_t n_scored, int mode, double *ts,
                             double *td, uint32_t *ac, hipStream_t stream) {
	if (n_scored == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_score_tables, dim3((n_scored + 255) / 256), dim3(256), 0, stream, counts, flip, n_scored,
	                   mode, ts, td, ac);
	return hipGetLastError();
}

template <int NCOLS>
static hipError_t LaunchAccumulateN(const RowView &view, const uint32_t *vlist, uint32_t n_scored,
                                    const double *weights, uint32_t w_stride, const double *ts, const double *td,
                                    const uint32_t *ac, double *score, uint32_t out_stride, double *dosage_sum,
                                    uint32_t *allele_ct, hipStream_t stream) {
	const uint32_t sample_blocks = (view.sample_ct + 255) / 256;
	uint32_t want_slices = (2048 + sample_blocks - 1) / sample_blocks;
	uint32_t slice_len = (n_scored + want_slices - 1) / want_slices;
	slice_len = ((slice_len + 63) / 64) * 64;
	uint32_t slices = (n_scored + slice_len - 1) / slice_len;
	if (slices > 65535u) {
		slices = 65535u;
		slice_len = (n_scored + slices - 1) / slices;
	}
	hipLaunchKernelGGL((k_score_accumulate<NCOLS>), dim3(sample_blocks, slices), dim3(256), 0, stream, view.rows,
	                   view.pitch, view.sample_ct, vlist, n_scored, slice_len, weights, w_stride, out_stride, ts, td,
	                   ac, score, dosage_sum, allele_ct);
	return hipGetLastError();
}

template <int NCT, int NQ>
static hipError_t LaunchAccumulateMfma(const RowView &view, const uint32_t *vlist, uint32_t n_var,
                                       const double *weights, uint32_t w_stride, uint32_t n_cols, const double *ts,
                                       bool track_dosage, double *out, uint32_t out_stride, double *dosage_sum,
                                       hipStream_t stream) {
	const uint32_t sample_blocks = (view.sample_ct + 255) / 256;
	// >= ~16k workgroups (each CU holds ~6; many short ones keep the tail of the launch
	// small); slices are multiples of the 64-variant stage and at least 8 stages long
	uint32_t want_slices = (16384 + sample_blocks - 1) / sample_blocks;
	uint32_t slice_len = (n_var + want_slices - 1) / want_slices;
	slice_len = ((slice_len + 63) / 64) * 64;
	if (slice_len < 512) {
		slice_len = 512;
	}
	uint32_t slices = (n_var + slice_len - 1) / slice_len;
	if (slices > 65535u) {
		slices = 65535u;
		slice_len = ((n_var + slices - 1) / slices + 63) / 64 * 64;
		slices = (n_var + slice_len - 1) / slice_len;
	}
	if (track_dosage) {
		hipLaunchKernelGGL((k_accumulate_mfma<NCT, NQ, true>), dim3(sample_blocks, slices), dim3(256), 0, stream,
		                   view.rows, view.pitch, view.sample_ct, vlist, n_var, slice_len, weights, w_stride, n_cols,
		                   ts, out, out_stride, dosage_sum);
	} else {
		hipLaunchKernelGGL((k_accumulate_mfma<NCT, NQ, false>), dim3(sample_blocks, slices), dim3(256), 0, stream,
		                   view.rows, view.pitch, view.sample_ct, vlist, n_var, slice_len, weights, w_stride, n_cols,
		                   ts, out, out_stride, dosage_sum);
	}
	return hipGetLastError();
}

hipError_t LaunchTableAccumulate(const RowView &view, const uint32_t *vlist, uint32_t n_var, const double *weights,
                                 uint32_t w_stride, uint32_t n_cols, const double *ts, const double *td,
                                 const uint32_t *ac, bool track_dosage, double *out, uint32_t out_stride,
                                 double *dosage_sum, uint32_t *allele_ct, hipStream_t stream) {
	if (n_var == 0) {
		return hipSuccess;
	}
	if (!track_dosage) {
		td = nullptr;
	}
	if (n_cols >= 3) {
		// dense contraction: FP64 MFMA tiles, 32 columns (2 tiles) per pass, 16 for the tail.
		// The dosage-sum table equals the score table whenever it is tracked (non-centred
		// plink_score); allele counts are integer bookkeeping done by the caller (k_allele_ct).
		(void)ac;
		(void)allele_ct;
		uint32_t c0 = 0;
		hipError_t e = hipSuccess;
		while (c0 < n_cols && e == hipSuccess) {
			const uint32_t left = n_cols - c0;
			const bool track = c0 == 0 && track_dosage && dosage_sum != nullptr;
#define PGH_ACC(NCT, NQ, WIDTH)                                                                                        \
	e = LaunchAccumulateMfma<NCT, NQ>(view, vlist, n_var, weights + c0, w_stride, left < (WIDTH) ? left : (WIDTH), ts, \
	                                  track, out + c0, out_stride, dosage_sum, stream);                                \
	c0 += (WIDTH)
			// full 32-column passes, then the tail as one tile + the quarter tiles it needs
			if (left >= 32 || left > 28) {
				PGH_ACC(2, 0, 32);
			} else if (left > 24) {
				PGH_ACC(1, 3, 28);
			} else if (left > 20) {
				PGH_ACC(1, 2, 24);
			} else if (left > 16) {
				PGH_ACC(1, 1, 20);
			} else {
				PGH_ACC(1, 0, 16);
			}
#undef PGH_ACC
		}
		return e;
	}
	if (n_cols == 1) {
		// GEMV: table-lookup kernel, HBM/LDS-bound
		const uint32_t n_dwords = (view.sample_ct + 15) / 16;
		const uint32_t col_blocks = (n_dwords + 255) / 256;
		uint32_t want_slices = (4096 + col_blocks - 1) / col_blocks;
		uint32_t slice_len = (n_var + want_slices - 1) / want_slices;
		slice_len = ((slice_len + 15) / 16) * 16;
		if (slice_len < 256) {
			slice_len = 256;
		}
		uint32_t slices = (n_var + slice_len - 1) / slice_len;
		if (slices > 65535u) {
			slices = 65535u;
			slice_len = ((n_var + slices - 1) / slices + 15) / 16 * 16;
			slices = (n_var + slice_len - 1) / slice_len;
		}
		hipLaunchKernelGGL(k_score_gemv_pairs, dim3(col_blocks, slices), dim3(256), 0, stream, view.rows, view.pitch,
		                   view.sample_ct, vlist, n_var, slice_len, weights, w_stride, ts, td, out, out_stride,
		                   dosage_sum);
		return hipGetLastError();
	}
	// 2 columns: plain FP64 FMAs
	uint32_t c0 = 0;
	hipError_t e = hipSuccess;
	while (c0 < n_cols && e == hipSuccess) {
		const uint32_t left = n_cols - c0;
		const double *td_b = c0 == 0 ? td : nullptr;
		double *ds_b = c0 == 0 ? dosage_sum : nullptr;
		if (left >= 2) {
			e = LaunchAccumulateN<2>(view, vlist, n_var, weights + c0, w_stride, ts, td_b, nullptr, out + c0, out_stride,
			                         ds_b, nullptr, stream);
			c0 += 2;
		} else {
			e = LaunchAccumulateN<1>(view, vlist, n_var, weights + c0, w_stride, ts, td_b, nullptr, out + c0, out_stride,
			                         ds_b, nullptr, stream);
			c0 += 1;
		}
	}
	return e;
}

hipError_t LaunchAlleleCt(const uint32_t *ac, uint32_t n_scored, const uint32_t *miss, uint32_t sample_ct,
                          uint32_t *allele_ct, hipStream_t stream) {
	uint32_t blocks = (sample_ct + 255) / 256;
	if (blocks > 1024) {
		blocks = 1024;
	}
	hipLaunchKernelGGL(k_allele_ct, dim3(blocks ? blocks : 1), dim3(256), 0, stream, ac, n_scored, miss, sample_ct,
	                   allele_ct);
	return hipGetLastError();
}

hipError_t LaunchScoreAccumulate(const RowView &view, const uint32_t *vlist, uint32_t n_scored, const double *weights,
                                 uint32_t n_cols, const double *ts, const double *td, const uint32_t *ac,
                                 bool track_dosage, double *score, double *dosage_sum, uint32_t *allele_ct,
                                 hipStream_t stream) {
	return LaunchTableAccumulate(view, vlist, n_scored, weights, n_cols, n_cols, ts, td, ac, track_dosage, score,
	                             n_cols, dosage_sum, allele_ct, stream);
}

hipError_t LaunchNormTables(const double *center, const double *inv_stdev, uint32_t n, double *ts,
                            hipStream_t stream) {
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_norm_tables, dim3((n + 255) / 256), dim3(256), 0, stream, center, inv_stdev, n, ts);
	return hipGetLastError();
}

template <int NCOLS>
static hipError_t LaunchVariantReduceN(const RowView &view, const uint32_t *vlist, uint32_t n_var, const double *ts,
                                       const double *G, uint32_t g_stride, double *out, uint32_t out_stride,
                                       hipStream_t stream) {
	constexpr int VT = 8;
	hipLaunchKernelGGL((k_variant_reduce<NCOLS, VT>), dim3((n_var + VT - 1) / VT), dim3(256), 0, stream, view.rows,
	                   view.pitch, view.sample_ct, vlist, n_var, ts, G, g_stride, out, out_stride);
	return hipGetLastError();
}

hipError_t LaunchVariantReduce(const RowView &view, const uint32_t *vlist, uint32_t n_var, const double *ts,
                               const double *G, uint32_t g_stride, uint32_t n_cols, double *out, uint32_t out_stride,
                               hipStream_t stream) {
	if (n_var == 0) {
		return hipSuccess;
	}
	// The MFMA form reads whole 32-byte (128-call) pieces of a row: the row pitch must cover
	// ceil(N/128) of them, which holds for 128-byte-aligned pitches (rows >= 512 bytes).
	const bool mfma_ok = n_cols >= 3 && view.pitch % 32 == 0 &&
	                     static_cast<uint64_t>((view.sample_ct + 127) / 128) * 32 <= view.pitch;
	uint32_t c0 = 0;
	hipError_t e = hipSuccess;
	while (c0 < n_cols && e == hipSuccess) {
		const uint32_t left = n_cols - c0;
		if (mfma_ok) {
			// One workgroup per 128 variants is too coarse for a few hundred thousand variants
			// (781 workgroups on 256 CUs leave a quarter of the matrix pipes idle at the end):
			// split the sample axis until there are >= ~12 workgroups per CU.
			const uint32_t blocks = (n_var + 127) / 128;
			const uint32_t all_chunks = (view.sample_ct + 127) / 128;
			uint32_t splits = blocks >= 3072 ? 1 : (3072 + blocks - 1) / blocks;
			splits = std::min(splits, std::max(1u, all_chunks / 64)); // keep >= 64 chunks (8192 samples) per run
			const uint32_t chunks_per_split = (all_chunks + splits - 1) / splits;
			splits = (all_chunks + chunks_per_split - 1) / chunks_per_split;
			const uint32_t width = left > 28 ? 32 : (left > 24 ? 28 : (left > 20 ? 24 : (left > 16 ? 20 : 16)));
			if (splits > 1) {
				e = hipMemset2DAsync(out + c0, sizeof(double) * out_stride, 0, sizeof(double) * std::min(left, width),
				                     n_var, stream);
				if (e != hipSuccess) {
					break;
				}
			}
#define PGH_VR(NCT, NQ, WIDTH)                                                                                         \
	hipLaunchKernelGGL((k_variant_reduce_mfma<NCT, NQ>), dim3(blocks, splits), dim3(256), 0, stream, view.rows,        \
	                   view.pitch, view.sample_ct, vlist, n_var, ts, G + c0, g_stride,                                 \
	                   left < (WIDTH) ? left : (WIDTH), out + c0, out_stride, chunks_per_split);                       \
	c0 += (WIDTH)
			if (left > 28) {
				PGH_VR(2, 0, 32);
			} else if (left > 24) {
				PGH_VR(1, 3, 28);
			} else if (left > 20) {
				PGH_VR(1, 2, 24);
			} else if (left > 16) {
				PGH_VR(1, 1, 20);
			} else {
				PGH_VR(1, 0, 16);
			}
#undef PGH_VR
			e = hipGetLastError();
		} else if (left >= 8) {
			e = LaunchVariantReduceN<8>(view, vlist, n_var, ts, G + c0, g_stride, out + c0, out_stride, stream);
			c0 += 8;
		} else if (left >= 4) {
			e = LaunchVariantReduceN<4>(view, vlist, n_var, ts, G + c0, g_stride, out + c0, out_stride, stream);
			c0 += 4;
		} else if (left >= 2) {
			e = LaunchVariantReduceN<2>(view, vlist, n_var, ts, G + c0, g_stride, out + c0, out_stride, stream);
			c0 += 2;
		} else {
			e = LaunchVariantReduceN<1>(view, vlist, n_var, ts, G + c0, g_stride, out + c0, out_stride, stream);
			c0 += 1;
		}
	}
	return e;
}

hipError_t LaunchMaskRows(double *m, uint32_t n_rows, uint32_t stride, uint32_t n_cols, const uint8_t *mask2,
                          hipStream_t stream) {
	const uint64_t total = static_cast<uint64_t>(n_rows) * n_cols;
	if (total == 0 || !mask2) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_mask_rows, dim3(static_cast<uint32_t>((total + 255) / 256)), dim3(256), 0, stream, m, n_rows,
	                   stride, n_cols, mask2);
	return hipGetLastError();
}

hipError_t LaunchScale(double *m, uint64_t n, double f, hipStream_t stream) {
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_scale, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, stream, m, n, f);
	return hipGetLastError();
}

hipError_t LaunchTallGram(const double *A, uint32_t lda, uint32_t na, const double *B, uint32_t ldb, uint32_t nb,
                          uint64_t m, double *C, uint32_t ldc, hipStream_t stream) {
	if (m == 0 || na == 0 || nb == 0) {
		return hipSuccess;
	}
	const uint32_t tiles = ((na + 15) / 16) * ((nb + 15) / 16);
	// ~1024 waves per tile pair at most, >= 256 rows per wave
	uint32_t rows_per_wave = 1024;
	uint64_t chunks = (m + 4ull * rows_per_wave - 1) / (4ull * rows_per_wave);
	while (chunks > 65535) {
		rows_per_wave *= 2;
		chunks = (m + 4ull * rows_per_wave - 1) / (4ull * rows_per_wave);
	}
	hipLaunchKernelGGL(k_tall_gram, dim3(tiles, static_cast<uint32_t>(chunks)), dim3(256), 0, stream, A, lda, na, B, ldb,
	                   nb, m, rows_per_wave, C, ldc);
	return hipGetLastError();
}

hipError_t LaunchTallTimesSmall(const double *A, uint32_t lda, uint32_t na, const double *C, uint32_t ldc, uint32_t nb,
                                double alpha, double beta, const double *B, uint32_t ldb, double *out, uint32_t ldo,
                                uint64_t m, hipStream_t stream) {
	if (m == 0 || nb == 0) {
		return hipSuccess;
	}
	const uint64_t blocks = (m + 3) / 4;
	hipLaunchKernelGGL(k_tall_times_small, dim3(static_cast<uint32_t>(blocks)), dim3(256), 4 * na * sizeof(double), stream,
	                   A, lda, na, C, ldc, nb, alpha, beta, B, ldb, out, ldo, m);
	return hipGetLastError();
}

hipError_t LaunchCopyCols(const double *src, uint32_t ld_src, double *dst, uint32_t ld_dst, uint32_t n, uint64_t m,
                          hipStream_t stream) {
	const uint64_t total = m * n;
	if (total == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_copy_cols, dim3(static_cast<uint32_t>((total + 255) / 256)), dim3(256), 0, stream, src, ld_src,
	                   dst, ld_dst, n, m);
	return hipGetLastError();
}

hipError_t LaunchHweBatch(const uint32_t *counts, uint32_t n, uint32_t midp, double *ln_p, hipStream_t stream) {
	if (n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_hwe_batch, dim3((n + 63) / 64), dim3(64), 0, stream, counts, n, midp, ln_p);
	return hipGetLastError();
}

} // namespace pgh

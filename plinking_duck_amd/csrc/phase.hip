// phase.hip -- phase tracks out of the staged record bytes (gfx950).
//
// Track layout (vrtype bit 0x10, after the main track): ceil((1 + hets) / 8) bytes -- bit 0 says whether
// the following `hets` bits are phase-PRESENT flags (1) or, every het being phased, the phase bits
// themselves (0); with flags, ceil(flagged / 8) bytes of phase bits follow, one per flagged het.
// The reference reads it through PgrGetP (src/pgen_reader.cpp:700-715) into phasepresent / phaseinfo bit
// arrays over the samples.  Here one workgroup per record deposits the per-het bits at the het samples'
// positions: a wave owns a 64-sample word, a ballot of (call == het) is the word's het mask, the number
// of hets before the word comes from a block prefix sum kept in LDS, and lane l's bit index is
// rank[word] + popcount(mask below l).  With flags the same again one level down for the phase bits.
#include "phase.hpp"

#include "decode_device.hpp"
#include "device_utils.hpp"

namespace pgh {

namespace {

constexpr uint32_t kMaxLdsBytes = 160u * 1024u - 1024u;

// exclusive prefix sums of v[0..n) in place; returns the total.  All 256 threads call it.
__device__ uint32_t BlockExclusiveScan(uint32_t *v, uint32_t n, uint32_t *s_wave) {
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t carry = 0;
	for (uint32_t base = 0; base < n; base += 256u) {
		const uint32_t i = base + threadIdx.x;
		const uint32_t c = i < n ? v[i] : 0u;
		const uint32_t incl = InclusiveScan(c, lane);
		__syncthreads();
		if (lane == 63u) {
			s_wave[wave] = incl;
		}
		__syncthreads();
		uint32_t before = carry;
		for (uint32_t k = 0; k < wave; k++) {
			before += s_wave[k];
		}
		if (i < n) {
			v[i] = before + incl - c;
		}
		carry += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
	}
	__syncthreads();
	return carry;
}

__global__ __launch_bounds__(256) void k_phase_extract(PhaseIngest b) {
	extern __shared__ uint32_t s_tables[]; // [words] hets before each word | [words] flagged hets before each word
	__shared__ uint32_t s_wave[4];
	const uint32_t r = blockIdx.x;
	const int32_t pr = b.ph_row[r];
	if (pr < 0) {
		return;
	}
	const Src src {b.bytes, b.bytes_len};
	const uint32_t N = b.sample_ct, words = b.words;
	const uint64_t rec_end = b.rec_begin[r + 1];
	const uint64_t cur = b.aux_at[r];
	auto fail = [&]() {
		if (threadIdx.x == 0) {
			atomicCAS(b.error, 0, static_cast<int>(b.variant0 + r) + 1);
		}
	};
	if ((b.vrtype[r] & 0x08u) || cur >= rec_end || cur < b.rec_begin[r]) {
		fail();
		return;
	}
	const uint32_t *row32 = reinterpret_cast<const uint32_t *>(b.rows + static_cast<uint64_t>(b.row0 + r) * b.pitch);
	uint64_t *present = b.present + static_cast<uint64_t>(pr) * words;
	uint64_t *info = b.info + static_cast<uint64_t>(pr) * words;
	uint32_t *het_before = s_tables, *flag_before = s_tables + words;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint64_t below = (1ull << lane) - 1ull;
	auto het_mask = [&](uint32_t w) {
		const uint32_t s = 64u * w + lane;
		const uint32_t code = s < N ? (row32[s >> 4] >> (2u * (s & 15u))) & 3u : 0u;
		return __ballot(code == 1u);
	};
	auto track_bit = [&](uint64_t first_byte, uint32_t bit) { return (src.Byte(first_byte + (bit >> 3)) >> (bit & 7u)) & 1u; };
	for (uint32_t w = wave; w < words; w += 4u) {
		const uint64_t hm = het_mask(w);
		if (lane == 0) {
			het_before[w] = static_cast<uint32_t>(__popcll(hm));
		}
	}
	__syncthreads();
	const uint32_t hets = BlockExclusiveScan(het_before, words, s_wave);
	const uint64_t head = (1ull + hets + 7ull) / 8ull;
	if (cur + head > rec_end) {
		fail();
		return;
	}
	const bool flags = src.Byte(cur) & 1u;
	for (uint32_t w = wave; w < words; w += 4u) {
		const uint64_t hm = het_mask(w);
		const bool is_het = (hm >> lane) & 1ull;
		const uint32_t bit = is_het ? track_bit(cur, 1u + het_before[w] + static_cast<uint32_t>(__popcll(hm & below))) : 0u;
		const uint64_t set = __ballot(bit != 0u);
		if (lane == 0) {
			if (flags) {
				present[w] = set;
				flag_before[w] = static_cast<uint32_t>(__popcll(set));
			} else {
				present[w] = hm;
				info[w] = set;
			}
		}
	}
	if (!flags) {
		return;
	}
	__threadfence_block();
	__syncthreads();
	const uint32_t flagged = BlockExclusiveScan(flag_before, words, s_wave);
	const uint64_t phase_bits = cur + head;
	if (phase_bits + (static_cast<uint64_t>(flagged) + 7ull) / 8ull > rec_end) {
		fail();
		return;
	}
	for (uint32_t w = wave; w < words; w += 4u) {
		const uint64_t pm = present[w];
		const bool has = (pm >> lane) & 1ull;
		const uint32_t bit = has ? track_bit(phase_bits, flag_before[w] + static_cast<uint32_t>(__popcll(pm & below))) : 0u;
		const uint64_t set = __ballot(bit != 0u);
		if (lane == 0) {
			info[w] = set;
		}
	}
}

} // namespace

uint32_t PhaseIngestMaxSamples() {
	return kMaxLdsBytes / 8u * 64u;
}

hipError_t LaunchPhaseIngest(const PhaseIngest &batch, hipStream_t stream) {
	if (batch.n == 0) {
		return hipSuccess;
	}
	const uint32_t lds = batch.words * 8u;
	if (lds > kMaxLdsBytes) {
		return hipErrorInvalidValue;
	}
	if (lds > 64u * 1024u) {
		hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_phase_extract),
		                                   hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
		if (e != hipSuccess) {
			return e;
		}
	}
	hipLaunchKernelGGL(k_phase_extract, dim3(batch.n), dim3(256), lds, stream, batch);
	return hipGetLastError();
}

} // namespace pgh

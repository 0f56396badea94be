// decode_device.hpp -- device-side helpers for reading .pgen record bytes (decode.hip, dosage.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pgh {

namespace {

// the staged file bytes of a batch of records
struct Src {
	const uint8_t *bytes;
	uint64_t len; // readable bytes (the staging buffer is padded past this by 16 zero bytes)

	__device__ uint32_t Byte(uint64_t at) const {
		return at < len ? bytes[at] : 0u;
	}
	// little-endian 32-bit word at any byte offset
	__device__ uint32_t Word(uint64_t at) const {
		if (at + 4 > len) {
			return Byte(at) | (Byte(at + 1) << 8) | (Byte(at + 2) << 16) | (Byte(at + 3) << 24);
		}
		const uint64_t base = at & ~3ull;
		const uint32_t sh = static_cast<uint32_t>(at & 3) * 8;
		const uint32_t lo = *reinterpret_cast<const uint32_t *>(bytes + base);
		if (sh == 0) {
			return lo;
		}
		const uint32_t hi = *reinterpret_cast<const uint32_t *>(bytes + base + 4); // inside the pad at worst
		return (lo >> sh) | (hi << (32 - sh));
	}
	__device__ uint32_t Le(uint64_t at, uint32_t n) const {
		uint32_t v = 0;
		for (uint32_t i = 0; i < n; i++) {
			v |= Byte(at + i) << (8 * i);
		}
		return v;
	}
};

__device__ inline uint32_t InclusiveScan(uint32_t v, uint32_t lane) {
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t up = __shfl_up(v, d);
		if (lane >= static_cast<uint32_t>(d)) {
			v += up;
		}
	}
	return v;
}

// One wave (64 lanes, all active) walks the sample ids of a difflist that starts at `cur` with its varint
// length.  Layout: varint len; ceil(len/64) group-first ids (id_bytes each); groups-1 group-length bytes;
// with_values: ceil(len/4) bytes of 2-bit values; then per group its other entries as varint gaps.
// apply(entry, id, values_at) runs on the lane that owns the entry (values_at: byte offset of the value
// section).  The gap stream is decoded 64 bytes per trip: a lane owns a byte, a ballot of the terminator
// bytes ranks the varints, and ONE inclusive scan over the bytes' shifted 7-bit payloads yields every
// running sample id at its terminator lane.  Returns false on a malformed list; on success `cur` is the
// first byte after it and `len` its entry count.
template <class Apply>
__device__ bool WalkDifflistIds(const Src &src, uint64_t &cur, uint64_t rec_end, uint32_t sample_ct, uint32_t id_bytes,
                                bool with_values, uint32_t &len, Apply apply) {
	const uint32_t lane = threadIdx.x & 63u;
	len = 0;
	{
		uint32_t shift = 0;
		while (true) {
			const uint32_t byte = src.Byte(cur++);
			len |= (byte & 0x7fu) << shift;
			if (!(byte & 0x80u)) {
				break;
			}
			shift += 7;
			if (shift > 28 || cur > rec_end) {
				return false;
			}
		}
	}
	if (len > sample_ct || cur > rec_end) {
		return false;
	}
	if (len == 0) {
		return true;
	}
	const uint32_t groups = (len + 63) / 64;
	const uint64_t first_ids = cur;
	const uint64_t values = first_ids + static_cast<uint64_t>(groups) * id_bytes + (groups - 1);
	const uint64_t gaps = values + (with_values ? (len + 3) / 4 : 0u);
	if (gaps > rec_end) {
		return false;
	}
	bool bad = false;
	// each group's first entry carries its sample id outright
	for (uint32_t g = lane; g < groups; g += 64) {
		const uint32_t id = src.Le(first_ids + static_cast<uint64_t>(g) * id_bytes, id_bytes);
		if (id >= sample_ct) {
			bad = true;
		} else {
			apply(g * 64, id, values);
		}
	}
	// the other entries are varint gaps: varint k belongs to group k / 63, entry 64*(k/63) + k%63 + 1
	const uint32_t n_gaps = len - groups;
	uint32_t k_base = 0, carry_id = 0;
	uint64_t pos = gaps;
	const uint64_t lt_mask = (1ull << lane) - 1ull;
	while (k_base < n_gaps) {
		const bool in_rec = pos + lane < rec_end;
		const uint32_t byte = in_rec ? src.Byte(pos + lane) : 0x80u;
		const bool term = in_rec && !(byte & 0x80u);
		const uint64_t terms = __ballot(term);
		const uint64_t before = terms & lt_mask;
		const uint32_t rank = static_cast<uint32_t>(__popcll(before));
		const uint32_t start = before ? 64u - static_cast<uint32_t>(__clzll(before)) : 0u;
		const uint32_t sh = 7u * (lane - start);
		const bool take = term && rank < n_gaps - k_base;
		const uint64_t taken = __ballot(take);
		if (taken == 0 || __ballot(in_rec && sh > 28u && rank < n_gaps - k_base) != 0) {
			bad = true; // no complete gap in 64 bytes, or a gap longer than five bytes
			break;
		}
		const uint32_t last = 63u - static_cast<uint32_t>(__clzll(taken));
		// payload of bytes past the last taken terminator must not leak into the scan of the next trip;
		// inside this trip they sit above every taken lane, so they never reach one
		const uint32_t sum = InclusiveScan(sh <= 28u ? (byte & 0x7fu) << sh : 0u, lane);
		const uint32_t g0 = k_base / 63;
		const uint32_t in_g0 = 63u * (g0 + 1) - k_base; // gaps of this trip that still belong to g0
		// running sum at the end of g0's part of this trip (0 when g0 does not end here)
		const uint64_t edge = __ballot(take && rank + 1 == in_g0);
		const uint32_t sum_g0 = edge ? __shfl(sum, static_cast<int>(__ffsll(static_cast<long long>(edge)) - 1)) : 0u;
		uint32_t id = 0;
		if (take) {
			const uint32_t k = k_base + rank;
			const uint32_t g = k / 63;
			if (g == g0) {
				const uint32_t from =
				    (k_base % 63 == 0) ? src.Le(first_ids + static_cast<uint64_t>(g0) * id_bytes, id_bytes) : carry_id;
				id = from + sum;
			} else {
				id = src.Le(first_ids + static_cast<uint64_t>(g) * id_bytes, id_bytes) + (sum - sum_g0);
			}
			if (id >= sample_ct) {
				bad = true;
			} else {
				apply(g * 64 + k % 63 + 1, id, values);
			}
		}
		carry_id = __shfl(id, static_cast<int>(last));
		k_base += static_cast<uint32_t>(__popcll(taken));
		pos += last + 1;
	}
	cur = n_gaps ? pos : gaps;
	return __ballot(bad) == 0;
}

} // namespace

} // namespace pgh

// decode.hip -- .pgen variant records -> pitched 2-bit rows, on the device (gfx950).
//
// pgenlib decodes a record inside every PgrGet* call (the reference's per-variant
// loops, src/plink_freq.cpp:482 etc.); here a file's records are expanded ONCE, at
// pgh_open, straight into the HBM-resident matrix, so the compressed bytes are what
// crosses PCIe.  The host twin is pgen_file.cpp:Normalizer (kept for the aux tracks
// and for an LD run that starts before the opened range).
//
// One 256-lane workgroup per record:
//   1. all lanes write the record's base row, one 32-bit word (16 samples) per lane per
//      trip: the literal 2-bit bytes (type 0), the 1-bit array spread to 2-bit slots
//      (type 1), a constant (types 4/6/7), or the LD base row, inverted for type 3;
//   2. wave 0 walks the difflist.  Its varint gap stream is decoded 64 bytes per trip:
//      a lane owns a byte, a ballot of the terminator bytes ranks the varints, and ONE
//      inclusive scan over the bytes' shifted 7-bit payloads yields every running
//      sample id at its terminator lane (ids are prefix sums of the gaps, gaps are sums
//      of payloads).  Each entry then flips its 2-bit slot with one atomicXor of
//      (base value ^ new value): entries never share a slot, so no ordering is needed.
// LD records (types 2/3) read another row of the matrix, so they run in a second
// launch after every other record of the batch is in place.
#include "decode.hpp"

namespace pgh {

namespace {

constexpr int kDecodeThreads = 256;

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

// 16 presence bits -> one bit in every even position of a 32-bit word
__device__ inline uint32_t Spread16(uint32_t x) {
	x = (x | (x << 8)) & 0x00ff00ffu;
	x = (x | (x << 4)) & 0x0f0f0f0fu;
	x = (x | (x << 2)) & 0x33333333u;
	x = (x | (x << 1)) & 0x55555555u;
	return x;
}

// hom-ref <-> hom-alt in every slot (00 <-> 10; het and missing keep their code)
__device__ inline uint32_t InvertWord(uint32_t x) {
	return x ^ ((~x & 0x55555555u) << 1);
}
__device__ inline uint32_t InvertCode(uint32_t g) {
	return g ^ ((~g & 1u) << 1);
}

__device__ inline uint32_t InclusiveScan(uint32_t v, uint32_t lane) {
	for (int d = 1; d < 64; d <<= 1) {
		uint32_t up = __shfl_up(v, d);
		if (lane >= static_cast<uint32_t>(d)) {
			v += up;
		}
	}
	return v;
}

template <bool LD_PASS>
__global__ __launch_bounds__(kDecodeThreads) void k_decode_records(DecodeBatch b) {
	const uint32_t r = blockIdx.x;
	const uint32_t kind = b.vrtype[r] & 7u;
	const bool is_ld = kind == 2 || kind == 3;
	if (is_ld != LD_PASS) {
		return;
	}
	const Src src {b.bytes, b.bytes_len};
	const uint32_t N = b.sample_ct;
	const uint32_t rb = (N + 3) / 4;
	const uint64_t rec = b.rec_begin[r];
	const uint64_t rec_end = b.rec_begin[r + 1];
	uint32_t *row = reinterpret_cast<uint32_t *>(b.rows + static_cast<uint64_t>(b.row0 + r) * b.pitch);
	const uint32_t *base_row = nullptr;
	if (LD_PASS) {
		const uint32_t br = b.ld_row[r];
		if (br == 0xffffffffu) {
			if (threadIdx.x == 0) {
				atomicCAS(b.error, 0, static_cast<int>(b.variant0 + r) + 1);
			}
			return;
		}
		base_row = reinterpret_cast<const uint32_t *>(b.rows + static_cast<uint64_t>(br) * b.pitch);
	}

	// ---- 1. base row -----------------------------------------------------------------
	uint64_t cur = rec; // start of the difflist, once the base is known
	uint32_t low = 0, delta = 0;
	uint64_t bits_at = 0;
	uint32_t fill = 0;
	switch (kind) {
	case 0:
		cur = rec + rb;
		break;
	case 1: {
		const uint32_t code = src.Byte(rec);
		low = code >> 2;
		delta = code & 3u;
		bits_at = rec + 1;
		cur = bits_at + (N + 7) / 8;
		break;
	}
	case 4:
		fill = 0x00000000u;
		break;
	case 6:
		fill = 0xaaaaaaaau;
		break;
	case 7:
		fill = 0xffffffffu;
		break;
	default:
		break;
	}
	if (kind == 5 || cur > rec_end) {
		if (threadIdx.x == 0) {
			atomicCAS(b.error, 0, static_cast<int>(b.variant0 + r) + 1);
		}
		return;
	}
	const uint32_t words = static_cast<uint32_t>(b.pitch / 4);
	for (uint32_t w = threadIdx.x; w < words; w += kDecodeThreads) {
		const uint32_t first = w * 16;
		uint32_t out = 0;
		if (first < N) {
			switch (kind) {
			case 0:
				out = src.Word(rec + 4ull * w);
				break;
			case 1: {
				const uint32_t bits16 = src.Byte(bits_at + 2ull * w) | (src.Byte(bits_at + 2ull * w + 1) << 8);
				out = low * 0x55555555u + Spread16(bits16) * delta;
				break;
			}
			case 2:
				out = base_row[w];
				break;
			case 3:
				out = InvertWord(base_row[w]);
				break;
			default:
				out = fill;
				break;
			}
			const uint32_t live = N - first;
			if (live < 16) {
				out &= (1u << (2 * live)) - 1u;
			}
		}
		row[w] = out;
	}
	if (kind == 0) {
		return;
	}
	__threadfence();
	__syncthreads();
	if (threadIdx.x >= 64) {
		return;
	}

	// ---- 2. difflist (wave 0) ----------------------------------------------------------
	const uint32_t lane = threadIdx.x;
	bool bad = false;
	uint32_t len = 0;
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
				bad = true;
				break;
			}
		}
	}
	if (bad || len > N) {
		if (lane == 0) {
			atomicCAS(b.error, 0, static_cast<int>(b.variant0 + r) + 1);
		}
		return;
	}
	if (len == 0) {
		return;
	}
	const uint32_t groups = (len + 63) / 64;
	const uint64_t first_ids = cur;
	const uint64_t values = first_ids + static_cast<uint64_t>(groups) * b.id_bytes + (groups - 1);
	const uint64_t gaps = values + (len + 3) / 4;
	if (gaps > rec_end) {
		if (lane == 0) {
			atomicCAS(b.error, 0, static_cast<int>(b.variant0 + r) + 1);
		}
		return;
	}
	// value the base row holds at sample `id`, in the code space of the finished row
	auto base_code = [&](uint32_t id) -> uint32_t {
		switch (kind) {
		case 1:
			return low + delta * ((src.Byte(bits_at + (id >> 3)) >> (id & 7u)) & 1u);
		case 2:
			return (base_row[id >> 4] >> (2 * (id & 15u))) & 3u;
		case 3:
			return InvertCode((base_row[id >> 4] >> (2 * (id & 15u))) & 3u);
		default:
			return fill & 3u;
		}
	};
	auto apply = [&](uint32_t entry, uint32_t id) {
		if (id >= N) {
			bad = true;
			return;
		}
		uint32_t val = (src.Byte(values + (entry >> 2)) >> (2 * (entry & 3u))) & 3u;
		if (kind == 3) {
			val = InvertCode(val); // the reference patches, then inverts the whole row
		}
		const uint32_t flip = base_code(id) ^ val;
		if (flip) {
			atomicXor(&row[id >> 4], flip << (2 * (id & 15u)));
		}
	};
	// each group's first entry carries its sample id outright
	for (uint32_t g = lane; g < groups; g += 64) {
		apply(g * 64, src.Le(first_ids + static_cast<uint64_t>(g) * b.id_bytes, b.id_bytes));
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
				const uint32_t from = (k_base % 63 == 0)
				                          ? src.Le(first_ids + static_cast<uint64_t>(g0) * b.id_bytes, b.id_bytes)
				                          : carry_id;
				id = from + sum;
			} else {
				id = src.Le(first_ids + static_cast<uint64_t>(g) * b.id_bytes, b.id_bytes) + (sum - sum_g0);
			}
			apply(g * 64 + k % 63 + 1, id);
		}
		carry_id = __shfl(id, static_cast<int>(last));
		k_base += static_cast<uint32_t>(__popcll(taken));
		pos += last + 1;
	}
	if (__ballot(bad) != 0 && lane == 0) {
		atomicCAS(b.error, 0, static_cast<int>(b.variant0 + r) + 1);
	}
}

} // namespace

hipError_t LaunchDecodeRecords(const DecodeBatch &batch, bool any_ld, hipStream_t stream) {
	if (batch.n == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_decode_records<false>, dim3(batch.n), dim3(kDecodeThreads), 0, stream, batch);
	hipError_t e = hipGetLastError();
	if (e == hipSuccess && any_ld) {
		hipLaunchKernelGGL(k_decode_records<true>, dim3(batch.n), dim3(kDecodeThreads), 0, stream, batch);
		e = hipGetLastError();
	}
	return e;
}

} // namespace pgh

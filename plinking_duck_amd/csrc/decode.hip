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

#include "decode_device.hpp"

namespace pgh {

namespace {

constexpr int kDecodeThreads = 256;

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
		if (b.aux_at && threadIdx.x == 0) {
			b.aux_at[r] = cur; // aux tracks (phase, dosage) follow the literal bytes
		}
		return;
	}
	__threadfence();
	__syncthreads();
	if (threadIdx.x >= 64) {
		return;
	}

	// ---- 2. difflist (wave 0) ----------------------------------------------------------
	const uint32_t lane = threadIdx.x;
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
	// Each entry flips its 2-bit slot with one atomicXor of (base value ^ new value): entries never share
	// a slot, so no ordering is needed.
	auto apply = [&](uint32_t entry, uint32_t id, uint64_t values) {
		uint32_t val = (src.Byte(values + (entry >> 2)) >> (2 * (entry & 3u))) & 3u;
		if (kind == 3) {
			val = InvertCode(val); // the reference patches, then inverts the whole row
		}
		const uint32_t flip = base_code(id) ^ val;
		if (flip) {
			atomicXor(&row[id >> 4], flip << (2 * (id & 15u)));
		}
	};
	uint32_t len = 0;
	const bool ok = WalkDifflistIds(src, cur, rec_end, N, b.id_bytes, true, len, apply);
	if (lane == 0) {
		if (!ok) {
			atomicCAS(b.error, 0, static_cast<int>(b.variant0 + r) + 1);
		} else if (b.aux_at) {
			b.aux_at[r] = cur;
		}
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

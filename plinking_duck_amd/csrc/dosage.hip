// dosage.hip -- kernels over the resident dosage tracks (gfx950).
//
// The reference reads dosages per variant through pgenlib: PgrGetD fills a dosage_present
// bit array + packed uint16 values, Dosage16ToDoublesMinus9 widens them to N doubles
// (src/plink_score.cpp:587-600, src/pgen_reader.cpp:694-705), PgrGetDCounts sums them
// (src/plink_freq.cpp:475).  Here the tracks stay in HBM in that same packed form
// (dosage.hpp:DosageView) and a lane resolves its sample with one popcount:
//     value index = rank[word] + popcount(present[word] & bits below the sample).
// A wave covers one 64-sample word, so the presence word and the rank are wave-uniform
// loads and the values it touches are one contiguous run.
#include "dosage.hpp"

#include "decode_device.hpp"
#include "device_utils.hpp"
#include "synth.hpp"

#include <algorithm>

namespace pgh {

namespace {

constexpr uint32_t kNoDosage = 0xffffu; // neither an explicit dosage nor a call

struct DosageRow {
	const uint64_t *present; // NULL: the variant has hardcalls only
	const uint32_t *rank;
	const uint16_t *values;
};

__device__ __forceinline__ DosageRow RowOf(const DosageView &dos, uint32_t local_variant) {
	const int32_t r = dos.row_of ? dos.row_of[local_variant] : -1;
	if (r < 0) {
		return DosageRow {nullptr, nullptr, nullptr};
	}
	const uint64_t at = static_cast<uint64_t>(r) * dos.words;
	return DosageRow {dos.present + at, dos.rank + at, dos.values + dos.val_off[r]};
}

// the sample's ALT dosage on the 16384-per-copy scale: the explicit value, else its call
__device__ __forceinline__ uint32_t DosageOrCall(const DosageRow &row, const uint32_t *row32, uint32_t s) {
	if (row.present) {
		const uint32_t w = s >> 6, b = s & 63u;
		const uint64_t bits = row.present[w];
		if ((bits >> b) & 1ull) {
			return row.values[row.rank[w] + static_cast<uint32_t>(__popcll(bits & ((1ull << b) - 1ull)))];
		}
	}
	const uint32_t code = (row32[s >> 4] >> (2u * (s & 15u))) & 3u;
	return code == 3u ? kNoDosage : code << 14;
}

__device__ __forceinline__ uint64_t WaveSum64(uint64_t x) {
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) {
		x += __shfl_xor(x, off, 64);
	}
	return x;
}

__global__ __launch_bounds__(256) void k_dosage_rank(const uint64_t *__restrict__ present, uint32_t words,
                                                     uint32_t *__restrict__ rank) {
	__shared__ uint32_t s_wave[4];
	const uint64_t at = static_cast<uint64_t>(blockIdx.x) * words;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t carry = 0;
	for (uint32_t base = 0; base < words; base += 256u) {
		const uint32_t w = base + threadIdx.x;
		const uint32_t c = w < words ? static_cast<uint32_t>(__popcll(present[at + w])) : 0u;
		uint32_t incl = c;
		for (int d = 1; d < 64; d <<= 1) {
			const uint32_t up = __shfl_up(incl, d);
			if (lane >= static_cast<uint32_t>(d)) {
				incl += up;
			}
		}
		__syncthreads(); // s_wave of the previous trip has been read
		if (lane == 63u) {
			s_wave[wave] = incl;
		}
		__syncthreads();
		uint32_t before = carry;
		for (uint32_t k = 0; k < wave; k++) {
			before += s_wave[k];
		}
		if (w < words) {
			rank[at + w] = before + incl - c;
		}
		carry += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
	}
}

// ---- ingest: dosage tracks out of the staged record bytes ------------------------------------------
// A record is [main track][phase track if 0x10][dosage track: 0x20 id list / 0x40 dense / 0x60 bit array].
// The main track's end comes from the record decode (aux_at); the phase track's length depends on the
// number of hets in the finished row.  Pass 1 (k_dosage_locate) finds the track, writes the presence bits
// and counts the values; a one-block scan hands out value offsets; k_dosage_rank builds the ranks;
// pass 2 (k_dosage_values) copies the values -- for the dense 0x40 shape through the ranks, dropping its
// 65535 "no dosage" entries.
__device__ __forceinline__ uint32_t BlockSum256(uint32_t x, uint32_t *s_part) {
	x = WaveSum(x);
	__syncthreads();
	if ((threadIdx.x & 63u) == 0) {
		s_part[threadIdx.x >> 6] = x;
	}
	__syncthreads();
	return s_part[0] + s_part[1] + s_part[2] + s_part[3];
}

__device__ __forceinline__ void IngestFail(const DosageIngest &b, uint32_t r) {
	if (threadIdx.x == 0) {
		atomicCAS(b.error, 0, static_cast<int>(b.variant0 + r) + 1);
	}
}

__global__ __launch_bounds__(256) void k_dosage_locate(DosageIngest b) {
	__shared__ uint32_t s_part[4];
	__shared__ uint64_t s_cur;
	__shared__ uint32_t s_len;
	__shared__ int s_ok;
	const uint32_t r = blockIdx.x;
	const int32_t dr = b.dos_row[r];
	if (dr < 0) {
		return;
	}
	if (threadIdx.x == 0) {
		b.count[r] = 0;
		b.track[r] = 0;
	}
	const Src src {b.bytes, b.bytes_len};
	const uint32_t N = b.sample_ct;
	const uint32_t t = b.vrtype[r];
	const uint64_t rec_end = b.rec_begin[r + 1];
	uint64_t cur = b.aux_at[r];
	if ((t & 0x08u) || cur > rec_end || cur < b.rec_begin[r]) {
		IngestFail(b, r);
		return;
	}
	if (t & 0x10u) {
		// phase track: ceil((1 + hets) / 8) bytes; bit 0 set = they are phase-present flags and
		// ceil(flagged / 8) bytes of phase bits follow
		const uint32_t *row32 = reinterpret_cast<const uint32_t *>(b.rows + static_cast<uint64_t>(b.row0 + r) * b.pitch);
		uint32_t hets = 0;
		for (uint32_t w = threadIdx.x; w < (N + 15) / 16; w += 256u) {
			const uint32_t x = row32[w]; // slots past N are zero (hom-ref)
			hets += __popc(x & ~(x >> 1) & kLow);
		}
		hets = BlockSum256(hets, s_part);
		const uint32_t head = (1 + hets + 7) / 8;
		if (cur + head > rec_end) {
			IngestFail(b, r);
			return;
		}
		uint32_t flagged = 0;
		if (src.Byte(cur) & 1u) {
			for (uint32_t i = threadIdx.x; i < head; i += 256u) {
				uint32_t byte = src.Byte(cur + i);
				if (i == 0) {
					byte &= ~1u;
				}
				if (i == head - 1 && ((1 + hets) & 7u)) {
					byte &= (1u << ((1 + hets) & 7u)) - 1u;
				}
				flagged += __popc(byte);
			}
			flagged = BlockSum256(flagged, s_part);
			flagged = (flagged + 7) / 8;
		}
		cur += head + flagged;
		if (cur > rec_end) {
			IngestFail(b, r);
			return;
		}
	}
	uint64_t *present = b.present + static_cast<uint64_t>(dr) * b.words;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	const uint32_t kind = t & 0x60u;
	uint32_t count = 0;
	uint64_t track = cur;
	if (kind == 0x60u) {
		const uint64_t nb = (N + 7) / 8;
		if (cur + nb > rec_end) {
			IngestFail(b, r);
			return;
		}
		for (uint32_t w = threadIdx.x; w < b.words; w += 256u) {
			uint64_t bits = 0;
			for (uint32_t k = 0; k < 8; k++) {
				const uint64_t at = 8ull * w + k;
				if (at < nb) {
					bits |= static_cast<uint64_t>(src.Byte(cur + at)) << (8 * k);
				}
			}
			const uint32_t live = N - 64u * w;
			if (live < 64u) {
				bits &= (1ull << live) - 1ull;
			}
			present[w] = bits;
			count += static_cast<uint32_t>(__popcll(bits));
		}
		count = BlockSum256(count, s_part);
		track = cur + nb;
	} else if (kind == 0x40u) {
		if (cur + 2ull * N > rec_end) {
			IngestFail(b, r);
			return;
		}
		bool bad = false;
		for (uint32_t w = wave; w < b.words; w += 4u) {
			const uint32_t s = 64u * w + lane;
			const uint32_t v = s < N ? src.Le(cur + 2ull * s, 2) : 0xffffu;
			bad |= v > 32768u && v != 0xffffu;
			const uint64_t bits = __ballot(v != 0xffffu);
			if (lane == 0) {
				present[w] = bits;
				count += static_cast<uint32_t>(__popcll(bits));
			}
		}
		count = BlockSum256(count, s_part);
		if (__syncthreads_or(bad)) {
			IngestFail(b, r);
			return;
		}
	} else { // 0x20: the samples are listed (a difflist without its value section), their values follow in that order
		if (threadIdx.x == 0) {
			s_ok = 1;
		}
		__syncthreads();
		if (wave == 0) {
			uint32_t len = 0;
			auto apply = [&](uint32_t, uint32_t id, uint64_t) { atomicOr(reinterpret_cast<unsigned long long *>(&present[id >> 6]), 1ull << (id & 63u)); };
			const bool ok = WalkDifflistIds(src, cur, rec_end, N, b.id_bytes, false, len, apply);
			if (lane == 0) {
				s_ok = ok;
				s_cur = cur;
				s_len = len;
			}
		}
		__syncthreads();
		if (!s_ok) {
			IngestFail(b, r);
			return;
		}
		count = s_len;
		track = s_cur;
	}
	if (track + 2ull * (kind == 0x40u ? N : count) > rec_end) {
		IngestFail(b, r);
		return;
	}
	if (threadIdx.x == 0) {
		b.count[r] = count;
		b.track[r] = track;
	}
}

// value offsets of the batch's tracks, in record order, continuing the dataset's running total
__global__ __launch_bounds__(1024) void k_dosage_offsets(DosageIngest b) {
	__shared__ uint64_t s_wave[16];
	__shared__ uint64_t s_carry;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	if (threadIdx.x == 0) {
		s_carry = *b.total;
	}
	__syncthreads();
	for (uint32_t base = 0; base < b.n; base += 1024u) {
		const uint32_t r = base + threadIdx.x;
		const bool on = r < b.n && b.dos_row[r] >= 0;
		const uint64_t c = on ? b.count[r] : 0ull;
		uint64_t incl = c;
		for (int d = 1; d < 64; d <<= 1) {
			const uint64_t up = __shfl_up(incl, d);
			if (lane >= static_cast<uint32_t>(d)) {
				incl += up;
			}
		}
		if (lane == 63u) {
			s_wave[wave] = incl;
		}
		__syncthreads();
		uint64_t before = s_carry;
		for (uint32_t k = 0; k < wave; k++) {
			before += s_wave[k];
		}
		if (on) {
			b.val_off[b.dos_row[r]] = before + incl - c;
		}
		__syncthreads();
		if (threadIdx.x == 1023u) {
			s_carry = before + incl;
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		*b.total = s_carry;
		if (s_carry > b.capacity) {
			atomicCAS(b.error, 0, static_cast<int>(b.variant0) + 1);
		}
	}
}

__global__ __launch_bounds__(256) void k_dosage_values(DosageIngest b) {
	const uint32_t r = blockIdx.x;
	const int32_t dr = b.dos_row[r];
	if (dr < 0 || *b.error != 0 || *b.total > b.capacity) {
		return;
	}
	const Src src {b.bytes, b.bytes_len};
	const uint32_t N = b.sample_ct;
	const uint32_t count = b.count[r];
	const uint64_t track = b.track[r];
	const uint64_t *present = b.present + static_cast<uint64_t>(dr) * b.words;
	const uint32_t *rank = b.rank + static_cast<uint64_t>(dr) * b.words;
	uint16_t *out = b.values + b.val_off[dr];
	// a list that names a sample twice leaves fewer presence bits than it has values
	const uint32_t bits_set = rank[b.words - 1] + static_cast<uint32_t>(__popcll(present[b.words - 1]));
	if (bits_set != count) {
		IngestFail(b, r);
		return;
	}
	bool bad = false;
	if ((b.vrtype[r] & 0x60u) == 0x40u) {
		for (uint32_t s = threadIdx.x; s < N; s += 256u) {
			const uint64_t bits = present[s >> 6];
			if ((bits >> (s & 63u)) & 1ull) {
				out[rank[s >> 6] + static_cast<uint32_t>(__popcll(bits & ((1ull << (s & 63u)) - 1ull)))] =
				    static_cast<uint16_t>(src.Le(track + 2ull * s, 2));
			}
		}
	} else {
		for (uint32_t j = threadIdx.x; j < count; j += 256u) {
			const uint32_t v = src.Le(track + 2ull * j, 2);
			bad |= v > 32768u;
			out[j] = static_cast<uint16_t>(v);
		}
	}
	if (bad) {
		atomicCAS(b.error, 0, static_cast<int>(b.variant0 + r) + 1);
	}
}

// ---- synthetic tracks for the benchmarks (the file-free twin of LaunchSynthFill) ----------------
__global__ __launch_bounds__(256) void k_synth_dosage_bits(uint64_t *__restrict__ present, uint32_t words,
                                                           uint32_t sample_ct, uint32_t variant0, uint64_t seed,
                                                           uint32_t threshold) {
	const uint64_t key = Mix64(Mix64(seed) ^ (static_cast<uint64_t>(variant0 + blockIdx.x) << 32) ^ 0x5851f42d4c957f2dULL);
	for (uint32_t w = threadIdx.x; w < words; w += 256u) {
		uint64_t bits = 0;
		for (uint32_t b = 0; b < 64u; b++) {
			const uint32_t s = 64u * w + b;
			if (s < sample_ct && static_cast<uint32_t>(Mix64(key ^ s)) < threshold) {
				bits |= 1ull << b;
			}
		}
		present[static_cast<uint64_t>(blockIdx.x) * words + w] = bits;
	}
}

__global__ __launch_bounds__(256) void k_dosage_row_totals(const uint64_t *__restrict__ present,
                                                           const uint32_t *__restrict__ rank, uint32_t words,
                                                           uint32_t rows, uint64_t *__restrict__ totals) {
	const uint32_t r = blockIdx.x * 256u + threadIdx.x;
	if (r < rows) {
		const uint64_t last = static_cast<uint64_t>(r) * words + (words - 1);
		totals[r] = rank[last] + static_cast<uint64_t>(__popcll(present[last]));
	}
}

// values keyed by (variant, sample), the same draw the presence bit came from: a shard holds the same
// tracks as the whole, and pgh_synth_write_dosage_files writes them too
__global__ __launch_bounds__(256) void k_synth_dosage_values(const uint64_t *__restrict__ present,
                                                             const uint32_t *__restrict__ rank,
                                                             const uint64_t *__restrict__ val_off,
                                                             uint16_t *__restrict__ values, uint32_t words,
                                                             uint32_t sample_ct, uint32_t variant0, uint64_t seed) {
	const uint64_t key = Mix64(Mix64(seed) ^ (static_cast<uint64_t>(variant0 + blockIdx.x) << 32) ^ 0x5851f42d4c957f2dULL);
	const uint64_t at = static_cast<uint64_t>(blockIdx.x) * words;
	uint16_t *out = values + val_off[blockIdx.x];
	for (uint32_t s = threadIdx.x; s < sample_ct; s += 256u) {
		const uint64_t bits = present[at + (s >> 6)];
		if ((bits >> (s & 63u)) & 1ull) {
			const uint64_t h = Mix64(key ^ s);
			out[rank[at + (s >> 6)] + static_cast<uint32_t>(__popcll(bits & ((1ull << (s & 63u)) - 1ull)))] =
			    static_cast<uint16_t>(((h >> 32) * 32769ull) >> 32);
		}
	}
}

// 16 presence bits -> one bit in every even position of a 32-bit word (the 01 slot mask of 16 samples)
__device__ __forceinline__ uint32_t Spread16(uint32_t x) {
	x = (x | (x << 8)) & 0x00ff00ffu;
	x = (x | (x << 4)) & 0x0f0f0f0fu;
	x = (x | (x << 2)) & 0x33333333u;
	x = (x | (x << 1)) & 0x55555555u;
	return x;
}

// One workgroup per variant, two sweeps:
//   A. the hardcalls of the samples WITHOUT an explicit dosage: a lane takes 64 samples per trip
//      (16 bytes of the row + their presence word), spreads the presence bits over the 2-bit slots
//      and counts hets and hom-alts among the rest -- the tally kernel's popcount algebra;
//   B. the explicit values.  Without a sample subset they are one contiguous run of uint16, streamed
//      16 bytes per lane with no reference to which sample owns which; with a subset a lane owns a
//      64-sample word and walks its included presence bits through the rank table.
// sum = 16384 (hets + 2 hom-alts) + sum of values, and likewise for the squares and the count.
template <bool HAS_INCLUDE>
__global__ __launch_bounds__(256) void k_dosage_sums(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                     uint32_t sample_ct, DosageView dos, uint32_t v0,
                                                     const uint32_t *__restrict__ vlist,
                                                     const uint64_t *__restrict__ include,
                                                     uint64_t *__restrict__ out) {
	__shared__ uint64_t s_part[4][3];
	const uint32_t i = blockIdx.x;
	const uint32_t lv = vlist ? vlist[i] : v0 + i;
	const uint4 *row128 = reinterpret_cast<const uint4 *>(rows + static_cast<uint64_t>(lv) * pitch);
	const int32_t r = dos.row_of ? dos.row_of[lv] : -1;
	const uint64_t *present = r >= 0 ? dos.present + static_cast<uint64_t>(r) * dos.words : nullptr;
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint32_t het = 0, hom_alt = 0, called = 0, n_explicit = 0, vsum = 0;
	uint64_t vssq = 0;
	for (uint32_t w = threadIdx.x; w < dos.words; w += 256u) {
		const uint4 q = LoadStream(row128 + w);
		const uint64_t e = present ? present[w] : 0ull;
		const uint32_t live = sample_ct - 64u * w;
		uint64_t m = live >= 64u ? ~0ull : (1ull << live) - 1ull;
		if (HAS_INCLUDE) {
			m &= include[w];
		}
		const uint64_t keep = m & ~e;
		const uint32_t x[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
		for (int j = 0; j < 4; j++) {
			const uint32_t kk = Spread16(static_cast<uint32_t>(keep >> (16 * j)) & 0xffffu);
			const uint32_t lo = x[j] & kLow, hi = (x[j] >> 1) & kLow;
			het += __popc(lo & ~hi & kk);
			hom_alt += __popc(hi & ~lo & kk);
			called += __popc(kk & ~(lo & hi));
		}
		if (HAS_INCLUDE) {
			uint64_t take = e & m;
			const uint16_t *vals = dos.values + dos.val_off[r] + dos.rank[static_cast<uint64_t>(r) * dos.words + w];
			while (take) {
				const uint32_t b = static_cast<uint32_t>(__ffsll(static_cast<long long>(take))) - 1u;
				const uint32_t u = vals[__popcll(e & ((1ull << b) - 1ull))];
				vsum += u;
				vssq += static_cast<uint64_t>(u) * u;
				n_explicit++;
				take &= take - 1ull;
			}
		}
	}
	if (!HAS_INCLUDE && present) {
		const uint64_t last = static_cast<uint64_t>(r) * dos.words + (dos.words - 1);
		const uint64_t o0 = dos.val_off[r], o1 = o0 + dos.rank[last] + static_cast<uint64_t>(__popcll(dos.present[last]));
		for (uint64_t at = (o0 & ~7ull) + 8ull * threadIdx.x; at < o1; at += 8ull * 256u) {
			const uint4 q = LoadStream(reinterpret_cast<const uint4 *>(dos.values + at));
			const uint32_t x[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
			for (int j = 0; j < 8; j++) {
				const uint64_t idx = at + j;
				const uint32_t u = (x[j >> 1] >> (16 * (j & 1))) & 0xffffu;
				if (idx >= o0 && idx < o1) {
					vsum += u;
					vssq += static_cast<uint64_t>(u) * u;
				}
			}
		}
		if (threadIdx.x == 0) {
			n_explicit = static_cast<uint32_t>(o1 - o0);
		}
	}
	uint64_t sum = 16384ull * (het + 2ull * hom_alt) + vsum;
	uint64_t ssq = 16384ull * 16384ull * (het + 4ull * hom_alt) + vssq;
	uint64_t nm = static_cast<uint64_t>(called) + n_explicit;
	sum = WaveSum64(sum);
	ssq = WaveSum64(ssq);
	nm = WaveSum64(nm);
	if (lane == 0) {
		s_part[wave][0] = sum;
		s_part[wave][1] = ssq;
		s_part[wave][2] = nm;
	}
	__syncthreads();
	if (threadIdx.x < 3) {
		out[3ull * i + threadIdx.x] =
		    s_part[0][threadIdx.x] + s_part[1][threadIdx.x] + s_part[2][threadIdx.x] + s_part[3][threadIdx.x];
	}
}

__global__ __launch_bounds__(256) void k_dosage_unpack(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                       DosageView dos, uint32_t v0, const uint32_t *__restrict__ vlist,
                                                       const uint32_t *__restrict__ sel, uint32_t n_out,
                                                       double *__restrict__ out, uint64_t out_stride) {
	const uint32_t k = blockIdx.x * 256u + threadIdx.x;
	if (k >= n_out) {
		return;
	}
	const uint32_t i = blockIdx.y;
	const uint32_t lv = vlist ? vlist[i] : v0 + i;
	const uint32_t *row32 = reinterpret_cast<const uint32_t *>(rows + static_cast<uint64_t>(lv) * pitch);
	const uint32_t u = DosageOrCall(RowOf(dos, lv), row32, sel ? sel[k] : k);
	// value / 16384 is exact in binary: the doubles are the reference's (Dosage16ToDoublesMinus9)
	out[static_cast<uint64_t>(i) * out_stride + k] = u == kNoDosage ? -9.0 : static_cast<double>(u) * 0x1p-14;
}

// sample-major form of the same doubles (read_pfile orient := 'sample', dosages := true): out[k][j]
__global__ __launch_bounds__(256) void k_dosage_unpack_transposed(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                                  DosageView dos, const uint32_t *__restrict__ vlist,
                                                                  uint32_t n_var, const uint32_t *__restrict__ sel,
                                                                  uint32_t k_first, double *__restrict__ out,
                                                                  uint64_t out_stride) {
	const uint32_t j = blockIdx.x * 256u + threadIdx.x;
	if (j >= n_var) {
		return;
	}
	const uint32_t k = k_first + blockIdx.y;
	const uint32_t lv = vlist[j];
	const uint32_t *row32 = reinterpret_cast<const uint32_t *>(rows + static_cast<uint64_t>(lv) * pitch);
	const uint32_t u = DosageOrCall(RowOf(dos, lv), row32, sel ? sel[k] : k);
	out[static_cast<uint64_t>(blockIdx.y) * out_stride + j] = u == kNoDosage ? -9.0 : static_cast<double>(u) * 0x1p-14;
}

__global__ __launch_bounds__(256) void k_score_tables_dosage(const uint64_t *__restrict__ sums,
                                                             const uint8_t *__restrict__ flip, uint32_t n_scored,
                                                             int mode, double *__restrict__ ts, double *__restrict__ td,
                                                             double *__restrict__ lin, uint32_t *__restrict__ ac) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i >= n_scored) {
		return;
	}
	const uint64_t non_missing = sums[3ull * i + 2];
	double s[4] = {0.0, 0.0, 0.0, 0.0};
	double d[4] = {0.0, 0.0, 0.0, 0.0};
	double l[4] = {0.0, 0.0, 0.0, 0.0}; // contribution of dosage x: ((sgn * x + off) - centre) * scale
	uint32_t inc = 0;
	if (non_missing != 0) {
		const bool fl = flip && flip[i];
		// the reference adds the doubles in sample order; every partial sum is a multiple of 2^-14 below 2^39,
		// so its total is exactly this quotient (src/plink_score.cpp:602-611)
		const double sum_alt = static_cast<double>(sums[3ull * i]) * 0x1p-14;
		const double mean_alt = sum_alt / static_cast<double>(non_missing);
		l[0] = fl ? -1.0 : 1.0;
		l[1] = fl ? 2.0 : 0.0;
		l[3] = 1.0;
		if (mode == 2) { // center
			const double freq = mean_alt / 2.0;
			const double sd = sqrt(2.0 * freq * (1.0 - freq));
			if (sd != 0.0) {
				const double mean_scored = fl ? (2.0 - mean_alt) : mean_alt;
				for (int g = 0; g < 3; g++) {
					const double scored = fl ? (2.0 - static_cast<double>(g)) : static_cast<double>(g);
					s[g] = (scored - mean_scored) / sd;
				}
				l[2] = mean_scored;
				l[3] = 1.0 / sd;
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
		ts[4ull * i + g] = s[g];
		td[4ull * i + g] = d[g];
		lin[4ull * i + g] = inc ? l[g] : 0.0; // a skipped variant contributes nothing, dosage or call
	}
	ac[i] = inc;
}

// Sample-owning form for tracks with gaps, at a constant cost per variant whatever its density: a lane owns four
// consecutive samples (one nibble of a presence word, one byte of the 2-bit row) over a slice of the scored
// variants.  Its present samples' values are consecutive in the value run -- one 8-byte load at
// rank + popcount(bits below the nibble) covers them -- and each sample takes its explicit dosage through the
// affine map or its call through the code table.  Two variants' loads are in flight together (more costs
// occupancy: 82 VGPRs at two, 102 at four).
template <int NCOLS>
__global__ __launch_bounds__(256) void k_score_dosage(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                      uint32_t sample_ct, DosageView dos,
                                                      const uint32_t *__restrict__ vlist, uint32_t n_scored,
                                                      uint32_t slice_len, const double *__restrict__ weights,
                                                      uint32_t w_stride, uint32_t n_cols, uint32_t out_stride,
                                                      const double *__restrict__ ts, const double *__restrict__ lin,
                                                      const uint32_t *__restrict__ ac, int mode,
                                                      double *__restrict__ score, double *__restrict__ dosage_sum,
                                                      uint32_t *__restrict__ miss) {
	constexpr uint32_t kStage = 64;
	constexpr uint32_t kGroup = 2;
	constexpr uint32_t kPer = 4;
	__shared__ double s_ts[kStage][4];
	__shared__ double s_lin[kStage][4];
	__shared__ double s_w[kStage][NCOLS];
	__shared__ uint64_t s_row[kStage];  // byte offset of the variant's 2-bit row
	__shared__ uint64_t s_bits[kStage]; // word offset of its presence / rank row
	__shared__ uint64_t s_vals[kStage]; // where its values start
	__shared__ uint32_t s_on[kStage];   // ~0: scored and carrying a track; 0 otherwise (presence reads as empty)
	__shared__ uint32_t s_counts[kStage];
	const uint32_t s0 = (blockIdx.x * 256u + threadIdx.x) * kPer;
	const bool live = s0 < sample_ct;
	const uint32_t sl = live ? s0 : 0u; // lanes past the end read sample 0's words and write nothing
	const uint32_t w = sl >> 6, b0 = sl & 63u;
	const uint64_t below = (1ull << b0) - 1ull;
	const uint32_t shift = 2u * (sl & 15u);
	const uint32_t i_begin = blockIdx.y * slice_len;
	const uint32_t i_end = min(i_begin + slice_len, n_scored);
	double acc[kPer][NCOLS];
	double dsum[kPer];
	uint32_t missed[kPer];
#pragma unroll
	for (uint32_t q = 0; q < kPer; q++) {
		dsum[q] = 0.0;
		missed[q] = 0;
#pragma unroll
		for (int c = 0; c < NCOLS; c++) {
			acc[q][c] = 0.0;
		}
	}
	for (uint32_t base = i_begin; base < i_end; base += kStage) {
		const uint32_t cnt = min(kStage, i_end - base);
		__syncthreads();
		for (uint32_t k = threadIdx.x; k < kStage * 4u; k += 256u) {
			const bool in = (k >> 2) < cnt;
			s_ts[k >> 2][k & 3] = in ? ts[4ull * base + k] : 0.0;
			s_lin[k >> 2][k & 3] = in ? lin[4ull * base + k] : 0.0;
		}
		for (uint32_t k = threadIdx.x; k < kStage * NCOLS; k += 256u) {
			const uint32_t c = k % NCOLS, v = k / NCOLS;
			s_w[v][c] = (v < cnt && c < n_cols) ? weights[static_cast<uint64_t>(base + v) * w_stride + c] : 0.0;
		}
		for (uint32_t k = threadIdx.x; k < kStage; k += 256u) {
			// entries past the slice read the slice's first variant with empty tables: they add nothing
			const uint32_t lv = k < cnt ? vlist[base + k] : vlist[i_begin];
			const int32_t r = k < cnt ? dos.row_of[lv] : -1;
			const uint32_t on = k < cnt ? ac[base + k] : 0u;
			s_row[k] = static_cast<uint64_t>(lv) * pitch;
			s_bits[k] = static_cast<uint64_t>(r < 0 ? 0 : r) * dos.words;
			s_vals[k] = r < 0 ? 0ull : dos.val_off[r];
			s_on[k] = (r >= 0 && on != 0u) ? ~0u : 0u;
			s_counts[k] = on != 0u;
		}
		__syncthreads();
		for (uint32_t k0 = 0; k0 < cnt; k0 += kGroup) {
			uint64_t bits[kGroup], vals[kGroup];
			uint32_t word[kGroup], rk[kGroup];
#pragma unroll
			for (uint32_t j = 0; j < kGroup; j++) {
				const uint32_t k = k0 + j;
				bits[j] = dos.present[s_bits[k] + w];
				rk[j] = dos.rank[s_bits[k] + w]; // with the bits, not behind them: one dependent load fewer
				word[j] = reinterpret_cast<const uint32_t *>(rows + s_row[k])[sl >> 4];
			}
#pragma unroll
			for (uint32_t j = 0; j < kGroup; j++) {
				const uint32_t k = k0 + j;
				bits[j] &= static_cast<uint64_t>(static_cast<int64_t>(static_cast<int32_t>(s_on[k]))); // all ones or zero
				vals[j] = 0;
				if ((bits[j] >> b0) & 0xfull) {
					const uint32_t idx = rk[j] + static_cast<uint32_t>(__popcll(bits[j] & below));
					__builtin_memcpy(&vals[j], dos.values + s_vals[k] + idx, 8); // up to four values; the run is padded
				}
			}
#pragma unroll
			for (uint32_t j = 0; j < kGroup; j++) {
				const uint32_t k = k0 + j;
				const uint32_t nib = static_cast<uint32_t>(bits[j] >> b0) & 0xfu;
				const uint32_t codes = word[j] >> shift;
				uint64_t run = vals[j];
				const double l0 = s_lin[k][0], l1 = s_lin[k][1], l2 = s_lin[k][2], l3 = s_lin[k][3];
#pragma unroll
				for (uint32_t q = 0; q < kPer; q++) {
					// both candidates, then a select: no divergent branch per sample
					const bool has = (nib >> q) & 1u;
					const double d = static_cast<double>(static_cast<uint32_t>(run) & 0xffffu) * 0x1p-14;
					run >>= has ? 16u : 0u;
					const uint32_t g = (codes >> (2u * q)) & 3u;
					const double x = has ? ((l0 * d + l1) - l2) * l3 : s_ts[k][g];
					missed[q] += (!has && g == 3u) ? s_counts[k] : 0u;
					dsum[q] += x;
#pragma unroll
					for (int c = 0; c < NCOLS; c++) {
						acc[q][c] = fma(s_w[k][c], x, acc[q][c]);
					}
				}
			}
		}
	}
	if (!live) {
		return;
	}
#pragma unroll
	for (uint32_t q = 0; q < kPer; q++) {
		const uint32_t s = s0 + q;
		if (s >= sample_ct) {
			break;
		}
#pragma unroll
		for (int c = 0; c < NCOLS; c++) {
			if (static_cast<uint32_t>(c) < n_cols) {
				unsafeAtomicAdd(score + static_cast<uint64_t>(s) * out_stride + c, acc[q][c]);
			}
		}
		if (dosage_sum && mode != 2) {
			unsafeAtomicAdd(dosage_sum + s, dsum[q]);
		}
		if (miss && missed[q]) {
			atomicAdd(miss + s, missed[q]);
		}
	}
}

// Variants whose every sample carries an explicit dosage (densely imputed data): the value run is indexed by
// the sample itself, no presence bits, ranks or calls are read, and a lane's term is the affine map of its
// value.  Same shape as k_score_dosage: one lane per sample, eight variants' loads in flight.
template <int NCOLS>
__global__ __launch_bounds__(256) void k_score_dosage_full(uint32_t sample_ct, DosageView dos,
                                                           const uint32_t *__restrict__ vlist, uint32_t n_scored,
                                                           uint32_t slice_len, const double *__restrict__ weights,
                                                           uint32_t w_stride, uint32_t n_cols, uint32_t out_stride,
                                                           const double *__restrict__ lin,
                                                           const uint32_t *__restrict__ ac, int mode,
                                                           double *__restrict__ score,
                                                           double *__restrict__ dosage_sum) {
	constexpr uint32_t kStage = 64;
	constexpr uint32_t kGroup = 4;
	constexpr uint32_t kPer = 4; // samples per lane: one 8-byte load of the value run (2-byte aligned)
	__shared__ double s_lin[kStage][4];
	__shared__ double s_w[kStage][NCOLS];
	__shared__ uint64_t s_vals[kStage];
	const uint32_t s0 = (blockIdx.x * 256u + threadIdx.x) * kPer;
	// a lane past the end re-reads the last full group of four (its sums are dropped)
	const uint32_t s_load = s0 + kPer <= sample_ct ? s0 : (sample_ct >= kPer ? sample_ct - kPer : 0u);
	const uint32_t i_begin = blockIdx.y * slice_len;
	const uint32_t i_end = min(i_begin + slice_len, n_scored);
	double acc[kPer][NCOLS];
	double dsum[kPer];
#pragma unroll
	for (uint32_t q = 0; q < kPer; q++) {
		dsum[q] = 0.0;
#pragma unroll
		for (int c = 0; c < NCOLS; c++) {
			acc[q][c] = 0.0;
		}
	}
	for (uint32_t base = i_begin; base < i_end; base += kStage) {
		const uint32_t cnt = min(kStage, i_end - base);
		__syncthreads();
		for (uint32_t k = threadIdx.x; k < kStage * 4u; k += 256u) {
			// a skipped variant (ac == 0) has an all-zero map; entries past the slice likewise
			s_lin[k >> 2][k & 3] = (k >> 2) < cnt ? lin[4ull * base + k] : 0.0;
		}
		for (uint32_t k = threadIdx.x; k < kStage * NCOLS; k += 256u) {
			const uint32_t c = k % NCOLS, v = k / NCOLS;
			s_w[v][c] = (v < cnt && c < n_cols) ? weights[static_cast<uint64_t>(base + v) * w_stride + c] : 0.0;
		}
		for (uint32_t k = threadIdx.x; k < kStage; k += 256u) {
			const uint32_t lv = vlist[k < cnt ? base + k : i_begin];
			s_vals[k] = dos.val_off[dos.row_of[lv]];
		}
		__syncthreads();
		for (uint32_t k0 = 0; k0 < cnt; k0 += kGroup) {
			uint2 u[kGroup];
#pragma unroll
			for (uint32_t j = 0; j < kGroup; j++) {
				__builtin_memcpy(&u[j], dos.values + s_vals[k0 + j] + s_load, 8);
			}
#pragma unroll
			for (uint32_t j = 0; j < kGroup; j++) {
				const uint32_t k = k0 + j;
				const uint32_t v4[kPer] = {u[j].x & 0xffffu, u[j].x >> 16, u[j].y & 0xffffu, u[j].y >> 16};
#pragma unroll
				for (uint32_t q = 0; q < kPer; q++) {
					const double d = static_cast<double>(v4[q]) * 0x1p-14;
					const double x = ((s_lin[k][0] * d + s_lin[k][1]) - s_lin[k][2]) * s_lin[k][3];
					dsum[q] += x;
#pragma unroll
					for (int c = 0; c < NCOLS; c++) {
						acc[q][c] = fma(s_w[k][c], x, acc[q][c]);
					}
				}
			}
		}
	}
	(void)ac;
	if (s0 + kPer <= sample_ct || s0 < sample_ct) {
#pragma unroll
		for (uint32_t q = 0; q < kPer; q++) {
			// lanes that re-read the last group own only the samples at or past s0
			const uint32_t s = s_load + q;
			if (s < s0 || s >= sample_ct) {
				continue;
			}
#pragma unroll
			for (int c = 0; c < NCOLS; c++) {
				if (static_cast<uint32_t>(c) < n_cols) {
					unsafeAtomicAdd(score + static_cast<uint64_t>(s) * out_stride + c, acc[q][c]);
				}
			}
			if (dosage_sum && mode != 2) {
				unsafeAtomicAdd(dosage_sum + s, dsum[q]);
			}
		}
	}
}

// plink_score's single weight column, the fast way round.  A sample's contribution at a variant is
//     ts[call]                                   without an explicit dosage,
//     ts[call] + (affine(dosage) - ts[call])     with one,
// so the dosage-bearing variants first go through the hardcall kernel with their (dosage-mean) tables
// (score.hip, a few instructions per sample), and this kernel adds the bracket for the explicit
// entries only.  A lane owns a 64-sample word of a 4096-sample tile and walks its presence bits in
// order -- the values are consumed in the order they are stored, so no rank arithmetic per entry --
// and the tile's sums live in LDS (FP64 LDS atomics: lanes of one instruction never share a sample).
// Work is proportional to the explicit entries: ~30 instructions per entry instead of ~40 per sample.
template <bool TRACK>
__global__ __launch_bounds__(1024, 8) void k_score_dosage_fix(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                           uint32_t sample_ct, DosageView dos,
                                                           const uint32_t *__restrict__ vlist, uint32_t n_scored,
                                                           uint32_t slice_len, const double *__restrict__ weights,
                                                           uint32_t w_stride, const double *__restrict__ ts,
                                                           const double *__restrict__ lin,
                                                           const uint32_t *__restrict__ ac,
                                                           double *__restrict__ score, uint32_t out_stride,
                                                           double *__restrict__ dosage_sum,
                                                           uint32_t *__restrict__ miss) {
	// The tile's sums in LDS, [bit of the word][word]: a wave's atomic instruction has lane l at 8-byte element
	// 64 b_l + l, i.e. in bank pair l mod 32 whatever its bit b_l is -- the two-pass minimum of a 64 x 8-byte
	// access.  (Round 1 had [word][bit] with a row stride of 65, which sends the lanes to RANDOM banks: 22 LDS cycles
	// per FP64 atomic instruction instead of 7.4, tools/lds_atomic_probe.hip, profiles/r02_lds_atomic.txt.)
	constexpr uint32_t kWaves = 16;  // sixteen waves share one tile: the loop is a chain of memory latencies
	constexpr uint32_t kChunk = 128; // variants whose constants are staged in LDS at a time
	__shared__ double s_acc[64 * 64];
	__shared__ double s_dsum[TRACK ? 64 * 64 : 1];
	__shared__ uint64_t s_row[kChunk], s_bits[kChunk], s_vals[kChunk]; // row bytes / presence row / first value
	// an explicit dosage u (in 2^-14) replaces the call's term: delta = ((l0 d + l1) - l2) l3 - ts[call], d = u 2^-14,
	// staged as delta = a u + b[call] with a = l0 l3 2^-14 and b[call] = (l1 - l2) l3 - ts[call]
	__shared__ double s_wt[kChunk], s_a[kChunk], s_b[kChunk][4];
	__shared__ uint32_t s_on[kChunk];
	for (uint32_t t = threadIdx.x; t < 64 * 64; t += 64u * kWaves) {
		s_acc[t] = 0.0;
		if (TRACK) {
			s_dsum[t] = 0.0;
		}
	}
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t w = blockIdx.x * 64u + lane;
	const bool in_row = w < dos.words;
	const uint32_t i_begin = blockIdx.y * slice_len;
	const uint32_t i_end = min(i_begin + slice_len, n_scored);
	for (uint32_t base = i_begin; base < i_end; base += kChunk) {
		const uint32_t cnt = min(kChunk, i_end - base);
		__syncthreads();
		if (threadIdx.x < cnt) {
			const uint32_t i = base + threadIdx.x;
			const uint32_t lv = vlist[i];
			const int32_t r = dos.row_of[lv];
			s_on[threadIdx.x] = ac[i] != 0u && r >= 0;
			s_row[threadIdx.x] = static_cast<uint64_t>(lv) * pitch;
			s_bits[threadIdx.x] = static_cast<uint64_t>(r < 0 ? 0 : r) * dos.words;
			s_vals[threadIdx.x] = r < 0 ? 0ull : dos.val_off[r];
			s_wt[threadIdx.x] = weights[static_cast<uint64_t>(i) * w_stride];
			const double l0 = lin[4ull * i + 0], l1 = lin[4ull * i + 1], l2 = lin[4ull * i + 2], l3 = lin[4ull * i + 3];
			s_a[threadIdx.x] = l0 * l3 * 0x1p-14;
			const double shift = (l1 - l2) * l3;
#pragma unroll
			for (int c = 0; c < 4; c++) {
				s_b[threadIdx.x][c] = shift - ts[4ull * i + c];
			}
		}
		__syncthreads();
	for (uint32_t k = wave; k < cnt; k += kWaves) {
		if (!s_on[k]) {
			continue; // skipped by the reference (nobody observed, or no variance under center)
		}
		const uint64_t at = s_bits[k] + w;
		uint64_t e = in_row ? dos.present[at] : 0ull;
		const uint32_t rk = in_row ? dos.rank[at] : 0u;
		uint4 q = make_uint4(0, 0, 0, 0);
		if (in_row) {
			q = *reinterpret_cast<const uint4 *>(rows + s_row[k] + 16ull * w);
		}
		if (__ballot(e != 0ull) == 0ull) {
			continue; // no explicit dosage in this tile
		}
		const uint16_t *vals = dos.values + s_vals[k] + rk;
		const double wt = s_wt[k], a = s_a[k];
		const uint64_t q_lo = q.x | (static_cast<uint64_t>(q.y) << 32), q_hi = q.z | (static_cast<uint64_t>(q.w) << 32);
#define PGH_DOSAGE_ENTRY(U)                                                                                            \
	{                                                                                                                  \
		const uint32_t b = static_cast<uint32_t>(__ffsll(static_cast<long long>(e))) - 1u;                             \
		e &= e - 1ull;                                                                                                 \
		const uint32_t code = static_cast<uint32_t>((b < 32u ? q_lo : q_hi) >> (2u * (b & 31u))) & 3u;                 \
		const double delta = fma(a, static_cast<double>(U), s_b[k][code]);                                             \
		atomicAdd(&s_acc[b * 64u + lane], wt * delta);                                                                 \
		if (TRACK) {                                                                                                   \
			atomicAdd(&s_dsum[b * 64u + lane], delta);                                                                 \
		}                                                                                                              \
		if (miss && code == 3u) {                                                                                      \
			atomicSub(miss + 64u * w + b, 1u); /* it has a dosage: not missing after all */                            \
		}                                                                                                              \
	}
		// The word's first sixteen values arrive in two 16-byte loads (2-byte aligned: the hardware takes
		// unaligned global loads), so a word costs one memory latency, not one per entry; they may run past
		// the word's own run, into its neighbours' or the array's padding, and only `have` of them are used.
		uint4 pa = make_uint4(0, 0, 0, 0), pb = make_uint4(0, 0, 0, 0);
		if (e) {
			__builtin_memcpy(&pa, vals, 16);
			__builtin_memcpy(&pb, vals + 8, 16);
		}
#define PGH_STEP(P)                                                                                                    \
	if (e) {                                                                                                           \
		PGH_DOSAGE_ENTRY((P) & 0xffffu)                                                                                \
	}                                                                                                                  \
	if (e) {                                                                                                           \
		PGH_DOSAGE_ENTRY((P) >> 16)                                                                                    \
	}
		PGH_STEP(pa.x) PGH_STEP(pa.y) PGH_STEP(pa.z) PGH_STEP(pa.w)
		PGH_STEP(pb.x) PGH_STEP(pb.y) PGH_STEP(pb.z) PGH_STEP(pb.w)
#undef PGH_STEP
		uint32_t n = 16;
		while (e) {
			const uint32_t u = vals[n++];
			PGH_DOSAGE_ENTRY(u)
		}
#undef PGH_DOSAGE_ENTRY
	}
	}
	__syncthreads();
	for (uint32_t t = threadIdx.x; t < 4096u; t += 64u * kWaves) {
		const uint32_t s = blockIdx.x * 4096u + t;
		const uint32_t at = (t & 63u) * 64u + (t >> 6); // (a 64-way bank conflict, eight times per slice: noise)
		if (s < sample_ct) {
			if (s_acc[at] != 0.0) {
				unsafeAtomicAdd(score + static_cast<uint64_t>(s) * out_stride, s_acc[at]);
			}
			if (TRACK && s_dsum[at] != 0.0) {
				unsafeAtomicAdd(dosage_sum + s, s_dsum[at]);
			}
		}
	}
}


// ---- entry records: the explicit dosages of sparse tracks, laid out for plink_score -----------------------------
// One 32-bit record per explicit dosage:
//     [31:16] the value (0 .. 32768)   [14:3] 64 * bit + word: the sample's element in the 4096-sample tile's LDS
//     image, pre-multiplied by 8          [1:0] the sample's hardcall
// A (row, tile) run holds the same entries as values[] does -- it starts at rank[row][64 tile] -- but ROUND-MAJOR:
// first the first entry of every word of the tile (words ascending), then every word's second entry, and so on.
// Any 64 consecutive records therefore belong to (nearly) 64 different words, and with the tile stored
// [bit][word] in LDS a wave's 64 atomic adds fall into 64 different 8-byte bank slots -- while every lane has an
// entry to work on (the bit walk of k_score_dosage_fix leaves half the lanes idle behind the fullest word).
// One wave per (row, tile); a round is one ballot.
__global__ __launch_bounds__(256) void k_dosage_records(const uint8_t *__restrict__ rows, uint64_t pitch, DosageView dos,
                                                        const uint32_t *__restrict__ row_variant,
                                                        const uint64_t *__restrict__ rec_off,
                                                        uint32_t *__restrict__ rec) {
	const uint32_t r = blockIdx.y;
	const uint64_t first = rec_off[r];
	if (rec_off[r + 1] == first) {
		return; // a track too dense for records (or an empty one)
	}
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6);
	if (tile * 64u >= dos.words) {
		return;
	}
	const uint32_t w = tile * 64u + lane;
	const bool in_row = w < dos.words;
	const uint64_t at = static_cast<uint64_t>(r) * dos.words + w;
	uint64_t e = in_row ? dos.present[at] : 0ull;
	const uint32_t rk = in_row ? dos.rank[at] : 0u;
	uint4 q = make_uint4(0, 0, 0, 0);
	if (in_row) {
		q = *reinterpret_cast<const uint4 *>(rows + static_cast<uint64_t>(row_variant[r]) * pitch + 16ull * w);
	}
	const uint64_t q_lo = q.x | (static_cast<uint64_t>(q.y) << 32), q_hi = q.z | (static_cast<uint64_t>(q.w) << 32);
	const uint16_t *vals = dos.values + dos.val_off[r] + rk;
	uint32_t *out = rec + first + __shfl(rk, 0); // lane 0 holds the tile's first word
	uint32_t base = 0;
	for (uint32_t i = 0;; i++) {
		const uint64_t mask = __ballot(e != 0ull);
		if (mask == 0ull) {
			break;
		}
		const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(mask >> 32),
		                                                      __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(mask), 0u));
		if (e != 0ull) {
			const uint32_t b = static_cast<uint32_t>(__ffsll(static_cast<long long>(e))) - 1u;
			e &= e - 1ull;
			const uint32_t code = static_cast<uint32_t>((b < 32u ? q_lo : q_hi) >> (2u * (b & 31u))) & 3u;
			out[pos] = (static_cast<uint32_t>(vals[i]) << 16) | ((b * 64u + lane) << 3) | code;
		}
		base += static_cast<uint32_t>(__popcll(mask));
	}
}

// plink_score's explicit-entry step over the records (same contract as k_score_dosage_fix, which stays as the
// path for datasets whose records did not fit).  Sixteen waves share a 4096-sample tile in LDS; a wave takes every
// sixteenth variant of the slice and streams that variant's run of the tile: 256 B per wave-instruction, the next
// 512 entries in registers while the current ones are added.  Per 64 entries: one load, ~10 vector instructions,
// one LDS read (the term of the hardcall the dosage replaces) and one or two FP64 LDS atomics without conflicts.
constexpr uint32_t kRecNone = 0xffffffffu; // not a record: values stop at 32768
constexpr uint32_t kRecGroup = 8;          // wave-loads of 64 records in flight per wave

template <bool TRACK>
__global__ __launch_bounds__(1024) void k_score_dosage_records(uint32_t sample_ct, DosageView dos,
                                                                const uint32_t *__restrict__ vlist, uint32_t n_scored,
                                                                uint32_t slice_len, const double *__restrict__ weights,
                                                                uint32_t w_stride, const double *__restrict__ ts,
                                                                const double *__restrict__ lin,
                                                                const uint32_t *__restrict__ ac,
                                                                double *__restrict__ score, uint32_t out_stride,
                                                                double *__restrict__ dosage_sum,
                                                                uint32_t *__restrict__ miss) {
	constexpr uint32_t kWaves = 16;
	constexpr uint32_t kChunk = 128; // variants whose constants are staged in LDS at a time
	__shared__ double s_acc[64 * 64]; // [bit][word]
	__shared__ double s_dsum[TRACK ? 64 * 64 : 1];
	__shared__ uint64_t s_first[kChunk]; // the tile's first record of each variant
	__shared__ uint32_t s_count[kChunk]; // and how many there are (0: nothing to add)
	__shared__ double s_wt[kChunk], s_a[kChunk], s_b[kChunk][4]; // delta = a u + b[call] (k_score_dosage_fix)
	for (uint32_t t = threadIdx.x; t < 64 * 64; t += 64u * kWaves) {
		s_acc[t] = 0.0;
		if (TRACK) {
			s_dsum[t] = 0.0;
		}
	}
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
	const uint32_t w0 = blockIdx.x * 64u; // first word of the tile
	const uint32_t i_begin = blockIdx.y * slice_len;
	const uint32_t i_end = min(i_begin + slice_len, n_scored);
	for (uint32_t base = i_begin; base < i_end; base += kChunk) {
		const uint32_t cnt = min(kChunk, i_end - base);
		__syncthreads();
		if (threadIdx.x < cnt) {
			const uint32_t i = base + threadIdx.x;
			const int32_t r = dos.row_of[vlist[i]];
			uint32_t n = 0;
			uint64_t first = 0;
			if (ac[i] != 0u && r >= 0) { // (skipped by the reference otherwise: nobody observed, or no variance)
				const uint64_t row_first = dos.rec_off[r], row_n = dos.rec_off[r + 1] - row_first;
				const uint64_t rk = static_cast<uint64_t>(r) * dos.words + w0;
				const uint32_t start = dos.rank[rk];
				const uint32_t end = w0 + 64u < dos.words ? dos.rank[rk + 64u] : static_cast<uint32_t>(row_n);
				first = row_first + start;
				n = row_n ? end - start : 0u;
			}
			s_first[threadIdx.x] = first;
			s_count[threadIdx.x] = n;
			s_wt[threadIdx.x] = weights[static_cast<uint64_t>(i) * w_stride];
			const double l0 = lin[4ull * i + 0], l1 = lin[4ull * i + 1], l2 = lin[4ull * i + 2], l3 = lin[4ull * i + 3];
			s_a[threadIdx.x] = l0 * l3 * 0x1p-14;
			const double shift = (l1 - l2) * l3;
#pragma unroll
			for (int c = 0; c < 4; c++) {
				s_b[threadIdx.x][c] = shift - ts[4ull * i + c];
			}
		}
		__syncthreads();
		// this wave's work items: (variant k, group g of 512 records), walked with the next item's records in flight
		auto fetch = [&](uint32_t k, uint32_t g, uint32_t(&dst)[kRecGroup]) {
			const uint32_t n = s_count[k] - g * (64u * kRecGroup);
			const uint32_t *p = dos.rec + s_first[k] + g * (64u * kRecGroup);
#pragma unroll
			for (uint32_t j = 0; j < kRecGroup; j++) {
				const uint32_t idx = 64u * j + lane;
				dst[j] = idx < n ? __builtin_nontemporal_load(p + idx) : kRecNone;
			}
		};
		auto next_variant = [&](uint32_t k) {
			while (k < cnt && s_count[k] == 0u) {
				k += kWaves;
			}
			return k;
		};
		uint32_t k = next_variant(wave), g = 0;
		uint32_t cur[kRecGroup], nxt[kRecGroup];
		if (k < cnt) {
			fetch(k, 0, cur);
		}
		while (k < cnt) {
			uint32_t k2 = k, g2 = g + 1u;
			if (g2 * (64u * kRecGroup) >= s_count[k]) {
				k2 = next_variant(k + kWaves);
				g2 = 0;
			}
			if (k2 < cnt) {
				fetch(k2, g2, nxt);
			}
			const double wt = s_wt[k], a = s_a[k];
			// (the four terms of the variant held in registers and picked by two selects instead of this per-entry
			// LDS read: no faster, 14.8 vs 14.5 ms -- the atomics are what fills the LDS pipe)
			const uint8_t *bk = reinterpret_cast<const uint8_t *>(&s_b[k][0]);
#pragma unroll
			for (uint32_t j = 0; j < kRecGroup; j++) {
				const uint32_t rv = cur[j];
				if (rv != kRecNone) {
					const uint32_t code = rv & 3u;
					const double delta = fma(a, static_cast<double>(rv >> 16), *reinterpret_cast<const double *>(bk + 8u * code));
					const uint32_t at = rv & 0x7ff8u;
					atomicAdd(reinterpret_cast<double *>(reinterpret_cast<uint8_t *>(s_acc) + at), wt * delta);
					if (TRACK) {
						atomicAdd(reinterpret_cast<double *>(reinterpret_cast<uint8_t *>(s_dsum) + at), delta);
					}
					if (miss && code == 3u) { // it has a dosage: not missing after all
						const uint32_t el = at >> 3;
						atomicSub(miss + 64u * (w0 + (el & 63u)) + (el >> 6), 1u);
					}
				}
			}
#pragma unroll
			for (uint32_t j = 0; j < kRecGroup; j++) {
				cur[j] = nxt[j];
			}
			k = k2;
			g = g2;
		}
	}
	__syncthreads();
	for (uint32_t t = threadIdx.x; t < 4096u; t += 64u * kWaves) {
		const uint32_t s = blockIdx.x * 4096u + t;
		const uint32_t at = (t & 63u) * 64u + (t >> 6); // (a 64-way bank conflict, eight times per slice: noise)
		if (s < sample_ct) {
			if (s_acc[at] != 0.0) {
				unsafeAtomicAdd(score + static_cast<uint64_t>(s) * out_stride, s_acc[at]);
			}
			if (TRACK && s_dsum[at] != 0.0) {
				unsafeAtomicAdd(dosage_sum + s, s_dsum[at]);
			}
		}
	}
}

} // namespace

hipError_t LaunchDosageRank(const uint64_t *present, uint32_t rows, uint32_t words, uint32_t *rank,
                            hipStream_t stream) {
	if (rows == 0 || words == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_dosage_rank, dim3(rows), dim3(256), 0, stream, present, words, rank);
	return hipGetLastError();
}

hipError_t LaunchDosageIngest(const DosageIngest &batch, uint32_t first_row, uint32_t n_rows, hipStream_t stream) {
	if (batch.n == 0 || n_rows == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_dosage_locate, dim3(batch.n), dim3(256), 0, stream, batch);
	hipLaunchKernelGGL(k_dosage_offsets, dim3(1), dim3(1024), 0, stream, batch);
	hipLaunchKernelGGL(k_dosage_rank, dim3(n_rows), dim3(256), 0, stream,
	                   batch.present + static_cast<uint64_t>(first_row) * batch.words, batch.words,
	                   batch.rank + static_cast<uint64_t>(first_row) * batch.words);
	hipLaunchKernelGGL(k_dosage_values, dim3(batch.n), dim3(256), 0, stream, batch);
	return hipGetLastError();
}

hipError_t LaunchSynthDosageBits(uint64_t *present, uint32_t rows, uint32_t words, uint32_t sample_ct, uint32_t variant0,
                                 uint64_t seed, double rate, hipStream_t stream) {
	if (rows == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_synth_dosage_bits, dim3(rows), dim3(256), 0, stream, present, words, sample_ct, variant0, seed,
	                   SynthMissThreshold(rate));
	return hipGetLastError();
}

hipError_t LaunchDosageRowTotals(const uint64_t *present, const uint32_t *rank, uint32_t rows, uint32_t words,
                                 uint64_t *totals, hipStream_t stream) {
	if (rows == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_dosage_row_totals, dim3((rows + 255) / 256), dim3(256), 0, stream, present, rank, words, rows,
	                   totals);
	return hipGetLastError();
}

hipError_t LaunchSynthDosageValues(const uint64_t *present, const uint32_t *rank, const uint64_t *val_off,
                                   uint16_t *values, uint32_t rows, uint32_t words, uint32_t sample_ct,
                                   uint32_t variant0, uint64_t seed, hipStream_t stream) {
	if (rows == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_synth_dosage_values, dim3(rows), dim3(256), 0, stream, present, rank, val_off, values, words,
	                   sample_ct, variant0, seed);
	return hipGetLastError();
}

hipError_t LaunchDosageSums(const RowView &view, const DosageView &dos, uint32_t v0, const uint32_t *vlist,
                            uint32_t n_var, const uint64_t *include, uint64_t *out, hipStream_t stream) {
	if (n_var == 0) {
		return hipSuccess;
	}
	if (include) {
		hipLaunchKernelGGL(k_dosage_sums<true>, dim3(n_var), dim3(256), 0, stream, view.rows, view.pitch, view.sample_ct,
		                   dos, v0, vlist, include, out);
	} else {
		hipLaunchKernelGGL(k_dosage_sums<false>, dim3(n_var), dim3(256), 0, stream, view.rows, view.pitch, view.sample_ct,
		                   dos, v0, vlist, include, out);
	}
	return hipGetLastError();
}

hipError_t LaunchDosageUnpack(const RowView &view, const DosageView &dos, uint32_t v0, const uint32_t *vlist,
                              uint32_t n_var, const uint32_t *sel, uint32_t n_out, double *out, uint64_t out_stride,
                              hipStream_t stream) {
	if (n_var == 0 || n_out == 0) {
		return hipSuccess;
	}
	for (uint32_t done = 0; done < n_var; done += 65535u) { // grid.y limit
		const uint32_t n = min(65535u, n_var - done);
		hipLaunchKernelGGL(k_dosage_unpack, dim3((n_out + 255) / 256, n), dim3(256), 0, stream, view.rows, view.pitch, dos,
		                   v0 + done, vlist ? vlist + done : nullptr, sel, n_out, out + static_cast<uint64_t>(done) * out_stride,
		                   out_stride);
	}
	return hipGetLastError();
}

hipError_t LaunchDosageUnpackTransposed(const RowView &view, const DosageView &dos, const uint32_t *vlist, uint32_t n_var,
                                        const uint32_t *sel, uint32_t k_first, uint32_t k_count, double *out,
                                        uint64_t out_stride, hipStream_t stream) {
	if (n_var == 0 || k_count == 0) {
		return hipSuccess;
	}
	for (uint32_t done = 0; done < k_count; done += 65535u) { // grid.y limit
		const uint32_t n = min(65535u, k_count - done);
		hipLaunchKernelGGL(k_dosage_unpack_transposed, dim3((n_var + 255) / 256, n), dim3(256), 0, stream, view.rows,
		                   view.pitch, dos, vlist, n_var, sel, k_first + done, out + static_cast<uint64_t>(done) * out_stride,
		                   out_stride);
	}
	return hipGetLastError();
}

hipError_t LaunchScoreTablesDosage(const uint64_t *sums, const uint8_t *flip, uint32_t n_scored, int mode, double *ts,
                                   double *td, double *lin, uint32_t *ac, hipStream_t stream) {
	if (n_scored == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_score_tables_dosage, dim3((n_scored + 255) / 256), dim3(256), 0, stream, sums, flip, n_scored,
	                   mode, ts, td, lin, ac);
	return hipGetLastError();
}

hipError_t LaunchScoreDosage(const RowView &view, const DosageView &dos, const uint32_t *vlist, uint32_t n_scored,
                             const double *weights, uint32_t w_stride, uint32_t n_cols, const double *ts,
                             const double *lin, const uint32_t *ac, int mode, double *score, uint32_t out_stride,
                             double *dosage_sum, uint32_t *miss, hipStream_t stream) {
	if (n_scored == 0 || n_cols == 0) {
		return hipSuccess;
	}
	const uint32_t sample_blocks = (view.sample_ct + 1023) / 1024; // four samples per lane
	const uint32_t want_slices = (2048 + sample_blocks - 1) / sample_blocks;
	uint32_t slice_len = (n_scored + want_slices - 1) / want_slices;
	slice_len = ((slice_len + 63) / 64) * 64;
	uint32_t slices = (n_scored + slice_len - 1) / slice_len;
	if (slices > 65535u) {
		slices = 65535u;
		slice_len = ((n_scored + slices - 1) / slices + 63) / 64 * 64;
		slices = (n_scored + slice_len - 1) / slice_len;
	}
	// weight columns four at a time; the dosage sum and the missing tally ride with the first pass
	for (uint32_t c0 = 0; c0 < n_cols; c0 += 4) {
		const uint32_t cols = min(4u, n_cols - c0);
		double *dsum = c0 == 0 ? dosage_sum : nullptr;
		uint32_t *ms = c0 == 0 ? miss : nullptr;
		if (cols == 1) {
			hipLaunchKernelGGL((k_score_dosage<1>), dim3(sample_blocks, slices), dim3(256), 0, stream, view.rows, view.pitch,
			                   view.sample_ct, dos, vlist, n_scored, slice_len, weights + c0, w_stride, cols, out_stride, ts,
			                   lin, ac, mode, score + c0, dsum, ms);
		} else {
			hipLaunchKernelGGL((k_score_dosage<4>), dim3(sample_blocks, slices), dim3(256), 0, stream, view.rows, view.pitch,
			                   view.sample_ct, dos, vlist, n_scored, slice_len, weights + c0, w_stride, cols, out_stride, ts,
			                   lin, ac, mode, score + c0, dsum, ms);
		}
	}
	return hipGetLastError();
}

} // namespace pgh

namespace pgh {

hipError_t LaunchScoreDosageFix(const RowView &view, const DosageView &dos, const uint32_t *vlist, uint32_t n_scored,
                                const double *weights, uint32_t w_stride, const double *ts, const double *lin,
                                const uint32_t *ac, double *score, uint32_t out_stride, double *dosage_sum,
                                uint32_t *miss, hipStream_t stream) {
	if (n_scored == 0) {
		return hipSuccess;
	}
	const uint32_t tiles = (dos.words + 63) / 64;
	const uint32_t want_slices = (2048 + tiles - 1) / tiles;
	uint32_t slice_len = (n_scored + want_slices - 1) / want_slices;
	slice_len = ((slice_len + 127) / 128) * 128;
	uint32_t slices = (n_scored + slice_len - 1) / slice_len;
	if (slices > 65535u) {
		slices = 65535u;
		slice_len = ((n_scored + slices - 1) / slices + 127) / 128 * 128;
		slices = (n_scored + slice_len - 1) / slice_len;
	}
	if (dosage_sum) {
		hipLaunchKernelGGL(k_score_dosage_fix<true>, dim3(tiles, slices), dim3(1024), 0, stream, view.rows, view.pitch,
		                   view.sample_ct, dos, vlist, n_scored, slice_len, weights, w_stride, ts, lin, ac, score, out_stride,
		                   dosage_sum, miss);
	} else {
		hipLaunchKernelGGL(k_score_dosage_fix<false>, dim3(tiles, slices), dim3(1024), 0, stream, view.rows, view.pitch,
		                   view.sample_ct, dos, vlist, n_scored, slice_len, weights, w_stride, ts, lin, ac, score, out_stride,
		                   dosage_sum, miss);
	}
	return hipGetLastError();
}

hipError_t LaunchDosageRecords(const RowView &view, const DosageView &dos, uint32_t rows, const uint32_t *row_variant,
                               const uint64_t *rec_off, uint32_t *rec, hipStream_t stream) {
	if (rows == 0 || dos.words == 0) {
		return hipSuccess;
	}
	const uint32_t tiles = (dos.words + 63) / 64;
	for (uint32_t done = 0; done < rows; done += 65535u) { // grid.y limit
		const uint32_t n = min(65535u, rows - done);
		DosageView part = dos;
		part.present += static_cast<uint64_t>(done) * dos.words;
		part.rank += static_cast<uint64_t>(done) * dos.words;
		part.val_off += done;
		hipLaunchKernelGGL(k_dosage_records, dim3((tiles + 3) / 4, n), dim3(256), 0, stream, view.rows, view.pitch, part,
		                   row_variant + done, rec_off + done, rec);
	}
	return hipGetLastError();
}

hipError_t LaunchScoreDosageRecords(const RowView &view, const DosageView &dos, const uint32_t *vlist, uint32_t n_scored,
                                    const double *weights, uint32_t w_stride, const double *ts, const double *lin,
                                    const uint32_t *ac, double *score, uint32_t out_stride, double *dosage_sum,
                                    uint32_t *miss, hipStream_t stream) {
	if (n_scored == 0) {
		return hipSuccess;
	}
	if (!dos.rec || !dos.rec_off) {
		return hipErrorInvalidValue;
	}
	const uint32_t tiles = (dos.words + 63) / 64;
	const uint32_t want_slices = (2048 + tiles - 1) / tiles;
	uint32_t slice_len = (n_scored + want_slices - 1) / want_slices;
	slice_len = ((slice_len + 127) / 128) * 128;
	uint32_t slices = (n_scored + slice_len - 1) / slice_len;
	if (slices > 65535u) {
		slices = 65535u;
		slice_len = ((n_scored + slices - 1) / slices + 127) / 128 * 128;
		slices = (n_scored + slice_len - 1) / slice_len;
	}
	if (dosage_sum) {
		hipLaunchKernelGGL(k_score_dosage_records<true>, dim3(tiles, slices), dim3(1024), 0, stream, view.sample_ct, dos,
		                   vlist, n_scored, slice_len, weights, w_stride, ts, lin, ac, score, out_stride, dosage_sum, miss);
	} else {
		hipLaunchKernelGGL(k_score_dosage_records<false>, dim3(tiles, slices), dim3(1024), 0, stream, view.sample_ct, dos,
		                   vlist, n_scored, slice_len, weights, w_stride, ts, lin, ac, score, out_stride, dosage_sum, miss);
	}
	return hipGetLastError();
}

hipError_t LaunchScoreDosageFull(const RowView &view, const DosageView &dos, const uint32_t *vlist, uint32_t n_scored,
                                 const double *weights, uint32_t w_stride, uint32_t n_cols, const double *lin,
                                 const uint32_t *ac, int mode, double *score, uint32_t out_stride, double *dosage_sum,
                                 hipStream_t stream) {
	if (n_scored == 0 || n_cols == 0) {
		return hipSuccess;
	}
	const uint32_t sample_blocks = (view.sample_ct + 1023) / 1024; // four samples per lane
	const uint32_t want_slices = (2048 + sample_blocks - 1) / sample_blocks;
	uint32_t slice_len = (n_scored + want_slices - 1) / want_slices;
	slice_len = ((slice_len + 63) / 64) * 64;
	uint32_t slices = (n_scored + slice_len - 1) / slice_len;
	if (slices > 65535u) {
		slices = 65535u;
		slice_len = ((n_scored + slices - 1) / slices + 63) / 64 * 64;
		slices = (n_scored + slice_len - 1) / slice_len;
	}
	for (uint32_t c0 = 0; c0 < n_cols; c0 += 4) {
		const uint32_t cols = std::min(4u, n_cols - c0);
		double *dsum = c0 == 0 ? dosage_sum : nullptr;
		if (cols == 1) {
			hipLaunchKernelGGL((k_score_dosage_full<1>), dim3(sample_blocks, slices), dim3(256), 0, stream, view.sample_ct, dos,
			                   vlist, n_scored, slice_len, weights + c0, w_stride, cols, out_stride, lin, ac, mode, score + c0,
			                   dsum);
		} else {
			hipLaunchKernelGGL((k_score_dosage_full<4>), dim3(sample_blocks, slices), dim3(256), 0, stream, view.sample_ct, dos,
			                   vlist, n_scored, slice_len, weights + c0, w_stride, cols, out_stride, lin, ac, mode, score + c0,
			                   dsum);
		}
	}
	return hipGetLastError();
}

} // namespace pgh

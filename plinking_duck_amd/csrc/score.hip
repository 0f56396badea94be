// score.hip -- plink_score's contraction of the packed rows with weight columns (gfx950):
// per-variant contribution tables, the pair-table one-column kernel, and the FP64-MFMA
// table-accumulate kernel that plink_pca's Step B / phase 3 share.
//
// Data layout: the genotype matrix is variant-major; row v holds ceil(N/4)
// bytes of packed 2-bit calls (00 hom-ref, 01 het, 10 hom-alt, 11 missing;
// sample s in bits 2*(s%4) of byte s/4) followed by zero bytes up to `pitch`
// (a multiple of 16, so every row can be streamed as whole 16-byte lanes and
// the pad decodes as hom-ref, which every kernel cancels against N).
#include "device_utils.hpp"
#include "kernels.hpp"

namespace pgh {

namespace {

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
// TRACK: the dosage half of an entry is wanted too (16-byte lookups and two adds per pair; without it
// 8-byte lookups of the same table and one add -- half the LDS cycles and half the FP64 adds)
template <int BUF, bool TRACK>
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
	if (TRACK) {                                                                                                       \
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
	} else {                                                                                                           \
		acc_s[4 * B] += *reinterpret_cast<const double *>(tab + ByteAnd##B(even_lo, kF0));                            \
		acc_s[4 * B + 1] += *reinterpret_cast<const double *>(tab + ByteAnd##B(odd_lo, kF0));                         \
		acc_s[4 * B + 2] += *reinterpret_cast<const double *>(tab + ByteAnd##B(even, kF0));                           \
		acc_s[4 * B + 3] += *reinterpret_cast<const double *>(tab + ByteAnd##B(odd, kF0));                            \
	}
		PGH_LOOKUP(0)
		PGH_LOOKUP(1)
		PGH_LOOKUP(2)
		PGH_LOOKUP(3)
#undef PGH_LOOKUP
	}
}

template <bool TRACK>
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
		GemvPairsStage<0, TRACK>(s_tab, w_a, acc_s, acc_d);
		__syncthreads();
		if (!more_b) {
			break;
		}
		if (base + 2 * kStage < i_end) {
			load_words(base + 2 * kStage, w_a);
			build(base + 2 * kStage, 0);
		}
		GemvPairsStage<1, TRACK>(s_tab, w_b, acc_s, acc_d);
		__syncthreads();
	}
	if (live) {
#pragma unroll
		for (int j = 0; j < 16; j++) {
			const uint32_t s0 = d * 16u + j;
			if (s0 < sample_ct) {
				unsafeAtomicAdd(score + static_cast<uint64_t>(s0) * out_stride, acc_s[j]);
				if (TRACK && dosage_sum) {
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
	// double-buffered stage of 64 variants: their tables, their weights and the workgroup's
	// 64 bytes (256 samples) of each of their rows.  The whole stage is fetched into registers
	// while the previous one is being multiplied, so the multiply loop touches no global memory
	// and an HBM miss has a full stage (16 groups x 4 MFMAs per wave) to land.
	__shared__ double s_ts[2][kStage][4];
	__shared__ double s_w[2][kStage][kCols];
	__shared__ uint4 s_geno[2][kStage][4];
	const uint32_t lane = threadIdx.x & 63u;
	const uint32_t wave = threadIdx.x >> 6;
	const uint32_t li = lane & 15u; // sample within tile (A), column within tile (B, D)
	const uint32_t lk = lane >> 4;  // variant within the group of 4 (A, B); row group (D)
	const uint32_t sample_base = (blockIdx.x * 4u + wave) * 64u;
	const bool wave_live = sample_base < sample_ct; // wave-uniform
	const uint32_t shift = 2u * li;
	const uint32_t i_begin = blockIdx.y * slice_len;
	const uint32_t i_end = min(i_begin + slice_len, n_var);
	// staging role of this thread: 16-byte piece (tid & 3) -- the 64 samples of wave (tid & 3) --
	// of stage variant (tid >> 2)
	const uint32_t piece = threadIdx.x & 3u;
	const bool piece_live = (blockIdx.x * 4u + piece) * 64u < sample_ct;
	const uint8_t *piece_ptr = rows + (static_cast<uint64_t>(blockIdx.x) * 4u + piece) * 16u;

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

	double r_ts = 0.0;
	double r_w[kWPerThread];
	uint4 r_geno = make_uint4(0, 0, 0, 0);
	auto fetch = [&](uint32_t base) {
		const uint32_t cnt = min(kStage, i_end - base);
		const uint32_t k = threadIdx.x; // kStage * 4 == 256
		r_ts = (k >> 2) < cnt ? ts[4 * static_cast<uint64_t>(base) + k] : 0.0;
		r_geno = make_uint4(0, 0, 0, 0);
		if ((k >> 2) < cnt && piece_live) {
			r_geno = *reinterpret_cast<const uint4 *>(piece_ptr + static_cast<uint64_t>(vlist[base + (k >> 2)]) * pitch);
		}
#pragma unroll
		for (uint32_t j = 0; j < kWPerThread; j++) {
			const uint32_t e = threadIdx.x + 256u * j;
			const uint32_t v = e / kCols, c = e % kCols;
			r_w[j] = (v < cnt && c < n_cols) ? weights[static_cast<uint64_t>(base + v) * w_stride + c] : 0.0;
		}
	};
	auto commit = [&](uint32_t buf) {
		s_ts[buf][threadIdx.x >> 2][threadIdx.x & 3] = r_ts;
		s_geno[buf][threadIdx.x >> 2][threadIdx.x & 3] = r_geno;
#pragma unroll
		for (uint32_t j = 0; j < kWPerThread; j++) {
			const uint32_t e = threadIdx.x + 256u * j;
			s_w[buf][e / kCols][e % kCols] = r_w[j];
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
			// operands of group g: the 16 bytes of variant 4g + lk that hold this wave's 64 samples
			// (one broadcast LDS read per 16 lanes), four table lookups, the weight entries
			auto operands = [&](uint32_t g, double a[4], double b[NCT + kQ]) {
				const uint32_t k = g * 4u + lk;
				const uint4 w = s_geno[buf][k][wave];
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
					a[t] = s_ts[buf][k][__builtin_amdgcn_ubfe(wt[t], shift, 2u)]; // one v_bfe_u32, not shift + and
				}
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
				// full stage, fully unrolled: while group g is on the matrix pipe the operands of
				// g + 1 are being read out of LDS (double-buffered in registers)
				double a[2][4], b[2][NCT + kQ];
				operands(0, a[0], b[0]);
#pragma unroll
				for (uint32_t g4 = 0; g4 < kStage / 4; g4++) {
					if (g4 + 1 < kStage / 4) {
						operands(g4 + 1u, a[(g4 + 1) % 2], b[(g4 + 1) % 2]);
					}
					multiply(a[g4 % 2], b[g4 % 2]);
				}
			} else {
				// ragged last stage of a slice
				for (uint32_t g4 = 0; g4 < groups; g4++) {
					double a[4], b[NCT + kQ];
					operands(g4, a, b);
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

} // namespace

// ---------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------

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
		if (td && dosage_sum) {
			hipLaunchKernelGGL(k_score_gemv_pairs<true>, dim3(col_blocks, slices), dim3(256), 0, stream, view.rows,
			                   view.pitch, view.sample_ct, vlist, n_var, slice_len, weights, w_stride, ts, td, out,
			                   out_stride, dosage_sum);
		} else {
			hipLaunchKernelGGL(k_score_gemv_pairs<false>, dim3(col_blocks, slices), dim3(256), 0, stream, view.rows,
			                   view.pitch, view.sample_ct, vlist, n_var, slice_len, weights, w_stride, ts, nullptr, out,
			                   out_stride, nullptr);
		}
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

} // namespace pgh

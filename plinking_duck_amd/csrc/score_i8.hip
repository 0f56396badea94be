// score_i8.hip -- plink_score's genotype x weight contraction on the int8 matrix cores (gfx950),
// exact in fixed point.
//
//   score[s][c] = sum_v  W[v][c] * T_v[g(v,s)]          (src/plink_score.cpp:575-654)
//
// T_v is affine in the call for g = 0, 1, 2 in every mode of the reference (dosage g or 2 - g, centred or
// not) and takes its own value for a missing call (the variant's mean, or nothing).  With the raw 2-bit
// code c(v,s) in {0,1,2,3} and miss = [c == 3]:
//
//   W T_v[g] = W t0  +  c * (W d)  +  miss * W (t3 - t0 - 3 d),          d = t1 - t0
//              -----     ---------     --------------------------
//              K0[c]      alpha          beta
//
// so the whole sum is two integer-matrix x real-vector products: the code plane and the missing plane.
// The real coefficients are cut into signed base-256 digits of a fixed-point number (one power-of-two
// scale per output column -- the Ozaki splitting), the planes are int8 matrices, and
// v_mfma_i32_16x16x64_i8 accumulates every digit column EXACTLY in int32.  Sixteen digit columns ride in
// one instruction: one weight column (7 digits, 54 bits below the column's largest coefficient), the
// unweighted NAMED_ALLELE_DOSAGE_SUM (5 digits) and the count of missing calls (ALLELE_CT) fit in a single
// 16-column tile, so the reference's one-column SQL contract costs two matrix instructions per
// 64 variants x 16 samples and reads every byte of the matrix once; sixteen weight columns (BASELINE
// config 4) take eight tiles.  Rounding happens once, where a coefficient is cut to 54 bits; the sums
// themselves are exact, i.e. closer to the real-number result than a double accumulation in any order.
//
// Kernel shape (k_score_i8<NT, TS>): a workgroup of four waves (eight with TS = 4) owns 16*TS samples per wave and walks a slice of
// the scored variants 64 at a time.  All lanes fetch the tile's packed bytes with 16-byte loads (whole
// 16*TS-byte row segments), park them in LDS, and each lane reads back the 16 variants of its matrix k-group
// for one 4-byte word of samples.  Four 4x4 transposes on 2-bit fields (16 bitfield ops) turn four such words
// into bytes that hold one sample's four calls; a multiply and a shift-or spread such a byte into four int8
// codes (three ops per four calls), a byte permute derives the missing plane (one op), and the two matrix
// instructions consume them against the tile's digit bytes -- ~1.25 vector ops per call, under the HBM time.
#include "device_utils.hpp"
#include "score_i8.hpp"

#include <type_traits>

namespace pgh {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

constexpr uint32_t kTileVariants = 64; // K of v_mfma_i32_16x16x64_i8

// ---------------------------------------------------------------------------
// preparation: scales, digits, constants (once per plan)
// ---------------------------------------------------------------------------

struct Coef {
	double alpha, beta, k0;
};

// coefficients of target column t at variant i: t < n_cols a weight column, t == n_cols the dosage sum
__device__ __forceinline__ Coef CoefOf(uint32_t i, uint32_t t, uint32_t n_cols, const double *__restrict__ weights,
                                       uint32_t w_stride, const double *__restrict__ ts,
                                       const double *__restrict__ td, int table_mode) {
	const double w = (t < n_cols) ? weights[static_cast<uint64_t>(i) * w_stride + t] : 1.0;
	if (table_mode != kI8Tables) {
		// bare planes (plink_pca's X G1 over the transposed matrix): the code itself, or the missing indicator
		Coef c;
		c.alpha = table_mode == kI8CodePlane ? w : 0.0;
		c.beta = table_mode == kI8MissingPlane ? w : 0.0;
		c.k0 = 0.0;
		return c;
	}
	const double *tab = (t < n_cols) ? ts : td;
	const double t0 = tab[4ull * i], t1 = tab[4ull * i + 1], t3 = tab[4ull * i + 3];
	const double d = t1 - t0;
	Coef c;
	c.alpha = w * d;
	c.beta = w * (t3 - t0 - 3.0 * d);
	c.k0 = w * t0;
	return c;
}

__device__ __forceinline__ double BlockMax(double v, double *s_red) {
	for (int off = 32; off > 0; off >>= 1) {
		v = fmax(v, __shfl_xor(v, off, 64));
	}
	if ((threadIdx.x & 63u) == 0) {
		s_red[threadIdx.x >> 6] = v;
	}
	__syncthreads();
	v = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
	__syncthreads();
	return v;
}

__device__ __forceinline__ double BlockSum(double v, double *s_red) {
	for (int off = 32; off > 0; off >>= 1) {
		v += __shfl_xor(v, off, 64);
	}
	if ((threadIdx.x & 63u) == 0) {
		s_red[threadIdx.x >> 6] = v;
	}
	__syncthreads();
	v = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
	__syncthreads();
	return v;
}

// colmax[t] = max over variants of max(|alpha|, |beta|) (as the bit pattern of a non-negative double:
// monotone under integer max); k0[t] = sum over variants of W t0.  grid.y = target column.
__global__ __launch_bounds__(256) void k_i8_ranges(uint32_t n_var, uint32_t n_cols, const double *__restrict__ weights,
                                                   uint32_t w_stride, const double *__restrict__ ts,
                                                   const double *__restrict__ td, int table_mode,
                                                   unsigned long long *__restrict__ colmax,
                                                   double *__restrict__ k0) {
	__shared__ double s_red[4];
	const uint32_t t = blockIdx.y;
	double mx = 0.0, sum = 0.0;
	for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n_var; i += gridDim.x * 256u) {
		const Coef c = CoefOf(i, t, n_cols, weights, w_stride, ts, td, table_mode);
		mx = fmax(mx, fmax(fabs(c.alpha), fabs(c.beta)));
		sum += c.k0;
	}
	mx = BlockMax(mx, s_red);
	sum = BlockSum(sum, s_red);
	if (threadIdx.x == 0) {
		atomicMax(colmax + t, static_cast<unsigned long long>(__double_as_longlong(mx)));
		unsafeAtomicAdd(k0 + t, sum);
	}
}

// Digit columns: weight column c owns columns [7c, 7c+7); with `extras` (plink_score) the dosage sum takes the
// next 5 and the missing count 1.
__host__ __device__ constexpr uint32_t DigitColumns(uint32_t n_cols, bool extras) {
	return kI8WeightDigits * n_cols + (extras ? kI8DosageDigits + 1u : 0u);
}

// mult[J] = value of one unit of digit column J; target[J] = output column (n_cols: dosage sum,
// n_cols + 1: missing count, 0xffffffff: unused)
__global__ void k_i8_columns(uint32_t n_cols, int extras, uint32_t n_tiles16, const unsigned long long *__restrict__ colmax,
                             double *__restrict__ scale_exp, double *__restrict__ mult,
                             uint32_t *__restrict__ target) {
	const uint32_t J = blockIdx.x * blockDim.x + threadIdx.x;
	if (J >= n_tiles16 * 16u) {
		return;
	}
	const uint32_t used = DigitColumns(n_cols, extras != 0);
	if (J >= used) {
		mult[J] = 0.0;
		target[J] = 0xffffffffu;
		return;
	}
	uint32_t t, d, digits;
	if (J < kI8WeightDigits * n_cols) {
		t = J / kI8WeightDigits;
		d = J % kI8WeightDigits;
		digits = kI8WeightDigits;
	} else if (J < kI8WeightDigits * n_cols + kI8DosageDigits) {
		t = n_cols;
		d = J - kI8WeightDigits * n_cols;
		digits = kI8DosageDigits;
	} else {
		mult[J] = 1.0;
		target[J] = n_cols + 1u;
		return;
	}
	// the column's coefficients are cut at 2^e / 2^(8 digits - 2), 2^e >= the largest of them: the top digit
	// then stays within +-65
	const double mx = __longlong_as_double(static_cast<long long>(colmax[t]));
	int e = 0;
	if (mx > 0.0) {
		(void)frexp(mx, &e); // mx = f * 2^e, f in [0.5, 1)
	}
	if (d == 0) {
		scale_exp[t] = static_cast<double>(e);
	}
	mult[J] = ldexp(1.0, e - static_cast<int>(8u * digits - 2u) + static_cast<int>(8u * d));
	target[J] = t;
}

// signed base-256 digits of x (|x| <= 2^(8n-2)), least significant first
__device__ __forceinline__ void Digits(long long x, uint32_t n, int8_t *out) {
	for (uint32_t d = 0; d < n; d++) {
		const long long q = ((x + 128) & 255) - 128;
		out[d] = static_cast<int8_t>(q);
		x = (x - q) >> 8;
	}
}

// bmat layout: [tile][plane][tile16][k-group g][digit column x][16 variants of the group] int8, so that a
// matrix-instruction lane (x, g) reads its 16 operand bytes as one 16-byte vector.
__device__ __forceinline__ uint64_t BmatOffset(uint32_t i, uint32_t plane, uint32_t J, uint32_t n_tiles16) {
	const uint32_t tile = i / kTileVariants, k = i % kTileVariants;
	return ((((static_cast<uint64_t>(tile) * 2u + plane) * n_tiles16 + J / 16u) * 4u + k / 16u) * 16u + J % 16u) * 16u +
	       k % 16u;
}

// one thread per (variant, target column); rows past n_var (tile padding) keep the zeros of the memset
__global__ __launch_bounds__(256) void k_i8_digits(uint32_t n_var, uint32_t n_cols, uint32_t n_tiles16,
                                                   const double *__restrict__ weights, uint32_t w_stride,
                                                   const double *__restrict__ ts, const double *__restrict__ td,
                                                   const uint32_t *__restrict__ ac, int count_missing, int table_mode,
                                                   const double *__restrict__ scale_exp, int8_t *__restrict__ bmat) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	const uint32_t t = blockIdx.y; // 0..n_cols: weight columns, dosage; n_cols + 1: the missing count
	if (i >= n_var) {
		return;
	}
	if (t == n_cols + 1u) {
		// ALLELE_CT bookkeeping (src/plink_score.cpp:632-651): variants whose missing calls take 2 alleles
		// away are those that count 2 for a present call and nothing for a missing one
		const uint32_t a = ac[i];
		const bool counts = count_missing && (a & 0xffu) != 0u && ((a >> 8) & 0xffu) == 0u;
		bmat[BmatOffset(i, 1, DigitColumns(n_cols, true) - 1u, n_tiles16)] = counts ? 1 : 0;
		return;
	}
	const Coef c = CoefOf(i, t, n_cols, weights, w_stride, ts, td, table_mode);
	const uint32_t digits = t < n_cols ? kI8WeightDigits : kI8DosageDigits;
	const uint32_t J0 = t < n_cols ? kI8WeightDigits * t : kI8WeightDigits * n_cols;
	const int e = static_cast<int>(scale_exp[t]);
	const int shift = static_cast<int>(8u * digits - 2u) - e;
	int8_t qa[8], qb[8];
	Digits(llrint(ldexp(c.alpha, shift)), digits, qa);
	Digits(llrint(ldexp(c.beta, shift)), digits, qb);
	for (uint32_t d = 0; d < digits; d++) {
		bmat[BmatOffset(i, 0, J0 + d, n_tiles16)] = qa[d];
		bmat[BmatOffset(i, 1, J0 + d, n_tiles16)] = qb[d];
	}
}

// ---------------------------------------------------------------------------
// the contraction
// ---------------------------------------------------------------------------

// NT 16-column digit tiles; TS samples per lane (16, 8 or 4): accumulators = TS * NT * 4 registers.  A workgroup is
// kWaves waves, each owning 16 TS samples, i.e. a stripe of 4 TS kWaves bytes of every row.  Eight waves (one
// workgroup per CU instead of two of four) where TS = 4: the stripe is then 128 bytes -- a whole cache line per row
// and DMA lane group instead of half of one -- and a tile's digit bytes are fetched once per 512 samples instead of
// once per 256 (the same 8 columns with 64-byte stripes instead of 128-byte ones: 49.1 vs 41.7 ms).
template <int NT, int TS>
struct I8Shape {
	static constexpr uint32_t kWaves = TS == 4 ? 8u : 4u;
	static constexpr uint32_t kThreads = 64u * kWaves;
	static constexpr uint32_t kRowBytes = 4u * TS * kWaves;    // bytes of one row that a workgroup owns
	static constexpr uint32_t kChunksPerRow = kRowBytes / 16u; // 16-byte chunks
	static constexpr uint32_t kChunksPerThread = TS / 4;       // 64 rows * kChunksPerRow chunks / kThreads
	static constexpr uint32_t kWordsPerRow = kRowBytes / 4u;
	static constexpr uint32_t kSwizzleChunks = kWordsPerRow >= 32 ? 4u : kWordsPerRow / 8u; // XOR for odd k-groups
	static constexpr uint32_t kGenoBytes = kTileVariants * kRowBytes;
	static constexpr uint32_t kBBytes = 2u * NT * 1024u;
	static constexpr uint32_t kBChunks = kBBytes / 16u; // 128 NT
	static constexpr uint32_t kSamplesPerGroup = 16u * TS * kWaves;
};

// LDS-DMA (global_load_lds_*): 64 lanes x 16 (4) bytes from per-lane global addresses to consecutive LDS bytes
// starting at the wave-uniform LDS address in M0.  Issued from inline asm so that the compiler's wait-count
// bookkeeping does not see them: with the builtin form hipcc drains vmcnt to zero in front of every LDS read that
// might alias a pending DMA, which is every one here.  Completion is counted by hand (PGH_WAIT_VM below).
__device__ __forceinline__ void Glds16Stream(const void *gsrc, uint32_t lds_dst) {
	uint32_t keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep)
	             : "v"(gsrc), "s"(lds_dst)
	             : "memory");
}
__device__ __forceinline__ void Glds16(const void *gsrc, uint32_t lds_dst) {
	uint32_t keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep)
	             : "v"(gsrc), "s"(lds_dst)
	             : "memory");
}
__device__ __forceinline__ void Glds4(const void *gsrc, uint32_t lds_dst) {
	uint32_t keep;
	asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
	             : "=&s"(keep)
	             : "v"(gsrc), "s"(lds_dst)
	             : "memory");
}
__device__ __forceinline__ uint32_t LdsAddress(const void *p) {
	return static_cast<uint32_t>(reinterpret_cast<uintptr_t>((const __attribute__((address_space(3))) void *)p));
}

// LDS slots per workgroup: the tile being multiplied + the ones on their way from HBM.  Four where the workgroups
// of a CU (two of four waves, or one of eight) still fit its 160 KB, else three (the many-column shapes are
// matrix-bound: a trip is long).
// Which shapes run the software-pipelined loop (k_score_i8): the matrix-bound ones, where a second set of operand
// registers still fits next to the accumulators (both planes, ten tiles, row-major words: 34 registers spilled).
template <int NT, int TS, int PLANES, bool TILED>
constexpr bool I8Pipelined() {
	return TS == 4 && (TILED || PLANES != 3 || NT <= 9);
}

template <int NT, int TS, int PLANES, bool TILED>
constexpr uint32_t RingSlots() {
	using S = I8Shape<NT, TS>;
	if (I8Pipelined<NT, TS, PLANES, TILED>()) {
		// the software-pipelined loop reads TWO landed tiles per trip (this tile's digit bytes, the next tile's
		// genotype words): a fifth slot keeps three tiles on their way as before (<= 5 x 28 KB + row numbers)
#ifdef PGH_I8_RING
		return PGH_I8_RING * (S::kGenoBytes + S::kBBytes) + 4096u <= 160u * 1024u ? PGH_I8_RING : 5u;
#else
		return 5u;
#endif
	}
	return 4u * (S::kGenoBytes + S::kBBytes) + 4096u <= (S::kWaves == 8 ? 160u : 80u) * 1024u ? 4u : 3u;
}

// PLANES: 3 = code plane and missing plane (plink_score, plink_pca's X^T Y); 1 / 2 = one of them alone (the two
// products of plink_pca's X G1, whose per-variant normalisation is applied afterwards).
//
// Knock-out builds (tools/i8_experiment.sh, profiles/r02_i8_knockout.txt): -DPGH_I8_NO_BUILD takes the operand
// building out of the loop, -DPGH_I8_NO_DMA / _NO_DMA_G / _NO_DMA_B the LDS-DMA (all of it / the genotype piece /
// the digit pieces) after the first tiles.  Results are then wrong; only the launch time is read.
//
// TILED: `rows` is the tile-major copy of the matrix (k_i8_tile_major below) -- for every (64-variant tile, sample
// group) the 8 KB the ring slot wants, contiguous and already swizzled -- instead of the variant-major rows.  Every
// genotype DMA instruction then reads ONE contiguous kilobyte instead of 64 bytes of each of 16 rows (128 of 8
// with eight waves), which was most of what kept the matrix pipe of the many-column shapes a third idle
// (profiles/r02_i8_knockout.txt: the single genotype piece cost 17 of the DMA's 23 ms, a contiguous one 4.5),
// and the tile's row numbers need not travel at all.  TS = 4 shapes only (five digit tiles and more).
template <int NT, int TS, int PLANES, bool TILED>
__global__ __launch_bounds__((I8Shape<NT, TS>::kThreads)) void k_score_i8(const uint8_t *__restrict__ rows, uint64_t pitch, uint32_t sample_ct,
                                                  const uint32_t *__restrict__ rowidx, uint32_t n_tiles,
                                                  uint32_t tiles_per_slice, const int8_t *__restrict__ bmat,
                                                  const double *__restrict__ mult,
                                                  const uint32_t *__restrict__ target, uint32_t n_cols,
                                                  double *__restrict__ score, uint32_t out_stride,
                                                  double *__restrict__ dosage_sum,
                                                  uint32_t *__restrict__ missing_ct) {
	using S = I8Shape<NT, TS>;
	constexpr uint32_t kRing = RingSlots<NT, TS, PLANES, TILED>();
	constexpr uint32_t kSlotBytes = S::kGenoBytes + S::kBBytes;
	__shared__ __attribute__((aligned(16))) uint8_t s_ring[kRing][kSlotBytes];
	const uint32_t tid = threadIdx.x;
	const uint32_t lane = tid & 63u, wave = tid >> 6;
	const uint32_t x = lane & 15u, g = lane >> 4;
	const uint32_t tile_begin = blockIdx.y * tiles_per_slice;
	const uint32_t tile_end = min(tile_begin + tiles_per_slice, n_tiles);
	if (tile_begin >= tile_end) {
		return;
	}
	// Which sample group this workgroup takes.  With TS = 4 a stripe is 64 bytes of every row, half a 128-byte cache
	// line: consecutive workgroup ids go to consecutive XCDs (eight L2s), so the two halves of a line would be fetched
	// from HBM twice (PMC: 271 GB per launch over 125 GB of rows).  Inside every run of 16 ids, ids i and i + 8 --
	// the same XCD, dispatched together -- therefore take the stripes 2 (i % 8) and 2 (i % 8) + 1.
	uint32_t group = blockIdx.x;
	if (S::kRowBytes == 64u && (group | 15u) < gridDim.x) {
		group = (group & ~15u) + ((group & 7u) << 1) + ((group >> 3) & 1u);
	}
	const uint64_t group_byte = static_cast<uint64_t>(group) * S::kRowBytes; // first byte of this workgroup's stripe

	// ---- staging: HBM -> LDS without a register stop (global_load_lds_dwordx4) ----
	// One such instruction writes 64 x 16 B to consecutive LDS bytes (lane l at base + 16 l), so the slot's image
	// is filled piece by piece in linear order: piece p = kWaves n + wave (n < TS / 4) holds chunks 64 p .. 64 p + 63
	// of the tile, chunk c = row c / TS, position c % TS.  Position q of a row of an odd k-group holds the row's
	// chunk q ^ swizzle (the lanes that read k-groups 0/1 resp. 2/3 together then hit different banks); the
	// swizzle is applied to the SOURCE address here and to the read address below.
	// The resident row of each of the tile's 64 variants (variant lists need not be contiguous) travels the
	// same way, 4 bytes per lane, into a small ring of its own, six tiles ahead: by the time a tile's pieces are
	// issued its row numbers are plain LDS reads.  Nothing in the loop is an ordinary vector load, so the only
	// vmcnt wait is the counted one that retires the oldest tile in flight.
	constexpr uint32_t kGenoPieces = S::kChunksPerThread;      // per wave and tile
	// digit bytes: 1 KiB pieces, NT per plane; only the planes this instantiation multiplies are fetched
	constexpr uint32_t kBFirstPiece = (PLANES & 1) ? 0u : NT;
	constexpr uint32_t kBPieceCount = PLANES == 3 ? 2u * NT : NT;
	constexpr uint32_t kBPieces = (kBPieceCount + S::kWaves - 1u) / S::kWaves; // per wave and tile
#if defined(PGH_I8_NO_DMA_B)
	constexpr uint32_t kPieces = kGenoPieces + 1u;
#elif defined(PGH_I8_NO_DMA_G)
	constexpr uint32_t kPieces = kBPieces + 1u;
#else
	constexpr uint32_t kPieces = kGenoPieces + kBPieces + (TILED ? 0u : 1u); // + the row numbers
#endif
	constexpr uint32_t kRowAhead = 2u * kRing - 2u;            // tiles between a row-number DMA and its use
	static_assert(kPieces * (kRing - 2) < 64, "vmcnt field");
	__shared__ uint32_t s_rows[16][kTileVariants]; // by tile number mod 16: a slot is rewritten 16 tiles (>= 6 barriers) after its last read
	const uint32_t wave_u = __builtin_amdgcn_readfirstlane(wave);
	const uint32_t st_pos = lane % S::kChunksPerRow;
	uint32_t st_row[kGenoPieces];
	uint64_t st_col[kGenoPieces]; // byte offset inside a row of the chunk this lane fetches for piece n
#pragma unroll
	for (uint32_t n = 0; n < kGenoPieces; n++) {
		const uint32_t row = ((S::kWaves * n + wave) * 64u + lane) / S::kChunksPerRow;
		const uint32_t logical = st_pos ^ (((row >> 4) & 1u) * S::kSwizzleChunks);
		const uint64_t want = group_byte + 16ull * logical;
		st_row[n] = row;
		st_col[n] = want < pitch ? want : 0ull; // past the row: some valid bytes; such samples are never stored
	}
	const uint32_t last_tile = tile_end - 1u;
	// (pinned in a scalar register: re-read from the dispatch packet inside the loop, its wait drained the LDS queue)
	const uint32_t n_groups = __builtin_amdgcn_readfirstlane(gridDim.x);
	const uint32_t ring_lds = __builtin_amdgcn_readfirstlane(LdsAddress(&s_ring[0][0]));
	const uint32_t rows_lds = __builtin_amdgcn_readfirstlane(LdsAddress(&s_rows[0][0]));
	auto issue_rows = [&](uint32_t tile) {
		if (TILED) {
			return;
		}
		const uint32_t t = min(tile, last_tile);
		Glds4(rowidx + static_cast<uint64_t>(t) * kTileVariants + lane, rows_lds + ((tile - tile_begin) & 15u) * 256u);
	};
	// the resident rows this lane fetches for `tile` (plain LDS reads: the numbers landed trips ago).  The loop asks
	// for them one trip before the DMA that needs them, so the read's latency is not in front of that DMA.
	auto read_rows = [&](uint32_t tile, uint32_t(&r)[kGenoPieces]) {
		if (TILED) {
			return;
		}
		const uint32_t *rp = &s_rows[(tile - tile_begin) & 15u][0];
#pragma unroll
		for (uint32_t n = 0; n < kGenoPieces; n++) {
			r[n] = rp[st_row[n]];
		}
	};
	auto issue = [&](uint32_t tile, uint32_t slot, const uint32_t(&r)[kGenoPieces]) {
#ifdef PGH_I8_NO_DMA
		if (tile > tile_begin + 8u) {
			return;
		}
#endif
		// (past the slice: a harmless repeat of its last tile keeps the counts uniform)
		issue_rows(tile + kRowAhead);
		const uint32_t base = ring_lds + slot * kSlotBytes;

		const int8_t *bsrc = bmat + static_cast<uint64_t>(min(tile, last_tile)) * S::kBBytes;
#ifndef PGH_I8_NO_DMA_B
#pragma unroll
		for (uint32_t n = 0; n < kBPieces; n++) {
			// every wave issues every piece (the vmcnt arithmetic wants equal counts): pieces past the digit bytes
			// repeat the last one
			const uint32_t p = kBFirstPiece + min(S::kWaves * n + wave_u, kBPieceCount - 1u);
			Glds16(bsrc + 1024ull * p + 16u * lane, base + S::kGenoBytes + p * 1024u);
		}
#else
		(void)bsrc;
#endif
#ifndef PGH_I8_NO_DMA_G
		// (the genotype piece goes last: 16 rows x 64 B per instruction with TS = 4, the slow one of the batch)
#pragma unroll
		for (uint32_t n = 0; n < kGenoPieces; n++) {
			const uint32_t piece = S::kWaves * n + wave_u;
			const uint8_t *src =
			    TILED ? rows + (static_cast<uint64_t>(min(tile, last_tile)) * n_groups + blockIdx.x) * S::kGenoBytes +
			                1024u * piece + 16u * lane
			          : rows + static_cast<uint64_t>(r[n]) * pitch + st_col[n];
			Glds16Stream(src, base + piece * 1024u);
		}
#endif
	};

	// ---- compute roles ----
	// this lane's word of every row of its k-group: wave `wave` owns words [wave * TS, (wave + 1) * TS) of the
	// workgroup's stripe, lane x the word x / (16 / TS) of those; with TS < 16 it uses bytes b0 .. of the
	// transposed words
	constexpr uint32_t kLanesPerWord = 16u / TS;
	const uint32_t word = wave * TS + x / kLanesPerWord;
	const uint32_t word_phys = word ^ ((g & 1u) * S::kSwizzleChunks * 4u);
	// TILED: the slot holds the BYTE-MAJOR image of the tile (k_i8_tile_major): the sixteen bytes "byte column b of
	// variants 16 g .. 16 g + 15" are contiguous, at b * 64 + 16 * ((g + b / 4) % 4) -- the rotation spreads the
	// sixteen lanes of a k-group over all banks.  This lane's byte column is 16 * wave + x.
	const uint32_t byte_col = 16u * wave + x;
	const uint32_t geno_off =
	    TILED ? byte_col * 64u + 16u * ((g + (byte_col >> 2)) & 3u) : (16u * g) * S::kRowBytes + 4u * word_phys;

	v4i acc[TS][NT];
#pragma unroll
	for (int t = 0; t < TS; t++) {
#pragma unroll
		for (int nt = 0; nt < NT; nt++) {
			acc[t][nt] = v4i {0, 0, 0, 0};
		}
	}

	// Operand bytes.  The matrix instruction wants, per lane, sixteen variants of ONE sample as sixteen bytes;
	// a packed word holds sixteen samples of ONE variant.  Byte c of such a word is samples 4c .. 4c+3, so two
	// byte permutes gather byte c of four variants' words into one register G (byte j = variant 4q + j); the call
	// of sample 4c + e then sits at bits 2e, 2e+1 of EVERY byte and a single AND leaves the instruction's operand
	// -- scaled by 4^e, which is the same for the whole matrix row (one sample) and is divided out of that
	// sample's sums at the end.  e = 3 would reach 192: it is taken from G >> 1 instead (scale 32).  The
	// missing plane is one three-input AND of G, G >> 1 and a bit mask, at the same scale.  ~12 vector ops per
	// sixteen calls and both planes.
	constexpr uint32_t kBytesPerLane = TS / 4u; // bytes of each word this lane turns into operands
	uint32_t sel_first = 0;                     // TS < 16: which bytes -- lane-dependent permute selectors
	if (TS == 8) {
		sel_first = (x & 1u) ? 0x07030602u : 0x05010400u;
	} else if (TS == 4) {
		sel_first = ((4u + (x & 3u)) << 8) | (x & 3u);
	}
	// this lane's sixteen words of the tile in `slot`: read at the top of a trip, in front of the DMA issue, so that
	// the LDS round trip runs under the issue's scalar work instead of after it
	auto load_words = [&](uint32_t slot, uint32_t(&wd)[16]) {
		const uint8_t *gp = &s_ring[slot][geno_off];
		if (TILED) {
#ifdef PGH_I8_NO_BUILD
			wd[0] = lane, wd[1] = lane * 3u, wd[2] = lane * 5u, wd[3] = lane * 7u;
#else
			const uint4 w = *reinterpret_cast<const uint4 *>(gp); // wd[q] is already G[q]: byte j = variant 4 q + j
			wd[0] = w.x, wd[1] = w.y, wd[2] = w.z, wd[3] = w.w;
#endif
			return;
		}
#pragma unroll
		for (int k = 0; k < 16; k++) {
#ifdef PGH_I8_NO_BUILD
			wd[k] = lane * (k + 3u);
			(void)gp;
#else
			wd[k] = *reinterpret_cast<const uint32_t *>(gp + k * S::kRowBytes);
#endif
		}
	};
	auto compute = [&](uint32_t slot, const uint32_t(&wd)[16]) {
		uint32_t G[4][kBytesPerLane];
#pragma unroll
		for (int q = 0; q < 4; q++) {
			if (TILED) {
				G[q][0] = wd[q];
				continue;
			}
			const uint32_t w0 = wd[4 * q + 0], w1 = wd[4 * q + 1], w2 = wd[4 * q + 2], w3 = wd[4 * q + 3];
			if (TS == 16) {
				const uint32_t pa = __builtin_amdgcn_perm(w1, w0, 0x05010400u), pb = __builtin_amdgcn_perm(w1, w0, 0x07030602u);
				const uint32_t qa = __builtin_amdgcn_perm(w3, w2, 0x05010400u), qb = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
				G[q][0] = __builtin_amdgcn_perm(qa, pa, 0x05040100u);
				G[q][1 % kBytesPerLane] = __builtin_amdgcn_perm(qa, pa, 0x07060302u);
				G[q][2 % kBytesPerLane] = __builtin_amdgcn_perm(qb, pb, 0x05040100u);
				G[q][3 % kBytesPerLane] = __builtin_amdgcn_perm(qb, pb, 0x07060302u);
			} else {
				const uint32_t pa = __builtin_amdgcn_perm(w1, w0, sel_first), qa = __builtin_amdgcn_perm(w3, w2, sel_first);
				G[q][0] = __builtin_amdgcn_perm(qa, pa, 0x05040100u);
				if (TS == 8) {
					G[q][1 % kBytesPerLane] = __builtin_amdgcn_perm(qa, pa, 0x07060302u);
				}
			}
		}
		// operand bytes of the whole tile, once: TS samples x (code plane, missing plane) x 4 registers
		v4i U[TS], Mi[TS];
#pragma unroll
		for (int cc = 0; cc < static_cast<int>(kBytesPerLane); cc++) {
			uint32_t h1[4], h2[4];
#pragma unroll
			for (int q = 0; q < 4; q++) {
				h1[q] = G[q][cc] >> 1;
				h2[q] = G[q][cc] >> 2;
			}
#pragma unroll
			for (int e = 0; e < 4; e++) {
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const uint32_t gq = G[q][cc];
					if (e < 3) {
						U[cc * 4 + e][q] = static_cast<int>(gq & (0x03030303u << (2 * e)));
						Mi[cc * 4 + e][q] = static_cast<int>(gq & h1[q] & (0x01010101u << (2 * e)));
					} else {
						U[cc * 4 + e][q] = static_cast<int>(h1[q] & 0x60606060u);
						Mi[cc * 4 + e][q] = static_cast<int>(h1[q] & h2[q] & 0x20202020u);
					}
				}
			}
		}
		// digit operands: two tiles at a time stay in registers across the samples, and the next two are read
		// from LDS while these multiply.  (What keeps the matrix pipe at ~70 % with many tiles is not this read and
		// not the operand building either -- without the LDS-DMA the launch runs at the matrix-instruction floor;
		// the knock-out table is in DESIGN.md section 6.)
		const v4i *bp = reinterpret_cast<const v4i *>(&s_ring[slot][S::kGenoBytes]);
		constexpr int kHold = NT >= 2 ? 2 : 1;
		v4i bg[kHold], bm[kHold], ng[kHold], nm[kHold];
		auto load_b = [&](v4i *dg, v4i *dm, int nt_first) {
#pragma unroll
			for (int h = 0; h < kHold; h++) {
				if (nt_first + h < NT) {
					if (PLANES & 1) {
						dg[h] = bp[((0 * NT + nt_first + h) * 4 + g) * 16 + x];
					}
					if (PLANES & 2) {
						dm[h] = bp[((1 * NT + nt_first + h) * 4 + g) * 16 + x];
					}
				}
			}
		};
		load_b(bg, bm, 0);
#pragma unroll
		for (int nt0 = 0; nt0 < NT; nt0 += kHold) {
			if (nt0 + kHold < NT) {
				load_b(ng, nm, nt0 + kHold);
			}
#pragma unroll
			for (int t = 0; t < TS; t++) {
#pragma unroll
				for (int h = 0; h < kHold; h++) {
					if (nt0 + h < NT) {
						if (PLANES & 1) {
							acc[t][nt0 + h] = __builtin_amdgcn_mfma_i32_16x16x64_i8(U[t], bg[h], acc[t][nt0 + h], 0, 0, 0);
						}
						if (PLANES & 2) {
							acc[t][nt0 + h] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Mi[t], bm[h], acc[t][nt0 + h], 0, 0, 0);
						}
					}
				}
			}
#pragma unroll
			for (int h = 0; h < kHold; h++) {
				bg[h] = ng[h];
				bm[h] = nm[h];
			}
		}
	};

	// ---- the matrix-bound shapes (I8Pipelined: TS = 4, five digit tiles and more): the software-pipelined loop ----
	// In the loop further down a trip is [read words, issue DMA, build operands, multiply, barrier]: the barrier puts
	// the two waves of a SIMD in step, so both build (vector pipe, ~50 instructions each) while the matrix pipe has
	// nothing queued, then both multiply -- a third of the cycles of the 16-column launch had no matrix instruction
	// in flight.  Here trip t multiplies tile t with operands that trip t-1 built, and builds tile t+1's between its
	// own matrix instructions (a quarter of the build after every group of them); the first digit registers of tile
	// t+1 are asked for before the barrier.  Ring: slot t = tile t (digit bytes), slot t+1 = tile t+1 (words), tiles
	// t+2 .. t+kRing-2 on their way, tile t+kRing-1 issued into the slot trip t-1 finished with.
	if constexpr (I8Pipelined<NT, TS, PLANES, TILED>()) {
#define PGH_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
		// tuning knobs of this loop (tools/i8_variants.sh builds and times the combinations)
#ifndef PGH_I8_SKEW
#define PGH_I8_SKEW 0 // 1: waves 4 .. 7 run half a trip behind waves 0 .. 3
#endif
#ifndef PGH_I8_HOLD
#define PGH_I8_HOLD 1 // digit tiles per group of matrix instructions
#endif
#ifndef PGH_I8_FILL
#define PGH_I8_FILL 1 // vector instructions scheduled behind every matrix instruction (0: the compiler's order)
#endif
#ifndef PGH_I8_JITW
#define PGH_I8_JITW 1 // genotype words read a quarter at a time, one group ahead of their use
#endif
#ifndef PGH_I8_ISSUE_AT
#define PGH_I8_ISSUE_AT 2 // the early waves' DMA pieces go out behind this group (-1: at the top of the trip)
#endif
		constexpr bool kSkew = PGH_I8_SKEW != 0;
		static_assert(kRing >= (kSkew ? 5u : 4u), "two landed tiles + their successors on the way");
		auto load_words_part = [&](uint32_t slot, int q, uint32_t(&wd)[16]) {
			if (TILED) {
				if (q == 0) {
					load_words(slot, wd); // one 16-byte read holds all four quarters
				}
				return;
			}
			const uint8_t *gp = &s_ring[slot][geno_off];
#pragma unroll
			for (int k = 4 * q; k < 4 * q + 4; k++) {
#ifdef PGH_I8_NO_BUILD
				wd[k] = lane * (k + 3u);
				(void)gp;
#else
				wd[k] = *reinterpret_cast<const uint32_t *>(gp + k * S::kRowBytes);
#endif
			}
		};
		auto build_part = [&](const uint32_t(&wd)[16], int q, v4i(&U)[TS], v4i(&Mi)[TS]) {
			uint32_t gq;
			if (TILED) {
				gq = wd[q];
			} else {
				const uint32_t pa = __builtin_amdgcn_perm(wd[4 * q + 1], wd[4 * q + 0], sel_first);
				const uint32_t qa = __builtin_amdgcn_perm(wd[4 * q + 3], wd[4 * q + 2], sel_first);
				gq = __builtin_amdgcn_perm(qa, pa, 0x05040100u);
			}
			const uint32_t h1 = gq >> 1, h2 = gq >> 2;
#pragma unroll
			for (int e = 0; e < 3; e++) {
				U[e][q] = static_cast<int>(gq & (0x03030303u << (2 * e)));
				Mi[e][q] = static_cast<int>(gq & h1 & (0x01010101u << (2 * e)));
			}
			U[3][q] = static_cast<int>(h1 & 0x60606060u);
			Mi[3][q] = static_cast<int>(h1 & h2 & 0x20202020u);
		};
		constexpr int kHold = PGH_I8_HOLD;
		constexpr int kGroups = (NT + kHold - 1) / kHold; // groups of matrix instructions per trip
		auto load_b = [&](uint32_t slot, v4i *dg, v4i *dm, int nt_first) {
			const v4i *bp = reinterpret_cast<const v4i *>(&s_ring[slot][S::kGenoBytes]);
#pragma unroll
			for (int h = 0; h < kHold; h++) {
				if (nt_first + h < NT) {
					if (PLANES & 1) {
						dg[h] = bp[((0 * NT + nt_first + h) * 4 + g) * 16 + x];
					}
					if (PLANES & 2) {
						dm[h] = bp[((1 * NT + nt_first + h) * 4 + g) * 16 + x];
					}
				}
			}
		};
		uint32_t r_cur[kGenoPieces] = {}, r_nxt[kGenoPieces] = {};
		v4i bg[kHold] = {}, bm[kHold] = {};
		// Early and late waves (kSkew).  A workgroup's barrier puts the two waves of a SIMD (waves w and w + 4) in
		// step: both read words and issue their DMA pieces right behind it, both wait in front of it.  With kSkew
		// waves 4 .. 7 run half a trip behind: their barrier sits in the MIDDLE of their groups of matrix
		// instructions and their DMA issue behind that, while waves 0 .. 3 have theirs at the trip's end and start.
		// The late waves read tile t - 1's digits until barrier t, so its slot is free for tile t + kRing - 1 only
		// behind that barrier: the late waves issue that tile in the second half of their trip t, the early waves
		// in their trip t + 1 (each trip of theirs issues tile + kRing - 2).  Either way a wave has issued up to
		// tile t + kRing - 2 when it reaches barrier t and waits for all but the youngest kRing - 4 of them: tile
		// t + 2 has landed, which is what the half trips behind the barrier read.  Without kSkew every wave is an
		// "early" one that issues tile t + kRing - 1 in trip t and waits for all but kRing - 3 tiles.
		auto trip = [&](auto late_c, uint32_t tile, uint32_t slot, v4i(&Uc)[TS], v4i(&Mc)[TS], v4i(&Un)[TS], v4i(&Mn)[TS]) {
			constexpr bool kLate = decltype(late_c)::value;
			constexpr int kMid = kGroups / 2;
			constexpr int kIssueAfter = kLate ? kMid : PGH_I8_ISSUE_AT; // the DMA pieces go out behind this group
			constexpr uint32_t kAhead = (kLate || !kSkew) ? kRing - 1u : kRing - 2u;
			constexpr uint32_t kKeep = kSkew ? kRing - 4u : kRing - 3u; // tiles still on their way behind the barrier
			const uint32_t nslot = slot + 1u == kRing ? 0u : slot + 1u;
			uint32_t wd[16];
			read_rows(tile + 1u + kAhead, r_nxt);
			if (kIssueAfter < 0) {
				issue(tile + kAhead, (slot + kAhead) % kRing, r_cur);
			}
#pragma unroll
			for (int gi = 0; gi < kGroups; gi++) {
				const int nt0 = gi * kHold;
#pragma unroll
				for (int q = 0; q < 4; q++) {
					const int build_group = (q * kGroups) / 4;
					if (gi == (PGH_I8_JITW && build_group > 0 ? build_group - 1 : 0)) {
						load_words_part(nslot, q, wd);
					}
				}
				if (kLate && gi == kMid) {
					PGH_WAIT_VM(kPieces * kKeep);
					asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
					__builtin_amdgcn_s_barrier();
				}
				v4i ng[kHold] = {}, nm[kHold] = {};
				if (gi + 1 < kGroups) {
					load_b(slot, ng, nm, nt0 + kHold);
				} else {
					load_b(nslot, ng, nm, 0); // the next trip's first digit registers
				}
#pragma unroll
				for (int t = 0; t < TS; t++) {
#pragma unroll
					for (int h = 0; h < kHold; h++) {
						if (nt0 + h < NT) {
							if (PLANES & 1) {
								acc[t][nt0 + h] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Uc[t], bg[h], acc[t][nt0 + h], 0, 0, 0);
							}
							if (PLANES & 2) {
								acc[t][nt0 + h] = __builtin_amdgcn_mfma_i32_16x16x64_i8(Mc[t], bm[h], acc[t][nt0 + h], 0, 0, 0);
							}
						}
					}
				}
				// the next tile's operands, a quarter behind each of four groups spread over the trip
#pragma unroll
				for (int q = 0; q < 4; q++) {
					if (gi == (q * kGroups) / 4) {
						build_part(wd, q, Un, Mn);
					}
				}
#ifndef PGH_I8_FREE_SCHEDULE
				// PGH_I8_FILL vector instructions of the build (and one LDS read, while there are any) behind every
				// matrix instruction: an MFMA holds the SIMD's vector issue for 8 of its 16 cycles and the two waves
				// of a SIMD alternate, so a gap has room for about two fillers per wave -- clustered fillers cost
				// their full issue time (MI355X_MICROARCH.md, 'vector-instruction ISSUE cost').
				if (PGH_I8_FILL > 0) {
#pragma unroll
					for (int i = 0; i < 8 * kHold; i++) {
						__builtin_amdgcn_sched_group_barrier(0x008, 1, 0); // MFMA
						__builtin_amdgcn_sched_group_barrier(0x002, PGH_I8_FILL, 0); // VALU
						if (i < 6) {
							__builtin_amdgcn_sched_group_barrier(0x100, 1, 0); // DS read
						}
					}
				}
				__builtin_amdgcn_sched_barrier(0); // keep the groups in this order: the build rides under the multiplies
#endif
				if (gi == kIssueAfter) {
					issue(tile + kAhead, (slot + kAhead) % kRing, r_cur);
				}
#pragma unroll
				for (int h = 0; h < kHold; h++) {
					bg[h] = ng[h];
					bm[h] = nm[h];
				}
			}
			if (!kLate) {
				PGH_WAIT_VM(kPieces * kKeep);
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				__builtin_amdgcn_s_barrier();
			}
#pragma unroll
			for (uint32_t n = 0; n < kGenoPieces; n++) {
				r_cur[n] = r_nxt[n];
			}
		};
#pragma unroll
		for (uint32_t d = 0; d < kRowAhead; d++) {
			issue_rows(tile_begin + d);
		}
		PGH_WAIT_VM(0);
		__builtin_amdgcn_s_barrier();
		const bool late = kSkew && (wave_u & 4u) != 0u;
		constexpr uint32_t kFill = kSkew ? kRing - 2u : kRing - 1u; // tiles every wave issues before the loop
#pragma unroll
		for (uint32_t d = 0; d < kFill; d++) {
			read_rows(tile_begin + d, r_cur);
			issue(tile_begin + d, d, r_cur);
		}
		if (late) { // (the early waves issue tile kRing - 2 in their first trip)
			read_rows(tile_begin + kFill, r_cur);
			issue(tile_begin + kFill, kFill, r_cur);
			read_rows(tile_begin + kFill + 1u, r_cur);
			PGH_WAIT_VM(kPieces * (kRing - 3)); // tiles 0 and 1 of the slice have landed
		} else {
			read_rows(tile_begin + kFill, r_cur);
			PGH_WAIT_VM(kPieces * (kFill - 2u));
		}
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		v4i Ua[TS], Ma[TS], Ub[TS], Mb[TS];
		{
			uint32_t wd[16];
			load_words(0, wd);
#pragma unroll
			for (int q = 0; q < 4; q++) {
				build_part(wd, q, Ua, Ma);
			}
			load_b(0, bg, bm, 0);
		}
		// (slices hold an even number of tiles: LaunchI8 rounds tiles per slice up, ScoreI8Bytes the tile count)
		uint32_t slot = 0;
		if (late) {
			for (uint32_t tile = tile_begin; tile < tile_end; tile += 2) {
				trip(std::true_type {}, tile, slot, Ua, Ma, Ub, Mb);
				slot = slot + 1u == kRing ? 0u : slot + 1u;
				trip(std::true_type {}, tile + 1u, slot, Ub, Mb, Ua, Ma);
				slot = slot + 1u == kRing ? 0u : slot + 1u;
			}
		} else {
			for (uint32_t tile = tile_begin; tile < tile_end; tile += 2) {
				trip(std::false_type {}, tile, slot, Ua, Ma, Ub, Mb);
				slot = slot + 1u == kRing ? 0u : slot + 1u;
				trip(std::false_type {}, tile + 1u, slot, Ub, Mb, Ua, Ma);
				slot = slot + 1u == kRing ? 0u : slot + 1u;
			}
		}
		PGH_WAIT_VM(0); // the tail's repeats must land before the LDS is handed back
#undef PGH_WAIT_VM
	} else {
	// Ring protocol (kRing = 4).  Trip t multiplies slot t % 4 while the DMAs of tiles t+1 .. t+3 are in flight or landed;
	// it first issues tile t+3 into the slot trip t-1 finished with (every wave passed that trip's barrier).
	// At the end each wave waits until all but its 2 x kPieces youngest DMAs have landed -- its share of tile
	// t+1, and the row numbers it asked for three trips ago -- and the barrier makes that true of every wave's
	// share before anyone reads them.
#define PGH_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#pragma unroll
	for (uint32_t d = 0; d < kRowAhead; d++) {
		issue_rows(tile_begin + d);
	}
	PGH_WAIT_VM(0);
	__builtin_amdgcn_s_barrier();
	uint32_t r_cur[kGenoPieces] = {}, r_nxt[kGenoPieces] = {};
#pragma unroll
	for (uint32_t d = 0; d + 1 < kRing; d++) {
		read_rows(tile_begin + d, r_cur);
		issue(tile_begin + d, d, r_cur);
	}
	read_rows(tile_begin + (kRing - 1), r_cur);
	PGH_WAIT_VM(kPieces * (kRing - 2));
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__builtin_amdgcn_s_barrier();
	uint32_t slot = 0;
	for (uint32_t tile = tile_begin; tile < tile_end; tile++) {
		uint32_t wd[16];
		load_words(slot, wd);
		read_rows(tile + kRing, r_nxt); // for the next trip's DMA (its slot of s_rows was filled 2 kRing - 3 trips ago)
		issue(tile + (kRing - 1), (slot + kRing - 1) % kRing, r_cur);
		compute(slot, wd);
		PGH_WAIT_VM(kPieces * (kRing - 2));
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		slot = (slot + 1) % kRing;
#pragma unroll
		for (uint32_t n = 0; n < kGenoPieces; n++) {
			r_cur[n] = r_nxt[n];
		}
	}
	PGH_WAIT_VM(0); // the tail's repeats must land before the LDS is handed back
#undef PGH_WAIT_VM
	}

	// ---- epilogue: lane (j = lane & 15, g) holds digit column 16 nt + j of the sample slots 4 g + r ----
	// The digits of one output column sit in up to seven NEIGHBOURING lanes of a 16-lane row: they are scaled,
	// summed across those lanes (three shuffle steps inside the row, gated on "same output column") and the lane
	// that holds the column's lowest digit of this tile issues ONE atomic add per sample.
	const uint32_t wave_sample0 = group * S::kSamplesPerGroup + wave * 16u * TS;
#pragma unroll
	for (int nt = 0; nt < NT; nt++) {
		const uint32_t J = 16u * nt + x;
		const uint32_t tg = target[J];
		const double mu = tg == 0xffffffffu ? 0.0 : mult[J];
		// which of the next 1, 2, 4 lanes of this row carry a digit of the same output column
		bool same[3];
#pragma unroll
		for (int st = 0; st < 3; st++) {
			const uint32_t off = 1u << st;
			same[st] = tg != 0xffffffffu && x + off < 16u && target[J + off] == tg;
		}
		const bool head = tg != 0xffffffffu && (x == 0u || target[J - 1] != tg);
#pragma unroll
		for (int t = 0; t < TS; t++) {
			// sample t of the lane's span was multiplied at scale {1, 4, 16, 32}[t & 3]: exact powers of two
			const int scale = (t & 3) == 3 ? 32 : (1 << (2 * (t & 3)));
			const double mu_t = mu / static_cast<double>(scale);
#pragma unroll
			for (int r = 0; r < 4; r++) {
				const uint32_t s = wave_sample0 + TS * (4u * g + r) + t;
				const int v = acc[t][nt][r];
				if (tg == n_cols + 1u) { // the missing count: an integer column of its own
					if (missing_ct && s < sample_ct && v != 0) {
						atomicAdd(missing_ct + s, static_cast<uint32_t>(v / scale));
					}
					continue;
				}
				double val = mu_t * static_cast<double>(v);
#pragma unroll
				for (int st = 0; st < 3; st++) {
					const double up = __shfl_down(val, 1u << st, 16);
					val += same[st] ? up : 0.0;
				}
				if (head && s < sample_ct && val != 0.0) {
					if (tg < n_cols) {
						unsafeAtomicAdd(score + static_cast<uint64_t>(s) * out_stride + tg, val);
					} else if (dosage_sum) {
						unsafeAtomicAdd(dosage_sum + s, val);
					}
				}
			}
		}
	}
}

// score[s][c] += k0[c]; dosage_sum[s] += k0[n_cols]
__global__ __launch_bounds__(256) void k_i8_constants(uint32_t sample_ct, uint32_t n_cols, const double *__restrict__ k0,
                                                      double *__restrict__ score, uint32_t out_stride,
                                                      double *__restrict__ dosage_sum) {
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s >= sample_ct) {
		return;
	}
	for (uint32_t c = 0; c < n_cols; c++) {
		const double k = k0[c];
		if (k != 0.0) {
			score[static_cast<uint64_t>(s) * out_stride + c] += k;
		}
	}
	if (dosage_sum && k0[n_cols] != 0.0) {
		dosage_sum[s] += k0[n_cols];
	}
}

__global__ __launch_bounds__(256) void k_i8_rowidx(const uint32_t *__restrict__ vlist, uint32_t n_var, uint32_t n_pad,
                                                   uint32_t *__restrict__ rowidx) {
	const uint32_t i = blockIdx.x * 256u + threadIdx.x;
	if (i < n_pad) {
		rowidx[i] = i < n_var ? vlist[i] : vlist[n_var - 1]; // padding rows carry all-zero digits
	}
}

// The tile-major copy: image (tile, group) = the 8 KB a ring slot of k_score_i8<., 4, ., true> holds, BYTE-MAJOR: for
// byte column b (0 .. 127) of the group's 128-byte stripe and k-group g (variants 16 g .. 16 g + 15 of the tile) the
// sixteen bytes [listed row 64 tile + 16 g + i][128 group + b], i = 0 .. 15, sit together at
// b * 64 + 16 * ((g + b / 4) % 4).  A lane of the contraction wants exactly those sixteen bytes (one sample quartet
// of sixteen variants): one 16-byte LDS read and no byte permutes, where the row-major image costs sixteen word
// reads and twelve permutes per lane and tile.
__global__ __launch_bounds__(512) void k_i8_tile_major(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                       const uint32_t *__restrict__ vlist, uint32_t n_var,
                                                       uint8_t *__restrict__ out) {
	using S = I8Shape<5, 4>; // every TS = 4 shape shares the stripe geometry
	__shared__ __attribute__((aligned(16))) uint8_t s_tile[kTileVariants][S::kRowBytes + 16]; // (+16: rows on different banks)
	const uint32_t group = blockIdx.x, tile = blockIdx.y;
	{
		const uint32_t row = threadIdx.x / S::kChunksPerRow, pos = threadIdx.x % S::kChunksPerRow;
		const uint32_t i = tile * kTileVariants + row;
		const uint64_t col = static_cast<uint64_t>(group) * S::kRowBytes + 16ull * pos;
		uint4 w = make_uint4(0, 0, 0, 0);
		if (i < n_var && col + 16 <= pitch) { // padding rows carry all-zero digits; so do the bytes past the row
			w = LoadStream(reinterpret_cast<const uint4 *>(rows + static_cast<uint64_t>(vlist[i]) * pitch + col));
		}
		*reinterpret_cast<uint4 *>(&s_tile[row][16u * pos]) = w;
	}
	__syncthreads();
	const uint32_t b = threadIdx.x >> 2, slot = threadIdx.x & 3u;
	const uint32_t g = (slot - (b >> 2)) & 3u;
	uint32_t o[4];
#pragma unroll
	for (int q = 0; q < 4; q++) {
		uint32_t v = 0;
#pragma unroll
		for (int j = 0; j < 4; j++) {
			v |= static_cast<uint32_t>(s_tile[16u * g + 4u * q + j][b]) << (8 * j);
		}
		o[q] = v;
	}
	StoreStream(reinterpret_cast<uint4 *>(out + (static_cast<uint64_t>(tile) * gridDim.x + group) * S::kGenoBytes) +
	                threadIdx.x,
	            make_uint4(o[0], o[1], o[2], o[3]));
}

} // namespace

bool ScoreI8UsesTiles(uint32_t n_cols, bool extras) {
	return ScoreI8Tiles16(n_cols, extras) >= 5u; // the TS = 4 shapes of LaunchI8Shape
}

size_t ScoreI8TiledBytes(uint32_t n_var, uint32_t sample_ct) {
	using S = I8Shape<5, 4>;
	const uint64_t n_tiles = ((static_cast<uint64_t>(n_var) + kTileVariants - 1) / kTileVariants + 1u) & ~1ull;
	const uint64_t groups = (static_cast<uint64_t>(sample_ct) + S::kSamplesPerGroup - 1) / S::kSamplesPerGroup;
	return static_cast<size_t>(n_tiles * groups * S::kGenoBytes);
}

hipError_t LaunchScoreI8TileMajor(const RowView &view, const uint32_t *vlist, uint32_t n_var, uint8_t *out,
                                  hipStream_t stream) {
	using S = I8Shape<5, 4>;
	if (n_var == 0) {
		return hipSuccess;
	}
	const uint32_t n_tiles = ((n_var + kTileVariants - 1) / kTileVariants + 1u) & ~1u;
	const uint32_t groups = (view.sample_ct + S::kSamplesPerGroup - 1) / S::kSamplesPerGroup;
	if (n_tiles > 65535u * 16u) {
		return hipErrorInvalidValue;
	}
	// grid.y <= 65535: a launch per 65,535 tiles (4 M variants)
	for (uint32_t t0 = 0; t0 < n_tiles; t0 += 65535u) {
		const uint32_t nt = n_tiles - t0 < 65535u ? n_tiles - t0 : 65535u;
		hipLaunchKernelGGL(k_i8_tile_major, dim3(groups, nt), dim3(512), 0, stream, view.rows, view.pitch,
		                   vlist + static_cast<uint64_t>(t0) * kTileVariants,
		                   n_var > t0 * kTileVariants ? n_var - t0 * kTileVariants : 1u,
		                   out + static_cast<uint64_t>(t0) * groups * S::kGenoBytes);
	}
	return hipGetLastError();
}

uint32_t ScoreI8Tiles16(uint32_t n_cols, bool extras) {
	return (DigitColumns(n_cols, extras) + 15u) / 16u;
}

ScoreI8Sizes ScoreI8Bytes(uint32_t n_var, uint32_t n_cols, bool extras) {
	ScoreI8Sizes z;
	z.n_tiles = ((n_var + kTileVariants - 1) / kTileVariants + 1u) & ~1u; // an even number: the kernel walks tile pairs
	z.n_tiles16 = ScoreI8Tiles16(n_cols, extras);
	z.bmat = static_cast<size_t>(z.n_tiles) * 2u * z.n_tiles16 * 1024u;
	z.rowidx = sizeof(uint32_t) * static_cast<size_t>(z.n_tiles) * kTileVariants;
	z.cols = static_cast<size_t>(z.n_tiles16) * 16u * (sizeof(double) + sizeof(uint32_t));
	z.small = sizeof(double) * 3u * (n_cols + 2u);
	return z;
}

hipError_t LaunchScoreI8Prepare(const uint32_t *vlist, uint32_t n_var, const double *weights, uint32_t w_stride,
                                uint32_t n_cols, const double *ts, const double *td, const uint32_t *ac,
                                bool count_missing, bool extras, int table_mode, const ScoreI8Buffers &b,
                                hipStream_t stream) {
	if (n_var == 0) {
		return hipSuccess;
	}
	const ScoreI8Sizes z = ScoreI8Bytes(n_var, n_cols, extras);
	hipError_t e = hipMemsetAsync(b.bmat, 0, z.bmat, stream);
	// (the three small arrays need not be adjacent: a caller may size one block for its widest pass)
	if (e == hipSuccess) {
		e = hipMemsetAsync(b.colmax, 0, sizeof(unsigned long long) * (n_cols + 2u), stream);
	}
	if (e == hipSuccess) {
		e = hipMemsetAsync(b.k0, 0, sizeof(double) * (n_cols + 2u), stream);
	}
	if (e == hipSuccess) {
		e = hipMemsetAsync(b.scale_exp, 0, sizeof(double) * (n_cols + 2u), stream);
	}
	if (e != hipSuccess) {
		return e;
	}
	const uint32_t blocks = (n_var + 255) / 256;
	hipLaunchKernelGGL(k_i8_rowidx, dim3((z.n_tiles * kTileVariants + 255) / 256), dim3(256), 0, stream, vlist, n_var,
	                   z.n_tiles * kTileVariants, b.rowidx);
	const uint32_t n_targets = n_cols + (extras ? 1u : 0u); // real-valued columns: the weights (+ the dosage sum)
	hipLaunchKernelGGL(k_i8_ranges, dim3(blocks < 1024 ? blocks : 1024, n_targets), dim3(256), 0, stream, n_var, n_cols,
	                   weights, w_stride, ts, td, table_mode, b.colmax, b.k0);
	hipLaunchKernelGGL(k_i8_columns, dim3((z.n_tiles16 * 16 + 63) / 64), dim3(64), 0, stream, n_cols, extras ? 1 : 0,
	                   z.n_tiles16, b.colmax, b.scale_exp, b.mult, b.target);
	hipLaunchKernelGGL(k_i8_digits, dim3(blocks, n_targets + (extras ? 1u : 0u)), dim3(256), 0, stream, n_var, n_cols,
	                   z.n_tiles16, weights, w_stride, ts, td, ac, count_missing ? 1 : 0, table_mode, b.scale_exp, b.bmat);
	return hipGetLastError();
}

template <int NT, int TS, int PLANES>
static hipError_t LaunchI8(const RowView &view, uint32_t n_tiles, uint32_t n_cols, const ScoreI8Buffers &b, double *score,
                           uint32_t out_stride, double *dosage_sum, uint32_t *missing_ct, hipStream_t stream) {
	using S = I8Shape<NT, TS>;
	const uint32_t groups = (view.sample_ct + S::kSamplesPerGroup - 1) / S::kSamplesPerGroup;
	// enough workgroups to fill the chip several times over (every slice ends in one atomic add per sample and
	// digit column, so no more than that); a slice keeps the int32 sums far from overflow (operand bytes reach
	// 96, digits 128) and its digit bytes inside one XCD's L2
	uint32_t want = (4096u * 4u / S::kWaves + groups - 1) / groups; // (the same number of waves either way)
	uint32_t tps = (n_tiles + want - 1) / want;
	// 1,024 .. 128,000 variants per slice: |int32 sum| <= 16,384 per variant (operand bytes reach 96 + 32, digits
	// 128) stays below 2^31.  Long slices win: every slice costs a ring fill and one atomic add per sample and
	// column (64-tile slices ran plink_pca 2.7x slower than 1,024-tile ones).
	const uint32_t tps_min = 16;
	uint32_t tps_max = 2000;
	static const int tps_env = [] {
		const char *e = getenv("PGH_I8_TPS_MAX"); // tuning knob: tiles per slice
		return e ? atoi(e) : 0;
	}();
	if (tps_env > 0 && tps_env < 2000) {
		tps_max = static_cast<uint32_t>(tps_env);
	}
	tps = tps < tps_min ? tps_min : (tps > tps_max ? tps_max : tps);
	tps = (tps + 1u) & ~1u;
	uint32_t slices = (n_tiles + tps - 1) / tps;
	if (slices > 65535u) {
		return hipErrorInvalidValue;
	}
	if (TS == 4 && b.tiled) {
		hipLaunchKernelGGL((k_score_i8<NT, TS, PLANES, TS == 4>), dim3(groups, slices), dim3(S::kThreads), 0, stream, b.tiled,
		                   view.pitch, view.sample_ct, b.rowidx, n_tiles, tps, b.bmat, b.mult, b.target, n_cols, score,
		                   out_stride, dosage_sum, missing_ct);
		return hipGetLastError();
	}
	hipLaunchKernelGGL((k_score_i8<NT, TS, PLANES, false>), dim3(groups, slices), dim3(S::kThreads), 0, stream, view.rows, view.pitch,
	                   view.sample_ct, b.rowidx, n_tiles, tps, b.bmat, b.mult, b.target, n_cols, score, out_stride,
	                   dosage_sum, missing_ct);
	return hipGetLastError();
}

template <int PLANES>
static hipError_t LaunchI8Shape(uint32_t nt, const RowView &view, uint32_t n_tiles, uint32_t n_cols,
                                const ScoreI8Buffers &b, double *score, uint32_t out_stride, double *dosage_sum,
                                uint32_t *missing_ct, hipStream_t stream) {
#define PGH_I8(NT, TS)                                                                                                 \
	case NT:                                                                                                           \
		return LaunchI8<NT, TS, PLANES>(view, n_tiles, n_cols, b, score, out_stride, dosage_sum, missing_ct, stream)
	switch (nt) {
		PGH_I8(1, 16);
		PGH_I8(2, 16);
		PGH_I8(3, 8);
		PGH_I8(4, 8);
		PGH_I8(5, 4);
		PGH_I8(6, 4);
		PGH_I8(7, 4);
		PGH_I8(8, 4);
		PGH_I8(9, 4);
		PGH_I8(10, 4);
	default:
		return hipErrorInvalidValue; // the caller splits wider weight sets into passes of <= kI8MaxCols columns
	}
#undef PGH_I8
}

hipError_t LaunchScoreI8(const RowView &view, uint32_t n_var, uint32_t n_cols, bool extras, int planes,
                         const ScoreI8Buffers &b, double *score, uint32_t out_stride, double *dosage_sum,
                         uint32_t *missing_ct, hipStream_t stream) {
	if (n_var == 0) {
		return hipSuccess;
	}
	const ScoreI8Sizes z = ScoreI8Bytes(n_var, n_cols, extras);
	hipError_t e;
	if (planes == kI8CodePlane) {
		e = LaunchI8Shape<1>(z.n_tiles16, view, z.n_tiles, n_cols, b, score, out_stride, dosage_sum, missing_ct, stream);
	} else if (planes == kI8MissingPlane) {
		e = LaunchI8Shape<2>(z.n_tiles16, view, z.n_tiles, n_cols, b, score, out_stride, dosage_sum, missing_ct, stream);
	} else {
		e = LaunchI8Shape<3>(z.n_tiles16, view, z.n_tiles, n_cols, b, score, out_stride, dosage_sum, missing_ct, stream);
	}
	if (e != hipSuccess) {
		return e;
	}
	if (planes == kI8Tables) {
		hipLaunchKernelGGL(k_i8_constants, dim3((view.sample_ct + 255) / 256), dim3(256), 0, stream, view.sample_ct,
		                   n_cols, b.k0, score, out_stride, dosage_sum);
	}
	return hipGetLastError();
}

} // namespace pgh

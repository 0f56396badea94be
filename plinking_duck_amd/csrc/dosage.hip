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

#include "device_utils.hpp"

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

__global__ __launch_bounds__(256) void k_dosage_sums(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                     uint32_t sample_ct, DosageView dos, uint32_t v0,
                                                     const uint32_t *__restrict__ vlist,
                                                     const uint64_t *__restrict__ include,
                                                     uint64_t *__restrict__ out) {
	__shared__ uint64_t s_part[4][3];
	const uint32_t i = blockIdx.x;
	const uint32_t lv = vlist ? vlist[i] : v0 + i;
	const uint32_t *row32 = reinterpret_cast<const uint32_t *>(rows + static_cast<uint64_t>(lv) * pitch);
	const DosageRow row = RowOf(dos, lv);
	const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
	uint64_t sum = 0, ssq = 0, nm = 0;
	for (uint32_t w = wave; w < dos.words; w += 4u) {
		const uint32_t s = 64u * w + lane;
		if (s >= sample_ct || (include && !((include[w] >> lane) & 1ull))) {
			continue;
		}
		const uint64_t u = DosageOrCall(row, row32, s);
		if (u != kNoDosage) {
			sum += u;
			ssq += u * u;
			nm++;
		}
	}
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
		lin[4ull * i + g] = l[g];
	}
	ac[i] = inc;
}

// One lane per sample, a slice of the scored variants per workgroup row (the shape of
// score.hip:k_score_accumulate); per variant the lane takes its explicit dosage through the
// affine map, or its call through the code table.
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
	__shared__ double s_ts[kStage][4];
	__shared__ double s_lin[kStage][4];
	__shared__ double s_w[kStage][NCOLS];
	__shared__ uint32_t s_ac[kStage];
	__shared__ uint32_t s_v[kStage];
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	const bool live = s < sample_ct;
	const uint32_t i_begin = blockIdx.y * slice_len;
	const uint32_t i_end = min(i_begin + slice_len, n_scored);
	double acc[NCOLS];
#pragma unroll
	for (int c = 0; c < NCOLS; c++) {
		acc[c] = 0.0;
	}
	double dsum = 0.0;
	uint32_t missed = 0;
	for (uint32_t base = i_begin; base < i_end; base += kStage) {
		const uint32_t cnt = min(kStage, i_end - base);
		__syncthreads();
		for (uint32_t k = threadIdx.x; k < cnt * 4u; k += 256u) {
			s_ts[k >> 2][k & 3] = ts[4ull * base + k];
			s_lin[k >> 2][k & 3] = lin[4ull * base + k];
		}
		for (uint32_t k = threadIdx.x; k < cnt * NCOLS; k += 256u) {
			const uint32_t c = k % NCOLS;
			s_w[k / NCOLS][c] = c < n_cols ? weights[static_cast<uint64_t>(base + k / NCOLS) * w_stride + c] : 0.0;
		}
		for (uint32_t k = threadIdx.x; k < cnt; k += 256u) {
			s_ac[k] = ac[base + k];
			s_v[k] = vlist[base + k];
		}
		__syncthreads();
		if (!live) {
			continue;
		}
		for (uint32_t k = 0; k < cnt; k++) {
			if (s_ac[k] == 0) {
				continue; // nobody observed, or no variance under center: the reference skips the variant
			}
			const uint32_t lv = s_v[k];
			const uint32_t *row32 = reinterpret_cast<const uint32_t *>(rows + static_cast<uint64_t>(lv) * pitch);
			const DosageRow row = RowOf(dos, lv);
			double x;
			bool explicit_dosage = false;
			uint32_t u = 0;
			if (row.present) {
				const uint32_t w = s >> 6, b = s & 63u;
				const uint64_t bits = row.present[w];
				if ((bits >> b) & 1ull) {
					explicit_dosage = true;
					u = row.values[row.rank[w] + static_cast<uint32_t>(__popcll(bits & ((1ull << b) - 1ull)))];
				}
			}
			if (explicit_dosage) {
				const double d = static_cast<double>(u) * 0x1p-14;
				x = ((s_lin[k][0] * d + s_lin[k][1]) - s_lin[k][2]) * s_lin[k][3];
			} else {
				const uint32_t g = (row32[s >> 4] >> (2u * (s & 15u))) & 3u;
				x = s_ts[k][g];
				missed += g == 3u;
			}
			dsum += x;
#pragma unroll
			for (int c = 0; c < NCOLS; c++) {
				acc[c] = fma(s_w[k][c], x, acc[c]);
			}
		}
	}
	if (live) {
#pragma unroll
		for (int c = 0; c < NCOLS; c++) {
			if (static_cast<uint32_t>(c) < n_cols) {
				unsafeAtomicAdd(score + static_cast<uint64_t>(s) * out_stride + c, acc[c]);
			}
		}
		if (dosage_sum && mode != 2) {
			unsafeAtomicAdd(dosage_sum + s, dsum);
		}
		if (miss && missed) {
			atomicAdd(miss + s, missed);
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

hipError_t LaunchDosageSums(const RowView &view, const DosageView &dos, uint32_t v0, const uint32_t *vlist,
                            uint32_t n_var, const uint64_t *include, uint64_t *out, hipStream_t stream) {
	if (n_var == 0) {
		return hipSuccess;
	}
	hipLaunchKernelGGL(k_dosage_sums, dim3(n_var), dim3(256), 0, stream, view.rows, view.pitch, view.sample_ct, dos, v0,
	                   vlist, include, out);
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
	const uint32_t sample_blocks = (view.sample_ct + 255) / 256;
	const uint32_t want_slices = (2048 + sample_blocks - 1) / sample_blocks;
	uint32_t slice_len = (n_scored + want_slices - 1) / want_slices;
	slice_len = ((slice_len + 63) / 64) * 64;
	uint32_t slices = (n_scored + slice_len - 1) / slice_len;
	if (slices > 65535u) {
		slices = 65535u;
		slice_len = (n_scored + slices - 1) / slices;
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

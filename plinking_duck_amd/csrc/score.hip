// score.hip -- plink_score's per-variant contribution tables and allele-count bookkeeping (gfx950).  The
// contraction of the packed rows with the weight columns itself runs on the int8 matrix cores: score_i8.hip.
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

// Non-finite weights (NaN, +-Inf) cannot be cut into fixed-point digits; the plan zeroes them for the matrix-core
// contraction and hands them here: the reference's own double arithmetic, score[s][c] += w * scored(s)
// (src/plink_score.cpp:621-651), term by term -- NaN where the scored value is 0 or w is NaN, +-Inf elsewhere,
// nothing for a sample or a variant the mode skips.  special[j] = {position in the plan's list, column, weight}.
__global__ __launch_bounds__(256) void k_score_nonfinite(const uint8_t *__restrict__ rows, uint64_t pitch,
                                                         uint32_t sample_ct, const uint32_t *__restrict__ vlist,
                                                         const double *__restrict__ ts, const uint32_t *__restrict__ ac,
                                                         const ScoreSpecial *__restrict__ special, uint32_t n_cols,
                                                         double *__restrict__ score) {
	const ScoreSpecial sp = special[blockIdx.y];
	const uint32_t s = blockIdx.x * 256u + threadIdx.x;
	if (s >= sample_ct) {
		return;
	}
	const uint32_t inc = ac[sp.pos];
	if ((inc & 0xffu) == 0) {
		return; // the variant is skipped (no observation, or zero variance in center mode)
	}
	const uint32_t code = (rows[static_cast<uint64_t>(vlist[sp.pos]) * pitch + (s >> 2)] >> (2u * (s & 3u))) & 3u;
	if (code == 3u && ((inc >> 8) & 0xffu) == 0) {
		return; // a missing call contributes only under mean imputation
	}
	atomicAdd(score + static_cast<uint64_t>(s) * n_cols + sp.col, sp.weight * ts[4ull * sp.pos + code]);
}

} // namespace

// ---------------------------------------------------------------------------
// launch wrappers
// ---------------------------------------------------------------------------

hipError_t LaunchScoreNonFinite(const RowView &view, const uint32_t *vlist, const double *ts, const uint32_t *ac,
                                const ScoreSpecial *special, uint32_t n_special, uint32_t n_cols, double *score,
                                hipStream_t stream) {
	if (n_special == 0 || view.sample_ct == 0) {
		return hipSuccess;
	}
	for (uint32_t j0 = 0; j0 < n_special; j0 += 65535u) { // grid.y limit
		const uint32_t nj = n_special - j0 < 65535u ? n_special - j0 : 65535u;
		hipLaunchKernelGGL(k_score_nonfinite, dim3((view.sample_ct + 255) / 256, nj), dim3(256), 0, stream, view.rows,
		                   view.pitch, view.sample_ct, vlist, ts, ac, special + j0, n_cols, score);
	}
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

} // namespace pgh

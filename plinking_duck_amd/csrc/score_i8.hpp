// score_i8.hpp -- plink_score's contraction on the int8 matrix cores (definitions in score_i8.hip).
#pragma once

#include "kernels.hpp"

#include <cstddef>

namespace pgh {

constexpr uint32_t kI8WeightDigits = 7; // base-256 digits per weight column: 54 bits below the column's largest coefficient
constexpr uint32_t kI8DosageDigits = 5; // NAMED_ALLELE_DOSAGE_SUM: terms are >= 0 and <= 3, 38 bits suffice
constexpr uint32_t kI8MaxCols = 17;     // weight columns per pass: 7 * 17 + 5 + 1 = 125 <= 128 digit columns (8 tiles)

struct ScoreI8Sizes {
	uint32_t n_tiles = 0;   // 64-variant tiles
	uint32_t n_tiles16 = 0; // 16-column digit tiles
	size_t bmat = 0, rowidx = 0, cols = 0, small = 0;
};
ScoreI8Sizes ScoreI8Bytes(uint32_t n_var, uint32_t n_cols);
uint32_t ScoreI8Tiles16(uint32_t n_cols);

// Device buffers of one prepared pass (sizes from ScoreI8Bytes):
//   bmat    digit bytes of both planes, tile by tile           (bmat bytes)
//   rowidx  resident row of every variant, padded to tiles     (rowidx bytes)
//   mult    double[16 n_tiles16], target uint32[16 n_tiles16]  (cols bytes together)
//   colmax  uint64[n_cols + 2], k0 double[n_cols + 2], scale_exp double[n_cols + 2], contiguous (small bytes)
struct ScoreI8Buffers {
	int8_t *bmat = nullptr;
	uint32_t *rowidx = nullptr;
	double *mult = nullptr;
	uint32_t *target = nullptr;
	unsigned long long *colmax = nullptr;
	double *k0 = nullptr;
	double *scale_exp = nullptr;
};

// Cuts the coefficients of n_cols (<= kI8MaxCols) weight columns, of the dosage sum (td) and of the
// missing-call count into digits.  ts / td: the per-variant tables of LaunchScoreTables (four doubles each);
// ac: its allele-count increments; count_missing: ALLELE_CT loses 2 per missing call (every mode but mean
// imputation).  Runs once per plan.
hipError_t LaunchScoreI8Prepare(const uint32_t *vlist, uint32_t n_var, const double *weights, uint32_t w_stride,
                                uint32_t n_cols, const double *ts, const double *td, const uint32_t *ac,
                                bool count_missing, const ScoreI8Buffers &b, hipStream_t stream);

// score[s * out_stride + c] += sum_v W[v][c] T_v[g(v,s)] for the prepared columns; dosage_sum[s] += the same with
// td and unit weights (NULL: not wanted); missing_ct[s] += missing calls of s at the variants that count
// (NULL: not wanted).  Outputs are raw-sample order and are added to.
hipError_t LaunchScoreI8(const RowView &view, uint32_t n_var, uint32_t n_cols, const ScoreI8Buffers &b, double *score,
                         uint32_t out_stride, double *dosage_sum, uint32_t *missing_ct, hipStream_t stream);

} // namespace pgh

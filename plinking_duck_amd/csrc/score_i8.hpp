// score_i8.hpp -- plink_score's contraction on the int8 matrix cores (definitions in score_i8.hip).
#pragma once

#include "kernels.hpp"

#include <cstddef>

namespace pgh {

constexpr uint32_t kI8WeightDigits = 7; // base-256 digits per weight column: 54 bits below the column's largest coefficient
constexpr uint32_t kI8DosageDigits = 5; // NAMED_ALLELE_DOSAGE_SUM: terms are >= 0 and <= 3, 38 bits suffice
constexpr uint32_t kI8MaxCols = 22;     // weight columns per pass: 7 * 22 + 5 + 1 = 160 digit columns (10 tiles)
constexpr uint32_t kI8MaxColsBare = 22; // ... without the dosage-sum / missing-count columns: 7 * 22 = 154

// What the two integer planes are multiplied with.  kI8Tables: per-variant contribution tables (ts / td), both
// planes; kI8CodePlane / kI8MissingPlane: the bare 2-bit code resp. the missing indicator times the weight, one
// plane only -- the two products plink_pca's X G1 is made of (its normalisation is per OUTPUT row).
enum { kI8Tables = 0, kI8CodePlane = 1, kI8MissingPlane = 2 };

struct ScoreI8Sizes {
	uint32_t n_tiles = 0;   // 64-variant tiles
	uint32_t n_tiles16 = 0; // 16-column digit tiles
	size_t bmat = 0, rowidx = 0, cols = 0, small = 0;
};
ScoreI8Sizes ScoreI8Bytes(uint32_t n_var, uint32_t n_cols, bool extras);
uint32_t ScoreI8Tiles16(uint32_t n_cols, bool extras);

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
	// optional: the tile-major copy of the listed rows (LaunchScoreI8TileMajor over the SAME list and view the
	// pass was prepared with); used by the shapes ScoreI8UsesTiles names, ignored by the others
	const uint8_t *tiled = nullptr;
};

// The many-column shapes (five 16-column digit tiles and more: a lane holds four samples, a workgroup a 128-byte
// stripe of every row) stream the genotype bytes of a 64-variant tile as one contiguous 8 KB image per workgroup
// when the caller keeps a TILE-MAJOR copy of the listed rows: ScoreI8TiledBytes bytes (the rows' own size, rounded
// up to 128-byte stripes and 128-variant pairs of tiles), filled by LaunchScoreI8TileMajor in one pass over the
// rows.  Worth it where the same list is contracted several times (plink_pca: every pass; a kept score plan).
bool ScoreI8UsesTiles(uint32_t n_cols, bool extras);
size_t ScoreI8TiledBytes(uint32_t n_var, uint32_t sample_ct);
hipError_t LaunchScoreI8TileMajor(const RowView &view, const uint32_t *vlist, uint32_t n_var, uint8_t *out,
                                  hipStream_t stream);

// Cuts the coefficients of n_cols (<= kI8MaxCols) weight columns, of the dosage sum (td) and of the
// missing-call count into digits.  ts / td: the per-variant tables of LaunchScoreTables (four doubles each);
// ac: its allele-count increments; count_missing: ALLELE_CT loses 2 per missing call (every mode but mean
// imputation).  Runs once per plan.
// extras: also the dosage-sum and missing-count columns (n_cols <= kI8MaxCols; without them kI8MaxColsBare,
// td / ac unused).  table_mode: kI8Tables, or one bare plane (ts / td unused).
hipError_t LaunchScoreI8Prepare(const uint32_t *vlist, uint32_t n_var, const double *weights, uint32_t w_stride,
                                uint32_t n_cols, const double *ts, const double *td, const uint32_t *ac,
                                bool count_missing, bool extras, int table_mode, const ScoreI8Buffers &b,
                                hipStream_t stream);

// score[s * out_stride + c] += sum_v W[v][c] T_v[g(v,s)] for the prepared columns; dosage_sum[s] += the same with
// td and unit weights (NULL: not wanted); missing_ct[s] += missing calls of s at the variants that count
// (NULL: not wanted).  Outputs are raw-sample order and are added to.
hipError_t LaunchScoreI8(const RowView &view, uint32_t n_var, uint32_t n_cols, bool extras, int planes,
                         const ScoreI8Buffers &b, double *score, uint32_t out_stride, double *dosage_sum,
                         uint32_t *missing_ct, hipStream_t stream);

} // namespace pgh

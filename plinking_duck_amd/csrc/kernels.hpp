// kernels.hpp -- launch wrappers of the gfx950 kernels (definitions in tally.hip, unpack.hip, score.hip, pca.hip,
// pca_i8.hip, reduce.hip; the int8 contraction has its own header, score_i8.hpp).
// All pointers are device pointers unless named h_*.  Every wrapper only
// enqueues work on `stream` and returns the hipError_t of the launch.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pgh {

// Geometry of the resident genotype matrix: row r starts at rows + r*pitch.
struct RowView {
	const uint8_t *rows;   // first resident row
	uint64_t pitch;        // bytes between rows (multiple of 16)
	uint32_t sample_ct;    // raw N
	uint32_t record_bytes; // ceil(N/4)
};

// ---- synthetic generator (host twin in synth.hpp) -------------------------
hipError_t LaunchSynthFill(uint8_t *rows, uint64_t pitch, uint32_t sample_ct, uint32_t variant_begin,
                           uint32_t variant_ct, uint64_t seed, uint32_t miss_threshold, hipStream_t stream);

// zero genotype bits past sample_ct and the pad bytes up to pitch
hipError_t LaunchSanitizeTail(uint8_t *rows, uint64_t pitch, uint32_t sample_ct, uint32_t variant_ct,
                              hipStream_t stream);

// ---- genotype-class tally --------------------------------------------------
// out[i] = {hom_ref, het, hom_alt, missing} of row (vlist ? vlist[i] : v_first + i)
// over the samples selected by mask2 (NULL = all).  mask2: one row of `pitch`
// bytes holding 01 in the 2-bit slot of every included sample.  n_eff = number of
// included samples.
hipError_t LaunchCounts(const RowView &view, uint32_t v_first, const uint32_t *vlist, uint32_t v_count,
                        const uint8_t *mask2, uint32_t n_eff, uint32_t *out, hipStream_t stream);

// alt_freq / obs_ct per variant from its counts (NaN where obs == 0)
hipError_t LaunchFreqFromCounts(const uint32_t *counts, uint32_t n, double *alt_freq, int32_t *obs_ct,
                                hipStream_t stream);

// ---- per-sample missing tally ----------------------------------------------
// out[s] = number of rows in [v_first, v_first + v_count) where sample s is missing
// (out: uint32[N]).  scratch: MissingPerSampleScratchBytes() bytes of device memory
// holding one partial row per slice of variants.
size_t MissingPerSampleScratchBytes(uint32_t record_bytes, uint32_t v_count);
// row_flags (optional, one per row): rows whose low byte is zero are left out.
hipError_t LaunchMissingPerSample(const RowView &view, uint32_t v_first, const uint32_t *vlist, uint32_t v_count,
                                  const uint32_t *row_flags, uint32_t *scratch, uint32_t *out, hipStream_t stream);

// The same column tally for any one genotype code: 1 het, 2 hom-alt, 3 missing.
hipError_t LaunchClassPerSample(const RowView &view, int genotype_class, uint32_t v_first, const uint32_t *vlist,
                                uint32_t v_count, const uint32_t *row_flags, uint32_t *scratch, uint32_t *out,
                                hipStream_t stream);

// All three non-reference codes in ONE pass: out[c * out_stride + s] for c = het, hom-alt, missing.
// scratch: ClassCounts3ScratchBytes() bytes of device memory (byte-counter slabs).
size_t ClassCounts3ScratchBytes(uint32_t record_bytes);
hipError_t LaunchClassCounts3(const RowView &view, uint32_t v_first, const uint32_t *vlist, uint32_t v_count,
                              uint8_t *scratch, uint32_t *out, uint32_t out_stride, hipStream_t stream);

// Both reductions in one pass over [v_first, v_first + v_count): counts[i] = class tallies
// of row i (all samples), missing_per_sample[s] as above.  Same scratch size as
// LaunchMissingPerSample.  accumulate: missing_per_sample[s] += this range's tally instead of being
// overwritten (a pass that walks the matrix batch by batch, api_tally.cpp).
hipError_t LaunchFusedTally(const RowView &view, uint32_t v_first, uint32_t v_count, uint32_t *scratch,
                            uint32_t *counts, uint32_t *missing_per_sample, hipStream_t stream,
                            bool accumulate = false);

// ---- 2-bit -> int8 unpack --------------------------------------------------
// out row i: int8[N] (+pad to out_pitch, multiple of 16) with missing -> fill;
// validity row i (optional): ceil(N/64) uint64 words, bit = non-missing.
hipError_t LaunchUnpack(const RowView &view, uint32_t v_first, uint32_t v_count, int8_t *out, uint64_t out_pitch,
                        uint64_t *validity, int8_t fill, hipStream_t stream);
// Measurement aid: the unpack's traffic shape (16 B read -> 64 B + 8 B written per lane) with no arithmetic;
// src: n_vec x 16 B, dst: n_vec x 64 B, val: n_vec x 8 B.
hipError_t LaunchUnpackShapeProbe(const void *src, size_t n_vec, void *dst, void *val, hipStream_t stream);
// Subset form: sel[k] = raw index of the k-th included sample, n_out entries.
hipError_t LaunchUnpackSubset(const RowView &view, uint32_t v_first, uint32_t v_count, const uint32_t *sel,
                              uint32_t n_out, int8_t *out, uint64_t out_pitch, uint64_t *validity, int8_t fill,
                              hipStream_t stream);
// Sample-major form: out[k - k_first][j] (rows of out_stride bytes) = call of output sample k at variant
// vlist[j] (local indices), missing -> fill, for k in [k_first, k_first + k_count); k_first a multiple of 64.
// sel == NULL: output samples are the raw samples.
hipError_t LaunchUnpackTransposed(const RowView &view, const uint32_t *vlist, uint32_t n_var, const uint32_t *sel,
                                  uint32_t k_first, uint32_t k_count, int8_t *out, uint64_t out_stride, int8_t fill,
                                  hipStream_t stream);

// ---- plink_score ------------------------------------------------------------
// Per scored variant i, from its counts: the scored-dosage table ts[i][g], the
// dosage-sum table td[i][g] and the allele-count increments ac[i] (byte 0: g<3,
// byte 1: g==3).  Skipped variants get all-zero tables.
hipError_t LaunchScoreTables(const uint32_t *counts, const uint8_t *flip, uint32_t n_scored, int mode, double *ts,
                             double *td, uint32_t *ac, hipStream_t stream);
// One non-finite weight of a plan: position in the plan's variant list, weight column, the weight itself.
struct ScoreSpecial {
	uint32_t pos, col;
	double weight;
};
// score[s * n_cols + col] += weight * ts[pos][call of s] in plain double arithmetic, for the plan's non-finite
// weights (vlist: local row of each list position; ts / ac: the plan's tables)
hipError_t LaunchScoreNonFinite(const RowView &view, const uint32_t *vlist, const double *ts, const uint32_t *ac,
                                const ScoreSpecial *special, uint32_t n_special, uint32_t n_cols, double *score,
                                hipStream_t stream);
// allele_ct[s] = sum_i (ac[i] & 0xff) - 2 * miss[s]   (miss == NULL: every sample gets the full sum)
hipError_t LaunchAlleleCt(const uint32_t *ac, uint32_t n_scored, const uint32_t *miss, uint32_t sample_ct,
                          uint32_t *allele_ct, hipStream_t stream);

// ---- plink_pca ----------------------------------------------------------------
// ts[i] = {(0-c)is, (1-c)is, (2-c)is, 0} (NormalizeGenotypes)
hipError_t LaunchNormTables(const double *center, const double *inv_stdev, uint32_t n, double *ts,
                            hipStream_t stream);
// zero the rows of samples whose slot in mask2 is clear
hipError_t LaunchMaskRows(double *m, uint32_t n_rows, uint32_t stride, uint32_t n_cols, const uint8_t *mask2,
                          hipStream_t stream);
hipError_t LaunchScale(double *m, uint64_t n, double f, hipStream_t stream);

// tall-skinny FP64 helpers (row-major, leading dimensions in elements)
// C[na x nb] += A^T B over m rows (C zeroed by the caller)
hipError_t LaunchTallGram(const double *A, uint32_t lda, uint32_t na, const double *B, uint32_t ldb, uint32_t nb,
                          uint64_t m, double *C, uint32_t ldc, hipStream_t stream);
// out = beta * B + alpha * A C   (A: m x na, C: na x nb, B/out: m x nb; out may alias B, not A)
hipError_t LaunchTallTimesSmall(const double *A, uint32_t lda, uint32_t na, const double *C, uint32_t ldc, uint32_t nb,
                                double alpha, double beta, const double *B, uint32_t ldb, double *out, uint32_t ldo,
                                uint64_t m, hipStream_t stream);
hipError_t LaunchCopyCols(const double *src, uint32_t ld_src, double *dst, uint32_t ld_dst, uint32_t n, uint64_t m,
                          hipStream_t stream);

// ---- plink_pca on the int8 contraction (pca_i8.hip) -----------------------------
// The packed matrix of the listed variants, sample-major: out row s = calls of sample s at vlist[0..n_var), 2 bits
// each, rows TransposedPitch(n_var) bytes apart (zero padded); out holds view.sample_ct rows.
uint64_t TransposedPitch(uint32_t n_var);
hipError_t LaunchTranspose2bit(const RowView &view, const uint32_t *vlist, uint32_t n_var, uint8_t *out,
                               hipStream_t stream);
hipError_t LaunchIota(uint32_t *p, uint32_t n, hipStream_t stream);
// out[c] = sum_r m[r * stride + c]
hipError_t LaunchColumnSums(const double *m, uint64_t n_rows, uint32_t stride, uint32_t n_cols, double *out,
                            hipStream_t stream);
// y[v * y_stride + c] = s_v (a[v][c] - 3 mm[v][c]) - c_v s_v (colsum[c] - mm[v][c]);  a, mm: n_var x n_cols, dense
hipError_t LaunchPcaCombine(const double *a, const double *mm, const double *colsum, const double *center,
                            const double *inv_stdev, uint64_t n_var, uint32_t n_cols, double *y, uint32_t y_stride,
                            hipStream_t stream);

// ---- shard-group combine (reduce.hip) -----------------------------------------
// dst[i] += src[i]; both 16-byte aligned device buffers of the current device
hipError_t LaunchAddF64(double *dst, const double *src, uint64_t n, hipStream_t stream);
hipError_t LaunchAddU32(uint32_t *dst, const uint32_t *src, uint64_t n, hipStream_t stream);

// ---- HWE --------------------------------------------------------------------
// order_scratch: HweOrderScratchBytes(n) bytes the launch may use on `stream`, or NULL.  With it, batches of
// kHweOrderMin variants and more are tested in order of their minor-allele fraction (waves of equally long walks:
// tally.hip); the results are the same doubles at the same positions either way.
constexpr uint32_t kHweOrderMin = 8192;
size_t HweOrderScratchBytes(uint32_t n);
hipError_t LaunchHweBatch(const uint32_t *counts, uint32_t n, uint32_t midp, double *ln_p, hipStream_t stream,
                          void *order_scratch = nullptr);
// chrX: strata[i] = {female_hets, female_hom1, female_hom2, male1, male2}; one workgroup per variant
hipError_t LaunchHweXchrBatch(const int32_t *strata, uint32_t n, uint32_t midp, double *ln_p, hipStream_t stream);

} // namespace pgh

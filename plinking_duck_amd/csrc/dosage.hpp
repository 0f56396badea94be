// dosage.hpp -- resident dosage tracks and the kernels that read them (gfx950).
#pragma once

#include "kernels.hpp"

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pgh {

//! A dataset's explicit dosages in HBM, in the file's own bit-array form (vrtype 0x60;
//! the 0x20 list and 0x40 dense forms are brought to it at pgh_open): per dosage-bearing
//! variant a presence bit per sample, and the present samples' 16-bit values (0..32768 =
//! 0..2 ALT copies) packed in sample order.  `rank` holds the number of presence bits
//! before each 64-sample word, so any lane finds its value with one popcount.
struct DosageView {
	const int32_t *row_of = nullptr;   // per resident variant: row below, or -1 (hardcalls only)
	const uint64_t *present = nullptr; // rows x words
	const uint32_t *rank = nullptr;    // rows x words
	const uint64_t *val_off = nullptr; // rows + 1: where each row's values start, and where the last one ends
	const uint16_t *values = nullptr;  // 16-byte aligned, padded by 16 bytes: rows are streamed in aligned 16-byte loads
	uint32_t words = 0; // ceil(sample_ct / 64)
	// Entry records of the sparse tracks (LaunchDosageRecords; NULL until the first plink_score plan that needs
	// them): rec_off[rows + 1], a row's records at rec[rec_off[r] .. rec_off[r + 1]) -- none for the dense tracks.
	const uint32_t *rec = nullptr;
	const uint64_t *rec_off = nullptr;
};

//! One staged run of records (decode.hpp:DecodeBatch) whose dosage tracks go into the resident form.
//! All pointers are device pointers; the scratch arrays live for the batch only.
struct DosageIngest {
	const uint8_t *bytes; // the records' file bytes (16 readable zero bytes follow bytes_len)
	uint64_t bytes_len;
	const uint64_t *rec_begin; // [n + 1]
	const uint8_t *vrtype;     // [n]
	const uint64_t *aux_at;    // [n] first byte after each record's main track (written by the record decode)
	const int32_t *dos_row;    // [n] row of the record in the arrays below, or -1: no dosage track
	const uint8_t *rows;       // the finished 2-bit rows (a phase track's length depends on the het count)
	uint64_t pitch;
	uint32_t row0, variant0, n, sample_ct, id_bytes;
	uint64_t *present; // rows x words, zero-filled
	uint32_t *rank;
	uint32_t words;
	uint64_t *val_off; // [rows]
	uint16_t *values;
	uint64_t capacity; // values the allocation holds
	uint64_t *total;   // running count of stored values (one device counter per dataset)
	uint32_t *count;   // [n] scratch: explicit dosages of each record
	uint64_t *track;   // [n] scratch: offset into bytes of each record's value section
	int *error;        // set (once) to 1 + the variant index of a malformed record
};

//! Locates every dosage track of the batch behind its record's main (and phase) track, writes presence bits,
//! ranks, value offsets and values.  first_row / n_rows: the contiguous run of dosage rows the batch fills.
hipError_t LaunchDosageIngest(const DosageIngest &batch, uint32_t first_row, uint32_t n_rows, hipStream_t stream);

//! rank[r][w] = number of presence bits of row r before word w
hipError_t LaunchDosageRank(const uint64_t *present, uint32_t rows, uint32_t words, uint32_t *rank,
                            hipStream_t stream);

//! Seeded synthetic tracks: presence bits Bernoulli(rate) and values uniform on 0..32768, both keyed by
//! (seed, variant, sample) -- the draws pgh_synth_write_dosage_files makes; row totals.
hipError_t LaunchSynthDosageBits(uint64_t *present, uint32_t rows, uint32_t words, uint32_t sample_ct, uint32_t variant0,
                                 uint64_t seed, double rate, hipStream_t stream);
hipError_t LaunchDosageRowTotals(const uint64_t *present, const uint32_t *rank, uint32_t rows, uint32_t words,
                                 uint64_t *totals, hipStream_t stream);
hipError_t LaunchSynthDosageValues(const uint64_t *present, const uint32_t *rank, const uint64_t *val_off,
                                   uint16_t *values, uint32_t rows, uint32_t words, uint32_t sample_ct,
                                   uint32_t variant0, uint64_t seed, hipStream_t stream);

//! PgrGetDCounts (src/plink_freq.cpp:475): per variant {sum of dosages, sum of squares, samples with a
//! dosage or a call} on the 16384-per-copy scale, over the included samples; hardcalls count as 0 / 16384 / 32768.
//! Variants are v0 + i, or vlist[i] when vlist != NULL (indices local to the resident range).
hipError_t LaunchDosageSums(const RowView &view, const DosageView &dos, uint32_t v0, const uint32_t *vlist,
                            uint32_t n_var, const uint64_t *include, uint64_t *out, hipStream_t stream);

//! PgrGetD + Dosage16ToDoublesMinus9 (src/pgen_reader.cpp:694, src/plink_score.cpp:587): one double per
//! output sample, -9 where the sample has neither a dosage nor a call.  sel == NULL: every sample.
hipError_t LaunchDosageUnpack(const RowView &view, const DosageView &dos, uint32_t v0, const uint32_t *vlist,
                              uint32_t n_var, const uint32_t *sel, uint32_t n_out, double *out, uint64_t out_stride,
                              hipStream_t stream);

//! The sample-major form: out[k - k_first][j] = dosage of output sample k at variant vlist[j] (local indices).
hipError_t LaunchDosageUnpackTransposed(const RowView &view, const DosageView &dos, const uint32_t *vlist, uint32_t n_var,
                                        const uint32_t *sel, uint32_t k_first, uint32_t k_count, double *out,
                                        uint64_t out_stride, hipStream_t stream);

//! Per scored variant, from LaunchDosageSums' output: the contribution tables of the hardcall codes
//! (ts / td as LaunchScoreTables writes them, with the mean taken over dosages), the affine map of an
//! explicit dosage lin = {scale, shift}: contribution = (d * scale + shift), and the ALLELE_CT increment.
hipError_t LaunchScoreTablesDosage(const uint64_t *sums, const uint8_t *flip, uint32_t n_scored, int mode, double *ts,
                                   double *td, double *lin, uint32_t *ac, hipStream_t stream);

//! plink_score over variants that carry dosages (src/plink_score.cpp:586-652): adds into score / dosage_sum and
//! counts, per sample, the scored variants at which it has neither dosage nor call (miss, for ALLELE_CT).
hipError_t LaunchScoreDosage(const RowView &view, const DosageView &dos, const uint32_t *vlist, uint32_t n_scored,
                             const double *weights, uint32_t w_stride, uint32_t n_cols, const double *ts,
                             const double *lin, const uint32_t *ac, int mode, double *score, uint32_t out_stride,
                             double *dosage_sum, uint32_t *miss, hipStream_t stream);

//! The single-column form in two steps: the caller first runs the hardcall contraction (LaunchScoreI8)
//! over the same variants with the same tables; this adds (affine(dosage) - ts[call]) for the explicit entries
//! and takes the samples that have a dosage but a missing call back out of `miss` (NULL: not tracked).
hipError_t LaunchScoreDosageFix(const RowView &view, const DosageView &dos, const uint32_t *vlist, uint32_t n_scored,
                                const double *weights, uint32_t w_stride, const double *ts, const double *lin,
                                const uint32_t *ac, double *score, uint32_t out_stride, double *dosage_sum,
                                uint32_t *miss, hipStream_t stream);

//! The entry records of the rows with rec_off[r + 1] > rec_off[r] (dosage.hip: value, tile element and hardcall of
//! every explicit dosage in 32 bits, round-major inside each 4096-sample tile).  row_variant[r]: the resident row
//! of dosage row r.  dos.present / rank / val_off / values are read; dos.rec / rec_off are not.
hipError_t LaunchDosageRecords(const RowView &view, const DosageView &dos, uint32_t rows, const uint32_t *row_variant,
                               const uint64_t *rec_off, uint32_t *rec, hipStream_t stream);

//! LaunchScoreDosageFix's contract over the entry records (dos.rec): every listed variant's row must have them.
hipError_t LaunchScoreDosageRecords(const RowView &view, const DosageView &dos, const uint32_t *vlist, uint32_t n_scored,
                                    const double *weights, uint32_t w_stride, const double *ts, const double *lin,
                                    const uint32_t *ac, double *score, uint32_t out_stride, double *dosage_sum,
                                    uint32_t *miss, hipStream_t stream);

//! Variants at which EVERY sample has an explicit dosage: contribution = affine map of the sample's value, read
//! straight from the value run (no presence bits, ranks or calls).  Nobody is missing at such a variant.
hipError_t LaunchScoreDosageFull(const RowView &view, const DosageView &dos, const uint32_t *vlist, uint32_t n_scored,
                                 const double *weights, uint32_t w_stride, uint32_t n_cols, const double *lin,
                                 const uint32_t *ac, int mode, double *score, uint32_t out_stride, double *dosage_sum,
                                 hipStream_t stream);

} // namespace pgh

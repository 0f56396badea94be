/*
 * pgenhip.h -- C ABI of libpgenhip: the MI355X (gfx950) replacement for the
 * plink-ng pgenlib calls on PlinkingDuck's .pgen decode-and-analyse hot path.
 *
 * Every entry point names the reference interface it replaces (file:line into
 * teaguesterling/plinking_duck).  Conventions, all taken from pgenlib's own:
 *   - every call returns an int status (PGH_OK == 0) and, where it can fail for
 *     a reason worth reporting, writes a NUL-terminated message into a caller
 *     buffer of PGH_ERRBUF_LEN bytes (pgenlib: PglErr + errstr_buf[kPglErrstrBufBlen]);
 *   - no exception, torch type or C++ type crosses this boundary;
 *   - the caller owns every output buffer; handles are opaque;
 *   - a pgh_dataset is immutable after creation and may be read concurrently
 *     by any number of scan threads; a pgh_reader belongs to one thread
 *     (pgenlib: one PgenReader per thread, src/plink_freq.cpp:342-390);
 *   - sample subsets follow pgenlib semantics: a bitmask over the raw samples,
 *     outputs compacted to the included samples in ascending file order
 *     (src/plink_common.cpp:1222-1250).
 *
 * Pointers named d_* are device (HBM) pointers owned by the caller; `stream`
 * is a hipStream_t passed as void* (NULL = HIP's default stream).  The
 * *_dev entry points only enqueue work; the host-buffer forms run on a stream
 * the library keeps for the calling thread, synchronise and copy the result
 * back.  Entry points that need device scratch keep one block per calling
 * thread, device and stream (grown on demand, released when the thread ends);
 * nothing is allocated from HIP's stream-ordered pool (hipMallocAsync), which
 * was seen to lose kernel-written data on this runtime.
 */
#ifndef PGENHIP_H_
#define PGENHIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PGH_ERRBUF_LEN 256

enum {
	PGH_OK = 0,
	PGH_ERR_OPEN = 1,        /* file cannot be opened/read      (IOException at the shell) */
	PGH_ERR_FORMAT = 2,      /* malformed or unsupported .pgen  (IOException)              */
	PGH_ERR_ARG = 3,         /* bad argument / out of range     (InvalidInputException)    */
	PGH_ERR_DEVICE = 4,      /* HIP runtime failure             (IOException)              */
	PGH_ERR_NOMEM = 5,
	PGH_ERR_UNSUPPORTED = 6  /* reserved: a track kind no path decodes (none at present: multiallelic tracks are stepped over, phased-dosage tracks are not read, as PgrGetD does not read them) */
};

typedef struct pgh_dataset pgh_dataset; /* packed 2-bit genotype matrix resident in HBM */
typedef struct pgh_subset pgh_subset;   /* sample-include mask, staged on the device    */
typedef struct pgh_reader pgh_reader;   /* per-scan-thread view (stream + staging)      */

typedef struct pgh_info {
	uint32_t raw_variant_ct;   /* pgfi.raw_variant_ct  (src/plink_freq.cpp:184) */
	uint32_t raw_sample_ct;    /* pgfi.raw_sample_ct   (src/plink_freq.cpp:185) */
	uint32_t variant_begin;    /* first variant resident on this device         */
	uint32_t variant_end;      /* one past the last resident variant            */
	uint32_t has_dosage;       /* gflags & kfPgenGlobalDosagePresent (src/plink_freq.cpp:201) */
	uint32_t has_phase;        /* gflags & kfPgenGlobalHardcallPhasePresent     */
	uint32_t max_record_bytes; /* max_vrec_width (src/plink_freq.cpp:193-197)   */
	uint32_t record_bytes;     /* ceil(N/4): bytes of one normalised 2-bit record */
	uint64_t pitch_bytes;      /* device row stride of one record               */
	uint32_t vrtype_hist[8];   /* number of records per main-track type (vrtype & 7) */
	int32_t device;            /* HIP device ordinal                            */
	uint32_t dosage_variant_ct; /* resident variants that carry a dosage track (vrtype & 0x60) */
	uint64_t dosage_value_ct;   /* explicit dosages held for them                */
} pgh_info;

/* ---- library / device --------------------------------------------------- */

/* Version string "pgenhip <n> gfx950". */
const char *pgh_version(void);
/* Number of visible HIP devices (0 without a GPU).  Never fails. */
int pgh_device_count(void);
/* Select the device used by datasets created afterwards from this thread. */
int pgh_set_device(int device, char *errbuf);

/* ---- dataset lifecycle --------------------------------------------------- */

/* Replaces PreinitPgfi + PgfiInitPhase1 + PgfiInitPhase2 + PgrInit
 * (src/plink_freq.cpp:168-208,344-390; same sequence in pgen_reader.cpp:227-266,
 * plink_hardy/missing/score/pca).  Parses the header and record tables, expands
 * every record of [variant_begin, variant_end) to a plain 2-bit row and leaves
 * the rows resident in HBM.  variant_end == UINT32_MAX means "to the last
 * variant".  pgi_path may be NULL (then "<pgen_path>.pgi" is tried for mode 0x20). */
int pgh_open(const char *pgen_path, const char *pgi_path, uint32_t variant_begin, uint32_t variant_end,
             pgh_dataset **out, char *errbuf);

/* Header probe only: no device work, works without a GPU (bind-time use:
 * src/plink_freq.cpp:168-208). */
int pgh_probe(const char *pgen_path, const char *pgi_path, pgh_info *out, char *errbuf);

/* The host half of pgh_open on its own: expand the records of [variant_begin,
 * variant_end) to plain 2-bit rows in HOST memory (row r at rows + r*row_stride,
 * row_stride >= ceil(N/4), pad bytes zeroed).  No device work; used by ingest
 * pipelines that stage rows themselves and by the CPU-only tests. */
int pgh_normalize_range_host(const char *pgen_path, const char *pgi_path, uint32_t variant_begin,
                             uint32_t variant_end, uint8_t *rows, size_t row_stride, char *errbuf);

/* Dataset over caller-supplied plain 2-bit rows in HOST memory (row v at
 * rows + v*row_stride, ceil(N/4) meaningful bytes each). */
int pgh_from_host_rows(const uint8_t *rows, size_t row_stride, uint32_t variant_ct, uint32_t sample_ct,
                       pgh_dataset **out, char *errbuf);

/* Seeded synthetic dataset written directly into HBM (BASELINE.md section 3:
 * p_v ~ U(0.01,0.5), g ~ Binomial(2,p_v), missing with probability
 * missing_rate).  Variant v of the generator lands in row v - variant_begin, so
 * ranks that own disjoint variant ranges hold slices of one global matrix. */
int pgh_synth_create(uint32_t variant_begin, uint32_t variant_end, uint32_t sample_ct, uint64_t seed,
                     double missing_rate, pgh_dataset **out, char *errbuf);
/* The same fileset with a dosage track (vrtype 0x60: presence bits + uint16 values) behind every record:
 * each sample explicit with probability dosage_rate, values uniform on 0..32768.  Ingest benchmark input. */
int pgh_synth_write_dosage_files(const char *prefix, uint32_t variant_ct, uint32_t sample_ct, uint64_t seed,
                                 double missing_rate, double dosage_rate, char *errbuf);
/* Gives every resident variant of a dataset without dosage tracks a seeded synthetic one:
 * each sample carries an explicit dosage with probability `rate`, values uniform on 0..32768.
 * Benchmark input of the shape `plink2 --import-dosage` leaves behind (vrtype 0x60). */
int pgh_synth_add_dosage(pgh_dataset *ds, double rate, uint64_t seed, char *errbuf);
/* The same generator on the host: one record (ceil(N/4) bytes) of variant v. */
int pgh_synth_record_host(uint32_t v, uint32_t sample_ct, uint64_t seed, double missing_rate, uint8_t *out);
/* Writes <prefix>.pgen (mode 0x10, vrtype-0 records), .pvar and .psam. */
int pgh_synth_write_files(const char *prefix, uint32_t variant_ct, uint32_t sample_ct, uint64_t seed,
                          double missing_rate, char *errbuf);

/* Copy resident rows [v_begin, v_end) back to HOST memory as plain 2-bit rows
 * (row r at rows + r*row_stride, ceil(N/4) bytes each). */
int pgh_copy_rows_to_host(const pgh_dataset *ds, uint32_t v_begin, uint32_t v_end, uint8_t *rows, size_t row_stride,
                          char *errbuf);

int pgh_get_info(const pgh_dataset *ds, pgh_info *out);
/* Device pointer of resident row 0 (pitch in pgh_info.pitch_bytes). */
const void *pgh_device_rows(const pgh_dataset *ds);
/* CleanupPgr + CleanupPgfi (src/plink_freq.cpp:109-115). */
void pgh_close(pgh_dataset *ds);

/* ---- shard groups: one process, several devices -------------------------------
 * The reference parallelises inside ONE process and merges per-thread partial sums under a mutex
 * (src/plink_score.cpp:657-664, src/plink_missing.cpp:614-619, src/plink_pca.cpp:940-954).  A shard group is
 * the same shape with devices in the place of threads: contiguous, ascending variant ranges of one file, one
 * resident dataset per device, behind ONE pgh_dataset handle.  Every host-buffer entry point of this header
 * accepts a group handle: per-variant outputs (pgh_counts_range, pgh_unpack_range, pgh_dosage_*, the pgh_get_*
 * reader calls) are filled shard by shard with no exchange; per-sample outputs are reduced per shard on its device
 * and merged across devices -- pgh_score's partials by an RCCL reduce onto the first shard, pgh_pca's by an RCCL
 * all-reduce per pass, each on the shards' own streams over xGMI (one communicator per shard, made at the first
 * collective; shards that share a device fall back to device-to-device copies and a sum on the first shard's
 * device), pgh_missing_per_sample / pgh_sample_counts (4 bytes per sample) on the host; pgh_ld_pairs computes the
 * pairs that straddle a shard boundary on a scratch dataset built from the rows they name.  Subsets and readers
 * created on a group handle are groups themselves.  The *_dev / plan entry points and pgh_device_rows take one
 * device's dataset: hand them pgh_shard(group, k).
 *
 * pgh_open_sharded: pgh_open of n_devices near-equal ranges of [variant_begin, variant_end), concurrently, shard k
 * on devices[k] (a device may be named more than once).  pgh_group_create: a group over datasets the caller made
 * (pgh_open / pgh_synth_create / pgh_from_host_rows, each with pgh_set_device in effect); it takes ownership:
 * pgh_close(group) closes them. */
int pgh_open_sharded(const char *pgen_path, const char *pgi_path, uint32_t variant_begin, uint32_t variant_end,
                     const int *devices, uint32_t n_devices, pgh_dataset **out, char *errbuf);
int pgh_group_create(pgh_dataset *const *shards, uint32_t n_shards, pgh_dataset **out, char *errbuf);
/* 1 when the group's per-sample merges run as RCCL collectives (one communicator per shard, made by this call if
 * the group had none yet: shards on distinct devices and librccl loadable), 0 when they use device-to-device copies
 * (shards sharing a device, PGH_GROUP_RCCL=0) or ds is not a group. */
int pgh_group_uses_rccl(const pgh_dataset *ds);
/* 0 for a plain dataset. */
uint32_t pgh_shard_count(const pgh_dataset *ds);
const pgh_dataset *pgh_shard(const pgh_dataset *ds, uint32_t k);

/* ---- sample subsets ------------------------------------------------------ */

/* Replaces BuildSampleSubset / PgrSetSampleSubsetIndex (src/plink_common.cpp:1222-1250,
 * src/plink_freq.cpp:393-397).  sample_include: ceil(N/64) words, bit s = sample s kept. */
int pgh_subset_create(const pgh_dataset *ds, const uint64_t *sample_include, pgh_subset **out, char *errbuf);
uint32_t pgh_subset_size(const pgh_subset *ss);
void pgh_subset_destroy(pgh_subset *ss);

/* ---- batched device calls (the fast path of the table functions) -------- */

/* PgrGetCounts over a variant range (src/plink_freq.cpp:482, plink_hardy.cpp:510,
 * plink_pca.cpp:394, pgen_reader.cpp:673,864): out[v - v_begin] = {hom_ref, het,
 * hom_alt, missing} over the (subset of) samples.  subset may be NULL. */
int pgh_counts_range(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin, uint32_t v_end,
                     uint32_t (*out)[4], char *errbuf);
int pgh_counts_range_dev(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin, uint32_t v_end,
                         void *d_out, void *stream, char *errbuf);

/* plink_freq's arithmetic on the device (src/plink_freq.cpp:495-544), from a
 * device counts array: d_alt_freq[i] = (het + 2 hom_alt) / (2 obs) as double, NaN
 * where the reference emits NULL (obs == 0); d_obs_ct[i] = 2 obs (int32). */
int pgh_freq_from_counts_dev(const void *d_counts, uint32_t n, void *d_alt_freq, void *d_obs_ct, void *stream,
                             char *errbuf);

/* PgrGetMissingness + PopcountWords (src/plink_missing.cpp:479-486) is column 3
 * of pgh_counts_range.  The per-sample form replaces the phase-1 accumulation of
 * plink_missing sample mode (src/plink_missing.cpp:585-619):
 * out[k] = number of variants in [v_begin, v_end) at which included sample k is missing. */
int pgh_missing_per_sample(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin, uint32_t v_end,
                           uint32_t *out, char *errbuf);
/* d_out: uint32[raw_sample_ct] (raw order, no compaction), zeroed by the call. */
int pgh_missing_per_sample_dev(const pgh_dataset *ds, uint32_t v_begin, uint32_t v_end, void *d_out, void *stream,
                               char *errbuf);

/* plink_freq + plink_hardy + plink_missing (variant and sample mode) off ONE pass over
 * the rows: d_counts uint32[v_end-v_begin][4] as pgh_counts_range_dev (all samples) and
 * d_missing uint32[raw_sample_ct] as pgh_missing_per_sample_dev, each byte read once. */
int pgh_fused_tally_dev(const pgh_dataset *ds, uint32_t v_begin, uint32_t v_end, void *d_counts, void *d_missing,
                        void *stream, char *errbuf);

/* ---- tally pass: one asynchronous walk of the matrix that serves several table functions ----------
 * The reference's plink_freq, plink_hardy and plink_missing each scan the file for themselves
 * (src/plink_freq.cpp:434-488, src/plink_hardy.cpp:472-516, src/plink_missing.cpp:463-486 and :585-619), one
 * blocking PgrGetCounts / PgrGetMissingness per variant.  A tally pass enqueues the whole range at once -- batch by
 * batch on streams it owns, results landing in pinned host memory the pass keeps -- and returns; scan threads then
 * wait only for the rows they are about to emit, while the device works on the batches behind them.  A pass is
 * immutable once its products have landed and may be read by any number of threads, so a caller that keeps it
 * (the shells cache it per dataset, subset and range) serves later functions without touching the matrix again:
 * the three scans of BASELINE config 3 cost one read of each byte.
 *
 * Products (PGH_TALLY_*): COUNTS is always made: {hom_ref, het, hom_alt, missing} per variant over the (subset of)
 * samples.  SAMPLE_MISSING: per included sample, the number of variants of the pass at which it is missing
 * (src/plink_missing.cpp:599-609); without a subset it comes out of the same kernel pass as the counts
 * (k_fused_tally), with one it is a second sweep.  HWE / HWE_MIDP: plink2::HweLnP of every variant's counts
 * (src/plink_hardy.cpp:52-79), on a side stream behind each batch's tally.
 * pgh_tally_start enqueues the products named in `products`; pgh_tally_request adds products later (a no-op for the
 * ones already there; HWE then runs from the resident counts, SAMPLE_MISSING re-reads the rows).  Both only enqueue.
 * pgh_tally_wait blocks until the named products of variants [v_begin, v_end) have landed (SAMPLE_MISSING: the whole
 * pass).  The accessors return pass-owned pinned arrays indexed by (variant - the pass's v_begin); rows are valid
 * once waited for.  Accepts a shard group: every shard walks its own range on its own device. */
typedef struct pgh_tally pgh_tally;
enum { PGH_TALLY_COUNTS = 1, PGH_TALLY_SAMPLE_MISSING = 2, PGH_TALLY_HWE = 4, PGH_TALLY_HWE_MIDP = 8 };
int pgh_tally_start(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin, uint32_t v_end,
                    uint32_t products, pgh_tally **out, char *errbuf);
int pgh_tally_request(pgh_tally *t, uint32_t products, char *errbuf);
int pgh_tally_wait(pgh_tally *t, uint32_t products, uint32_t v_begin, uint32_t v_end, char *errbuf);
const uint32_t (*pgh_tally_counts(const pgh_tally *t))[4];
/* ln p of the exact test per variant (product HWE or HWE_MIDP), NULL when that product was never requested. */
const double *pgh_tally_hwe_lnp(const pgh_tally *t, uint32_t midp);
/* Waits for the product; out[k] for the n_out included samples in ascending file order. */
int pgh_tally_sample_missing(pgh_tally *t, uint32_t *out, char *errbuf);
/* Waits for everything the pass has enqueued, then releases it. */
void pgh_tally_destroy(pgh_tally *t);
/* Passes started by this process so far (a diagnostic: the tests assert that plink_hardy after plink_freq starts none). */
uint64_t pgh_tally_passes_started(void);

/* Page-locked host memory for callers that hand output buffers to the host-buffer entry points again and again
 * (read_pgen's chunk buffers: a device-to-host copy into pinned memory runs at the link's rate and without a
 * staging hop).  Portable across the node's devices.  pgh_host_free(NULL) is a no-op. */
int pgh_host_alloc(size_t bytes, void **out, char *errbuf);
void pgh_host_free(void *p);

/* Call-scoped device work blocks of 64 MB and more (plink_pca's transposed and tile-major matrices, wide score
 * outputs) are kept on a per-device free list between calls instead of going back to the driver -- hipMalloc of
 * tens of gigabytes costs seconds once a process has done it a few times.  The list holds at
 * most PGH_BLOCK_CACHE_GB (environment, default 64, 0 = keep nothing) per device, is emptied by pgh_close and
 * whenever an allocation of the library fails, and by this call: a host that wants the memory back for its own
 * allocations right now.  No reference counterpart (the reference allocates no device memory). */
void pgh_trim_device_cache(void);

/* PgrGet + GenoarrToBytesMinus9 over a variant range (src/pgen_reader.cpp:727-733)
 * plus the validity fill of the ARRAY/LIST child (src/pgen_reader.cpp:1009-1047).
 * out: int8 [v_end-v_begin][n_out] with n_out = subset size or N; a missing call is
 * stored as missing_code (-9 for pgenlib parity, 0 for the DuckDB child vector).
 * validity (may be NULL): ceil(n_out/64) words per variant, bit set = non-missing. */
int pgh_unpack_range(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin, uint32_t v_end, int8_t *out,
                     uint64_t *validity, int missing_code, char *errbuf);
/* d_out row stride = out_pitch bytes (>= n_out, multiple of 16); d_validity row
 * stride = ceil(n_out/64) words; either may be NULL. */
int pgh_unpack_range_dev(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin, uint32_t v_end,
                         void *d_out, size_t out_pitch, void *d_validity, int missing_code, void *stream,
                         char *errbuf);

/* Measurement aid for pgh_unpack_range_dev: a bare kernel with that kernel's traffic shape -- a lane reads 16 bytes
 * and writes 64 + 8 -- and no arithmetic, on the current device.  d_src: n_vec x 16 B, d_dst: n_vec x 64 B, d_val:
 * n_vec x 8 B.  bench.py times it beside the unpack and reports it as roofline.store_ceiling. */
int pgh_probe_unpack_shape_dev(const void *d_src, size_t n_vec, void *d_dst, void *d_val, void *stream, char *errbuf);

/* plink_score phase 1 (src/plink_score.cpp:575-654) for n_scored variants and
 * n_cols weight columns (the reference has one; BASELINE config 4 uses 16).
 *   vidx[i]      variant index (ascending; src/plink_score.cpp:407-408)
 *   weights      [n_scored][n_cols] doubles, row-major
 *   flip[i]      scored allele is REF (dosage 2 - alt)      (may be NULL)
 *   mode         PGH_SCORE_MEAN_IMPUTE | _NO_MEAN_IMPUTATION | _CENTER
 * Outputs over the included samples (ascending file order):
 *   score_sum    [n_out][n_cols], dosage_sum [n_out], allele_ct [n_out].
 * dosage_sum (d_dosage_sum in the device forms) may be NULL when NAMED_ALLELE_DOSAGE_SUM is not
 * wanted: the one-column kernel then looks up 8-byte instead of 16-byte entries and does half
 * the adds (the reference always accumulates it; skipping it is projection pushdown). */
enum { PGH_SCORE_MEAN_IMPUTE = 0, PGH_SCORE_NO_MEAN_IMPUTATION = 1, PGH_SCORE_CENTER = 2 };
int pgh_score(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_scored, const uint32_t *vidx,
              const double *weights, const uint8_t *flip, uint32_t n_cols, int mode, double *score_sum,
              double *dosage_sum, uint32_t *allele_ct, char *errbuf);
/* pgh_score for a caller that already holds the scored variants' class tallies -- a tally pass's rows
 * (pgh_tally_counts), as plink_score after plink_freq on the same file and subset does: counts[i] =
 * {hom_ref, het, hom_alt, missing} of vidx[i] over the included samples.  The means / variances the reference
 * derives per variant (src/plink_score.cpp:598-620) then cost no read of the rows; NULL = pgh_score. */
int pgh_score_counts(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_scored, const uint32_t *vidx,
                     const double *weights, const uint8_t *flip, uint32_t n_cols, int mode, const uint32_t (*counts)[4],
                     double *score_sum, double *dosage_sum, uint32_t *allele_ct, char *errbuf);
/* Device form: raw-sample order (no compaction), outputs are caller-owned device
 * buffers of raw_sample_ct rows, overwritten. */
int pgh_score_dev(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_scored, const uint32_t *vidx,
                  const double *weights, const uint8_t *flip, uint32_t n_cols, int mode, void *d_score_sum,
                  void *d_dosage_sum, void *d_allele_ct, void *stream, char *errbuf);

/* The same in two steps, for callers that score the same weight set repeatedly or
 * want an enqueue-only launch: the plan uploads vidx / weights / flip once and
 * runs the tally + table kernels; pgh_score_run_dev only enqueues the memsets
 * and the accumulate kernels on `stream`.
 * The first plan over a dataset with sparse dosage tracks also builds that
 * dataset's entry records (4 bytes per explicit dosage, resident until
 * pgh_close; skipped without error when they do not fit). */
typedef struct pgh_score_plan pgh_score_plan;
int pgh_score_plan_create(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_scored, const uint32_t *vidx,
                          const double *weights, const uint8_t *flip, uint32_t n_cols, int mode,
                          pgh_score_plan **out, char *errbuf);
int pgh_score_run_dev(const pgh_score_plan *plan, void *d_score_sum, void *d_dosage_sum, void *d_allele_ct,
                      void *stream, char *errbuf);
void pgh_score_plan_destroy(pgh_score_plan *plan);

/* ---- dosage tracks ------------------------------------------------------------
 * A file's explicit dosages (vrtype bits 0x20 / 0x40 / 0x60) are brought to one resident form at
 * pgh_open: a presence bit per sample and the present samples' uint16 values, 16384 per ALT copy.
 * A sample without an explicit dosage takes its hardcall (0 / 16384 / 32768); one with neither is
 * missing.  pgh_score scores dosage-bearing variants from these values as the reference does
 * through PgrGetD (src/plink_score.cpp:586-652).
 *
 * pgh_dosage_sums: PgrGetDCounts (src/plink_freq.cpp:475-480, :525-535): per variant
 *   sums[i] = {sum of dosages, sum of squared dosages, samples with a dosage or a call}
 * on the 16384 scale over the included samples: alt dosage sum = sums[0], ref = 2*16384*sums[2] - sums[0],
 * MaCH r2 from the first two moments.  Variants: [variant_begin, variant_begin + n_variants), or the
 * n_variants listed in vidx when vidx != NULL.
 *
 * pgh_dosage_unpack: PgrGetD + Dosage16ToDoublesMinus9 (src/pgen_reader.cpp:694-705): out[i][k] =
 * dosage of output sample k at variant i as a double, -9.0 when missing; rows of n_out doubles. */
int pgh_dosage_sums(const pgh_dataset *ds, const pgh_subset *subset, uint32_t variant_begin, uint32_t n_variants,
                    const uint32_t *vidx, uint64_t (*sums)[3], char *errbuf);
int pgh_dosage_unpack(const pgh_dataset *ds, const pgh_subset *subset, uint32_t variant_begin, uint32_t n_variants,
                      const uint32_t *vidx, double *out, char *errbuf);
/* Enqueue-only forms over [v_begin, v_end): d_sums = uint64[n][3]; d_out = double rows of out_stride elements. */
int pgh_dosage_sums_dev(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin, uint32_t v_end,
                        void *d_sums, void *stream, char *errbuf);
int pgh_dosage_unpack_dev(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin, uint32_t v_end,
                          void *d_out, size_t out_stride, void *stream, char *errbuf);

/* read_pfile orient := 'sample' (src/pfile_reader.cpp:1560-1720: the reference pre-reads every effective
 * variant with PgrGet / PgrGetD into a variants x samples matrix and emits one row per sample): the matrix
 * sample-major, out[k][j] = call (0/1/2, missing -> missing_code) or dosage (-9.0 = missing) of output sample
 * k at listed variant vidx[j]; rows of n_variants elements. */
int pgh_unpack_samples(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_variants, const uint32_t *vidx,
                       int8_t *out, int missing_code, char *errbuf);
int pgh_dosage_unpack_samples(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_variants,
                              const uint32_t *vidx, double *out, char *errbuf);

/* read_pfile's sample-orient aggregate (src/pfile_reader.cpp:3308-3400, the streaming
 * accumulate_dense loop): counts[k] = {hom_ref, het, hom_alt, missing} of output sample k over
 * the variants [variant_begin, variant_begin + n_var) or, with vidx != NULL, the n_var listed
 * variants.  Three column-tally passes over the packed rows (het, hom-alt, missing), hom-ref by
 * subtraction as the reference derives it at emit time. */
int pgh_sample_counts(const pgh_dataset *ds, const pgh_subset *subset, uint32_t variant_begin, uint32_t n_var,
                      const uint32_t *vidx, uint32_t (*counts)[4], char *errbuf);
/* Enqueue-only form over all raw samples: d_classes = uint32[3][ceil(N/64)*64] receives the het,
 * hom-alt and missing tallies (hom-ref = variants - the three). */
int pgh_sample_counts_dev(const pgh_dataset *ds, uint32_t variant_begin, uint32_t variant_end, void *d_classes,
                          void *stream, char *errbuf);

/* plink_ld's per-pair sums (src/plink_ld.cpp:52-84, ComputeLdStats' sample loop): for each
 * pair p of variants (vidx_a[p], vidx_b[p]) over the samples at which both calls are present
 * (and which the subset keeps),
 *   sums[p] = {n, sum_a, sum_b, sum_ab, sum_a2, sum_b2}
 * as exact integers; r2 / D' follow from them in the caller (the reference's double arithmetic
 * on the same values).  The two rows are reduced with popcounts on the packed planes -- no
 * PgrGet + per-sample decode.  Pairs that share an anchor and walk consecutive partners (the
 * windowed scan's order) read the anchor row once per four partners. */
int pgh_ld_pairs(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_pairs, const uint32_t *vidx_a,
                 const uint32_t *vidx_b, uint32_t (*sums)[6], char *errbuf);
/* Device-output form: sums land in d_sums (uint32[n_pairs][6], zeroed by the call), computed on `stream`;
 * vidx_a / vidx_b stay host arrays.  The task list built from them goes up through the calling thread's own
 * pinned staging buffer, stream-ordered; a thread's next call waits for this one's kernel before it reuses
 * that buffer.  The kernel refuses a task the host cannot have built (no partners, rows outside the resident
 * matrix) and reports it: pgh_ld_pairs_status -- and the thread's next pgh_ld_pairs(_dev) call, and
 * pgh_ld_pairs itself -- wait for the launch and return PGH_ERR_DEVICE if that happened. */
int pgh_ld_pairs_dev(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_pairs, const uint32_t *vidx_a,
                     const uint32_t *vidx_b, void *d_sums, void *stream, char *errbuf);
int pgh_ld_pairs_status(char *errbuf);

/* plink_pca's randomized subspace iteration (src/plink_pca.cpp:630-1080): n_pcs + 1
 * passes of Y = X G1 (Step A) and G1 = X^T Y / M (Step B) over the n_var effective
 * variants, thin SVD of the M x (n_pcs+1)*2*n_pcs Krylov block, then B = X^T U and
 * its thin SVD.  X is never materialised: both contractions read the packed 2-bit
 * rows and normalise on the fly (x = (g - center) * inv_stdev, missing -> 0).
 *   vidx/center/inv_stdev  effective variants and their norms, as the reference's
 *                          bind computes them (src/plink_pca.cpp:392-416)
 *   g1_init                [n_out][2*n_pcs] row-major start matrix (src/plink_pca.cpp:517-523)
 *   eigenvalues            [n_pcs]  (S^2 / n_var)
 *   eigenvectors           [n_out][n_pcs] row-major, defined up to sign
 * Everything tall stays on the device: the Krylov block is orthonormalised by block
 * Gram-Schmidt (any orthonormal basis of its column space serves where the reference
 * takes the left singular vectors), and the final SVD goes through the small
 * (n_pcs+1)*2*n_pcs square Gram matrix, whose eigen-decomposition runs on the host. */
int pgh_pca(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_var, const uint32_t *vidx,
            const double *center, const double *inv_stdev, uint32_t n_pcs, const double *g1_init,
            double *eigenvalues, double *eigenvectors, char *errbuf);

/* The same over variant shards, one process per GPU (variants
 * sharded, one exchange per pass).  Each rank passes ITS shard's effective variants
 * (n_var may be 0) and the job-wide count n_var_total; X is split by rows, so G2 = X^T Y,
 * the Gram matrices of the Krylov block and B = X^T U are sums of per-shard terms.
 * The library leaves the transport to the host: `allreduce` must sum `count` doubles at
 * device pointer `d_buf` in place over all ranks.  Work that produces d_buf has been
 * enqueued on `stream`; the callback either enqueues its collective there (RCCL:
 * ncclAllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, comm, stream)) or synchronises
 * the stream itself before a host transport, and returns 0 on success.  Calls happen in
 * the same order with the same counts on every rank: n_pcs of N*2k, O(n_pcs) small Gram
 * blocks, one of N*qq.  g1_init must be identical on all ranks; eigenvalues and
 * eigenvectors come back replicated.  allreduce == NULL requires n_var_total == n_var. */
typedef int (*pgh_allreduce_fn)(void *ctx, void *d_buf, uint64_t count, void *stream);
int pgh_pca_sharded(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_var, const uint32_t *vidx,
                    const double *center, const double *inv_stdev, uint64_t n_var_total, uint32_t n_pcs,
                    const double *g1_init, pgh_allreduce_fn allreduce, void *allreduce_ctx, double *eigenvalues,
                    double *eigenvectors, char *errbuf);

/* pgh_pca over a .pgen that is NOT resident -- a file beyond the HBM budget: the effective variants (ascending) are cut
 * into windows whose file span is at most `window_variants`, and every pass of the algorithm (n_pcs + 1 power
 * iterations, then phase 3: the reference walks its 240-variant blocks once per pass too, src/plink_pca.cpp:632-676)
 * opens the windows one after the other on the current device, uses each for that pass's two contractions and closes
 * it; the Krylov block (n_var x (n_pcs + 1) 2 n_pcs doubles) stays on the device throughout.  The file is therefore
 * read n_pcs + 2 times.  sample_include: the subset's bit mask over the raw samples, or NULL.  Same results as
 * pgh_pca on a resident dataset of the same variants up to the order of FP64 additions. */
int pgh_pca_streamed(const char *pgen_path, const char *pgi_path, const uint64_t *sample_include, uint32_t n_var,
                     const uint32_t *vidx, const double *center, const double *inv_stdev, uint32_t n_pcs,
                     const double *g1_init, uint64_t window_variants, double *eigenvalues, double *eigenvectors,
                     char *errbuf);

/* ---- per-variant calls mirroring pgenlib -------------------------------- */

/* PgrInit + PgrSetSampleSubsetIndex per scan thread (src/plink_freq.cpp:381-397). */
int pgh_reader_create(const pgh_dataset *ds, const pgh_subset *subset, pgh_reader **out, char *errbuf);
void pgh_reader_destroy(pgh_reader *rd);
/* PgrGet (src/pgen_reader.cpp:727): subset-compacted 2-bit genovec, ceil(n_out/32) words. */
int pgh_get_2bit(pgh_reader *rd, uint32_t vidx, uint64_t *genovec);
/* PgrGetCounts (src/plink_freq.cpp:482). */
int pgh_get_counts(pgh_reader *rd, uint32_t vidx, uint32_t out[4]);
/* PgrGetMissingness (src/plink_missing.cpp:479): ceil(n_out/64) words, bit set = missing. */
int pgh_get_missingness(pgh_reader *rd, uint32_t vidx, uint64_t *bits);
/* PgrGet + GenoarrToBytesMinus9 (src/plink_freq.cpp:463-469): {0,1,2,-9}. */
int pgh_get_int8(pgh_reader *rd, uint32_t vidx, int8_t *out);
/* PgrGetD + Dosage16ToDoublesMinus9 (src/plink_score.cpp:586-596): -9.0 = missing.  One row of
 * pgh_dosage_unpack through the reader's stream. */
int pgh_get_dosage_f64(pgh_reader *rd, uint32_t vidx, double *out);
/* PgrGetP (src/pgen_reader.cpp:715): genovec as pgh_get_2bit plus the
 * phasepresent / phaseinfo bitarrays (ceil(n_out/64) words each, zero for
 * variants without a phase track).  The track was expanded into two resident bit rows at pgh_open. */
int pgh_get_phased(pgh_reader *rd, uint32_t vidx, uint64_t *genovec, uint64_t *phasepresent, uint64_t *phaseinfo);
/* PgrGet + GenoarrToBytesMinus9 over a range (pgh_unpack_range), enqueue-and-return on the reader's stream: calls
 * and validity words of [v_begin, v_end) go to the caller's PAGE-LOCKED buffers (pgh_host_alloc), which must stay
 * untouched until pgh_reader_unpack_wait(rd, slot).  `slot` (0 or 1) names which of the reader's two staging blocks
 * the launch uses: a scan thread keeps chunk k + 1 on its way (kernel + copy over the host link) while it fills
 * its output vector from chunk k -- the reference's scan decodes and copies one variant at a time
 * (src/pgen_reader.cpp:727-733, :1009-1047).  A slot is reused only after it has been waited for. */
int pgh_reader_unpack_start(pgh_reader *rd, int slot, uint32_t v_begin, uint32_t v_end, int8_t *out, uint64_t *validity,
                            int missing_code);
int pgh_reader_unpack_wait(pgh_reader *rd, int slot);
const char *pgh_reader_error(const pgh_reader *rd);

/* ---- HWE exact tests (host) --------------------------------------------- */

/* plink2::HweLnP (src/plink_hardy.cpp:78): ln of the two-sided exact-test p. */
double pgh_hwe_lnp(int32_t obs_hets, int32_t obs_hom1, int32_t obs_hom2, uint32_t midp);
/* plink2::HweXchrLnP (src/plink_hardy.cpp:94). */
double pgh_hwe_xchr_lnp(int32_t female_hets, int32_t female_hom1, int32_t female_hom2, int32_t male1, int32_t male2,
                        uint32_t midp);
/* Batch of autosomal tests on the device, straight from a counts array:
 * ln_p[i] from counts[i] = {hom_ref, het, hom_alt, missing}. */
int pgh_hwe_lnp_batch(const uint32_t (*counts)[4], uint32_t n, uint32_t midp, double *ln_p, char *errbuf);

/* plink2::HweXchrLnP (src/plink_hardy.cpp:94) for n variants at once, one workgroup per variant:
 * strata[i] = {female_hets, female_hom1, female_hom2, male1, male2}.  The host form
 * pgh_hwe_xchr_lnp costs ~0.1 s per variant at 500k samples; this is what the plink_hardy shell
 * calls for the chrX variants of a device batch. */
int pgh_hwe_xchr_lnp_batch(const int32_t (*strata)[5], uint32_t n, uint32_t midp, double *ln_p, char *errbuf);
/* Same with device buffers: d_counts uint32[n][4] -> d_ln_p double[n]. */
int pgh_hwe_lnp_batch_dev(const void *d_counts, uint32_t n, uint32_t midp, void *d_ln_p, void *stream, char *errbuf);

#ifdef __cplusplus
}
#endif
#endif /* PGENHIP_H_ */

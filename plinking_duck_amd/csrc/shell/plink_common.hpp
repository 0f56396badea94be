// plink_common.hpp -- helpers shared by the table-function shells: companion
// metadata, sample subsets, regions, ploidy rules, thread caps, and the handle
// wrappers over libpgenhip.  Mirrors the on-path parts of the reference's
// src/plink_common.{hpp,cpp}; the text readers are
// minimal (pvar/bim/psam/fam only, local files) because metadata plumbing is out
// of scope for this path.
#pragma once

#include "../../../include/pgenhip.h"
#include "duck_api.hpp"

#include <condition_variable>
#include <functional>
#include <mutex>
#include <unordered_map>

namespace duckdb {

// ---- variant / sample metadata ---------------------------------------------------

//! The parsed columns of one .pvar / .bim file.  Immutable once built and shared: every bind of the same file
//! (same path, mtime and size) gets the same object out of a process-wide cache instead of re-parsing the text --
//! the reference's bind is dominated by LoadVariantMetadata (src/plink_common.cpp:171-375), and a DuckDB session
//! binds the same file again and again.
struct VariantColumns {
	vector<string> chroms, ids, refs, alts;
	vector<int32_t> positions;
	std::unordered_map<string, std::pair<idx_t, idx_t>> chrom_offsets; // contiguous CHROM runs
};

struct VariantMetadataIndex {
	shared_ptr<const VariantColumns> cols = make_shared<VariantColumns>();
	idx_t variant_ct = 0;
	bool is_bim = false;

	const vector<string> &chroms() const {
		return cols->chroms;
	}
	const vector<string> &ids() const {
		return cols->ids;
	}
	const vector<string> &refs() const {
		return cols->refs;
	}
	const vector<string> &alts() const {
		return cols->alts;
	}
	const vector<int32_t> &positions() const {
		return cols->positions;
	}
	const std::unordered_map<string, std::pair<idx_t, idx_t>> &chrom_offsets() const {
		return cols->chrom_offsets;
	}
	const string &GetChrom(idx_t v) const {
		return cols->chroms[v];
	}
	int32_t GetPos(idx_t v) const {
		return cols->positions[v];
	}
	const string &GetId(idx_t v) const {
		return cols->ids[v];
	}
	const string &GetRef(idx_t v) const {
		return cols->refs[v];
	}
	const string &GetAlt(idx_t v) const {
		return cols->alts[v];
	}
};

//! src/plink_common.cpp:171-375 (text path only); parsed once per (path, mtime, size) and process
VariantMetadataIndex LoadVariantMetadata(ClientContext &context, const string &path, const string &func_name);

struct SampleInfo {
	vector<string> iids;
	vector<string> fids;   // empty when the file has no FID column
	vector<uint8_t> sexes; // 1 male, 2 female, 0 unknown; empty when no SEX column
	idx_t sample_ct = 0;
	// the whole table, for read_pfile's sample-oriented rows (src/pfile_reader.cpp:241-330):
	// header names without the '#' (.fam: FID IID PAT MAT SEX PHENO1) and every field as text
	vector<string> column_names;
	vector<vector<string>> rows;
	std::unordered_map<string, idx_t> iid_to_idx;
	void EnsureIidMap(const string &source_label = "sample file");
};

//! src/psam_reader.cpp (.psam with #FID/#IID header, or headerless .fam).  Parsed once per (path, mtime, size) and
//! process, like the .pvar columns: at 500,000 samples the table is 1.5 million strings, a bind's largest cost once
//! the variants are cached.  The object is shared and immutable (its IID map is built once, under a lock).
shared_ptr<const SampleInfo> LoadSampleMetadata(ClientContext &context, const string &path);

//! src/plink_common.cpp:553-595 (native text companions only)
string FindCompanionFile(const string &pgen_path, const vector<string> &extensions);
bool FileExists(const string &path);

//! A resident synthetic fileset instead of files on disk: 'synth:<variants>x<samples>[:<seed>[:<missing rate>]]'
//! in the place of the .pgen path.  The matrix is pgh_synth_create's (written straight into HBM), the .pvar / .psam
//! columns are the ones pgh_synth_write_files would have written -- so every table function runs over BASELINE's
//! 1,000,000 x 500,000 shape through bind / init / scan without a 125 GB file (tools/shell_bench.py), and a small
//! spec can be compared with the same fileset on disk.  Not a reference feature; companions of a synth: path are
//! the path itself.
struct SynthSpec {
	uint32_t variants = 0, samples = 0;
	uint64_t seed = 20260807;
	double missing_rate = 0.02;
};
bool ParseSynthPath(const string &path, SynthSpec &out);

// ---- samples / regions ------------------------------------------------------------

//! src/plink_common.cpp:1161-1216
vector<uint32_t> ResolveSampleIndices(const Value &samples_val, uint32_t raw_sample_ct, const SampleInfo *sample_info,
                                      const string &func_name);

//! src/plink_common.hpp:373-396, cpp:1222-1250: the include bitmask; the
//! interleaved vector / cumulative popcounts of pgenlib live inside pgh_subset.
struct SampleSubset {
	uint32_t raw_sample_ct = 0;
	uint32_t subset_sample_ct = 0;
	vector<uint64_t> sample_include;
	vector<uint32_t> sorted_indices; // ascending file order == output order
};
SampleSubset BuildSampleSubset(uint32_t raw_sample_ct, const vector<uint32_t> &sample_indices);

struct VariantRange {
	bool has_filter = false;
	uint32_t start_idx = 0;
	uint32_t end_idx = 0;
};
//! src/plink_common.cpp:1256-1334
VariantRange ParseRegion(const string &region_str, const VariantMetadataIndex &variants, const string &func_name);

//! read_pgen's `variants :=` (src/plink_common.cpp:1787-1885): an index, an ID, a
//! 'CHROM:POS[:REF:ALT]' string, a {start, stop} range (inclusive), a {chrom, pos[, ref, alt]}
//! struct, or a list of one of those kinds.  Returns file variant indices in caller order;
//! duplicates are an error.
vector<uint32_t> ResolveVariantsParameter(const Value &val, const VariantMetadataIndex &variants,
                                          uint32_t raw_variant_ct, const string &func_name);

// ---- filters of read_pgen ------------------------------------------------------------

struct RangeFilter {
	bool active = false;
	double min = -std::numeric_limits<double>::infinity();
	double max = std::numeric_limits<double>::infinity();
	bool Passes(double v) const {
		return v >= min && v <= max;
	}
};
struct CountFilter {
	RangeFilter af_filter, ac_filter;
	bool HasFilter() const {
		return af_filter.active || ac_filter.active;
	}
};
struct GenotypeRangeFilter {
	bool active = false;
	bool allowed[3] = {false, false, false};
	bool include_missing = false;
	bool AllowsCall(double g) const {
		int i = static_cast<int>(g);
		return i >= 0 && i <= 2 && allowed[i];
	}
	void SetFromRange(const RangeFilter &r, bool inc_missing);
};
struct PreDecompFilterResult {
	bool skip = false;
	bool all_pass = true;
};
//! src/plink_common.cpp:1340-1391
RangeFilter ParseRangeFilter(const Value &val, const string &param_name, double valid_min, double valid_max,
                             const string &func_name, bool *include_missing_out = nullptr);
//! src/plink_common.cpp:1393-1436
void ParseIncludeGenotypes(const Value &val, GenotypeRangeFilter &out, const string &func_name);
//! src/plink_common.cpp:1494-1515
PreDecompFilterResult CheckPreDecompFilters(const CountFilter &count_filter, const GenotypeRangeFilter &genotype_filter,
                                            const uint32_t genocounts[4], uint32_t sample_ct);

enum class GenotypeMode { ARRAY, LIST, COLUMNS, STRUCT, COUNTS, STATS };
//! src/plink_common.cpp:17-57
GenotypeMode ResolveGenotypeMode(const string &mode_str, uint32_t sample_ct, const string &func_name);
inline bool IsAggregateGenotypeMode(GenotypeMode m) {
	return m == GenotypeMode::COUNTS || m == GenotypeMode::STATS;
}
LogicalType MakeGenotypeCountsType();
LogicalType MakeGenotypeStatsType();

// ---- PCA normalisation ----------------------------------------------------------------

struct VariantNorm {
	double center = 0.0;
	double inv_stdev = 0.0;
	bool skip = true;
};
//! src/plink_common.cpp:1521-1533
VariantNorm ComputeVariantNorm(double alt_freq);

// ---- threads ---------------------------------------------------------------------------

//! src/plink_common.cpp:1894-1911
uint32_t GetPlinkingMaxThreads(ClientContext &context);
idx_t ApplyMaxThreadsCap(idx_t computed, uint32_t config_max_threads);

// ---- ploidy / sex -----------------------------------------------------------------------

enum class ChromPloidy { AUTOSOMAL, CHR_X, CHR_Y, CHR_MT };
struct ParBounds {
	bool active = false;
	int32_t par1_end = 0, par2_start = 0, par2_end = 0;
};
//! src/plink_common.cpp:1928-1958
ParBounds ResolveParBounds(const string &build, const string &func_name);
//! src/plink_common.cpp:1960-1979 (one variant)
ChromPloidy ClassifyChromPloidy(const string &chrom, int32_t pos, const ParBounds &par);

//! The ploidy class of every variant of a file, decided once per CHROM run instead of once per row (a .pvar's
//! chromosomes are contiguous runs, LoadVariantMetadata refuses anything else): only rows of an X run consult the
//! PAR bounds.  Shared read-only by the scan threads.
class PloidyMap {
public:
	PloidyMap() = default;
	PloidyMap(const VariantMetadataIndex &variants, const ParBounds &par);
	ChromPloidy At(uint32_t vidx) const;
	//! [first, last) = the smallest range that holds every non-autosomal variant of [begin, end); false if none
	bool NonAutosomalSpan(uint32_t begin, uint32_t end, uint32_t &first, uint32_t &last) const;
	bool AnyNonAutosomal(uint32_t begin, uint32_t end) const {
		uint32_t a, b;
		return NonAutosomalSpan(begin, end, a, b);
	}

private:
	struct Run {
		uint32_t begin, end;
		ChromPloidy name_class; // CHR_X: subject to the PAR test per position
	};
	vector<Run> runs_; // the non-autosomal runs only, ascending
	shared_ptr<const VariantColumns> cols_;
	ParBounds par_;
};
//! src/plink_common.cpp:1981-1994
vector<uint8_t> BuildAlignedSex(const SampleInfo &sample_info, const vector<uint32_t> *subset_sorted);

struct SexAwareCounts {
	uint32_t obs_allele_ct = 0, alt_allele_ct = 0;
	uint32_t geno_hom_ref = 0, geno_het = 0, geno_hom_alt = 0, geno_missing = 0;
	uint32_t hwe_hom_ref = 0, hwe_het = 0, hwe_hom_alt = 0;
	bool hwe_defined = false;
	bool sex_unavailable = false;
};
//! ComputeSexAwareCounts (src/plink_common.cpp:1996-2108) restated over class
//! counts per sex stratum -- what the device tally produces with a male and a
//! female sample mask -- instead of a per-sample loop over decoded bytes.
//! total/male/female = {hom_ref, het, hom_alt, missing} of the stratum.
SexAwareCounts SexAwareFromStrata(ChromPloidy ploidy, const uint32_t total[4], const uint32_t male[4],
                                  const uint32_t female[4], bool have_sex);

// ---- libpgenhip handle wrappers ------------------------------------------------------------

//! Status + errbuf -> the reference's exception types (PGH_ERR_ARG ->
//! InvalidInputException, everything else -> IOException).
void ThrowOnPghError(int rc, const char *errbuf, const string &func_name, const string &what);

//! A genotype matrix resident in HBM, shared by every scan thread of a query and
//! kept across queries on the same file (process-wide cache).
class DeviceTally;
class DeviceSubset;
class DeviceDataset {
public:
	~DeviceDataset();
	pgh_dataset *handle = nullptr;
	pgh_info info;
	string path;
	//! The file's rows do not fit the HBM budget (PLINKING_HBM_CACHE_GB): nothing is resident, `handle` is null, and
	//! the functions whose reference counterpart streams the file anyway -- plink_freq, plink_hardy, plink_missing in
	//! both modes, read_pgen's counts / stats / filters -- get their tallies from a pass that walks the file window
	//! by window through HBM (DeviceTally, streamed form); read_pfile's per-sample counts add over the windows
	//! (ForEachWindow), as do plink_score's partial sums, and hardcall output unpacks one window at a time
	//! (LeaseRows), dosage and phased output, read_pfile's sample / genotype orients and plink_ld's pairs included;
	//! plink_pca walks the windows once per pass (pgh_pca_streamed).  LD pairs further apart than a window report that
	//! the file does not fit.
	bool streamed = false;
	pgh_dataset *Resident(const string &func_name) const;
	//! A streamed file (see `streamed`) window by window through HBM: opens variants [v0, v1) -- half the HBM budget
	//! each -- as a dataset of their own (rows keyed by the file's variant numbers), stages `sample_include` next to
	//! them when given, hands both to fn and closes them.  Throws IOException when a window cannot be opened.
	void ForEachWindow(uint32_t begin, uint32_t end, const vector<uint64_t> *sample_include, const string &func_name,
	                   const std::function<void(pgh_dataset *, pgh_subset *, uint32_t, uint32_t)> &fn) const;
	//! Variants per window of a streamed file: half the HBM budget.
	uint64_t WindowVariants() const;
	static shared_ptr<DeviceDataset> Acquire(const string &pgen_path, const string &func_name);

	//! The tally pass over [begin, end) for this sample mask (nullptr = every sample), started if nobody has one:
	//! plink_freq, plink_hardy, plink_missing and read_pgen's filters on the same file, subset and range share ONE
	//! walk of the matrix (pgh_tally_*), whichever of them comes first, in this query or an earlier one.  A pass
	//! that covers more than the asked range serves per-variant products too (`exact_range` = false); the
	//! per-sample product needs the range itself.  `products`: PGH_TALLY_* wanted now (more can be requested from
	//! the pass later).  Kept per dataset, least recently used dropped beyond a few entries; the option
	//! plinking_tally_cache = false (or PLINKING_TALLY_CACHE=0) gives every call a pass of its own.
	shared_ptr<DeviceTally> AcquireTally(const vector<uint64_t> *sample_include, uint32_t begin, uint32_t end,
	                                     uint32_t products, bool exact_range, bool use_cache, const string &func_name);
	//! A pass somebody already started that covers [begin, end) for this mask, or null: for callers that can use
	//! the tallies but would not walk the whole range for them (plink_score, plink_pca's AF prepass).
	shared_ptr<DeviceTally> FindTally(const vector<uint64_t> *sample_include, uint32_t begin, uint32_t end);

private:
	std::mutex tally_mutex_;
	vector<shared_ptr<DeviceTally>> tallies_; // most recently used last
};

//! RAII pgh_tally plus the subset it was started with (the enqueued work reads the subset's device mask).
class DeviceTally {
public:
	DeviceTally(DeviceDataset &ds, const vector<uint64_t> *sample_include, uint32_t begin, uint32_t end,
	            uint32_t products, const string &func_name);
	~DeviceTally();
	DeviceTally(const DeviceTally &) = delete;
	//! enqueue-only; idempotent
	void Request(uint32_t products, const string &func_name);
	//! blocks until the products of [v_begin, v_end) are in host memory
	void Wait(uint32_t products, uint32_t v_begin, uint32_t v_end, const string &func_name);
	const uint32_t *Counts(uint32_t vidx) const {
		return counts_[vidx - begin];
	}
	double LnP(uint32_t vidx, bool midp) const {
		return lnp_[midp ? 1 : 0][vidx - begin];
	}
	void SampleMissing(uint32_t *out, const string &func_name);

	pgh_tally *handle = nullptr; // null for the streamed form
	uint32_t begin = 0, end = 0;
	vector<uint64_t> mask; // empty = all samples

private:
	unique_ptr<DeviceSubset> subset_;
	const uint32_t (*counts_)[4] = nullptr;
	const double *lnp_[2] = {nullptr, nullptr};

	// Streamed form (DeviceDataset::streamed): a producer thread opens the file window by window (pgh_open of a
	// variant range: the ingest's 50 GB/s is the pace), runs a resident tally pass over each window with every
	// product, keeps the results on the host and closes the window.  Scan threads wait on a condition variable
	// for the variants they are about to emit -- the same contract as the resident form's event waits.
	struct Streamed;
	unique_ptr<Streamed> streamed_;
	void RunStream(const string &path, uint32_t sample_ct, uint64_t window_variants, const string &func_name);
};

//! The extension option `plinking_devices` (next to plinking_max_threads, the reference's only option:
//! src/plinking_duck_extension.cpp:49-86): which GPUs of the node hold a file's variants.  '' (default) = the
//! current device; '0,1,2,3' = one contiguous variant shard on each listed device (a device may repeat);
//! 'all' = every visible device.  Datasets opened afterwards are shard groups (pgh_open_sharded): the table
//! functions need no other change, the per-sample merges of plink_score / plink_missing / plink_pca become
//! device-to-device sums.  Throws InvalidInputException on a malformed list or an ordinal the node lacks.
void SetPlinkingDevices(const string &spec);
//! The current list (empty = single current device); PLINKING_DEVICES in the environment is the initial value.
vector<int> GetPlinkingDevices();

//! The extension option `plinking_tally_cache` (default true; PLINKING_TALLY_CACHE=0 in the environment turns it
//! off process-wide): whether tally passes are shared across table-function calls on the same file.
bool GetPlinkingTallyCache(ClientContext &context);

//! RAII pgh_subset
class DeviceSubset {
public:
	DeviceSubset(const DeviceDataset &ds, const vector<uint64_t> &include, const string &func_name);
	~DeviceSubset();
	DeviceSubset(const DeviceSubset &) = delete;
	pgh_subset *handle = nullptr;
};

//! The rows of a dataset for a scan that unpacks them span by span.  A resident dataset hands out its own handle; a
//! streamed one (DeviceDataset::streamed) keeps ONE window of the file resident at a time -- opened on demand at the
//! span a scan thread asks for, half the HBM budget long, with the sample subset staged next to it -- and replaces it
//! when a thread asks for rows beyond it and nobody holds the old one any more.  Claims move forward through the file,
//! so the windows do too; a straggler that still needs the previous window waits its turn and has it re-opened.
struct RowWindows {
	std::mutex m;
	std::condition_variable cv;
	pgh_dataset *ds = nullptr;
	pgh_subset *ss = nullptr;
	uint32_t begin = 0, end = 0;
	int users = 0;
	uint64_t opened = 0; // windows opened so far (a diagnostic)
	~RowWindows();
};

struct RowLease {
	pgh_dataset *ds = nullptr;
	pgh_subset *ss = nullptr;
	RowWindows *windows = nullptr;
	uint64_t window_id = 0; // which window of a streamed file (0: a resident dataset): a reader made on it is good until this changes
	RowLease() = default;
	RowLease(const RowLease &) = delete;
	RowLease &operator=(const RowLease &) = delete;
	RowLease(RowLease &&o) noexcept : ds(o.ds), ss(o.ss), windows(o.windows), window_id(o.window_id) {
		o.windows = nullptr;
	}
	~RowLease();
};

//! Rows [span_begin, span_end) of `dataset`: resident -> its handle and `subset`; streamed -> a lease on the window
//! that holds them (`sample_include`: the subset's mask, NULL for all samples).
RowLease LeaseRows(DeviceDataset &dataset, DeviceSubset *subset, RowWindows &windows, const vector<uint64_t> *sample_include,
                   uint32_t span_begin, uint32_t span_end, uint32_t file_end, const string &func_name);

//! A grow-only buffer of page-locked host memory (pgh_host_alloc): what a scan thread hands to the host-buffer
//! entry points chunk after chunk, so the device-to-host copies run at the link's rate with no staging hop
//! (the reference's counterpart is the per-thread AlignedBuffer, src/plink_common.hpp:65-118).
//! Blocks come from (and go back to) a process-wide pool: page-locking a gigabyte costs a few hundred
//! milliseconds, more than a short query's whole scan, and sixteen scan threads locking at once serialise in the
//! kernel.  The pool keeps what finished queries returned, up to PLINKING_PINNED_POOL_GB (default 24).
void *PinnedPoolAcquire(size_t bytes, size_t &got_bytes);
void PinnedPoolRelease(void *p, size_t bytes);

template <class T>
class PinnedBuffer {
public:
	PinnedBuffer() = default;
	PinnedBuffer(const PinnedBuffer &) = delete;
	PinnedBuffer &operator=(const PinnedBuffer &) = delete;
	~PinnedBuffer() {
		PinnedPoolRelease(p_, bytes_);
	}
	//! at least n elements; contents are not kept
	void resize(size_t n) {
		if (n * sizeof(T) <= bytes_) {
			return;
		}
		PinnedPoolRelease(p_, bytes_);
		p_ = nullptr;
		bytes_ = 0;
		p_ = static_cast<T *>(PinnedPoolAcquire(n * sizeof(T), bytes_));
	}
	T *data() {
		return p_;
	}
	const T *data() const {
		return p_;
	}

private:
	T *p_ = nullptr;
	size_t bytes_ = 0;
};

//! dst bits [dst_bit, dst_bit + n) = src bits [0, n) (bit set = valid), whole words at a time; the other bits
//! of dst keep their values.  dst must hold ceil((dst_bit + n) / 64) words.
inline void CopyValidityBits(uint64_t *dst, idx_t dst_bit, const uint64_t *src, idx_t n) {
	if (n == 0) {
		return;
	}
	const unsigned shift = static_cast<unsigned>(dst_bit & 63);
	uint64_t *d = dst + (dst_bit >> 6);
	const idx_t words = (n + 63) / 64;
	const unsigned tail = static_cast<unsigned>(n & 63); // valid bits of the last source word (0 = all 64)
	for (idx_t w = 0; w < words; w++) {
		const unsigned nb = (w + 1 == words && tail) ? tail : 64u;
		const uint64_t keep = nb == 64 ? ~0ull : ((1ull << nb) - 1);
		const uint64_t bits = src[w] & keep;
		// low part into d[w]
		d[w] = (d[w] & ~(keep << shift)) | (bits << shift);
		if (shift && nb + shift > 64) {
			const uint64_t hi_keep = keep >> (64 - shift);
			d[w + 1] = (d[w + 1] & ~hi_keep) | (bits >> (64 - shift));
		}
	}
}

//! Header probe (replaces the bind-time PgfiInitPhase1/2 of every function).
pgh_info ProbePgen(const string &pgen_path, const string &func_name);

} // namespace duckdb

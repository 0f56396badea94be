// variant_scan.hpp -- the per-variant scan skeleton shared by plink_freq,
// plink_hardy, plink_missing (variant mode) and read_pgen.
//
// Reference shape (src/plink_freq.cpp:434-488): each scan thread claims 128
// variants with fetch_add and calls PgrGetCounts once per variant, blocking on
// the file.  Here the tallies of the WHOLE scanned range are enqueued once, at
// init_global, as a tally pass (pgh_tally_*, shared per dataset / subset / range
// across table functions and queries: DeviceDataset::AcquireTally): the device
// walks the matrix batch by batch while the scan threads claim device batches
// (kDeviceBatch variants, a multiple of the reference's 128), wait only for the
// batch they are about to emit -- its rows are then in pinned host memory -- and
// fill their chunks (<= 2048 rows each) while the batches behind are still being
// tallied.  A wait always happens inside a Scan call, never by yielding an empty
// chunk (scan protocol).
#pragma once

#include "plink_common.hpp"

#include <atomic>
#include <unordered_map>

namespace duckdb {

//! Variants a scan thread claims at a time: two output chunks' worth -- the reference claims 128
//! (src/plink_freq.cpp:413); nothing is launched per claim here (the range's pass is already running), so the
//! claim only sets how finely the threads share the range.  read_pgen's genotype output claims one chunk's span.
constexpr uint32_t kDeviceBatch = 4096;

struct VariantScanGlobal {
	std::atomic<uint32_t> next_variant_idx {0};
	uint32_t start_variant_idx = 0;
	uint32_t end_variant_idx = 0;
	shared_ptr<DeviceDataset> dataset;     // null when no genotype-derived column is projected
	unique_ptr<DeviceSubset> subset;       // samples := [...] (the per-variant calls of list mode, dosage sums)
	uint32_t effective_sample_ct = 0;
	bool want_counts = true;               // false: only the batch claim is needed (read_pgen without filters)
	// The range's tallies (StartTallies): all included samples, and -- over the span that holds the chrX / chrY /
	// chrMT variants, when the file has any and the samples' sexes are known -- the male and the female stratum.
	shared_ptr<DeviceTally> tally, male_tally, female_tally;
	uint32_t strata_begin = 0, strata_end = 0;
	uint32_t wait_products = PGH_TALLY_COUNTS; // what a thread waits for per batch (plink_hardy adds the exact tests)
	uint32_t claim = kDeviceBatch;             // variants per claim
	// read_pgen's `variants :=` list: the scan walks list positions (caller order, as the
	// reference's effective_variant_indices does) and claims kListBatch of them at a time.
	bool has_variant_list = false;
	vector<uint32_t> variant_list;
	RowWindows row_windows; // read_pgen's genotype output over a file beyond the HBM budget (LeaseRows)

	//! Enqueue the range's tally pass(es).  `products`: PGH_TALLY_* beyond the counts; exact_range: the pass must
	//! cover exactly the scanned range (the per-sample product sums over it).
	void StartTallies(const SampleSubset *sample_subset, const SampleInfo *sexes_of, uint32_t raw_sample_ct,
	                  const PloidyMap *ploidy, uint32_t products, bool exact_range, bool use_cache,
	                  const string &func_name) {
		if (!dataset || !want_counts || has_variant_list) {
			return;
		}
		wait_products = PGH_TALLY_COUNTS | (products & (PGH_TALLY_HWE | PGH_TALLY_HWE_MIDP));
		tally = dataset->AcquireTally(sample_subset ? &sample_subset->sample_include : nullptr, start_variant_idx,
		                              end_variant_idx, PGH_TALLY_COUNTS | products, exact_range, use_cache, func_name);
		if (!ploidy || !sexes_of || sexes_of->sexes.empty() ||
		    !ploidy->NonAutosomalSpan(start_variant_idx, end_variant_idx, strata_begin, strata_end)) {
			strata_begin = strata_end = 0;
			return;
		}
		// sex strata masks (male / female, each ANDed with the sample subset if any)
		vector<uint64_t> male((raw_sample_ct + 63) / 64, 0), female((raw_sample_ct + 63) / 64, 0);
		for (uint32_t s = 0; s < raw_sample_ct && s < sexes_of->sexes.size(); s++) {
			if (sample_subset && !((sample_subset->sample_include[s >> 6] >> (s & 63)) & 1ull)) {
				continue;
			}
			if (sexes_of->sexes[s] == 1) {
				male[s >> 6] |= 1ull << (s & 63);
			} else if (sexes_of->sexes[s] == 2) {
				female[s >> 6] |= 1ull << (s & 63);
			}
		}
		male_tally = dataset->AcquireTally(&male, strata_begin, strata_end, PGH_TALLY_COUNTS, false, use_cache, func_name);
		female_tally =
		    dataset->AcquireTally(&female, strata_begin, strata_end, PGH_TALLY_COUNTS, false, use_cache, func_name);
	}
};

constexpr uint32_t kListBatch = 128;

struct VariantScanLocal {
	uint32_t batch_begin = 0, batch_end = 0, cursor = 0;
	bool have_strata = false; // the claimed batch reaches into the sex-strata span

	//! True once every variant of the claimed batch has been handed out: callers that defer
	//! work on a batch (read_pgen's chunk plan) stop here, because the next claim replaces
	//! the batch.
	bool BatchDrained() const {
		return cursor >= batch_end;
	}

	//! Advance to the next variant of this thread; returns false when the range is drained.
	bool Next(VariantScanGlobal &g, const string &func_name, uint32_t &vidx) {
		g_ = &g;
		if (g.has_variant_list) {
			return NextListed(g, func_name, vidx);
		}
		if (cursor >= batch_end) {
			uint32_t begin = g.next_variant_idx.fetch_add(g.claim);
			if (begin >= g.end_variant_idx) {
				return false;
			}
			uint32_t end = std::min<uint64_t>(g.end_variant_idx, static_cast<uint64_t>(begin) + g.claim);
			batch_begin = begin;
			batch_end = end;
			cursor = begin;
			have_strata = false;
			if (g.tally) {
				g.tally->Wait(g.wait_products, begin, end, func_name);
				const uint32_t sb = std::max(begin, g.strata_begin), se = std::min(end, g.strata_end);
				if (g.male_tally && sb < se) {
					g.male_tally->Wait(PGH_TALLY_COUNTS, sb, se, func_name);
					g.female_tally->Wait(PGH_TALLY_COUNTS, sb, se, func_name);
					have_strata = true;
				}
			}
		}
		vidx = cursor++;
		return true;
	}

	const uint32_t *Counts(uint32_t vidx) const {
		if (!list_slot.empty()) {
			return list_counts.data() + 4 * static_cast<size_t>(list_slot.at(vidx));
		}
		return g_->tally->Counts(vidx);
	}
	//! ln p of the variant's exact test (a pass that was asked for PGH_TALLY_HWE / _HWE_MIDP)
	double LnP(uint32_t vidx, bool midp) const {
		return g_->tally->LnP(vidx, midp);
	}
	//! Stratum tallies of a chrX / chrY / chrMT variant (zeros when the sexes are unknown)
	const uint32_t *MaleCounts(uint32_t vidx) const {
		return StratumCounts(g_->male_tally.get(), vidx);
	}
	const uint32_t *FemaleCounts(uint32_t vidx) const {
		return StratumCounts(g_->female_tally.get(), vidx);
	}

private:
	const VariantScanGlobal *g_ = nullptr;
	vector<uint32_t> list_counts;                     // list mode: [batch][4]
	std::unordered_map<uint32_t, uint32_t> list_slot; // list mode: variant index -> row of list_counts

	const uint32_t *StratumCounts(const DeviceTally *t, uint32_t vidx) const {
		static const uint32_t zero[4] = {0, 0, 0, 0};
		return (t && have_strata && vidx >= g_->strata_begin && vidx < g_->strata_end) ? t->Counts(vidx) : zero;
	}

	bool NextListed(VariantScanGlobal &g, const string &func_name, uint32_t &vidx) {
		if (cursor >= batch_end) {
			const uint32_t total = static_cast<uint32_t>(g.variant_list.size());
			uint32_t begin = g.next_variant_idx.fetch_add(kListBatch);
			if (begin >= total) {
				return false;
			}
			batch_begin = begin;
			batch_end = std::min<uint64_t>(total, static_cast<uint64_t>(begin) + kListBatch);
			cursor = begin;
			have_strata = false;
			list_slot.clear();
			if (g.dataset && g.want_counts) {
				list_counts.resize(4 * static_cast<size_t>(batch_end - batch_begin));
				char errbuf[PGH_ERRBUF_LEN] = {0};
				for (uint32_t pos = batch_begin; pos < batch_end; pos++) {
					const uint32_t v = g.variant_list[pos];
					list_slot[v] = pos - batch_begin;
					int rc = pgh_counts_range(g.dataset->Resident(func_name), g.subset ? g.subset->handle : nullptr, v, v + 1,
					                          reinterpret_cast<uint32_t(*)[4]>(list_counts.data() + 4 * (pos - batch_begin)),
					                          errbuf);
					if (rc != PGH_OK) {
						throw IOException("%s: PgrGetCounts failed for variant %u: %s", func_name, v, string(errbuf));
					}
				}
			}
		}
		vidx = g.variant_list[cursor++];
		return true;
	}
};

//! The five metadata columns every per-variant function emits first
//! (src/plink_freq.cpp:576-607): ID empty -> NULL, ALT "" or "." -> NULL.
inline bool FillVariantMetadataColumn(const VariantMetadataIndex &variants, idx_t file_col, uint32_t vidx, Vector &vec,
                                      idx_t row) {
	switch (file_col) {
	case 0:
		FlatVector::GetData<string_t>(vec)[row] = StringVector::AddString(vec, variants.GetChrom(vidx));
		return true;
	case 1:
		FlatVector::GetData<int32_t>(vec)[row] = variants.GetPos(vidx);
		return true;
	case 2: {
		auto &val = variants.GetId(vidx);
		if (val.empty()) {
			FlatVector::SetNull(vec, row, true);
		} else {
			FlatVector::GetData<string_t>(vec)[row] = StringVector::AddString(vec, val);
		}
		return true;
	}
	case 3:
		FlatVector::GetData<string_t>(vec)[row] = StringVector::AddString(vec, variants.GetRef(vidx));
		return true;
	case 4: {
		auto &val = variants.GetAlt(vidx);
		if (val.empty() || val == ".") {
			FlatVector::SetNull(vec, row, true);
		} else {
			FlatVector::GetData<string_t>(vec)[row] = StringVector::AddString(vec, val);
		}
		return true;
	}
	default:
		return false;
	}
}

//! FID / IID of one output row of the per-sample functions: FID is NULL when the file has no FID
//! column and, for plink_missing (src/plink_missing.cpp:651-664), when the field is empty.
inline void FillSampleIdColumn(const SampleInfo &info, bool fid, uint32_t file_idx, Vector &vec, idx_t row,
                               bool empty_fid_is_null = true) {
	const auto &ids = fid ? info.fids : info.iids;
	if (file_idx < ids.size() && !(fid && empty_fid_is_null && ids[file_idx].empty())) {
		FlatVector::GetData<string_t>(vec)[row] = StringVector::AddString(vec, ids[file_idx]);
	} else {
		FlatVector::SetNull(vec, row, true);
	}
}

//! Common bind work: companions, header probe, metadata, count checks, samples, region.
struct PgenBindCommon {
	string pgen_path, pvar_path, psam_path;
	VariantMetadataIndex variants;
	shared_ptr<const SampleInfo> sample_info_ptr = make_shared<SampleInfo>(); // shared with the process-wide cache
	const SampleInfo &sample_info() const {
		return *sample_info_ptr;
	}
	bool has_sample_info = false;
	uint32_t raw_variant_ct = 0, raw_sample_ct = 0;
	bool file_has_dosage = false, file_has_phase = false;
	bool has_sample_subset = false;
	unique_ptr<SampleSubset> sample_subset;
	uint32_t effective_sample_ct = 0;
	VariantRange variant_range;

	//! psam_required: plink_score / plink_pca / plink_missing sample mode need IIDs.
	void Bind(ClientContext &context, TableFunctionBindInput &input, const string &func_name, bool psam_required) {
		pgen_path = input.inputs[0].GetValue<string>();
		auto it = input.named_parameters.find("pvar");
		if (it != input.named_parameters.end()) {
			pvar_path = it->second.GetValue<string>();
		}
		it = input.named_parameters.find("psam");
		if (it != input.named_parameters.end()) {
			psam_path = it->second.GetValue<string>();
		}
		if (pvar_path.empty()) {
			pvar_path = FindCompanionFile(pgen_path, {".pvar", ".bim"});
			if (pvar_path.empty()) {
				throw InvalidInputException("%s: cannot find .pvar or .bim companion for '%s' "
				                            "(use pvar := 'path' to specify explicitly)",
				                            func_name, pgen_path);
			}
		}
		if (psam_path.empty()) {
			psam_path = FindCompanionFile(pgen_path, {".psam", ".fam"});
			if (psam_path.empty() && psam_required) {
				throw InvalidInputException("%s: cannot find .psam or .fam companion for '%s' "
				                            "(use psam := 'path' to specify explicitly)",
				                            func_name, pgen_path);
			}
		}
		pgh_info info = ProbePgen(pgen_path, func_name);
		raw_variant_ct = info.raw_variant_ct;
		raw_sample_ct = info.raw_sample_ct;
		file_has_dosage = info.has_dosage != 0;
		file_has_phase = info.has_phase != 0;

		variants = LoadVariantMetadata(context, pvar_path, func_name);
		if (variants.variant_ct != raw_variant_ct) {
			throw InvalidInputException("%s: variant count mismatch: .pgen has %u variants, "
			                            ".pvar/.bim '%s' has %llu variants",
			                            func_name, raw_variant_ct, pvar_path,
			                            static_cast<unsigned long long>(variants.variant_ct));
		}
		if (!psam_path.empty()) {
			sample_info_ptr = LoadSampleMetadata(context, psam_path);
			has_sample_info = true;
			if (static_cast<uint32_t>(sample_info().sample_ct) != raw_sample_ct) {
				throw InvalidInputException("%s: sample count mismatch: .pgen has %u samples, "
				                            ".psam/.fam '%s' has %llu samples",
				                            func_name, raw_sample_ct, psam_path,
				                            static_cast<unsigned long long>(sample_info().sample_ct));
			}
		}
		effective_sample_ct = raw_sample_ct;
		auto samples_it = input.named_parameters.find("samples");
		if (samples_it != input.named_parameters.end()) {
			auto indices = ResolveSampleIndices(samples_it->second, raw_sample_ct,
			                                    has_sample_info ? &sample_info() : nullptr, func_name);
			sample_subset = make_uniq<SampleSubset>(BuildSampleSubset(raw_sample_ct, indices));
			has_sample_subset = true;
			effective_sample_ct = sample_subset->subset_sample_ct;
		}
		auto region_it = input.named_parameters.find("region");
		if (region_it != input.named_parameters.end()) {
			variant_range = ParseRegion(region_it->second.GetValue<string>(), variants, func_name);
		}
	}

	uint32_t RangeStart() const {
		return variant_range.has_filter ? variant_range.start_idx : 0;
	}
	uint32_t RangeEnd() const {
		return variant_range.has_filter ? variant_range.end_idx : raw_variant_ct;
	}
};

} // namespace duckdb

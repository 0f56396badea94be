// pfile_reader.cpp -- read_pfile(prefix, pgen, pvar, psam, orient, genotypes, samples, region,
//                                variants, af_range, ac_range, include_genotypes, genotype_range,
//                                dosages, phased)
//
// The part of the reference's src/pfile_reader.cpp that sits on the genotype hot path, for one
// fileset (a prefix or explicit pgen/pvar/psam paths) or a LIST of prefixes that share their
// samples and are concatenated along the variant axis (shards):
//   orient := 'variant'  one row per variant -- the read_pgen scan (pgen_reader.cpp) under
//                        read_pfile's name, plus `region`;
//   orient := 'sample' with genotypes := 'counts' | 'stats'
//                        one row per sample: the psam columns and the sample's
//                        {hom_ref, het, hom_alt, missing} tallies over the effective variants
//                        -- the reference's streaming aggregate (src/pfile_reader.cpp:3308-3460:
//                        every thread decodes variant batches and bumps per-sample counters,
//                        merged under a mutex).  Here phase 1 is ONE pgh_sample_counts call by
//                        whichever thread scans first; phase 2 emits rows;
//   orient := 'sample' with genotypes := 'array' | 'list' | 'columns' | 'struct' (calls or dosages)
//                        one row per sample and its calls over the effective variants -- the
//                        reference pre-reads a variants x samples matrix with PgrGet per variant
//                        (src/pfile_reader.cpp:1560-1835); here pgh_unpack_samples hands the matrix
//                        back sample-major (a tiled transpose on the device), source by source.
//   orient := 'genotype'  one row per (effective variant, output sample): the variant columns, the psam
//                        columns and the scalar call (or dosage); threads claim runs of <= 64 variants,
//                        the device unpacks a run in one call, the genotype filter decides which rows
//                        exist (src/pfile_reader.cpp:2342-2760).
// Not carried over: parquet companions (combine_samples := 'union' | 'intersect' | 'concatenate' are "not yet
// implemented" in the reference as well).

#include "pgen_reader.hpp"

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <cstring>
#include <mutex>
#include <unordered_set>

namespace duckdb {

namespace {

string Lowered(string s) {
	for (auto &ch : s) {
		ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
	}
	return s;
}

//! read_pfile's own region grammar (src/pfile_reader.cpp:43-95): 'chr', 'chr:start-' and
//! 'chr:start-end'; returns the closed form the shared ParseRegion takes.
string CanonicalRegion(const string &region_str) {
	const long kOpenEnd = 2147483647L;
	auto colon = region_str.find(':');
	if (colon == string::npos) {
		return region_str + ":0-" + std::to_string(kOpenEnd); // the whole chromosome
	}
	if (colon == 0) {
		throw InvalidInputException("read_pfile: invalid region format '%s' (empty chromosome)", region_str);
	}
	const string range = region_str.substr(colon + 1);
	auto dash = range.find('-');
	if (dash == string::npos) {
		throw InvalidInputException("read_pfile: invalid region format '%s' (expected chr:start-end)", region_str);
	}
	const string start_str = range.substr(0, dash), end_str = range.substr(dash + 1);
	if (start_str.empty()) {
		throw InvalidInputException("read_pfile: invalid region format '%s' (empty start position)", region_str);
	}
	char *tail = nullptr;
	errno = 0;
	long start = std::strtol(start_str.c_str(), &tail, 10);
	if (tail == start_str.c_str() || *tail != '\0' || errno != 0 || start < 0) {
		throw InvalidInputException("read_pfile: invalid region start '%s' in '%s'", start_str, region_str);
	}
	long end = kOpenEnd;
	if (!end_str.empty()) {
		errno = 0;
		end = std::strtol(end_str.c_str(), &tail, 10);
		if (tail == end_str.c_str() || *tail != '\0' || errno != 0 || end < 0) {
			throw InvalidInputException("read_pfile: invalid region end '%s' in '%s'", end_str, region_str);
		}
	}
	if (start > end) {
		throw InvalidInputException("read_pfile: region start (%lld) > end (%lld) in '%s'", static_cast<long long>(start),
		                            static_cast<long long>(end), region_str);
	}
	return region_str.substr(0, colon) + ":" + std::to_string(start) + "-" + std::to_string(std::min(end, kOpenEnd));
}

bool PsamMissing(const string &v) {
	return v.empty() || v == "." || v == "NA" || v == "na";
}

} // namespace

//! One fileset of the call.
struct PfileSource {
	string pgen_path;
	TableFunctionBindInput inner;          // (pgen path; pvar, psam, samples, region, ...) for the shared binds
	unique_ptr<FunctionData> variant_bind; // orient := 'variant': the read_pgen bind under read_pfile's name
	PgenBindCommon c;                      // orient := 'sample'
	bool has_variant_list = false;
	vector<uint32_t> variant_indices;
	vector<uint32_t> effective; // orient := 'sample', per-element modes: the variants whose calls are emitted
};

struct PfileBindData : public TableFunctionData {
	bool sample_orient = false;
	vector<PfileSource> sources; // row-concatenated in list order; all share source 0's samples
	// orient := 'sample' aggregate
	GenotypeMode genotype_mode = GenotypeMode::COUNTS;
	CountFilter count_filter;
	GenotypeRangeFilter genotype_filter;
	idx_t genotypes_col = 0;
	idx_t sex_col = static_cast<idx_t>(-1);
	vector<idx_t> parent_cols;
	vector<uint32_t> output_samples; // file index of every output sample, ascending
	// orient := 'sample', per-element modes (ARRAY / LIST / COLUMNS / STRUCT of the effective variants)
	bool element_mode = false;
	bool dosages = false;
	bool phased = false; // cells carry code 3 for a het the phase track reads ALT|REF; elements are TINYINT[2]
	uint32_t effective_total = 0;
	vector<uint8_t> all_pass; // per effective variant, list order: no call of it falls outside the genotype filter
	idx_t first_geno_col = 0; // COLUMNS: the first per-variant column
	// orient := 'genotype': one row per (effective variant, output sample); columns are the five variant
	// columns, the psam columns, then `genotype`
	bool genotype_orient = false;
	vector<uint32_t> flat_source, flat_variant; // the effective variants of all sources, in list order
	vector<uint32_t> batch_starts;              // scan batches: runs of <= 64 of them inside one source
};

struct PfileGlobalState : public GlobalTableFunctionState {
	// orient := 'variant': one read_pgen scan state per source
	vector<unique_ptr<GlobalTableFunctionState>> variant_states;
	uint64_t total_variants = 0;
	// orient := 'sample'
	vector<column_t> column_ids;
	bool need_genotypes = false;
	uint32_t max_threads_config = 0;
	vector<shared_ptr<DeviceDataset>> datasets;
	vector<unique_ptr<DeviceSubset>> subsets;
	vector<unique_ptr<RowWindows>> row_windows; // per source: a file beyond the HBM budget, one window at a time
	std::mutex phase1_mutex;
	bool phase1_done = false;
	vector<uint32_t> counts;     // [output sample][4]
	vector<uint32_t> keep;       // output positions that pass the genotype row filter
	bool use_keep = false;
	vector<int8_t> calls;        // per-element modes: [output sample][effective variant], -9 = missing / filtered out
	vector<double> dosage_rows;  // the same for dosages := true
	std::atomic<uint32_t> next_variant {0}; // orient := 'genotype': the next unclaimed effective variant
	uint32_t effective_variants = 0;
	uint32_t candidate_variants = 0;
	std::atomic<uint32_t> next_idx {0};

	idx_t MaxThreads() const override {
		if (!variant_states.empty()) {
			return ApplyMaxThreadsCap(total_variants / 1000 + 1, max_threads_config);
		}
		// src/pfile_reader.cpp:576-581: phase 1 is parallel over variants in the reference
		uint32_t work = std::max<uint32_t>(static_cast<uint32_t>(counts.size() / 4), candidate_variants);
		return ApplyMaxThreadsCap(work / 1000 + 1, max_threads_config);
	}
};

struct PfileLocalState : public LocalTableFunctionState {
	// orient := 'genotype': the claimed batch of variants and where the next row comes from
	uint32_t batch_begin = 0, batch_cnt = 0, cur_variant = 0, cur_sample = 0;
	vector<int8_t> batch_calls;    // [variant in batch][output sample], -9 = missing
	vector<double> batch_dosages;
	// orient := 'variant': this thread's read_pgen scan state per source, and the source it is draining
	vector<unique_ptr<LocalTableFunctionState>> variant_states;
	size_t current = 0;
};

//! Paths of one fileset (src/pfile_reader.cpp:670-760): `prefix`.pgen or a full .pgen path, the
//! companions next to it unless named explicitly; returns the (pgen; named...) input the shared binds take.
static PfileSource ResolveSource(const string &prefix, const string &pgen_override, const TableFunctionBindInput &input,
                                 bool take_pvar_override, bool need_psam) {
	PfileSource src;
	SynthSpec synth_unused;
	src.pgen_path = pgen_override;
	string eff_prefix = prefix;
	if (src.pgen_path.empty()) {
		if (prefix.empty()) {
			throw InvalidInputException("read_pfile: no .pgen file path provided");
		}
		SynthSpec synth;
		if (ParseSynthPath(prefix, synth)) {
			src.pgen_path = prefix; // a resident synthetic fileset: the spec stands in for all three files
			eff_prefix.clear();
		} else if (FileExists(prefix + ".pgen")) {
			src.pgen_path = prefix + ".pgen";
		} else if (FileExists(prefix)) {
			src.pgen_path = prefix; // a full .pgen path given as the prefix
			if (prefix.size() > 5 && prefix.compare(prefix.size() - 5, 5, ".pgen") == 0) {
				eff_prefix = prefix.substr(0, prefix.size() - 5);
			}
		} else {
			throw InvalidInputException("read_pfile: cannot find .pgen file for prefix '%s' (tried '%s')", prefix,
			                            prefix + ".pgen");
		}
	}
	src.inner.inputs.push_back(Value::VARCHAR(src.pgen_path));
	for (auto &kv : input.named_parameters) {
		if (kv.first == "region") {
			src.inner.named_parameters[kv.first] = Value::VARCHAR(CanonicalRegion(kv.second.GetValue<string>()));
		} else if (kv.first == "pvar" && !take_pvar_override) {
			continue;
		} else if (kv.first != "pgen" && kv.first != "orient" && kv.first != "combine_samples") {
			src.inner.named_parameters[kv.first] = kv.second;
		}
	}
	// companions: next to the prefix, else next to the .pgen itself with its extension replaced (a full .pgen path
	// given as the prefix or through pgen :=); src/pfile_reader.cpp:715-756
	string pgen_stem = src.pgen_path;
	if (pgen_stem.size() > 5 && pgen_stem.compare(pgen_stem.size() - 5, 5, ".pgen") == 0) {
		pgen_stem.resize(pgen_stem.size() - 5);
	}
	auto companion = [&](const char *param, std::initializer_list<const char *> exts) {
		if (src.inner.named_parameters.count(param)) {
			return;
		}
		for (const string &stem : {eff_prefix, pgen_stem}) {
			for (const char *ext : exts) {
				if (!stem.empty() && FileExists(stem + ext)) {
					src.inner.named_parameters[param] = Value::VARCHAR(stem + ext);
					return;
				}
			}
		}
	};
	const bool synthetic = ParseSynthPath(src.pgen_path, synth_unused);
	const string shown = prefix.empty() ? src.pgen_path : prefix;
	companion("pvar", {".pvar", ".bim"});
	if (!synthetic && (take_pvar_override || !input.named_parameters.count("pvar")) &&
	    !src.inner.named_parameters.count("pvar")) {
		throw InvalidInputException("read_pfile: cannot find .pvar or .bim file for '%s' "
		                            "(use pvar := 'path' to specify explicitly)",
		                            shown);
	}
	companion("psam", {".psam", ".fam"});
	// read_pfile is the fileset reader: unlike read_pgen it does not go on without sample metadata (the first source
	// always needs it, every source under combine_samples := 'identical' without a psam override)
	if (need_psam && !synthetic && !src.inner.named_parameters.count("psam")) {
		throw InvalidInputException("read_pfile: cannot find .psam or .fam file for '%s' "
		                            "(use psam := 'path' to specify explicitly)",
		                            shown);
	}
	return src;
}

//! combine_samples := 'identical' (src/pfile_reader.cpp:1099-1131): every shard must carry the IIDs of the
//! first one, in the same order; 'implicit' trusts the positions.
static void CheckIdenticalSamples(const SampleInfo &ref, const string &ref_path, const SampleInfo &other,
                                  const string &other_path) {
	if (other.iids.size() != ref.iids.size()) {
		throw InvalidInputException("read_pfile: combine_samples := 'identical' but '%s' has %llu samples vs %llu in '%s'",
		                            other_path, static_cast<unsigned long long>(other.iids.size()),
		                            static_cast<unsigned long long>(ref.iids.size()), ref_path);
	}
	for (idx_t k = 0; k < ref.iids.size(); k++) {
		if (other.iids[k] != ref.iids[k]) {
			throw InvalidInputException(
			    "read_pfile: combine_samples := 'identical' but sample %llu differs: '%s' in '%s' vs '%s' in "
			    "'%s'. All shards must share the same IIDs in the same order (or use combine_samples := "
			    "'implicit' to trust positional alignment).",
			    static_cast<unsigned long long>(k), other.iids[k], other_path, ref.iids[k], ref_path);
		}
	}
}

static unique_ptr<FunctionData> PfileBind(ClientContext &context, TableFunctionBindInput &input,
                                          vector<LogicalType> &return_types, vector<string> &names) {
	auto bind_data = make_uniq<PfileBindData>();
	string pgen_override, orient_str = "variant", combine = "implicit";
	for (auto &kv : input.named_parameters) {
		if (kv.first == "pgen") {
			pgen_override = kv.second.GetValue<string>();
		} else if (kv.first == "orient") {
			orient_str = Lowered(kv.second.GetValue<string>());
		} else if (kv.first == "combine_samples") {
			combine = Lowered(kv.second.GetValue<string>());
		}
	}
	// --- the prefix, or the list of prefixes (ResolvePathList, src/plink_common.cpp:460-483) ---
	vector<string> prefixes;
	const Value &first = input.inputs[0];
	if (first.IsNull()) {
		if (pgen_override.empty()) { // a NULL positional with pgen := stands for the no-positional call
			throw InvalidInputException("read_pfile: empty file list provided");
		}
		prefixes.push_back(string());
	} else if (first.type().id() == LogicalTypeId::LIST) {
		for (auto &item : ListValue::GetChildren(first)) {
			if (!item.IsNull()) {
				prefixes.push_back(item.GetValue<string>());
			}
		}
		if (prefixes.empty()) {
			throw InvalidInputException("read_pfile: empty file list provided");
		}
	} else {
		prefixes.push_back(first.GetValue<string>());
	}
	const bool multi_file = prefixes.size() > 1;
	if (orient_str != "variant" && orient_str != "sample" && orient_str != "genotype") {
		throw InvalidInputException("read_pfile: invalid orient value '%s' (expected 'variant', 'genotype', or 'sample')",
		                            orient_str);
	}
	if (combine != "implicit" && combine != "identical") {
		if (combine == "union" || combine == "intersect" || combine == "concatenate") {
			throw InvalidInputException("read_pfile: combine_samples := '%s' is not yet implemented "
			                            "(only 'implicit' and 'identical' are supported)",
			                            combine);
		}
		throw InvalidInputException("read_pfile: unknown combine_samples '%s'", combine);
	}
	// 'identical' verifies the shards' IIDs; a psam override makes them identical by construction
	const bool check_iids = multi_file && combine == "identical" && !input.named_parameters.count("psam");
	if (multi_file) {
		if (!pgen_override.empty() || input.named_parameters.count("pvar")) {
			throw InvalidInputException(
			    "read_pfile: pgen/pvar overrides cannot be combined with a multi-file list (pgen is the per-shard "
			    "input and pvar differs per shard). A psam override is allowed — it applies to every shard.");
		}
		if (input.named_parameters.count("variants")) {
			throw InvalidInputException("read_pfile: variants := [...] with a multi-file list is not yet supported; "
			                            "use region := for selection across files");
		}
	}
	for (size_t i = 0; i < prefixes.size(); i++) {
		const bool need_psam = i == 0 || (combine == "identical" && !input.named_parameters.count("psam"));
		bind_data->sources.push_back(
		    ResolveSource(prefixes[i], i == 0 ? pgen_override : string(), input, !multi_file, need_psam));
	}

	string genotypes_str = "auto";
	auto genotypes_it = input.named_parameters.find("genotypes");
	if (genotypes_it != input.named_parameters.end()) {
		genotypes_str = Lowered(genotypes_it->second.GetValue<string>());
	}
	auto same_samples = [&](uint32_t have, uint32_t want, const PfileSource &src) {
		if (have != want) {
			throw InvalidInputException(
			    "read_pfile: sample count mismatch across files: '%s' has %u samples, but expected %u "
			    "(from '%s'). All listed pfiles must share the same samples.",
			    src.pgen_path, have, want, bind_data->sources[0].pgen_path);
		}
	};
	if (orient_str == "variant") {
		// every source binds as a read_pgen scan; the schema is source 0's
		uint32_t want = 0;
		for (size_t i = 0; i < bind_data->sources.size(); i++) {
			auto &src = bind_data->sources[i];
			vector<LogicalType> types_i;
			vector<string> names_i;
			src.variant_bind = PgenBindNamed(context, src.inner, i ? types_i : return_types, i ? names_i : names,
			                                 "read_pfile", true);
			const uint32_t have = src.variant_bind->Cast<PgenBindData>().c.raw_sample_ct;
			if (i == 0) {
				want = have;
			}
			same_samples(have, want, src);
			if (i && check_iids) {
				CheckIdenticalSamples(bind_data->sources[0].variant_bind->Cast<PgenBindData>().c.sample_info(),
				                      bind_data->sources[0].pgen_path,
				                      src.variant_bind->Cast<PgenBindData>().c.sample_info(), src.pgen_path);
			}
		}
		return std::move(bind_data);
	}
	if (orient_str == "genotype") {
		if (genotypes_str == "counts" || genotypes_str == "stats") {
			throw InvalidInputException("read_pfile: genotypes := '%s' is not compatible with orient := 'genotype' "
			                            "(aggregate modes require orient := 'variant' or 'sample')",
			                            genotypes_str);
		}
		if (genotypes_str == "columns" || genotypes_str == "struct") {
			throw InvalidInputException("read_pfile: genotypes := '%s' is not compatible with orient := 'genotype' "
			                            "(genotype mode already produces scalar output)",
			                            genotypes_str);
		}
		bind_data->genotype_orient = true;
	}

	// --- orient := 'sample' and orient := 'genotype': both are built on the samples of source 0 and the
	//     effective variants of every source ---
	bind_data->sample_orient = true;
	bool dosages = false, phased = false;
	for (auto &kv : input.named_parameters) {
		if (kv.first == "dosages") {
			dosages = kv.second.GetValue<bool>();
		} else if (kv.first == "phased") {
			phased = kv.second.GetValue<bool>();
		}
	}
	if (dosages && phased) {
		throw InvalidInputException("read_pfile: dosages and phased cannot both be true");
	}
	uint64_t total_variants = 0;
	for (size_t i = 0; i < bind_data->sources.size(); i++) {
		auto &src = bind_data->sources[i];
		src.c.Bind(context, src.inner, "read_pfile", i == 0);
		same_samples(src.c.raw_sample_ct, bind_data->sources[0].c.raw_sample_ct, src);
		if (i && check_iids) {
			CheckIdenticalSamples(bind_data->sources[0].c.sample_info(), bind_data->sources[0].pgen_path, src.c.sample_info(),
			                      src.pgen_path);
		}
		total_variants += src.c.raw_variant_ct;
	}
	auto &c = bind_data->sources[0].c;
	const bool aggregate = !bind_data->genotype_orient && (genotypes_str == "counts" || genotypes_str == "stats");
	if (aggregate) {
		bind_data->genotype_mode = ResolveGenotypeMode(genotypes_str, static_cast<uint32_t>(total_variants), "read_pfile");
		const char *label = bind_data->genotype_mode == GenotypeMode::COUNTS ? "counts" : "stats";
		if (phased) {
			throw InvalidInputException("read_pfile: genotypes := '%s' is incompatible with phased := true", label);
		}
		if (dosages) {
			throw InvalidInputException("read_pfile: genotypes := '%s' is incompatible with dosages := true", label);
		}
	}
	bind_data->phased = phased;
	auto variants_it = input.named_parameters.find("variants");
	if (variants_it != input.named_parameters.end()) { // single source only (guarded above)
		auto &src = bind_data->sources[0];
		src.variant_indices = ResolveVariantsParameter(variants_it->second, c.variants, c.raw_variant_ct, "read_pfile");
		std::sort(src.variant_indices.begin(), src.variant_indices.end());
		src.variant_indices.erase(std::unique(src.variant_indices.begin(), src.variant_indices.end()),
		                          src.variant_indices.end());
		src.has_variant_list = true;
	}
	auto af_it = input.named_parameters.find("af_range");
	if (af_it != input.named_parameters.end()) {
		bind_data->count_filter.af_filter = ParseRangeFilter(af_it->second, "af_range", 0.0, 1.0, "read_pfile");
	}
	auto ac_it = input.named_parameters.find("ac_range");
	if (ac_it != input.named_parameters.end()) {
		bind_data->count_filter.ac_filter = ParseRangeFilter(
		    ac_it->second, "ac_range", 0.0, static_cast<double>(2 * c.effective_sample_ct), "read_pfile");
	}
	auto ig_it = input.named_parameters.find("include_genotypes");
	auto gr_it = input.named_parameters.find("genotype_range");
	if (ig_it != input.named_parameters.end() && gr_it != input.named_parameters.end()) {
		throw InvalidInputException(
		    "read_pfile: specify only one of include_genotypes or genotype_range (genotype_range is the numeric "
		    "alias of include_genotypes)");
	}
	if ((ig_it != input.named_parameters.end() || gr_it != input.named_parameters.end()) && dosages) {
		throw InvalidInputException("read_pfile: %s is incompatible with dosages := true",
		                            ig_it != input.named_parameters.end() ? "include_genotypes" : "genotype_range");
	}
	if (ig_it != input.named_parameters.end()) {
		ParseIncludeGenotypes(ig_it->second, bind_data->genotype_filter, "read_pfile");
	} else if (gr_it != input.named_parameters.end()) {
		bool inc_missing = false;
		RangeFilter range = ParseRangeFilter(gr_it->second, "genotype_range", 0.0, 2.0, "read_pfile", &inc_missing);
		bind_data->genotype_filter.SetFromRange(range, inc_missing);
	}
	for (uint32_t s = 0; s < c.raw_sample_ct; s++) {
		if (!c.has_sample_subset || ((c.sample_subset->sample_include[s >> 6] >> (s & 63)) & 1ull)) {
			bind_data->output_samples.push_back(s);
		}
	}
	if (!aggregate) {
		// The effective variants fix the ARRAY dimension and the COLUMNS / STRUCT names, so they are settled
		// here (src/pfile_reader.cpp:1300-1430): region and variants from the metadata, the count and genotype
		// filters from one batched device tally per source.
		bind_data->element_mode = true;
		bind_data->dosages = dosages;
		const bool filtered = bind_data->count_filter.HasFilter() || bind_data->genotype_filter.active;
		for (auto &src : bind_data->sources) {
			vector<uint32_t> candidates;
			if (src.has_variant_list) {
				for (auto v : src.variant_indices) {
					if (!src.c.variant_range.has_filter || (v >= src.c.RangeStart() && v < src.c.RangeEnd())) {
						candidates.push_back(v);
					}
				}
			} else {
				for (uint32_t v = src.c.RangeStart(); v < src.c.RangeEnd(); v++) {
					candidates.push_back(v);
				}
			}
			if (filtered && !candidates.empty()) {
				auto dataset = DeviceDataset::Acquire(src.c.pgen_path, "read_pfile");
				unique_ptr<DeviceSubset> subset;
				if (c.has_sample_subset) {
					subset = make_uniq<DeviceSubset>(*dataset, c.sample_subset->sample_include, "read_pfile");
				}
				// the candidates (ascending) inside [w0, w1) on `ds`: the whole file when it is resident, one window
				// at a time when it is beyond the HBM budget
				auto tally_candidates = [&](pgh_dataset *ds, pgh_subset *ss, uint32_t w0, uint32_t w1) {
					char errbuf[PGH_ERRBUF_LEN] = {0};
					uint32_t gc[4];
					uint32_t run_begin = static_cast<uint32_t>(std::lower_bound(candidates.begin(), candidates.end(), w0) - candidates.begin());
					const uint32_t stop = static_cast<uint32_t>(std::lower_bound(candidates.begin(), candidates.end(), w1) - candidates.begin());
					vector<uint32_t> tallies;
					while (run_begin < stop) { // one device call per run of consecutive candidates
						uint32_t run_end = run_begin + 1;
						while (run_end < stop && candidates[run_end] == candidates[run_end - 1] + 1) {
							run_end++;
						}
						tallies.resize(4 * static_cast<size_t>(run_end - run_begin));
						if (pgh_counts_range(ds, ss, candidates[run_begin], candidates[run_end - 1] + 1,
						                     reinterpret_cast<uint32_t(*)[4]>(tallies.data()), errbuf) != PGH_OK) {
							throw IOException("read_pfile: PgrGetCounts failed for variant %u during count filter: %s",
							                  candidates[run_begin], string(errbuf));
						}
						for (uint32_t i = run_begin; i < run_end; i++) {
							std::memcpy(gc, tallies.data() + 4 * static_cast<size_t>(i - run_begin), sizeof gc);
							auto pf = CheckPreDecompFilters(bind_data->count_filter, bind_data->genotype_filter, gc,
							                                c.effective_sample_ct);
							if (!pf.skip) {
								src.effective.push_back(candidates[i]);
								bind_data->all_pass.push_back(pf.all_pass);
							}
						}
						run_begin = run_end;
					}
				};
				if (dataset->streamed) {
					dataset->ForEachWindow(candidates.front(), candidates.back() + 1,
					                       c.has_sample_subset ? &c.sample_subset->sample_include : nullptr, "read_pfile",
					                       tally_candidates);
				} else {
					tally_candidates(dataset->Resident("read_pfile"), subset ? subset->handle : nullptr, 0u, 0xffffffffu);
				}
			} else {
				src.effective = std::move(candidates);
				bind_data->all_pass.insert(bind_data->all_pass.end(), src.effective.size(), 1);
			}
			bind_data->effective_total += static_cast<uint32_t>(src.effective.size());
		}
		if (bind_data->genotype_orient) {
			// schema (src/pfile_reader.cpp:1272-1294): variant columns, psam columns, the scalar genotype
			for (size_t si = 0; si < bind_data->sources.size(); si++) {
				for (auto v : bind_data->sources[si].effective) {
					if (bind_data->batch_starts.empty() || bind_data->flat_source.empty() ||
					    bind_data->flat_source.back() != si ||
					    bind_data->flat_source.size() - bind_data->batch_starts.back() >= 64) {
						bind_data->batch_starts.push_back(static_cast<uint32_t>(bind_data->flat_source.size()));
					}
					bind_data->flat_source.push_back(static_cast<uint32_t>(si));
					bind_data->flat_variant.push_back(v);
				}
			}
			bind_data->batch_starts.push_back(static_cast<uint32_t>(bind_data->flat_source.size()));
			names = {"CHROM", "POS", "ID", "REF", "ALT"};
			return_types = {LogicalType::VARCHAR, LogicalType::INTEGER, LogicalType::VARCHAR, LogicalType::VARCHAR,
			                LogicalType::VARCHAR};
			for (idx_t i = 0; i < c.sample_info().column_names.size(); i++) {
				const string &name = c.sample_info().column_names[i];
				names.push_back(name);
				if (name == "SEX") {
					return_types.push_back(LogicalType::INTEGER);
					bind_data->sex_col = i;
				} else {
					return_types.push_back(LogicalType::VARCHAR);
					if (name == "PAT" || name == "MAT") {
						bind_data->parent_cols.push_back(i);
					}
				}
			}
			bind_data->genotypes_col = names.size();
			names.push_back("genotype");
			return_types.push_back(phased    ? LogicalType::ARRAY(LogicalType::TINYINT, 2)
			                       : dosages ? LogicalType(LogicalType::DOUBLE)
			                                 : LogicalType(LogicalType::TINYINT));
			return std::move(bind_data);
		}
		bind_data->genotype_mode = ResolveGenotypeMode(genotypes_str, bind_data->effective_total, "read_pfile");
		const uint64_t matrix_size =
		    static_cast<uint64_t>(bind_data->effective_total) * static_cast<uint64_t>(bind_data->output_samples.size());
		int64_t max_elements = 16LL * 1024 * 1024 * 1024;
		Value max_elements_val;
		if (context.TryGetCurrentSetting("plinking_max_matrix_elements", max_elements_val)) {
			max_elements = max_elements_val.GetValue<int64_t>();
		}
		if (matrix_size > static_cast<uint64_t>(max_elements)) {
			throw InvalidInputException("read_pfile: orient := 'sample' would require %llu genotype values "
			                            "(%u variants x %u samples, limit: %lld). "
			                            "Use variants := [...] or samples := [...] to reduce, "
			                            "or SET plinking_max_matrix_elements = <higher value>.",
			                            static_cast<unsigned long long>(matrix_size), bind_data->effective_total,
			                            static_cast<uint32_t>(bind_data->output_samples.size()),
			                            static_cast<long long>(max_elements));
		}
	}
	// schema: every psam column of source 0 (SEX is INTEGER, the rest VARCHAR), then the genotypes
	for (idx_t i = 0; i < c.sample_info().column_names.size(); i++) {
		const string &name = c.sample_info().column_names[i];
		names.push_back(name);
		if (name == "SEX") {
			return_types.push_back(LogicalType::INTEGER);
			bind_data->sex_col = i;
		} else {
			return_types.push_back(LogicalType::VARCHAR);
			if (name == "PAT" || name == "MAT") {
				bind_data->parent_cols.push_back(i);
			}
		}
	}
	bind_data->genotypes_col = names.size();
	if (aggregate) {
		names.push_back("genotypes");
		return_types.push_back(bind_data->genotype_mode == GenotypeMode::COUNTS ? MakeGenotypeCountsType()
		                                                                         : MakeGenotypeStatsType());
		return std::move(bind_data);
	}
	// (scalar columns carry hardcalls or dosages: phase is not representable there, src/pfile_reader.cpp:1488)
	const LogicalType elem = phased && bind_data->genotype_mode != GenotypeMode::COLUMNS
	                             ? LogicalType::ARRAY(LogicalType::TINYINT, 2)
	                             : dosages ? LogicalType(LogicalType::DOUBLE) : LogicalType(LogicalType::TINYINT);
	if (bind_data->genotype_mode == GenotypeMode::COLUMNS || bind_data->genotype_mode == GenotypeMode::STRUCT) {
		// one column / field per effective variant, named by its ID, else CHROM:POS (src/pfile_reader.cpp:1443-1512)
		const char *label = bind_data->genotype_mode == GenotypeMode::COLUMNS ? "columns" : "struct";
		std::unordered_set<string> seen;
		child_list_t fields;
		bind_data->first_geno_col = names.size();
		for (auto &src : bind_data->sources) {
			for (auto v : src.effective) {
				string id = src.c.variants.GetId(v);
				if (id.empty()) {
					id = src.c.variants.GetChrom(v) + ":" + std::to_string(src.c.variants.GetPos(v));
				}
				if (!seen.insert(id).second) {
					throw InvalidInputException(
					    "read_pfile: genotypes := '%s' with orient := 'sample' requires unique variant identifiers, but "
					    "'%s' appears more than once. Use variants := [...] to select unique variants.",
					    label, id);
				}
				if (bind_data->genotype_mode == GenotypeMode::COLUMNS) {
					names.push_back(id);
					return_types.push_back(elem);
				} else {
					fields.push_back({id, elem});
				}
			}
		}
		if (bind_data->genotype_mode == GenotypeMode::STRUCT) {
			names.push_back("genotypes");
			return_types.push_back(LogicalType::STRUCT(std::move(fields)));
		}
	} else {
		names.push_back("genotypes");
		return_types.push_back(bind_data->genotype_mode == GenotypeMode::ARRAY
		                           ? LogicalType::ARRAY(elem, bind_data->effective_total)
		                           : LogicalType::LIST(elem));
	}
	return std::move(bind_data);
}

static unique_ptr<GlobalTableFunctionState> PfileInitGlobal(ClientContext &context, TableFunctionInitInput &input) {
	auto &bind_data = input.bind_data->Cast<PfileBindData>();
	auto state = make_uniq<PfileGlobalState>();
	state->max_threads_config = GetPlinkingMaxThreads(context);
	if (!bind_data.sample_orient) {
		for (auto &src : bind_data.sources) {
			TableFunctionInitInput inner = input;
			inner.bind_data = src.variant_bind.get();
			state->variant_states.push_back(PgenInitGlobal(context, inner));
			auto &scan = state->variant_states.back()->Cast<PgenGlobalState>().scan;
			state->total_variants += scan.has_variant_list ? scan.variant_list.size()
			                                               : scan.end_variant_idx - scan.start_variant_idx;
		}
		return std::move(state);
	}
	state->column_ids = input.column_ids;
	for (auto col_id : input.column_ids) {
		if (col_id != COLUMN_IDENTIFIER_ROW_ID && col_id >= bind_data.genotypes_col) {
			state->need_genotypes = true; // the genotypes column, or one of the per-variant columns
		}
	}
	if (bind_data.genotype_orient && bind_data.genotype_filter.active) {
		state->need_genotypes = true; // the filter decides which rows exist, whatever is projected
	}
	for (auto &src : bind_data.sources) {
		state->candidate_variants += src.has_variant_list ? static_cast<uint32_t>(src.variant_indices.size())
		                                                  : src.c.RangeEnd() - src.c.RangeStart();
	}
	if (!bind_data.element_mode) {
		state->counts.assign(4 * bind_data.output_samples.size(), 0);
	}
	// a row filter needs the tallies even when the struct itself is not projected
	if (state->need_genotypes || bind_data.genotype_filter.active) {
		for (auto &src : bind_data.sources) {
			state->datasets.push_back(DeviceDataset::Acquire(src.c.pgen_path, "read_pfile"));
			state->row_windows.push_back(make_uniq<RowWindows>());
			state->subsets.push_back(nullptr);
			if (bind_data.sources[0].c.has_sample_subset) {
				state->subsets.back() = make_uniq<DeviceSubset>(
				    *state->datasets.back(), bind_data.sources[0].c.sample_subset->sample_include, "read_pfile");
			}
		}
	}
	return std::move(state);
}

static unique_ptr<LocalTableFunctionState> PfileInitLocal(ExecutionContext &context, TableFunctionInitInput &input,
                                                          GlobalTableFunctionState *global_state) {
	auto &bind_data = input.bind_data->Cast<PfileBindData>();
	auto state = make_uniq<PfileLocalState>();
	if (!bind_data.sample_orient) {
		auto &gstate = global_state->Cast<PfileGlobalState>();
		for (size_t i = 0; i < bind_data.sources.size(); i++) {
			TableFunctionInitInput inner = input;
			inner.bind_data = bind_data.sources[i].variant_bind.get();
			state->variant_states.push_back(PgenInitLocal(context, inner, gstate.variant_states[i].get()));
		}
	}
	return std::move(state);
}

//! phased := true outside orient := 'variant' (UnpackPhasedGenotypes, src/plink_common.cpp:1549-1584): a het whose
//! phase track reads ALT|REF becomes code 3 in a matrix of calls {0, 1, 2, -9}; cell(j, k) is the j-th listed
//! variant's call of output sample k.  Every other het stays 1 = REF|ALT, the canonical order of an unphased one.
template <class Cell>
static void MarkAltFirstHets(pgh_dataset *ds, pgh_subset *ss, const uint32_t *variants, uint32_t n_var, uint32_t n_out,
                             Cell &&cell) {
	char errbuf[PGH_ERRBUF_LEN] = {0};
	pgh_reader *rd = nullptr;
	if (pgh_reader_create(ds, ss, &rd, errbuf) != PGH_OK) {
		throw IOException("read_pfile: PgrInit failed: %s", string(errbuf));
	}
	vector<uint64_t> genovec((n_out + 31) / 32 + 1), present((n_out + 63) / 64 + 1), info((n_out + 63) / 64 + 1);
	for (uint32_t j = 0; j < n_var; j++) {
		if (pgh_get_phased(rd, variants[j], genovec.data(), present.data(), info.data()) != PGH_OK) {
			pgh_reader_destroy(rd);
			throw IOException("read_pfile: PgrGetP failed for variant %u", variants[j]);
		}
		for (uint32_t w = 0; w < (n_out + 63) / 64; w++) {
			uint64_t bits = present[w] & info[w];
			while (bits) {
				const uint32_t k = w * 64 + static_cast<uint32_t>(__builtin_ctzll(bits));
				bits &= bits - 1;
				if (k < n_out && cell(j, k) == 1) {
					cell(j, k) = 3;
				}
			}
		}
	}
	pgh_reader_destroy(rd);
}

//! One element of a phased output: dst is ARRAY(TINYINT, 2); code as in MarkAltFirstHets, -9 = NULL.
static void PutPhasedPair(Vector &dst, idx_t slot, int8_t code) {
	auto *alleles = FlatVector::GetData<int8_t>(ArrayVector::GetEntry(dst));
	int8_t a0 = 0, a1 = 0;
	if (code == -9) {
		FlatVector::Validity(dst).SetInvalid(slot);
	} else if (code == 2) {
		a0 = a1 = 1;
	} else if (code == 1) {
		a1 = 1;
	} else if (code == 3) {
		a0 = 1;
	}
	alleles[2 * slot] = a0;
	alleles[2 * slot + 1] = a1;
}

//! Per-element modes, phase 1 (the reference's pre-read, src/pfile_reader.cpp:1560-1835): every source's
//! effective variants, sample-major, side by side in list order; then the genotype filter -- a sample stays
//! if any of its calls is allowed (or it has a missing call and missing is included), and calls outside the
//! filter read as NULL at variants where not every call passes.
static void RunSampleMatrixPhase1(const PfileBindData &bind_data, PfileGlobalState &gstate) {
	char errbuf[PGH_ERRBUF_LEN] = {0};
	const size_t n_out = bind_data.output_samples.size();
	const size_t total = bind_data.effective_total;
	if (bind_data.dosages) {
		gstate.dosage_rows.assign(n_out * total, 0.0);
	} else {
		gstate.calls.assign(n_out * total, 0);
	}
	size_t at = 0;
	vector<int8_t> part;
	vector<double> dpart;
	for (size_t si = 0; si < bind_data.sources.size(); si++) {
		const auto &eff = bind_data.sources[si].effective;
		if (eff.empty()) {
			continue;
		}
		// the effective variants [lo, hi) of this source on `ds`, into matrix columns at + lo ..: all of them on a
		// resident file (one source: straight into the matrix), a window's share on a file beyond the HBM budget
		auto read_columns = [&](pgh_dataset *ds, pgh_subset *ss, size_t lo, size_t hi) {
			const uint32_t n_var = static_cast<uint32_t>(hi - lo);
			if (n_var == 0) {
				return;
			}
			const bool whole = bind_data.sources.size() == 1 && lo == 0 && hi == eff.size();
			int rc;
			if (bind_data.dosages) {
				double *dst = gstate.dosage_rows.data();
				if (!whole) {
					dpart.resize(n_out * n_var);
					dst = dpart.data();
				}
				rc = pgh_dosage_unpack_samples(ds, ss, n_var, eff.data() + lo, dst, errbuf);
				for (size_t k = 0; !whole && rc == PGH_OK && k < n_out; k++) {
					std::memcpy(gstate.dosage_rows.data() + k * total + at + lo, dpart.data() + k * n_var, 8 * static_cast<size_t>(n_var));
				}
			} else {
				int8_t *dst = gstate.calls.data();
				if (!whole) {
					part.resize(n_out * n_var);
					dst = part.data();
				}
				rc = pgh_unpack_samples(ds, ss, n_var, eff.data() + lo, dst, -9, errbuf);
				for (size_t k = 0; !whole && rc == PGH_OK && k < n_out; k++) {
					std::memcpy(gstate.calls.data() + k * total + at + lo, part.data() + k * n_var, n_var);
				}
			}
			if (rc != PGH_OK) {
				throw IOException("read_pfile: %s failed during sample-orient pre-read: %s",
				                  bind_data.dosages ? "PgrGetD" : "PgrGet", string(errbuf));
			}
		};
		if (gstate.datasets[si]->streamed) {
			const auto &first = bind_data.sources[0].c;
			gstate.datasets[si]->ForEachWindow(
			    eff.front(), eff.back() + 1, first.has_sample_subset ? &first.sample_subset->sample_include : nullptr, "read_pfile",
			    [&](pgh_dataset *ds, pgh_subset *ss, uint32_t w0, uint32_t w1) {
				    read_columns(ds, ss, std::lower_bound(eff.begin(), eff.end(), w0) - eff.begin(),
				                 std::lower_bound(eff.begin(), eff.end(), w1) - eff.begin());
			    });
		} else {
			read_columns(gstate.datasets[si]->Resident("read_pfile"), gstate.subsets[si] ? gstate.subsets[si]->handle : nullptr, 0,
			             eff.size());
		}
		at += eff.size();
	}
	if (bind_data.genotype_filter.active && !bind_data.dosages) {
		const auto &gf = bind_data.genotype_filter;
		gstate.use_keep = true;
		for (uint32_t k = 0; k < n_out; k++) {
			int8_t *row = gstate.calls.data() + static_cast<size_t>(k) * total;
			bool in_range = false, has_missing = false;
			for (size_t j = 0; j < total; j++) {
				if (row[j] == -9) {
					has_missing = true;
				} else if (gf.AllowsCall(static_cast<double>(row[j]))) {
					in_range = true;
				} else if (!bind_data.all_pass[j]) {
					row[j] = -9;
				}
			}
			if (in_range || (gf.include_missing && has_missing)) {
				gstate.keep.push_back(k);
			}
		}
	}
	if (bind_data.phased && !bind_data.dosages && bind_data.genotype_mode != GenotypeMode::COLUMNS) {
		size_t first = 0;
		for (size_t si = 0; si < bind_data.sources.size(); si++) {
			const auto &eff = bind_data.sources[si].effective;
			if (!eff.empty() && gstate.datasets[si]->streamed) {
				const auto &c0 = bind_data.sources[0].c;
				gstate.datasets[si]->ForEachWindow(
				    eff.front(), eff.back() + 1, c0.has_sample_subset ? &c0.sample_subset->sample_include : nullptr, "read_pfile",
				    [&](pgh_dataset *ds, pgh_subset *ss, uint32_t w0, uint32_t w1) {
					    const size_t lo = std::lower_bound(eff.begin(), eff.end(), w0) - eff.begin();
					    const size_t hi = std::lower_bound(eff.begin(), eff.end(), w1) - eff.begin();
					    MarkAltFirstHets(ds, ss, eff.data() + lo, static_cast<uint32_t>(hi - lo), static_cast<uint32_t>(n_out),
					                     [&](uint32_t j, uint32_t k) -> int8_t & { return gstate.calls[k * total + first + lo + j]; });
				    });
			} else if (!eff.empty()) {
				MarkAltFirstHets(gstate.datasets[si]->Resident("read_pfile"), gstate.subsets[si] ? gstate.subsets[si]->handle : nullptr,
				                 eff.data(), static_cast<uint32_t>(eff.size()), static_cast<uint32_t>(n_out),
				                 [&](uint32_t j, uint32_t k) -> int8_t & { return gstate.calls[k * total + first + j]; });
			}
			first += eff.size();
		}
	}
}

//! Phase 1: per source, the effective variants (region, variants, af/ac filters) and every sample's
//! tallies over them; the sources share their samples, so the tallies simply add.
static void RunSamplePhase1(const PfileBindData &bind_data, PfileGlobalState &gstate) {
	char errbuf[PGH_ERRBUF_LEN] = {0};
	vector<uint32_t> part(gstate.counts.size());
	gstate.effective_variants = 0;
	for (size_t si = 0; si < bind_data.sources.size(); si++) {
		const auto &src = bind_data.sources[si];
		const auto &c = src.c;
		vector<uint32_t> all_listed;
		if (src.has_variant_list) {
			for (auto v : src.variant_indices) {
				if (!c.variant_range.has_filter || (v >= c.RangeStart() && v < c.RangeEnd())) {
					all_listed.push_back(v);
				}
			}
		}
		// the tallies of the candidates in [w_begin, w_end) on `ds`: the whole range on a resident file, one window
		// at a time on a file beyond the HBM budget (per-sample tallies simply add over the windows, as they do over
		// the sources -- the reference streams this mode too, src/pfile_reader.cpp:3287-3460)
		auto tally_window = [&](pgh_dataset *ds, pgh_subset *ss, uint32_t w_begin, uint32_t w_end) {
		vector<uint32_t> list;
		bool listed = src.has_variant_list;
		if (listed) {
			for (auto v : all_listed) {
				if (v >= w_begin && v < w_end) {
					list.push_back(v);
				}
			}
		}
		const uint32_t begin = w_begin;
		uint32_t n_var = listed ? static_cast<uint32_t>(list.size()) : w_end - w_begin;
		if (bind_data.count_filter.HasFilter() && n_var) {
			// per-variant tallies of the candidates decide which of them stay
			vector<uint32_t> vc(4 * static_cast<size_t>(n_var));
			vector<uint32_t> kept;
			const GenotypeRangeFilter none;
			auto consider = [&](uint32_t v, const uint32_t *gc) {
				if (!CheckPreDecompFilters(bind_data.count_filter, none, gc, bind_data.sources[0].c.effective_sample_ct)
				         .skip) {
					kept.push_back(v);
				}
			};
			for (uint32_t i = 0; i < (listed ? n_var : 1u); i++) {
				const uint32_t v0 = listed ? list[i] : begin, v1 = listed ? list[i] + 1 : begin + n_var;
				if (pgh_counts_range(ds, ss, v0, v1, reinterpret_cast<uint32_t(*)[4]>(vc.data() + 4 * static_cast<size_t>(i)),
				                     errbuf) != PGH_OK) {
					throw IOException("read_pfile: PgrGetCounts failed for variants [%u, %u): %s", v0, v1, string(errbuf));
				}
			}
			for (uint32_t i = 0; i < n_var; i++) {
				consider(listed ? list[i] : begin + i, vc.data() + 4 * static_cast<size_t>(i));
			}
			list.swap(kept);
			listed = true;
			n_var = static_cast<uint32_t>(list.size());
		}
		if (n_var == 0) {
			return; // nothing of this window is a candidate
		}
		if (pgh_sample_counts(ds, ss, listed ? 0 : begin, n_var, listed ? list.data() : nullptr,
		                      reinterpret_cast<uint32_t(*)[4]>(part.data()), errbuf) != PGH_OK) {
			throw IOException("read_pfile: PgrGet failed during sample-orient aggregation: %s", string(errbuf));
		}
		for (size_t k = 0; k < part.size(); k++) {
			gstate.counts[k] += part[k];
		}
		gstate.effective_variants += n_var;
		};
		if (c.RangeEnd() <= c.RangeStart()) {
			continue;
		}
		if (gstate.datasets[si]->streamed) {
			const auto &first = bind_data.sources[0].c; // (the sources share their samples: the subset is source 0's)
			gstate.datasets[si]->ForEachWindow(c.RangeStart(), c.RangeEnd(),
			                                   first.has_sample_subset ? &first.sample_subset->sample_include : nullptr,
			                                   "read_pfile", tally_window);
		} else {
			tally_window(gstate.datasets[si]->Resident("read_pfile"), gstate.subsets[si] ? gstate.subsets[si]->handle : nullptr,
			             c.RangeStart(), c.RangeEnd());
		}
	}
	if (bind_data.genotype_filter.active) {
		// keep a sample if any of its calls is allowed (src/pfile_reader.cpp:3452-3462)
		auto &gf = bind_data.genotype_filter;
		gstate.use_keep = true;
		for (uint32_t k = 0; k < bind_data.output_samples.size(); k++) {
			const uint32_t *sc = gstate.counts.data() + 4 * static_cast<size_t>(k);
			bool in_range = false;
			for (int g = 0; g < 3; g++) {
				in_range |= sc[g] > 0 && gf.AllowsCall(static_cast<double>(g));
			}
			if (in_range || (gf.include_missing && sc[3] > 0)) {
				gstate.keep.push_back(k);
			}
		}
	}
}

//! FillSampleMetadataValue (src/pfile_reader.cpp:2846-2885): SEX is an integer with 0 / NA -> NULL,
//! PAT / MAT "0" -> NULL, the usual missing tokens -> NULL.
static void PutSampleField(const PfileBindData &bind_data, idx_t psam_col, uint32_t sample, Vector &vec, idx_t r) {
	const auto &fields = bind_data.sources[0].c.sample_info().rows[sample];
	const string val = psam_col < fields.size() ? fields[psam_col] : string();
	if (psam_col == bind_data.sex_col) {
		char *end = nullptr;
		const long parsed = PsamMissing(val) ? 0 : std::strtol(val.c_str(), &end, 10);
		if (parsed == 0 || end == val.c_str()) {
			FlatVector::SetNull(vec, r, true);
		} else {
			FlatVector::GetData<int32_t>(vec)[r] = static_cast<int32_t>(parsed);
		}
		return;
	}
	const bool is_parent =
	    std::find(bind_data.parent_cols.begin(), bind_data.parent_cols.end(), psam_col) != bind_data.parent_cols.end();
	if (PsamMissing(val) || (is_parent && val == "0")) {
		FlatVector::SetNull(vec, r, true);
	} else {
		FlatVector::GetData<string_t>(vec)[r] = StringVector::AddString(vec, val);
	}
}

//! orient := 'genotype' (src/pfile_reader.cpp:2342-2760): one row per (effective variant, output sample).
//! A thread claims a batch of variants, has their calls (or dosages) unpacked by the device in one call per
//! run of consecutive variants, and fans them out; the genotype filter decides which rows exist.
static void GenotypeOrientScan(const PfileBindData &bind_data, PfileGlobalState &gstate, PfileLocalState &lstate,
                               DataChunk &output) {
	const uint32_t n_out = static_cast<uint32_t>(bind_data.output_samples.size());
	const auto &gf = bind_data.genotype_filter;
	// the rows of this chunk: (flat variant, output sample)
	vector<uint32_t> row_variant, row_sample;
	row_variant.reserve(STANDARD_VECTOR_SIZE);
	row_sample.reserve(STANDARD_VECTOR_SIZE);
	vector<int8_t> row_call;
	vector<double> row_dosage;
	while (row_variant.size() < STANDARD_VECTOR_SIZE && n_out != 0) {
		if (lstate.cur_variant >= lstate.batch_cnt) {
			// next batch: bind cut the effective variants into runs of <= 64 that stay inside one source
			const uint32_t b = gstate.next_variant.fetch_add(1);
			if (b + 1 >= bind_data.batch_starts.size()) {
				break;
			}
			const uint32_t begin = bind_data.batch_starts[b], cnt = bind_data.batch_starts[b + 1] - begin;
			lstate.batch_begin = begin;
			lstate.batch_cnt = cnt;
			lstate.cur_variant = 0;
			lstate.cur_sample = 0;
			if (gstate.need_genotypes) {
				const uint32_t si = bind_data.flat_source[begin];
				// The batch (<= 64 ascending variants of one source) in pieces whose span fits a window of a file beyond
				// the HBM budget -- one piece on a resident file, or where the effective variants lie close together.
				const auto &c0 = bind_data.sources[0].c;
				const uint64_t reach = gstate.datasets[si]->streamed ? gstate.datasets[si]->WindowVariants() : ~0ull;
				char errbuf[PGH_ERRBUF_LEN] = {0};
				int rc = PGH_OK;
				if (bind_data.dosages) {
					lstate.batch_dosages.resize(static_cast<size_t>(cnt) * n_out);
				} else {
					lstate.batch_calls.resize(static_cast<size_t>(cnt) * n_out);
				}
				const uint32_t *fv = bind_data.flat_variant.data() + begin;
				for (uint32_t p0 = 0; p0 < cnt && rc == PGH_OK;) {
					uint32_t p1 = p0 + 1;
					while (p1 < cnt && static_cast<uint64_t>(fv[p1]) + 1 - fv[p0] <= reach) {
						p1++;
					}
					RowLease rows = LeaseRows(*gstate.datasets[si], gstate.subsets[si].get(), *gstate.row_windows[si],
					                          c0.has_sample_subset ? &c0.sample_subset->sample_include : nullptr, fv[p0],
					                          fv[p1 - 1] + 1, bind_data.sources[si].c.raw_variant_ct, "read_pfile");
					if (bind_data.dosages) {
						rc = pgh_dosage_unpack(rows.ds, rows.ss, 0, p1 - p0, fv + p0,
						                       lstate.batch_dosages.data() + static_cast<size_t>(p0) * n_out, errbuf);
					} else {
						uint32_t k = p0;
						while (k < p1 && rc == PGH_OK) { // one device call per run of consecutive variants
							uint32_t e = k + 1;
							while (e < p1 && fv[e] == fv[e - 1] + 1) {
								e++;
							}
							rc = pgh_unpack_range(rows.ds, rows.ss, fv[k], fv[e - 1] + 1,
							                      lstate.batch_calls.data() + static_cast<size_t>(k) * n_out, nullptr, -9, errbuf);
							k = e;
						}
						if (rc == PGH_OK && bind_data.phased) {
							MarkAltFirstHets(rows.ds, rows.ss, fv + p0, p1 - p0, n_out, [&](uint32_t j, uint32_t k2) -> int8_t & {
								return lstate.batch_calls[static_cast<size_t>(p0 + j) * n_out + k2];
							});
						}
					}
					p0 = p1;
				}
				if (rc != PGH_OK) {
					throw IOException("read_pfile: %s failed for variant %u: %s", bind_data.dosages ? "PgrGetD" : "PgrGet",
					                  bind_data.flat_variant[begin], string(errbuf));
				}
			}
		}
		while (lstate.cur_variant < lstate.batch_cnt && row_variant.size() < STANDARD_VECTOR_SIZE) {
			const uint32_t j = lstate.cur_variant, k = lstate.cur_sample;
			bool keep = true;
			int8_t call = 0;
			double dose = 0.0;
			if (gstate.need_genotypes) {
				if (bind_data.dosages) {
					dose = lstate.batch_dosages[static_cast<size_t>(j) * n_out + k];
				} else {
					call = lstate.batch_calls[static_cast<size_t>(j) * n_out + k];
					if (gf.active) { // (3: a het read ALT|REF)
						keep = call == -9 ? gf.include_missing : gf.AllowsCall(static_cast<double>(call == 3 ? 1 : call));
					}
				}
			}
			if (keep) {
				row_variant.push_back(lstate.batch_begin + j);
				row_sample.push_back(k);
				row_call.push_back(call);
				row_dosage.push_back(dose);
			}
			if (++lstate.cur_sample == n_out) {
				lstate.cur_sample = 0;
				lstate.cur_variant++;
			}
		}
	}
	const idx_t n_rows = row_variant.size();
	for (idx_t out_col = 0; out_col < gstate.column_ids.size(); out_col++) {
		const auto file_col = gstate.column_ids[out_col];
		if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
			continue;
		}
		auto &vec = output.data[out_col];
		if (file_col < 5) {
			for (idx_t r = 0; r < n_rows; r++) {
				const auto &src = bind_data.sources[bind_data.flat_source[row_variant[r]]];
				FillVariantMetadataColumn(src.c.variants, file_col, bind_data.flat_variant[row_variant[r]], vec, r);
			}
		} else if (file_col < bind_data.genotypes_col) {
			for (idx_t r = 0; r < n_rows; r++) {
				PutSampleField(bind_data, file_col - 5, bind_data.output_samples[row_sample[r]], vec, r);
			}
		} else if (bind_data.dosages) {
			for (idx_t r = 0; r < n_rows; r++) {
				if (row_dosage[r] == -9.0) {
					FlatVector::SetNull(vec, r, true);
				} else {
					FlatVector::GetData<double>(vec)[r] = row_dosage[r];
				}
			}
		} else if (bind_data.phased) {
			for (idx_t r = 0; r < n_rows; r++) {
				PutPhasedPair(vec, r, row_call[r]);
			}
		} else {
			for (idx_t r = 0; r < n_rows; r++) {
				if (row_call[r] == -9) {
					FlatVector::SetNull(vec, r, true);
				} else {
					FlatVector::GetData<int8_t>(vec)[r] = row_call[r];
				}
			}
		}
	}
	CompatSetOutputCardinality(output, n_rows);
}

static void PfileScan(ClientContext &context, TableFunctionInput &data_p, DataChunk &output) {
	auto &bind_data = data_p.bind_data->Cast<PfileBindData>();
	auto &gstate = data_p.global_state->Cast<PfileGlobalState>();
	if (!bind_data.sample_orient) {
		// drain the sources in list order; a thread moves on when the source has nothing left for it
		auto &lstate = data_p.local_state->Cast<PfileLocalState>();
		while (lstate.current < bind_data.sources.size()) {
			TableFunctionInput inner = data_p;
			inner.bind_data = bind_data.sources[lstate.current].variant_bind.get();
			inner.global_state = gstate.variant_states[lstate.current].get();
			inner.local_state = lstate.variant_states[lstate.current].get();
			PgenScan(context, inner, output);
			if (output.size() > 0) {
				return;
			}
			lstate.current++;
			output.Reset();
		}
		CompatSetOutputCardinality(output, 0);
		return;
	}
	if (bind_data.genotype_orient) {
		GenotypeOrientScan(bind_data, gstate, data_p.local_state->Cast<PfileLocalState>(), output);
		return;
	}
	if (!gstate.datasets.empty()) {
		std::lock_guard<std::mutex> lock(gstate.phase1_mutex);
		if (!gstate.phase1_done) {
			if (bind_data.element_mode) {
				RunSampleMatrixPhase1(bind_data, gstate);
			} else {
				RunSamplePhase1(bind_data, gstate);
			}
			gstate.phase1_done = true;
		}
	}
	const auto &info = bind_data.sources[0].c.sample_info();
	const uint32_t total = gstate.use_keep ? static_cast<uint32_t>(gstate.keep.size())
	                                       : static_cast<uint32_t>(bind_data.output_samples.size());
	// a run of output rows per call, one loop per projected column
	const uint32_t first = gstate.next_idx.fetch_add(STANDARD_VECTOR_SIZE);
	const idx_t n_rows = first < total ? std::min<idx_t>(STANDARD_VECTOR_SIZE, total - first) : 0;
	auto position = [&](idx_t r) { return gstate.use_keep ? gstate.keep[first + r] : first + static_cast<uint32_t>(r); };
	for (idx_t out_col = 0; out_col < gstate.column_ids.size(); out_col++) {
		const auto file_col = gstate.column_ids[out_col];
		if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
			continue;
		}
		auto &vec = output.data[out_col];
		if (file_col < bind_data.genotypes_col) {
			// FillSampleMetadataValue (src/pfile_reader.cpp:2846-2885): SEX is an integer with 0 / NA -> NULL,
			// PAT / MAT "0" -> NULL, the usual missing tokens -> NULL
			const bool is_sex = file_col == bind_data.sex_col;
			const bool is_parent = std::find(bind_data.parent_cols.begin(), bind_data.parent_cols.end(), file_col) !=
			                       bind_data.parent_cols.end();
			for (idx_t r = 0; r < n_rows; r++) {
				const auto &fields = info.rows[bind_data.output_samples[position(r)]];
				const string val = file_col < fields.size() ? fields[file_col] : string();
				if (is_sex) {
					char *end = nullptr;
					const long parsed = PsamMissing(val) ? 0 : std::strtol(val.c_str(), &end, 10);
					if (parsed == 0 || end == val.c_str()) {
						FlatVector::SetNull(vec, r, true);
					} else {
						FlatVector::GetData<int32_t>(vec)[r] = static_cast<int32_t>(parsed);
					}
				} else if (PsamMissing(val) || (is_parent && val == "0")) {
					FlatVector::SetNull(vec, r, true);
				} else {
					FlatVector::GetData<string_t>(vec)[r] = StringVector::AddString(vec, val);
				}
			}
			continue;
		}
		if (bind_data.element_mode) {
			// one sample's calls (or dosages) over the effective variants: -9 reads as NULL
			const size_t total = bind_data.effective_total;
			auto put = [&](Vector &dst, idx_t slot, uint32_t sample_pos, size_t j) {
				if (bind_data.dosages) {
					const double d = gstate.dosage_rows[static_cast<size_t>(sample_pos) * total + j];
					if (d == -9.0) {
						FlatVector::Validity(dst).SetInvalid(slot);
						FlatVector::GetData<double>(dst)[slot] = 0.0;
					} else {
						FlatVector::GetData<double>(dst)[slot] = d;
					}
				} else if (bind_data.phased && bind_data.genotype_mode != GenotypeMode::COLUMNS) {
					PutPhasedPair(dst, slot, gstate.calls[static_cast<size_t>(sample_pos) * total + j]);
				} else {
					const int8_t g = gstate.calls[static_cast<size_t>(sample_pos) * total + j];
					if (g == -9) {
						FlatVector::Validity(dst).SetInvalid(slot);
						FlatVector::GetData<int8_t>(dst)[slot] = 0;
					} else {
						FlatVector::GetData<int8_t>(dst)[slot] = g;
					}
				}
			};
			if (bind_data.genotype_mode == GenotypeMode::COLUMNS) {
				const size_t j = file_col - bind_data.first_geno_col;
				for (idx_t r = 0; r < n_rows; r++) {
					put(vec, r, position(r), j);
				}
			} else if (bind_data.genotype_mode == GenotypeMode::STRUCT) {
				auto &fields = StructVector::GetEntries(vec);
				for (idx_t r = 0; r < n_rows; r++) {
					for (size_t j = 0; j < total; j++) {
						put(*fields[j], r, position(r), j);
					}
				}
			} else {
				const bool is_array = bind_data.genotype_mode == GenotypeMode::ARRAY;
				Vector &child = is_array ? ArrayVector::GetEntry(vec) : ListVector::GetEntry(vec);
				for (idx_t r = 0; r < n_rows; r++) {
					idx_t base = r * total;
					if (!is_array) {
						base = ListVector::GetListSize(vec);
						ListVector::Reserve(vec, base + total);
						auto *entries = FlatVector::GetData<list_entry_t>(vec);
						entries[r].offset = base;
						entries[r].length = total;
						ListVector::SetListSize(vec, base + total);
					}
					Vector &dst = is_array ? child : ListVector::GetEntry(vec);
					for (size_t j = 0; j < total; j++) {
						put(dst, base + j, position(r), j);
					}
				}
			}
			continue;
		}
		auto &entries = StructVector::GetEntries(vec);
		const double nan = std::numeric_limits<double>::quiet_NaN();
		for (idx_t r = 0; r < n_rows; r++) {
			const uint32_t *sc = gstate.counts.data() + 4 * static_cast<size_t>(position(r));
			for (int k = 0; k < 4; k++) {
				FlatVector::GetData<uint32_t>(*entries[k])[r] = sc[k];
			}
			if (bind_data.genotype_mode != GenotypeMode::STATS) {
				continue;
			}
			const uint32_t n = sc[0] + sc[1] + sc[2];
			const uint32_t all = n + sc[3];
			const double af = n ? (static_cast<double>(sc[1]) + 2.0 * sc[2]) / (2.0 * n) : nan;
			FlatVector::GetData<uint32_t>(*entries[4])[r] = n;
			FlatVector::GetData<double>(*entries[5])[r] = af;
			FlatVector::GetData<double>(*entries[6])[r] = n ? std::min(af, 1.0 - af) : nan;
			FlatVector::GetData<double>(*entries[7])[r] = all ? static_cast<double>(sc[3]) / static_cast<double>(all) : nan;
			FlatVector::GetData<uint32_t>(*entries[8])[r] = sc[1] + sc[2];
			FlatVector::GetData<double>(*entries[9])[r] = n ? static_cast<double>(sc[1]) / static_cast<double>(n) : nan;
		}
	}
	CompatSetOutputCardinality(output, n_rows);
}

void RegisterPfileReader(ExtensionLoader &loader) {
	TableFunction fn("read_pfile", {LogicalType::ANY}, PfileScan, PfileBind, PfileInitGlobal, PfileInitLocal);
	fn.projection_pushdown = true;
	fn.named_parameters["pgen"] = LogicalType::VARCHAR;
	fn.named_parameters["pvar"] = LogicalType::VARCHAR;
	fn.named_parameters["psam"] = LogicalType::VARCHAR;
	fn.named_parameters["orient"] = LogicalType::VARCHAR;
	fn.named_parameters["dosages"] = LogicalType::BOOLEAN;
	fn.named_parameters["phased"] = LogicalType::BOOLEAN;
	fn.named_parameters["region"] = LogicalType::VARCHAR;
	fn.named_parameters["samples"] = LogicalType::ANY;
	fn.named_parameters["variants"] = LogicalType::ANY;
	fn.named_parameters["genotypes"] = LogicalType::VARCHAR;
	fn.named_parameters["af_range"] = LogicalType::ANY;
	fn.named_parameters["ac_range"] = LogicalType::ANY;
	fn.named_parameters["genotype_range"] = LogicalType::ANY;
	fn.named_parameters["include_genotypes"] = LogicalType::LIST(LogicalType::VARCHAR);
	fn.named_parameters["combine_samples"] = LogicalType::VARCHAR;
	loader.RegisterFunction(fn);
}

} // namespace duckdb

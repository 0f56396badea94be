// pfile_reader.cpp -- read_pfile(prefix, pgen, pvar, psam, orient, genotypes, samples, region,
//                                variants, af_range, ac_range, include_genotypes, genotype_range,
//                                dosages, phased)
//
// The part of the reference's src/pfile_reader.cpp that sits on the genotype hot path, for ONE
// fileset (a prefix or explicit pgen/pvar/psam paths):
//   orient := 'variant'  one row per variant -- the read_pgen scan (pgen_reader.cpp) under
//                        read_pfile's name, plus `region`;
//   orient := 'sample' with genotypes := 'counts' | 'stats'
//                        one row per sample: the psam columns and the sample's
//                        {hom_ref, het, hom_alt, missing} tallies over the effective variants
//                        -- the reference's streaming aggregate (src/pfile_reader.cpp:3308-3460:
//                        every thread decodes variant batches and bumps per-sample counters,
//                        merged under a mutex).  Here phase 1 is ONE pgh_sample_counts call (three
//                        column-tally launches) by whichever thread scans first; phase 2 emits rows.
// Not carried over (they materialise a variant x sample matrix on the host or fan variants out
// to tidy rows -- no device work): orient := 'genotype', sample-oriented array/list/columns/
// struct output, multi-file lists, combine_samples, parquet companions.
#include "pgen_reader.hpp"

#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <mutex>

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

struct PfileBindData : public TableFunctionData {
	bool sample_orient = false;
	// orient := 'variant': the read_pgen bind under read_pfile's name
	unique_ptr<FunctionData> variant_bind;
	// orient := 'sample' aggregate
	PgenBindCommon c;
	GenotypeMode genotype_mode = GenotypeMode::COUNTS;
	CountFilter count_filter;
	GenotypeRangeFilter genotype_filter;
	bool has_variant_list = false;
	vector<uint32_t> variant_indices;
	idx_t genotypes_col = 0;
	idx_t sex_col = static_cast<idx_t>(-1);
	vector<idx_t> parent_cols;
	vector<uint32_t> output_samples; // file index of every output sample, ascending
};

struct PfileGlobalState : public GlobalTableFunctionState {
	// orient := 'variant'
	unique_ptr<GlobalTableFunctionState> variant_state;
	// orient := 'sample'
	vector<column_t> column_ids;
	bool need_genotypes = false;
	uint32_t max_threads_config = 0;
	shared_ptr<DeviceDataset> dataset;
	unique_ptr<DeviceSubset> subset;
	std::mutex phase1_mutex;
	bool phase1_done = false;
	vector<uint32_t> counts;     // [output sample][4]
	vector<uint32_t> keep;       // output positions that pass the genotype row filter
	bool use_keep = false;
	uint32_t effective_variants = 0;
	uint32_t candidate_variants = 0;
	std::atomic<uint32_t> next_idx {0};

	idx_t MaxThreads() const override {
		if (variant_state) {
			return variant_state->MaxThreads();
		}
		// src/pfile_reader.cpp:576-581: phase 1 is parallel over variants in the reference
		uint32_t work = std::max<uint32_t>(static_cast<uint32_t>(counts.size() / 4), candidate_variants);
		return ApplyMaxThreadsCap(work / 1000 + 1, max_threads_config);
	}
};

struct PfileLocalState : public LocalTableFunctionState {
	unique_ptr<LocalTableFunctionState> variant_state;
};

static unique_ptr<FunctionData> PfileBind(ClientContext &context, TableFunctionBindInput &input,
                                          vector<LogicalType> &return_types, vector<string> &names) {
	auto bind_data = make_uniq<PfileBindData>();
	if (input.inputs[0].type().id() == LogicalTypeId::LIST) {
		throw InvalidInputException("read_pfile: multi-file lists are not available in this build "
		                            "(pass one prefix per call)");
	}
	const string prefix = input.inputs[0].IsNull() ? string() : input.inputs[0].GetValue<string>();
	string pgen_path, orient_str = "variant";
	for (auto &kv : input.named_parameters) {
		if (kv.first == "pgen") {
			pgen_path = kv.second.GetValue<string>();
		} else if (kv.first == "orient") {
			orient_str = Lowered(kv.second.GetValue<string>());
		} else if (kv.first == "combine_samples") {
			throw InvalidInputException("read_pfile: combine_samples needs a multi-file list, which is not available "
			                            "in this build");
		}
	}
	if (orient_str != "variant" && orient_str != "sample" && orient_str != "genotype") {
		throw InvalidInputException("read_pfile: invalid orient value '%s' (expected 'variant', 'genotype', or 'sample')",
		                            orient_str);
	}
	// --- the three paths (src/pfile_reader.cpp:670-760) ---
	string eff_prefix = prefix;
	if (pgen_path.empty()) {
		if (prefix.empty()) {
			throw InvalidInputException("read_pfile: no .pgen file path provided");
		}
		if (FileExists(prefix + ".pgen")) {
			pgen_path = prefix + ".pgen";
		} else if (FileExists(prefix)) {
			pgen_path = prefix; // a full .pgen path given as the prefix
			if (prefix.size() > 5 && prefix.compare(prefix.size() - 5, 5, ".pgen") == 0) {
				eff_prefix = prefix.substr(0, prefix.size() - 5);
			}
		} else {
			throw InvalidInputException("read_pfile: cannot find .pgen file for prefix '%s' (tried '%s')", prefix,
			                            prefix + ".pgen");
		}
	}
	// the shared bind takes (pgen path; pvar, psam, ...): drop what only read_pfile knows
	TableFunctionBindInput inner;
	inner.inputs.push_back(Value::VARCHAR(pgen_path));
	for (auto &kv : input.named_parameters) {
		if (kv.first == "region") {
			inner.named_parameters[kv.first] = Value::VARCHAR(CanonicalRegion(kv.second.GetValue<string>()));
		} else if (kv.first != "pgen" && kv.first != "orient") {
			inner.named_parameters[kv.first] = kv.second;
		}
	}
	if (!inner.named_parameters.count("pvar") && !eff_prefix.empty()) {
		for (const char *ext : {".pvar", ".bim"}) {
			if (FileExists(eff_prefix + ext)) {
				inner.named_parameters["pvar"] = Value::VARCHAR(eff_prefix + ext);
				break;
			}
		}
	}
	if (!inner.named_parameters.count("psam") && !eff_prefix.empty()) {
		for (const char *ext : {".psam", ".fam"}) {
			if (FileExists(eff_prefix + ext)) {
				inner.named_parameters["psam"] = Value::VARCHAR(eff_prefix + ext);
				break;
			}
		}
	}

	string genotypes_str = "auto";
	auto genotypes_it = input.named_parameters.find("genotypes");
	if (genotypes_it != input.named_parameters.end()) {
		genotypes_str = Lowered(genotypes_it->second.GetValue<string>());
	}
	if (orient_str == "variant") {
		bind_data->variant_bind = PgenBindNamed(context, inner, return_types, names, "read_pfile", true);
		return std::move(bind_data);
	}
	if (orient_str == "genotype") {
		if (genotypes_str == "counts" || genotypes_str == "stats") {
			throw InvalidInputException("read_pfile: genotypes := '%s' is not compatible with orient := 'genotype' "
			                            "(aggregate modes require orient := 'variant' or 'sample')",
			                            genotypes_str);
		}
		throw InvalidInputException("read_pfile: orient := 'genotype' is not available in this build "
		                            "(use orient := 'variant', or orient := 'sample' with genotypes := 'counts'|'stats')");
	}

	// --- orient := 'sample' ---
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
	auto &c = bind_data->c;
	c.Bind(context, inner, "read_pfile", true);
	bind_data->genotype_mode = ResolveGenotypeMode(genotypes_str, c.raw_variant_ct, "read_pfile");
	if (!IsAggregateGenotypeMode(bind_data->genotype_mode)) {
		throw InvalidInputException("read_pfile: orient := 'sample' with genotypes := '%s' is not available in this "
		                            "build (the sample-oriented matrix is assembled on the host; use genotypes := "
		                            "'counts' or 'stats')",
		                            genotypes_str);
	}
	const char *label = bind_data->genotype_mode == GenotypeMode::COUNTS ? "counts" : "stats";
	if (phased) {
		throw InvalidInputException("read_pfile: genotypes := '%s' is incompatible with phased := true", label);
	}
	if (dosages) {
		throw InvalidInputException("read_pfile: genotypes := '%s' is incompatible with dosages := true", label);
	}
	auto variants_it = input.named_parameters.find("variants");
	if (variants_it != input.named_parameters.end()) {
		bind_data->variant_indices = ResolveVariantsParameter(variants_it->second, c.variants, c.raw_variant_ct, "read_pfile");
		std::sort(bind_data->variant_indices.begin(), bind_data->variant_indices.end());
		bind_data->has_variant_list = true;
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
	// schema: every psam column (SEX is INTEGER, the rest VARCHAR), then the aggregate struct
	for (idx_t i = 0; i < c.sample_info.column_names.size(); i++) {
		const string &name = c.sample_info.column_names[i];
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
	names.push_back("genotypes");
	return_types.push_back(bind_data->genotype_mode == GenotypeMode::COUNTS ? MakeGenotypeCountsType()
	                                                                         : MakeGenotypeStatsType());
	return std::move(bind_data);
}

static unique_ptr<GlobalTableFunctionState> PfileInitGlobal(ClientContext &context, TableFunctionInitInput &input) {
	auto &bind_data = input.bind_data->Cast<PfileBindData>();
	auto state = make_uniq<PfileGlobalState>();
	if (!bind_data.sample_orient) {
		TableFunctionInitInput inner = input;
		inner.bind_data = bind_data.variant_bind.get();
		state->variant_state = PgenInitGlobal(context, inner);
		return std::move(state);
	}
	state->column_ids = input.column_ids;
	state->max_threads_config = GetPlinkingMaxThreads(context);
	for (auto col_id : input.column_ids) {
		if (col_id == bind_data.genotypes_col) {
			state->need_genotypes = true;
		}
	}
	const auto &c = bind_data.c;
	state->candidate_variants = bind_data.has_variant_list ? static_cast<uint32_t>(bind_data.variant_indices.size())
	                                                       : c.RangeEnd() - c.RangeStart();
	state->counts.assign(4 * bind_data.output_samples.size(), 0);
	// a row filter needs the tallies even when the struct itself is not projected
	if (state->need_genotypes || bind_data.genotype_filter.active) {
		state->dataset = DeviceDataset::Acquire(c.pgen_path, "read_pfile");
		if (c.has_sample_subset) {
			state->subset = make_uniq<DeviceSubset>(*state->dataset, c.sample_subset->sample_include, "read_pfile");
		}
	}
	return std::move(state);
}

static unique_ptr<LocalTableFunctionState> PfileInitLocal(ExecutionContext &context, TableFunctionInitInput &input,
                                                          GlobalTableFunctionState *global_state) {
	auto &bind_data = input.bind_data->Cast<PfileBindData>();
	auto state = make_uniq<PfileLocalState>();
	if (!bind_data.sample_orient) {
		TableFunctionInitInput inner = input;
		inner.bind_data = bind_data.variant_bind.get();
		state->variant_state = PgenInitLocal(context, inner, global_state->Cast<PfileGlobalState>().variant_state.get());
	}
	return std::move(state);
}

//! Phase 1: the effective variants (region, variants, af/ac filters) and every sample's tallies.
static void RunSamplePhase1(const PfileBindData &bind_data, PfileGlobalState &gstate) {
	const auto &c = bind_data.c;
	pgh_dataset *ds = gstate.dataset->handle;
	pgh_subset *ss = gstate.subset ? gstate.subset->handle : nullptr;
	char errbuf[PGH_ERRBUF_LEN] = {0};
	vector<uint32_t> list;
	if (bind_data.has_variant_list) {
		for (auto v : bind_data.variant_indices) {
			if (!c.variant_range.has_filter || (v >= c.RangeStart() && v < c.RangeEnd())) {
				list.push_back(v);
			}
		}
	}
	const bool listed = bind_data.has_variant_list;
	uint32_t begin = c.RangeStart(), n_var = listed ? static_cast<uint32_t>(list.size()) : c.RangeEnd() - c.RangeStart();
	if (bind_data.count_filter.HasFilter() && n_var) {
		// per-variant tallies of the candidates decide which of them stay
		vector<uint32_t> vc(4 * static_cast<size_t>(n_var));
		vector<uint32_t> kept;
		auto consider = [&](uint32_t v, const uint32_t *gc) {
			GenotypeRangeFilter none;
			if (!CheckPreDecompFilters(bind_data.count_filter, none, gc, c.effective_sample_ct).skip) {
				kept.push_back(v);
			}
		};
		if (listed) {
			for (uint32_t i = 0; i < n_var; i++) {
				if (pgh_counts_range(ds, ss, list[i], list[i] + 1, reinterpret_cast<uint32_t(*)[4]>(vc.data() + 4 * i),
				                     errbuf) != PGH_OK) {
					throw IOException("read_pfile: PgrGetCounts failed for variant %u: %s", list[i], string(errbuf));
				}
				consider(list[i], vc.data() + 4 * i);
			}
		} else {
			if (pgh_counts_range(ds, ss, begin, begin + n_var, reinterpret_cast<uint32_t(*)[4]>(vc.data()), errbuf) !=
			    PGH_OK) {
				throw IOException("read_pfile: PgrGetCounts failed for variants [%u, %u): %s", begin, begin + n_var,
				                  string(errbuf));
			}
			for (uint32_t i = 0; i < n_var; i++) {
				consider(begin + i, vc.data() + 4 * i);
			}
		}
		list.swap(kept);
		n_var = static_cast<uint32_t>(list.size());
		gstate.effective_variants = n_var;
		if (pgh_sample_counts(ds, ss, 0, n_var, list.data(), reinterpret_cast<uint32_t(*)[4]>(gstate.counts.data()),
		                      errbuf) != PGH_OK) {
			throw IOException("read_pfile: PgrGet failed during sample-orient aggregation: %s", string(errbuf));
		}
	} else {
		gstate.effective_variants = n_var;
		if (pgh_sample_counts(ds, ss, listed ? 0 : begin, n_var, listed ? list.data() : nullptr,
		                      reinterpret_cast<uint32_t(*)[4]>(gstate.counts.data()), errbuf) != PGH_OK) {
			throw IOException("read_pfile: PgrGet failed during sample-orient aggregation: %s", string(errbuf));
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

static void PfileScan(ClientContext &context, TableFunctionInput &data_p, DataChunk &output) {
	auto &bind_data = data_p.bind_data->Cast<PfileBindData>();
	auto &gstate = data_p.global_state->Cast<PfileGlobalState>();
	if (!bind_data.sample_orient) {
		TableFunctionInput inner = data_p;
		inner.bind_data = bind_data.variant_bind.get();
		inner.global_state = gstate.variant_state.get();
		inner.local_state = data_p.local_state->Cast<PfileLocalState>().variant_state.get();
		PgenScan(context, inner, output);
		return;
	}
	if (gstate.dataset) {
		std::lock_guard<std::mutex> lock(gstate.phase1_mutex);
		if (!gstate.phase1_done) {
			RunSamplePhase1(bind_data, gstate);
			gstate.phase1_done = true;
		}
	}
	const auto &info = bind_data.c.sample_info;
	const uint32_t total = gstate.use_keep ? static_cast<uint32_t>(gstate.keep.size())
	                                       : static_cast<uint32_t>(bind_data.output_samples.size());
	idx_t rows = 0;
	while (rows < STANDARD_VECTOR_SIZE) {
		uint32_t claim = static_cast<uint32_t>(std::min<idx_t>(128, STANDARD_VECTOR_SIZE - rows));
		uint32_t first = gstate.next_idx.fetch_add(claim);
		if (first >= total) {
			break;
		}
		uint32_t last = std::min(first + claim, total);
		for (uint32_t idx = first; idx < last; idx++, rows++) {
			const uint32_t pos = gstate.use_keep ? gstate.keep[idx] : idx;
			const uint32_t file_idx = bind_data.output_samples[pos];
			for (idx_t out_col = 0; out_col < gstate.column_ids.size(); out_col++) {
				auto file_col = gstate.column_ids[out_col];
				if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
					continue;
				}
				auto &vec = output.data[out_col];
				if (file_col < bind_data.genotypes_col) {
					// FillSampleMetadataValue (src/pfile_reader.cpp:2846-2885)
					const auto &row = info.rows[file_idx];
					const string val = file_col < row.size() ? row[file_col] : string();
					if (file_col == bind_data.sex_col) {
						int32_t sex = 0;
						if (!PsamMissing(val)) {
							char *end = nullptr;
							long parsed = std::strtol(val.c_str(), &end, 10);
							sex = end != val.c_str() ? static_cast<int32_t>(parsed) : 0;
						}
						if (sex == 0) {
							FlatVector::SetNull(vec, rows, true);
						} else {
							FlatVector::GetData<int32_t>(vec)[rows] = sex;
						}
						continue;
					}
					bool is_parent = std::find(bind_data.parent_cols.begin(), bind_data.parent_cols.end(), file_col) !=
					                 bind_data.parent_cols.end();
					if (PsamMissing(val) || (is_parent && val == "0")) {
						FlatVector::SetNull(vec, rows, true);
					} else {
						FlatVector::GetData<string_t>(vec)[rows] = StringVector::AddString(vec, val);
					}
					continue;
				}
				const uint32_t *sc = gstate.counts.data() + 4 * static_cast<size_t>(pos);
				auto &entries = StructVector::GetEntries(vec);
				for (int k = 0; k < 4; k++) {
					FlatVector::GetData<uint32_t>(*entries[k])[rows] = sc[k];
				}
				if (bind_data.genotype_mode == GenotypeMode::STATS) {
					const double nan = std::numeric_limits<double>::quiet_NaN();
					const uint32_t n = sc[0] + sc[1] + sc[2];
					const uint32_t all = n + sc[3];
					FlatVector::GetData<uint32_t>(*entries[4])[rows] = n;
					const double af = n ? (static_cast<double>(sc[1]) + 2.0 * sc[2]) / (2.0 * n) : nan;
					FlatVector::GetData<double>(*entries[5])[rows] = af;
					FlatVector::GetData<double>(*entries[6])[rows] = n ? std::min(af, 1.0 - af) : nan;
					FlatVector::GetData<double>(*entries[7])[rows] =
					    all ? static_cast<double>(sc[3]) / static_cast<double>(all) : nan;
					FlatVector::GetData<uint32_t>(*entries[8])[rows] = sc[1] + sc[2];
					FlatVector::GetData<double>(*entries[9])[rows] =
					    n ? static_cast<double>(sc[1]) / static_cast<double>(n) : nan;
				}
			}
		}
	}
	CompatSetOutputCardinality(output, rows);
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

// pgen_reader.cpp -- read_pgen(path, pvar, psam, samples, genotypes, dosages, phased,
//                               af_range, ac_range, include_genotypes, genotype_range, orient)
//
// Surface of the reference's src/pgen_reader.cpp for the variant-oriented
// genotype column: ARRAY/LIST(TINYINT) hardcalls (missing -> NULL element),
// dosages, phased pairs, and the counts / stats aggregate structs.  Per chunk
// the hardcall path is one 2-bit -> int8 unpack launch (pgh_unpack_range) over
// the claimed variants instead of PgrGet + GenoarrToBytesMinus9 + a scalar
// per-sample copy (src/pgen_reader.cpp:727-733, 1009-1047).  Dosage and phase
// tracks are decoded per variant through the reader calls.  genotypes :=
// 'columns' / 'struct' lay the same unpacked bytes out as one scalar column (or
// struct field) per sample (src/pgen_reader.cpp:386-433, 781-845); `variants :=`
// scans a caller-ordered list of variant indices (src/pgen_reader.cpp:304-312).
#include "pgen_reader.hpp"

#include <cmath>
#include <limits>

namespace duckdb {

static unique_ptr<FunctionData> PgenBind(ClientContext &context, TableFunctionBindInput &input,
                                         vector<LogicalType> &return_types, vector<string> &names) {
	return PgenBindNamed(context, input, return_types, names, "read_pgen", false);
}

unique_ptr<FunctionData> PgenBindNamed(ClientContext &context, TableFunctionBindInput &input,
                                       vector<LogicalType> &return_types, vector<string> &names, const string &func,
                                       bool with_region) {
	auto bind_data = make_uniq<PgenBindData>();
	bind_data->func = func;
	const char *fn = bind_data->func.c_str();
	if (!with_region && input.named_parameters.count("region")) {
		throw BinderException("Invalid named parameter \"region\" for function %s", fn);
	}
	for (auto &kv : input.named_parameters) {
		if (kv.first == "dosages") {
			bind_data->include_dosages = kv.second.GetValue<bool>();
		} else if (kv.first == "phased") {
			bind_data->include_phased = kv.second.GetValue<bool>();
		} else if (kv.first == "orient") {
			auto v = kv.second.GetValue<string>();
			string lower = v;
			for (auto &ch : lower) {
				ch = static_cast<char>(std::tolower(static_cast<unsigned char>(ch)));
			}
			if (lower != "variant") {
				throw InvalidInputException("%s: orient := '%s' is not supported "
				                            "(read_pgen only supports orient := 'variant'; "
				                            "use read_pfile for orient := 'genotype' or 'sample')", fn,
				                            v);
			}
		}
	}
	if (bind_data->include_dosages && bind_data->include_phased) {
		throw InvalidInputException("%s: dosages and phased cannot both be true", fn);
	}
	auto &c = bind_data->c;
	c.Bind(context, input, bind_data->func, false);
	uint32_t output_sample_ct = c.effective_sample_ct;
	bind_data->output_sample_ct = output_sample_ct;

	auto variants_it = input.named_parameters.find("variants");
	if (variants_it != input.named_parameters.end()) {
		bind_data->variant_indices =
		    ResolveVariantsParameter(variants_it->second, c.variants, c.raw_variant_ct, bind_data->func);
		bind_data->has_variant_list = true;
		if (c.variant_range.has_filter) {
			// read_pfile: region and variants intersect
			vector<uint32_t> kept;
			for (auto v : bind_data->variant_indices) {
				if (v >= c.RangeStart() && v < c.RangeEnd()) {
					kept.push_back(v);
				}
			}
			bind_data->variant_indices.swap(kept);
		}
	}

	auto af_it = input.named_parameters.find("af_range");
	if (af_it != input.named_parameters.end()) {
		bind_data->count_filter.af_filter = ParseRangeFilter(af_it->second, "af_range", 0.0, 1.0, bind_data->func);
	}
	auto ac_it = input.named_parameters.find("ac_range");
	if (ac_it != input.named_parameters.end()) {
		bind_data->count_filter.ac_filter =
		    ParseRangeFilter(ac_it->second, "ac_range", 0.0, static_cast<double>(2 * output_sample_ct), bind_data->func);
	}
	auto ig_it = input.named_parameters.find("include_genotypes");
	auto gr_it = input.named_parameters.find("genotype_range");
	bool has_ig = ig_it != input.named_parameters.end();
	bool has_gr = gr_it != input.named_parameters.end();
	if (has_ig && has_gr) {
		throw InvalidInputException("%s: specify only one of include_genotypes or genotype_range (genotype_range is the numeric "
		    "alias of include_genotypes)", fn);
	}
	if ((has_ig || has_gr) && bind_data->include_dosages) {
		throw InvalidInputException("%s: %s is incompatible with dosages := true", fn,
		                            has_ig ? "include_genotypes" : "genotype_range");
	}
	if (has_ig) {
		ParseIncludeGenotypes(ig_it->second, bind_data->genotype_filter, bind_data->func);
	} else if (has_gr) {
		bool inc_missing = false;
		RangeFilter range = ParseRangeFilter(gr_it->second, "genotype_range", 0.0, 2.0, bind_data->func, &inc_missing);
		bind_data->genotype_filter.SetFromRange(range, inc_missing);
	}

	string genotypes_str = "auto";
	auto genotypes_it = input.named_parameters.find("genotypes");
	if (genotypes_it != input.named_parameters.end()) {
		genotypes_str = genotypes_it->second.GetValue<string>();
	}
	bind_data->genotype_mode = ResolveGenotypeMode(genotypes_str, output_sample_ct, bind_data->func);
	if (IsAggregateGenotypeMode(bind_data->genotype_mode)) {
		const char *label = bind_data->genotype_mode == GenotypeMode::COUNTS ? "counts" : "stats";
		if (bind_data->include_phased) {
			throw InvalidInputException("%s: genotypes := '%s' is incompatible with phased := true", fn, label);
		}
		if (bind_data->include_dosages) {
			throw InvalidInputException("%s: genotypes := '%s' is incompatible with dosages := true", fn, label);
		}
	}
	const bool per_sample_names =
	    bind_data->genotype_mode == GenotypeMode::COLUMNS || bind_data->genotype_mode == GenotypeMode::STRUCT;
	if (per_sample_names) {
		if (!c.has_sample_info) {
			throw InvalidInputException("%s: genotypes := '%s' requires a .psam/.fam file for sample IDs "
			                            "(no companion file found)", fn,
			                            bind_data->genotype_mode == GenotypeMode::COLUMNS ? "columns" : "struct");
		}
		// the subset is a mask, so fields come out in ascending file order whatever order was asked for
		for (uint32_t s = 0; s < c.raw_sample_ct; s++) {
			if (!c.has_sample_subset || ((c.sample_subset->sample_include[s >> 6] >> (s & 63)) & 1ull)) {
				bind_data->genotype_column_names.push_back(c.sample_info().iids[s]);
			}
		}
	}

	names = {"CHROM", "POS", "ID", "REF", "ALT"};
	return_types = {LogicalType::VARCHAR, LogicalType::INTEGER, LogicalType::VARCHAR, LogicalType::VARCHAR,
	                LogicalType::VARCHAR};
	LogicalType elem_type = bind_data->include_phased    ? LogicalType::ARRAY(LogicalType::TINYINT, 2)
	                        : bind_data->include_dosages ? LogicalType(LogicalType::DOUBLE)
	                                                     : LogicalType(LogicalType::TINYINT);
	if (bind_data->genotype_mode == GenotypeMode::COLUMNS) {
		// scalar columns carry hardcalls or dosages; phase is not representable here
		LogicalType col_type =
		    bind_data->include_dosages ? LogicalType(LogicalType::DOUBLE) : LogicalType(LogicalType::TINYINT);
		for (auto &iid : bind_data->genotype_column_names) {
			names.push_back(iid);
			return_types.push_back(col_type);
		}
		return std::move(bind_data);
	}
	names.push_back("genotypes");
	if (bind_data->genotype_mode == GenotypeMode::STRUCT) {
		child_list_t fields;
		for (auto &iid : bind_data->genotype_column_names) {
			fields.push_back({iid, elem_type});
		}
		return_types.push_back(LogicalType::STRUCT(std::move(fields)));
	} else if (bind_data->genotype_mode == GenotypeMode::COUNTS) {
		return_types.push_back(MakeGenotypeCountsType());
	} else if (bind_data->genotype_mode == GenotypeMode::STATS) {
		return_types.push_back(MakeGenotypeStatsType());
	} else {
		return_types.push_back(bind_data->genotype_mode == GenotypeMode::ARRAY
		                           ? LogicalType::ARRAY(elem_type, output_sample_ct)
		                           : LogicalType::LIST(elem_type));
	}
	return std::move(bind_data);
}

unique_ptr<GlobalTableFunctionState> PgenInitGlobal(ClientContext &context, TableFunctionInitInput &input) {
	auto &bind_data = input.bind_data->Cast<PgenBindData>();
	auto state = make_uniq<PgenGlobalState>();
	state->scan.start_variant_idx = bind_data.c.RangeStart(); // the whole file unless read_pfile passed a region
	state->scan.end_variant_idx = bind_data.c.RangeEnd();
	state->scan.next_variant_idx.store(state->scan.start_variant_idx);
	state->scan.effective_sample_ct = bind_data.c.effective_sample_ct;
	state->column_ids = input.column_ids;
	state->max_threads_config = GetPlinkingMaxThreads(context);
	if (bind_data.has_variant_list) {
		state->scan.has_variant_list = true;
		state->scan.variant_list = bind_data.variant_indices;
	}
	for (auto col_id : input.column_ids) {
		// columns mode: every column from COL_GENOTYPES on is one sample
		if (col_id != COLUMN_IDENTIFIER_ROW_ID &&
		    (col_id == COL_GENOTYPES || (bind_data.genotype_mode == GenotypeMode::COLUMNS && col_id > COL_GENOTYPES))) {
			state->need_genotypes = true;
		}
	}
	state->need_counts = bind_data.count_filter.HasFilter() || bind_data.genotype_filter.active ||
	                     (state->need_genotypes && IsAggregateGenotypeMode(bind_data.genotype_mode));
	state->scan.want_counts = state->need_counts;
	if (state->need_genotypes && !IsAggregateGenotypeMode(bind_data.genotype_mode)) {
		state->scan.claim = kUnpackSpan; // one output chunk per claim: every scan thread gets rows to unpack
	}
	if (state->need_genotypes || state->need_counts) {
		state->scan.dataset = DeviceDataset::Acquire(bind_data.c.pgen_path, bind_data.func);
		if (bind_data.c.has_sample_subset) {
			state->scan.subset =
			    make_uniq<DeviceSubset>(*state->scan.dataset, bind_data.c.sample_subset->sample_include, bind_data.func);
		}
		// the filters' and the counts / stats modes' tallies: the range's pass (shared with plink_freq & co.)
		state->scan.StartTallies(bind_data.c.sample_subset.get(), nullptr, bind_data.c.raw_sample_ct, nullptr, 0u, false,
		                         GetPlinkingTallyCache(context), bind_data.func);
	}
	return std::move(state);
}

unique_ptr<LocalTableFunctionState> PgenInitLocal(ExecutionContext &, TableFunctionInitInput &input,
                                                  GlobalTableFunctionState *global_state) {
	auto &bind_data = input.bind_data->Cast<PgenBindData>();
	const char *fn = bind_data.func.c_str();
	auto &gstate = global_state->Cast<PgenGlobalState>();
	auto state = make_uniq<PgenLocalState>();
	const bool phased_out = bind_data.include_phased && bind_data.genotype_mode != GenotypeMode::COLUMNS;
	const bool pipelined = gstate.need_genotypes && !IsAggregateGenotypeMode(bind_data.genotype_mode) &&
	                       !bind_data.include_dosages && !phased_out && !gstate.scan.has_variant_list;
	if (pipelined && !gstate.scan.dataset->streamed) { // (a streamed file's hardcalls come window by window: LeaseRows)
		char errbuf[PGH_ERRBUF_LEN] = {0};
		if (pgh_reader_create(gstate.scan.dataset->Resident(bind_data.func), gstate.scan.subset ? gstate.scan.subset->handle : nullptr,
		                      &state->reader, errbuf) != PGH_OK) {
			throw IOException("%s: thread init failed: %s", fn, string(errbuf));
		}
	}
	if (gstate.need_genotypes && (bind_data.include_dosages || phased_out)) {
		if (!gstate.scan.dataset->streamed) { // (streamed: the scan makes its reader on the window it leases)
			char errbuf[PGH_ERRBUF_LEN] = {0};
			int rc = pgh_reader_create(gstate.scan.dataset->Resident(bind_data.func),
			                           gstate.scan.subset ? gstate.scan.subset->handle : nullptr, &state->reader, errbuf);
			if (rc != PGH_OK) {
				throw IOException("%s: thread init failed: %s", fn, string(errbuf));
			}
		}
		uint32_t n = bind_data.output_sample_ct;
		state->dosage_doubles.resize(n);
		state->genovec.resize((n + 31) / 32);
		state->phasepresent.resize((n + 63) / 64);
		state->phaseinfo.resize((n + 63) / 64);
	}
	return std::move(state);
}

namespace {

// child element slot for row `row`: ARRAY rows are fixed stride, LIST rows append
idx_t BeginGenotypeRow(const PgenBindData &bind_data, Vector &vec, idx_t row, uint32_t n) {
	if (bind_data.genotype_mode == GenotypeMode::ARRAY) {
		return row * n;
	}
	idx_t offset = ListVector::GetListSize(vec);
	ListVector::Reserve(vec, offset + n);
	auto *entries = FlatVector::GetData<list_entry_t>(vec);
	entries[row].offset = offset;
	entries[row].length = n;
	ListVector::SetListSize(vec, offset + n);
	return offset;
}

Vector &GenotypeChild(const PgenBindData &bind_data, Vector &vec) {
	return bind_data.genotype_mode == GenotypeMode::ARRAY ? ArrayVector::GetEntry(vec) : ListVector::GetEntry(vec);
}

} // namespace

void PgenScan(ClientContext &, TableFunctionInput &data_p, DataChunk &output) {
	auto &bind_data = data_p.bind_data->Cast<PgenBindData>();
	const char *fn = bind_data.func.c_str();
	auto &gstate = data_p.global_state->Cast<PgenGlobalState>();
	auto &lstate = data_p.local_state->Cast<PgenLocalState>();
	auto &column_ids = gstate.column_ids;
	const uint32_t n = bind_data.output_sample_ct;
	const GenotypeMode mode = bind_data.genotype_mode;
	const bool listed = gstate.scan.has_variant_list;
	const bool phased_out = bind_data.include_phased && mode != GenotypeMode::COLUMNS;
	const bool dosage_rows = gstate.need_genotypes && bind_data.include_dosages;
	const bool per_variant_decode = gstate.need_genotypes && !dosage_rows && phased_out;
	const bool plain_hardcalls =
	    gstate.need_genotypes && !IsAggregateGenotypeMode(mode) && !bind_data.include_dosages && !phased_out;
	const auto t_plan0 = std::chrono::steady_clock::now();
	auto ms_since = [](std::chrono::steady_clock::time_point a) {
		return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count();
	};
	// 1. choose the variants of a chunk (filters run off the range's tallies).  A chunk never straddles
	//    two claims: its unpack span is one launch.
	auto plan_chunk = [&](vector<RowPlan> &plan) {
		plan.clear();
		plan.reserve(STANDARD_VECTOR_SIZE);
		uint32_t vidx = 0;
		while (plan.size() < STANDARD_VECTOR_SIZE) {
			if (!plan.empty() &&
			    (lstate.scan.BatchDrained() || (!listed && vidx + 1 - plan.front().vidx >= kUnpackSpan))) {
				break; // the next chunk continues
			}
			if (!lstate.scan.Next(gstate.scan, bind_data.func, vidx)) {
				break;
			}
			bool all_pass = true;
			if (bind_data.count_filter.HasFilter() || bind_data.genotype_filter.active) {
				auto pf = CheckPreDecompFilters(bind_data.count_filter, bind_data.genotype_filter,
				                                lstate.scan.Counts(vidx), n);
				if (pf.skip) {
					continue;
				}
				all_pass = pf.all_pass;
			}
			plan.push_back({vidx, all_pass});
		}
	};
	const size_t val_words = (n + 63) / 64;
	static const bool direct_env = [] {
		const char *e = std::getenv("PLINKING_UNPACK_DIRECT");
		return e && *e == '1';
	}();
	// PLINKING_UNPACK_PIPELINE=1: chunk k + 1 is unpacked and copied (pgh_reader_unpack_start) while chunk k is
	// filled.  Off by default: with sixteen scan threads the host side is the limit either way -- 21 GB/s with the
	// pipeline (two pinned gigabytes per thread) against 25-28 without at 500,000 samples -- it pays with few threads.
	const char *pipeline_env = std::getenv("PLINKING_UNPACK_PIPELINE");
	const bool pipelined = plain_hardcalls && !listed && !direct_env && pipeline_env && *pipeline_env == '1' &&
	                       lstate.reader != nullptr;
	auto launch_slot = [&](int k) {
		auto &sl = lstate.slot[k];
		if (sl.plan.empty()) {
			return;
		}
		const uint32_t b = sl.plan.front().vidx, e = sl.plan.back().vidx + 1;
		sl.bytes.resize(static_cast<size_t>(e - b) * n);
		sl.validity.resize(static_cast<size_t>(e - b) * val_words);
		if (pgh_reader_unpack_start(lstate.reader, k, b, e, sl.bytes.data(), sl.validity.data(), 0) != PGH_OK) {
			throw IOException("%s: PgrGet failed for variants [%u, %u): %s", fn, b, e,
			                  string(pgh_reader_error(lstate.reader)));
		}
	};
	vector<RowPlan> plan_one;
	if (pipelined) {
		if (!lstate.primed) {
			plan_chunk(lstate.slot[lstate.cur].plan);
			launch_slot(lstate.cur);
			lstate.primed = true;
		}
		// the chunk after this one goes on its way before this one is waited for
		plan_chunk(lstate.slot[lstate.cur ^ 1].plan);
		launch_slot(lstate.cur ^ 1);
	} else {
		plan_chunk(plan_one);
	}
	const vector<RowPlan> &plan = pipelined ? lstate.slot[lstate.cur].plan : plan_one;
	if (plan.empty()) {
		CompatSetOutputCardinality(output, 0);
		return;
	}

	lstate.plan_ms += ms_since(t_plan0);
	const auto t_unpack0 = std::chrono::steady_clock::now();
	// 2. unpack: one launch for the span the chunk covers, or one per listed variant
	const uint32_t span_begin = plan.front().vidx;
	const uint32_t span_end = plan.back().vidx + 1;
	const int8_t *chunk_bytes = nullptr;     // row r of the span at chunk_bytes + r * n
	const uint64_t *chunk_validity = nullptr;
	// PLINKING_UNPACK_DIRECT=1: a LIST / ARRAY chunk without gaps is unpacked straight into the output vector's
	// child buffer (pageable memory: the runtime stages the copy) instead of into this thread's pinned block and
	// from there with a memcpy.  Which one wins is a property of the host: measured in tools/shell_bench.py.
	int8_t *direct_dst = nullptr;
	if (plain_hardcalls && direct_env && !listed && (mode == GenotypeMode::LIST || mode == GenotypeMode::ARRAY) &&
	    plan.size() == span_end - span_begin) {
		for (idx_t out_col = 0; out_col < column_ids.size(); out_col++) {
			if (column_ids[out_col] == COL_GENOTYPES) {
				auto &vec = output.data[out_col];
				if (mode == GenotypeMode::LIST) {
					ListVector::Reserve(vec, plan.size() * static_cast<idx_t>(n));
				}
				direct_dst = FlatVector::GetData<int8_t>(GenotypeChild(bind_data, vec)); // row r at r * n
			}
		}
	}
	const vector<uint64_t> *mask = bind_data.c.has_sample_subset ? &bind_data.c.sample_subset->sample_include : nullptr;
	auto rows_of = [&](uint32_t b, uint32_t e) {
		return LeaseRows(*gstate.scan.dataset, gstate.scan.subset.get(), gstate.scan.row_windows, mask, b, e,
		                 bind_data.c.raw_variant_ct, bind_data.func);
	};
	const bool streamed = gstate.scan.dataset && gstate.scan.dataset->streamed;
	if (pipelined) {
		if (pgh_reader_unpack_wait(lstate.reader, lstate.cur) != PGH_OK) {
			throw IOException("%s: PgrGet failed for variants [%u, %u): %s", fn, span_begin, span_end,
			                  string(pgh_reader_error(lstate.reader)));
		}
		chunk_bytes = lstate.slot[lstate.cur].bytes.data();
		chunk_validity = lstate.slot[lstate.cur].validity.data();
	} else if (plain_hardcalls) {
		const size_t rows = listed ? plan.size() : span_end - span_begin;
		if (!direct_dst) {
			lstate.bytes.resize(rows * n);
		}
		lstate.validity.resize(rows * val_words);
		char errbuf[PGH_ERRBUF_LEN] = {0};
		int rc = PGH_OK;
		if (listed) {
			for (size_t r = 0; r < plan.size() && rc == PGH_OK; r++) {
				RowLease rows_r = rows_of(plan[r].vidx, plan[r].vidx + 1);
				rc = pgh_unpack_range(rows_r.ds, rows_r.ss, plan[r].vidx, plan[r].vidx + 1, lstate.bytes.data() + r * n,
				                      lstate.validity.data() + r * val_words, 0, errbuf);
			}
		} else {
			RowLease span = rows_of(span_begin, span_end);
			rc = pgh_unpack_range(span.ds, span.ss, span_begin, span_end, direct_dst ? direct_dst : lstate.bytes.data(),
			                      lstate.validity.data(), 0, errbuf);
		}
		if (rc != PGH_OK) {
			throw IOException("%s: PgrGet failed for variants [%u, %u): %s", fn, span_begin, span_end,
			                  string(errbuf));
		}
		chunk_bytes = direct_dst ? direct_dst : lstate.bytes.data();
		chunk_validity = lstate.validity.data();
	}

	const double *dose = nullptr; // the current row of the chunk's dosages
	if (dosage_rows) {
		// PgrGetD + Dosage16ToDoublesMinus9 for the chunk's variants in one device call
		vector<uint32_t> chunk_vidx(plan.size());
		for (size_t r = 0; r < plan.size(); r++) {
			chunk_vidx[r] = plan[r].vidx;
		}
		lstate.dosage_doubles.resize(plan.size() * static_cast<size_t>(n));
		char errbuf[PGH_ERRBUF_LEN] = {0};
		int rc = PGH_OK;
		if (streamed && listed) { // (a list need not be in file order: a window per variant, re-used while it lasts)
			for (size_t r = 0; r < plan.size() && rc == PGH_OK; r++) {
				RowLease rows_r = rows_of(chunk_vidx[r], chunk_vidx[r] + 1);
				rc = pgh_dosage_unpack(rows_r.ds, rows_r.ss, 0, 1, &chunk_vidx[r], lstate.dosage_doubles.data() + r * static_cast<size_t>(n),
				                       errbuf);
			}
		} else {
			RowLease span = rows_of(span_begin, span_end);
			rc = pgh_dosage_unpack(span.ds, span.ss, 0, static_cast<uint32_t>(plan.size()), chunk_vidx.data(),
			                       lstate.dosage_doubles.data(), errbuf);
		}
		if (rc != PGH_OK) {
			throw IOException("%s: PgrGetD failed for variants [%u, %u): %s", fn, span_begin, span_end, string(errbuf));
		}
	}

	lstate.unpack_ms += ms_since(t_unpack0);
	const auto t_fill0 = std::chrono::steady_clock::now();
	// per-call writers shared by the ARRAY / LIST / STRUCT / COLUMNS layouts: `slot` is the
	// element index inside `dst` (child offset, or the output row for scalar layouts)
	auto put_dosage = [&](Vector &dst, idx_t slot, uint32_t s) {
		double d = dose[s];
		if (d == -9.0) {
			FlatVector::Validity(dst).SetInvalid(slot);
			FlatVector::GetData<double>(dst)[slot] = 0.0;
		} else {
			FlatVector::GetData<double>(dst)[slot] = d;
		}
	};
	// UnpackPhasedGenotypes (src/plink_common.cpp:1549-1584): dst = ARRAY(TINYINT, 2)
	auto put_phased = [&](Vector &dst, idx_t slot, uint32_t s, bool null_out) {
		auto *alleles = FlatVector::GetData<int8_t>(ArrayVector::GetEntry(dst));
		uint32_t code = (lstate.genovec[s >> 5] >> (2 * (s & 31))) & 3u;
		bool ignore = null_out && !bind_data.genotype_filter.AllowsCall(static_cast<double>(code));
		int8_t a0 = 0, a1 = 0;
		if (code == 3 || ignore) {
			FlatVector::Validity(dst).SetInvalid(slot);
		} else if (code == 2) {
			a0 = a1 = 1;
		} else if (code == 1) {
			bool alt_first = ((lstate.phasepresent[s >> 6] >> (s & 63)) & 1ull) &&
			                 ((lstate.phaseinfo[s >> 6] >> (s & 63)) & 1ull);
			a0 = alt_first ? 1 : 0;
			a1 = alt_first ? 0 : 1;
		}
		alleles[2 * slot] = a0;
		alleles[2 * slot + 1] = a1;
	};
	auto put_hardcall = [&](Vector &dst, idx_t slot, const int8_t *src, const uint64_t *val, uint32_t s,
	                        bool null_out) {
		const bool present = (val[s >> 6] >> (s & 63)) & 1ull;
		if (!present || (null_out && !bind_data.genotype_filter.AllowsCall(static_cast<double>(src[s])))) {
			FlatVector::Validity(dst).SetInvalid(slot);
			FlatVector::GetData<int8_t>(dst)[slot] = 0;
		} else {
			FlatVector::GetData<int8_t>(dst)[slot] = src[s];
		}
	};

	// 3. fill the projected columns (a LIST child is sized for the whole chunk at once)
	if (mode == GenotypeMode::LIST && gstate.need_genotypes) {
		for (idx_t out_col = 0; out_col < column_ids.size(); out_col++) {
			if (column_ids[out_col] == COL_GENOTYPES) {
				ListVector::Reserve(output.data[out_col], plan.size() * static_cast<idx_t>(n));
			}
		}
	}
	for (idx_t row = 0; row < plan.size(); row++) {
		const uint32_t v = plan[row].vidx;
		const bool null_out = bind_data.genotype_filter.active && !plan[row].geno_range_all_pass;
		const size_t src_row = listed ? row : v - span_begin;
		const int8_t *src = plain_hardcalls ? chunk_bytes + src_row * n : nullptr;
		const uint64_t *val = plain_hardcalls ? chunk_validity + src_row * val_words : nullptr;
		dose = dosage_rows ? lstate.dosage_doubles.data() + row * static_cast<size_t>(n) : nullptr;
		if (per_variant_decode) {
			// streamed: variant v's window is held while its calls are read, and the thread's reader is one made
			// on that window (a new one when the window has changed)
			RowLease rows_v = streamed ? rows_of(v, v + 1) : RowLease();
			if (streamed && (!lstate.reader || lstate.reader_window != rows_v.window_id)) {
				if (lstate.reader) {
					pgh_reader_destroy(lstate.reader);
					lstate.reader = nullptr;
				}
				char errbuf[PGH_ERRBUF_LEN] = {0};
				if (pgh_reader_create(rows_v.ds, rows_v.ss, &lstate.reader, errbuf) != PGH_OK) {
					throw IOException("%s: thread init failed: %s", fn, string(errbuf));
				}
				lstate.reader_window = rows_v.window_id;
			}
			if (pgh_get_phased(lstate.reader, v, lstate.genovec.data(), lstate.phasepresent.data(),
			                          lstate.phaseinfo.data()) != PGH_OK) {
				throw IOException("%s: PgrGetP failed for variant %u: %s", fn, v,
				                  string(pgh_reader_error(lstate.reader)));
			}
		}
		for (idx_t out_col = 0; out_col < column_ids.size(); out_col++) {
			auto file_col = column_ids[out_col];
			if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
				continue;
			}
			auto &vec = output.data[out_col];
			if (FillVariantMetadataColumn(bind_data.c.variants, file_col, v, vec, row)) {
				continue;
			}
			if (mode == GenotypeMode::COLUMNS) {
				const uint32_t s = static_cast<uint32_t>(file_col - COL_GENOTYPES);
				if (s >= n) {
					continue;
				}
				if (bind_data.include_dosages) {
					put_dosage(vec, row, s);
				} else {
					put_hardcall(vec, row, src, val, s, null_out);
				}
				continue;
			}
			if (file_col != COL_GENOTYPES) {
				continue;
			}
			if (mode == GenotypeMode::STRUCT) {
				auto &entries = StructVector::GetEntries(vec);
				for (uint32_t s = 0; s < n; s++) {
					if (bind_data.include_dosages) {
						put_dosage(*entries[s], row, s);
					} else if (phased_out) {
						put_phased(*entries[s], row, s, null_out);
					} else {
						put_hardcall(*entries[s], row, src, val, s, null_out);
					}
				}
				continue;
			}
			if (IsAggregateGenotypeMode(mode)) {
				const uint32_t *gc = lstate.scan.Counts(v);
				auto &entries = StructVector::GetEntries(vec);
				for (int k = 0; k < 4; k++) {
					FlatVector::GetData<uint32_t>(*entries[k])[row] = gc[k];
				}
				if (mode == GenotypeMode::STATS) {
					const double nan = std::numeric_limits<double>::quiet_NaN();
					uint32_t nn = gc[0] + gc[1] + gc[2];
					uint32_t total = nn + gc[3];
					FlatVector::GetData<uint32_t>(*entries[4])[row] = nn;
					double af = nn ? (static_cast<double>(gc[1]) + 2.0 * gc[2]) / (2.0 * nn) : nan;
					FlatVector::GetData<double>(*entries[5])[row] = af;
					FlatVector::GetData<double>(*entries[6])[row] = nn ? std::min(af, 1.0 - af) : nan;
					FlatVector::GetData<double>(*entries[7])[row] =
					    total ? static_cast<double>(gc[3]) / static_cast<double>(total) : nan;
					FlatVector::GetData<uint32_t>(*entries[8])[row] = gc[1] + gc[2];
					FlatVector::GetData<double>(*entries[9])[row] =
					    nn ? static_cast<double>(gc[1]) / static_cast<double>(nn) : nan;
				}
				continue;
			}
			const idx_t base = BeginGenotypeRow(bind_data, vec, row, n);
			Vector &child = GenotypeChild(bind_data, vec);
			if (bind_data.include_dosages) {
				for (uint32_t s = 0; s < n; s++) {
					put_dosage(child, base + s, s);
				}
			} else if (phased_out) {
				for (uint32_t s = 0; s < n; s++) {
					put_phased(child, base + s, s, null_out);
				}
			} else {
				auto &child_validity = FlatVector::Validity(child);
				auto *dst = FlatVector::GetData<int8_t>(child);
				if (dst + base != src) {
					std::memcpy(dst + base, src, n); // missing calls are already stored as 0
				}
				// the row's validity bits, as the device wrote them, go into the child's mask whole words at a
				// time (the reference sets them sample by sample, src/pgen_reader.cpp:1009-1047)
				child_validity.EnsureCapacity(base + n);
				CopyValidityBits(child_validity.GetData(), base, val, n);
				if (null_out) {
					// genotype_range / include_genotypes: calls outside the filter become NULL too
					for (uint32_t s2 = 0; s2 < n; s2++) {
						if (!bind_data.genotype_filter.AllowsCall(static_cast<double>(src[s2]))) {
							child_validity.SetInvalid(base + s2);
							dst[base + s2] = 0;
						}
					}
				}
			}
		}
	}
	lstate.fill_ms += ms_since(t_fill0);
	lstate.chunks++;
	CompatSetOutputCardinality(output, plan.size());
	if (pipelined) {
		lstate.cur ^= 1; // (the slot just consumed is planned and launched again by the next call)
	}
}

void RegisterPgenReader(ExtensionLoader &loader) {
	TableFunction read_pgen("read_pgen", {LogicalType::VARCHAR}, PgenScan, PgenBind, PgenInitGlobal, PgenInitLocal);
	read_pgen.projection_pushdown = true;
	read_pgen.named_parameters["pvar"] = LogicalType::VARCHAR;
	read_pgen.named_parameters["psam"] = LogicalType::VARCHAR;
	read_pgen.named_parameters["dosages"] = LogicalType::BOOLEAN;
	read_pgen.named_parameters["phased"] = LogicalType::BOOLEAN;
	read_pgen.named_parameters["samples"] = LogicalType::ANY;
	read_pgen.named_parameters["genotypes"] = LogicalType::VARCHAR;
	read_pgen.named_parameters["orient"] = LogicalType::VARCHAR;
	read_pgen.named_parameters["af_range"] = LogicalType::ANY;
	read_pgen.named_parameters["ac_range"] = LogicalType::ANY;
	read_pgen.named_parameters["variants"] = LogicalType::ANY;
	read_pgen.named_parameters["genotype_range"] = LogicalType::ANY;
	read_pgen.named_parameters["include_genotypes"] = LogicalType::LIST(LogicalType::VARCHAR);
	loader.RegisterFunction(read_pgen);
}

} // namespace duckdb

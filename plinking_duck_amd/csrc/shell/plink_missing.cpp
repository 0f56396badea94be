// plink_missing.cpp -- plink_missing(path, pvar, psam, samples, region, mode)
//
// Surface of the reference's src/plink_missing.cpp.  Variant mode: MISSING_CT is
// column 3 of the batched device tally.  Sample mode: the reference's phase 1
// (every scan thread accumulating per-sample counters and merging under a mutex,
// src/plink_missing.cpp:585-619) is ONE per-sample column-sum launch over the
// variant range, run by whichever thread enters Scan first; phase 2 (row
// emission by next_sample_idx.fetch_add) is unchanged.
#include "variant_scan.hpp"

#include <mutex>

namespace duckdb {

// variant mode: CHROM POS ID REF ALT MISSING_CT(5) OBS_CT(6) F_MISS(7)
static constexpr idx_t VCOL_MISSING_CT = 5;
static constexpr idx_t VCOL_OBS_CT = 6;
static constexpr idx_t VCOL_F_MISS = 7;
// sample mode: FID(0) IID(1) MISSING_CT(2) OBS_CT(3) F_MISS(4)
static constexpr idx_t SCOL_FID = 0;
static constexpr idx_t SCOL_IID = 1;
static constexpr idx_t SCOL_MISSING_CT = 2;
static constexpr idx_t SCOL_OBS_CT = 3;
static constexpr idx_t SCOL_F_MISS = 4;

struct PlinkMissingBindData : public TableFunctionData {
	PgenBindCommon c;
	bool sample_mode = false;
};

struct PlinkMissingGlobalState : public GlobalTableFunctionState {
	VariantScanGlobal scan;
	vector<column_t> column_ids;
	bool need_missingness = false;
	uint32_t max_threads_config = 0;
	uint32_t db_thread_count = 1;
	// sample mode
	std::mutex phase1_mutex;
	bool variant_scan_done = false;
	vector<uint32_t> sample_missing_counts;
	std::atomic<uint32_t> next_sample_idx {0};
	uint32_t total_variant_ct = 0;

	idx_t MaxThreads() const override {
		uint32_t range = scan.end_variant_idx - scan.start_variant_idx;
		idx_t computed = std::min<idx_t>(range / 500 + 1, db_thread_count);
		return ApplyMaxThreadsCap(computed, max_threads_config);
	}
};

struct PlinkMissingLocalState : public LocalTableFunctionState {
	VariantScanLocal scan;
};

static unique_ptr<FunctionData> PlinkMissingBind(ClientContext &context, TableFunctionBindInput &input,
                                                 vector<LogicalType> &return_types, vector<string> &names) {
	auto bind_data = make_uniq<PlinkMissingBindData>();
	auto mode_it = input.named_parameters.find("mode");
	if (mode_it != input.named_parameters.end()) {
		auto mode_str = mode_it->second.GetValue<string>();
		if (mode_str == "variant") {
			bind_data->sample_mode = false;
		} else if (mode_str == "sample") {
			bind_data->sample_mode = true;
		} else {
			throw InvalidInputException("plink_missing: mode must be 'variant' or 'sample', got '%s'", mode_str);
		}
	}
	bind_data->c.Bind(context, input, "plink_missing", false);
	if (bind_data->sample_mode && bind_data->c.psam_path.empty()) {
		throw InvalidInputException("plink_missing: sample mode requires a .psam or .fam file "
		                            "(use psam := 'path' to specify explicitly)");
	}
	if (bind_data->sample_mode) {
		names = {"FID", "IID", "MISSING_CT", "OBS_CT", "F_MISS"};
		return_types = {LogicalType::VARCHAR, LogicalType::VARCHAR, LogicalType::INTEGER, LogicalType::INTEGER,
		                LogicalType::DOUBLE};
	} else {
		names = {"CHROM", "POS", "ID", "REF", "ALT", "MISSING_CT", "OBS_CT", "F_MISS"};
		return_types = {LogicalType::VARCHAR, LogicalType::INTEGER, LogicalType::VARCHAR, LogicalType::VARCHAR,
		                LogicalType::VARCHAR, LogicalType::INTEGER, LogicalType::INTEGER, LogicalType::DOUBLE};
	}
	return std::move(bind_data);
}

static unique_ptr<GlobalTableFunctionState> PlinkMissingInitGlobal(ClientContext &context,
                                                                   TableFunctionInitInput &input) {
	auto &bind_data = input.bind_data->Cast<PlinkMissingBindData>();
	auto state = make_uniq<PlinkMissingGlobalState>();
	state->scan.start_variant_idx = bind_data.c.RangeStart();
	state->scan.end_variant_idx = bind_data.c.RangeEnd();
	state->scan.next_variant_idx.store(state->scan.start_variant_idx);
	state->scan.effective_sample_ct = bind_data.c.effective_sample_ct;
	state->total_variant_ct = state->scan.end_variant_idx - state->scan.start_variant_idx;
	state->column_ids = input.column_ids;
	state->max_threads_config = GetPlinkingMaxThreads(context);
	state->db_thread_count = static_cast<uint32_t>(context.db_threads);
	const idx_t first = bind_data.sample_mode ? SCOL_MISSING_CT : VCOL_MISSING_CT;
	const idx_t last = bind_data.sample_mode ? SCOL_F_MISS : VCOL_F_MISS;
	for (auto col_id : input.column_ids) {
		if (col_id != COLUMN_IDENTIFIER_ROW_ID && col_id >= first && col_id <= last) {
			state->need_missingness = true;
			break;
		}
	}
	if (state->need_missingness) {
		state->scan.dataset = DeviceDataset::Acquire(bind_data.c.pgen_path, "plink_missing");
		// MISSING_CT per variant is column 3 of the range's tally pass, MISSING_CT per sample its per-sample
		// product: whichever of plink_freq / plink_hardy / plink_missing (either mode) saw this file, subset and
		// range first has already paid for the walk (src/plink_missing.cpp:479 and :593-609 are two more scans)
		state->scan.StartTallies(bind_data.c.sample_subset.get(), nullptr, bind_data.c.raw_sample_ct, nullptr,
		                         (bind_data.sample_mode || !bind_data.c.has_sample_subset) ? static_cast<uint32_t>(PGH_TALLY_SAMPLE_MISSING) : 0u,
		                         bind_data.sample_mode, GetPlinkingTallyCache(context), "plink_missing");
	}
	if (bind_data.sample_mode) {
		state->sample_missing_counts.assign(bind_data.c.effective_sample_ct, 0);
	}
	return std::move(state);
}

static unique_ptr<LocalTableFunctionState> PlinkMissingInitLocal(ExecutionContext &, TableFunctionInitInput &,
                                                                 GlobalTableFunctionState *) {
	return make_uniq<PlinkMissingLocalState>();
}

// Variant mode, columnar: the chunk's variants first, then one loop per projected column.
// MISSING_CT is column 3 of the batched tally; OBS_CT = samples - MISSING_CT; F_MISS over the
// effective sample count (src/plink_missing.cpp:486-490).
static void PlinkMissingScanVariant(const PlinkMissingBindData &bind_data, PlinkMissingGlobalState &gstate,
                                    PlinkMissingLocalState &lstate, DataChunk &output) {
	const uint32_t sample_ct = bind_data.c.effective_sample_ct;
	uint32_t vids[STANDARD_VECTOR_SIZE], missing[STANDARD_VECTOR_SIZE];
	idx_t n_rows = 0;
	uint32_t vidx;
	while (n_rows < STANDARD_VECTOR_SIZE && lstate.scan.Next(gstate.scan, "plink_missing", vidx)) {
		vids[n_rows] = vidx;
		missing[n_rows] = gstate.need_missingness ? lstate.scan.Counts(vidx)[3] : 0;
		n_rows++;
	}
	for (idx_t out_col = 0; out_col < gstate.column_ids.size(); out_col++) {
		const auto file_col = gstate.column_ids[out_col];
		if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
			continue;
		}
		auto &vec = output.data[out_col];
		if (file_col < VCOL_MISSING_CT) {
			for (idx_t r = 0; r < n_rows; r++) {
				FillVariantMetadataColumn(bind_data.c.variants, file_col, vids[r], vec, r);
			}
		} else if (file_col == VCOL_F_MISS) {
			auto *dst = FlatVector::GetData<double>(vec);
			for (idx_t r = 0; r < n_rows; r++) {
				dst[r] = sample_ct > 0 ? static_cast<double>(missing[r]) / static_cast<double>(sample_ct) : 0.0;
			}
		} else {
			auto *dst = FlatVector::GetData<int32_t>(vec);
			for (idx_t r = 0; r < n_rows; r++) {
				dst[r] = static_cast<int32_t>(file_col == VCOL_MISSING_CT ? missing[r] : sample_ct - missing[r]);
			}
		}
	}
	CompatSetOutputCardinality(output, n_rows);
}

static void PlinkMissingScanSample(const PlinkMissingBindData &bind_data, PlinkMissingGlobalState &gstate,
                                   DataChunk &output) {
	const uint32_t sample_ct = bind_data.c.effective_sample_ct;
	{
		// Phase 1: the range's tally pass (enqueued at init_global, or found in the dataset's cache) holds the
		// per-sample tallies; the first thread in waits for them, later threads go straight to phase 2.
		std::lock_guard<std::mutex> lock(gstate.phase1_mutex);
		if (!gstate.variant_scan_done) {
			if (gstate.need_missingness && gstate.total_variant_ct > 0 && sample_ct > 0) {
				gstate.scan.tally->SampleMissing(gstate.sample_missing_counts.data(), "plink_missing");
			}
			gstate.variant_scan_done = true;
		}
	}
	// Phase 2: a run of output samples per call (ascending file order within the subset); the
	// denominator is the number of variants in the region (src/plink_missing.cpp:643-646)
	const uint32_t first = gstate.next_sample_idx.fetch_add(STANDARD_VECTOR_SIZE);
	const idx_t n_rows = first < sample_ct ? std::min<idx_t>(STANDARD_VECTOR_SIZE, sample_ct - first) : 0;
	const uint32_t variant_ct = gstate.total_variant_ct;
	auto file_index = [&](idx_t r) {
		const uint32_t sidx = first + static_cast<uint32_t>(r);
		return bind_data.c.has_sample_subset ? bind_data.c.sample_subset->sorted_indices[sidx] : sidx;
	};
	auto missing_of = [&](idx_t r) { return gstate.need_missingness ? gstate.sample_missing_counts[first + r] : 0u; };
	for (idx_t out_col = 0; out_col < gstate.column_ids.size(); out_col++) {
		const auto file_col = gstate.column_ids[out_col];
		if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
			continue;
		}
		auto &vec = output.data[out_col];
		if (file_col == SCOL_FID || file_col == SCOL_IID) {
			for (idx_t r = 0; r < n_rows; r++) {
				FillSampleIdColumn(bind_data.c.sample_info(), file_col == SCOL_FID, file_index(r), vec, r);
			}
		} else if (file_col == SCOL_F_MISS) {
			auto *dst = FlatVector::GetData<double>(vec);
			for (idx_t r = 0; r < n_rows; r++) {
				dst[r] = variant_ct > 0 ? static_cast<double>(missing_of(r)) / static_cast<double>(variant_ct) : 0.0;
			}
		} else {
			auto *dst = FlatVector::GetData<int32_t>(vec);
			for (idx_t r = 0; r < n_rows; r++) {
				dst[r] = static_cast<int32_t>(file_col == SCOL_MISSING_CT ? missing_of(r) : variant_ct - missing_of(r));
			}
		}
	}
	CompatSetOutputCardinality(output, n_rows);
}

static void PlinkMissingScan(ClientContext &, TableFunctionInput &data_p, DataChunk &output) {
	auto &bind_data = data_p.bind_data->Cast<PlinkMissingBindData>();
	auto &gstate = data_p.global_state->Cast<PlinkMissingGlobalState>();
	auto &lstate = data_p.local_state->Cast<PlinkMissingLocalState>();
	if (bind_data.sample_mode) {
		PlinkMissingScanSample(bind_data, gstate, output);
	} else {
		PlinkMissingScanVariant(bind_data, gstate, lstate, output);
	}
}

void RegisterPlinkMissing(ExtensionLoader &loader) {
	TableFunction plink_missing("plink_missing", {LogicalType::VARCHAR}, PlinkMissingScan, PlinkMissingBind,
	                            PlinkMissingInitGlobal, PlinkMissingInitLocal);
	plink_missing.projection_pushdown = true;
	plink_missing.named_parameters["pvar"] = LogicalType::VARCHAR;
	plink_missing.named_parameters["psam"] = LogicalType::VARCHAR;
	plink_missing.named_parameters["samples"] = LogicalType::ANY;
	plink_missing.named_parameters["region"] = LogicalType::VARCHAR;
	plink_missing.named_parameters["mode"] = LogicalType::VARCHAR;
	loader.RegisterFunction(plink_missing);
}

} // namespace duckdb

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
		if (bind_data.c.has_sample_subset) {
			state->scan.subset = make_uniq<DeviceSubset>(*state->scan.dataset,
			                                             bind_data.c.sample_subset->sample_include, "plink_missing");
		}
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

static void PlinkMissingScanVariant(const PlinkMissingBindData &bind_data, PlinkMissingGlobalState &gstate,
                                    PlinkMissingLocalState &lstate, DataChunk &output) {
	auto &column_ids = gstate.column_ids;
	uint32_t sample_ct = bind_data.c.effective_sample_ct;
	auto no_strata = [](uint32_t, uint32_t) { return false; };
	idx_t rows_emitted = 0;
	uint32_t vidx;
	while (rows_emitted < STANDARD_VECTOR_SIZE && lstate.scan.Next(gstate.scan, "plink_missing", no_strata, vidx)) {
		uint32_t missing_ct = gstate.need_missingness ? lstate.scan.Counts(vidx)[3] : 0;
		uint32_t obs_ct = sample_ct - missing_ct;
		double f_miss = sample_ct > 0 ? static_cast<double>(missing_ct) / static_cast<double>(sample_ct) : 0.0;
		for (idx_t out_col = 0; out_col < column_ids.size(); out_col++) {
			auto file_col = column_ids[out_col];
			if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
				continue;
			}
			auto &vec = output.data[out_col];
			if (FillVariantMetadataColumn(bind_data.c.variants, file_col, vidx, vec, rows_emitted)) {
				continue;
			}
			switch (file_col) {
			case VCOL_MISSING_CT:
				FlatVector::GetData<int32_t>(vec)[rows_emitted] = static_cast<int32_t>(missing_ct);
				break;
			case VCOL_OBS_CT:
				FlatVector::GetData<int32_t>(vec)[rows_emitted] = static_cast<int32_t>(obs_ct);
				break;
			case VCOL_F_MISS:
				FlatVector::GetData<double>(vec)[rows_emitted] = f_miss;
				break;
			default:
				break;
			}
		}
		rows_emitted++;
	}
	CompatSetOutputCardinality(output, rows_emitted);
}

static void PlinkMissingScanSample(const PlinkMissingBindData &bind_data, PlinkMissingGlobalState &gstate,
                                   DataChunk &output) {
	uint32_t sample_ct = bind_data.c.effective_sample_ct;
	{
		// Phase 1: one device launch covers the whole variant range, so the first
		// thread in does it; later threads find it done and go straight to phase 2.
		std::lock_guard<std::mutex> lock(gstate.phase1_mutex);
		if (!gstate.variant_scan_done) {
			if (gstate.need_missingness && gstate.total_variant_ct > 0) {
				char errbuf[PGH_ERRBUF_LEN] = {0};
				int rc = pgh_missing_per_sample(gstate.scan.dataset->handle,
				                                gstate.scan.subset ? gstate.scan.subset->handle : nullptr,
				                                gstate.scan.start_variant_idx, gstate.scan.end_variant_idx,
				                                gstate.sample_missing_counts.data(), errbuf);
				if (rc != PGH_OK) {
					throw IOException("plink_missing: PgrGetMissingness failed: %s", string(errbuf));
				}
			}
			gstate.variant_scan_done = true;
		}
	}
	// Phase 2: emit sample rows (ascending file order within the subset)
	auto &column_ids = gstate.column_ids;
	uint32_t total_variant_ct = gstate.total_variant_ct;
	idx_t rows_emitted = 0;
	while (rows_emitted < STANDARD_VECTOR_SIZE) {
		uint32_t sidx = gstate.next_sample_idx.fetch_add(1);
		if (sidx >= sample_ct) {
			break;
		}
		uint32_t missing_ct = gstate.need_missingness ? gstate.sample_missing_counts[sidx] : 0;
		uint32_t obs_ct = total_variant_ct - missing_ct;
		double f_miss =
		    total_variant_ct > 0 ? static_cast<double>(missing_ct) / static_cast<double>(total_variant_ct) : 0.0;
		uint32_t orig_idx = bind_data.c.has_sample_subset ? bind_data.c.sample_subset->sorted_indices[sidx] : sidx;
		for (idx_t out_col = 0; out_col < column_ids.size(); out_col++) {
			auto file_col = column_ids[out_col];
			if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
				continue;
			}
			auto &vec = output.data[out_col];
			switch (file_col) {
			case SCOL_FID: {
				auto &fids = bind_data.c.sample_info.fids;
				if (!fids.empty() && orig_idx < fids.size() && !fids[orig_idx].empty()) {
					FlatVector::GetData<string_t>(vec)[rows_emitted] = StringVector::AddString(vec, fids[orig_idx]);
				} else {
					FlatVector::SetNull(vec, rows_emitted, true);
				}
				break;
			}
			case SCOL_IID:
				FlatVector::GetData<string_t>(vec)[rows_emitted] =
				    StringVector::AddString(vec, bind_data.c.sample_info.iids[orig_idx]);
				break;
			case SCOL_MISSING_CT:
				FlatVector::GetData<int32_t>(vec)[rows_emitted] = static_cast<int32_t>(missing_ct);
				break;
			case SCOL_OBS_CT:
				FlatVector::GetData<int32_t>(vec)[rows_emitted] = static_cast<int32_t>(obs_ct);
				break;
			case SCOL_F_MISS:
				FlatVector::GetData<double>(vec)[rows_emitted] = f_miss;
				break;
			default:
				break;
			}
		}
		rows_emitted++;
	}
	CompatSetOutputCardinality(output, rows_emitted);
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

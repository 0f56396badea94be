// plink_score.cpp -- plink_score(path, weights, pvar, psam, samples, region, center, no_mean_imputation)
//
// Surface of the reference's src/plink_score.cpp.  Weight resolution (positional
// list / ID-keyed structs, zero weights dropped, sorted by variant index) is bind
// code kept as is; phase 1 (per-thread accumulators + mutex merge,
// src/plink_score.cpp:575-664) is one call into libpgenhip's pgh_score, whose
// kernels tally the scored variants, derive the per-genotype contribution tables
// and accumulate score / dosage sum / allele count per sample.
#include "variant_scan.hpp"

#include <algorithm>
#include <cmath>
#include <mutex>

namespace duckdb {

static constexpr idx_t COL_FID = 0;
static constexpr idx_t COL_IID = 1;
static constexpr idx_t COL_ALLELE_CT = 2;
static constexpr idx_t COL_DENOM = 3;
static constexpr idx_t COL_NAMED_ALLELE_DOSAGE_SUM = 4;
static constexpr idx_t COL_SCORE_SUM = 5;
static constexpr idx_t COL_SCORE_AVG = 6;

struct ScoredVariant {
	uint32_t variant_idx;
	double weight;
	bool flip; // scored allele is REF
};

struct PlinkScoreBindData : public TableFunctionData {
	PgenBindCommon c;
	vector<ScoredVariant> scored_variants;
	bool center = false;
	bool no_mean_imputation = false;
	vector<uint32_t> sample_output_order; // output row -> original sample index
};

struct PlinkScoreGlobalState : public GlobalTableFunctionState {
	vector<double> score_sums, named_allele_dosage_sums;
	vector<uint32_t> allele_cts;
	std::mutex phase1_mutex;
	bool scoring_done = false;
	bool need_dosage_sum = false; // NAMED_ALLELE_DOSAGE_SUM projected: only then is it accumulated
	std::atomic<uint32_t> next_sample_idx {0};
	uint32_t total_samples = 0;
	uint32_t scored_variant_count = 0;
	vector<column_t> column_ids;
	uint32_t db_thread_count = 1;
	uint32_t max_threads_config = 0;
	shared_ptr<DeviceDataset> dataset;
	unique_ptr<DeviceSubset> subset;

	idx_t MaxThreads() const override {
		if (scored_variant_count < 100) {
			return 1;
		}
		idx_t computed = std::min<idx_t>(scored_variant_count / 16 + 1, db_thread_count);
		return ApplyMaxThreadsCap(computed, max_threads_config);
	}
};

struct PlinkScoreLocalState : public LocalTableFunctionState {};

static unique_ptr<FunctionData> PlinkScoreBind(ClientContext &context, TableFunctionBindInput &input,
                                               vector<LogicalType> &return_types, vector<string> &names) {
	auto bind_data = make_uniq<PlinkScoreBindData>();
	for (auto &kv : input.named_parameters) {
		if (kv.first == "center") {
			bind_data->center = kv.second.GetValue<bool>();
		} else if (kv.first == "no_mean_imputation") {
			bind_data->no_mean_imputation = kv.second.GetValue<bool>();
		}
	}
	if (bind_data->center && bind_data->no_mean_imputation) {
		throw InvalidInputException("plink_score: center and no_mean_imputation cannot both be true");
	}
	auto &c = bind_data->c;
	c.Bind(context, input, "plink_score", true);
	if (c.has_sample_subset) {
		bind_data->sample_output_order = c.sample_subset->sorted_indices;
	} else {
		bind_data->sample_output_order.resize(c.raw_sample_ct);
		for (uint32_t i = 0; i < c.raw_sample_ct; i++) {
			bind_data->sample_output_order[i] = i;
		}
	}

	auto weights_it = input.named_parameters.find("weights");
	if (weights_it == input.named_parameters.end()) {
		throw InvalidInputException("plink_score: weights parameter is required");
	}
	auto &weights_val = weights_it->second;
	if (weights_val.IsNull()) {
		throw InvalidInputException("plink_score: weights must not be NULL");
	}
	auto &weights_type = weights_val.type();
	uint32_t range_start = c.RangeStart();
	uint32_t range_end = c.RangeEnd();
	uint32_t variant_count = range_end - range_start;

	if (weights_type.id() != LogicalTypeId::LIST) {
		throw InvalidInputException("plink_score: weights must be a list (LIST(DOUBLE) for positional mode, "
		                            "or LIST(STRUCT(id, allele, weight)) for ID-keyed mode)");
	}
	auto &child_type = ListType::GetChildType(weights_type);
	auto &children = ListValue::GetChildren(weights_val);
	if (children.empty()) {
		throw InvalidInputException("plink_score: weights list is empty");
	}
	if (child_type.id() == LogicalTypeId::STRUCT) {
		// ID-keyed mode: LIST(STRUCT(id VARCHAR, allele VARCHAR, weight DOUBLE))
		auto &struct_children = StructType::GetChildTypes(child_type);
		constexpr idx_t kNone = static_cast<idx_t>(-1);
		idx_t id_idx = kNone, allele_idx = kNone, weight_idx = kNone;
		for (idx_t i = 0; i < struct_children.size(); i++) {
			if (struct_children[i].first == "id") {
				id_idx = i;
			} else if (struct_children[i].first == "allele") {
				allele_idx = i;
			} else if (struct_children[i].first == "weight") {
				weight_idx = i;
			}
		}
		if (id_idx == kNone || allele_idx == kNone || weight_idx == kNone) {
			throw InvalidInputException("plink_score: ID-keyed weights must be "
			                            "LIST(STRUCT(id VARCHAR, allele VARCHAR, weight DOUBLE))");
		}
		// variant ID -> index, restricted to the region; duplicate IDs: last one wins
		std::unordered_map<string, uint32_t> variant_id_map;
		for (uint32_t v = range_start; v < range_end; v++) {
			if (!c.variants.ids[v].empty()) {
				variant_id_map[c.variants.ids[v]] = v;
			}
		}
		for (auto &entry : children) {
			auto &struct_vals = StructValue::GetChildren(entry);
			string id = struct_vals[id_idx].GetValue<string>();
			string allele = struct_vals[allele_idx].GetValue<string>();
			double weight = struct_vals[weight_idx].GetValue<double>();
			auto it = variant_id_map.find(id);
			if (it == variant_id_map.end()) {
				continue; // unmatched IDs are skipped silently
			}
			uint32_t vidx = it->second;
			bool flip;
			if (allele == c.variants.GetAlt(vidx)) {
				flip = false;
			} else if (allele == c.variants.GetRef(vidx)) {
				flip = true;
			} else {
				continue; // unmatched allele
			}
			if (weight != 0.0) {
				bind_data->scored_variants.push_back({vidx, weight, flip});
			}
		}
		std::sort(bind_data->scored_variants.begin(), bind_data->scored_variants.end(),
		          [](const ScoredVariant &a, const ScoredVariant &b) { return a.variant_idx < b.variant_idx; });
	} else {
		if (static_cast<uint32_t>(children.size()) != variant_count) {
			throw InvalidInputException("plink_score: weights list length (%llu) must match variant count (%u)",
			                            static_cast<unsigned long long>(children.size()), variant_count);
		}
		for (idx_t i = 0; i < children.size(); i++) {
			double w = children[i].GetValue<double>();
			if (w != 0.0) {
				bind_data->scored_variants.push_back({range_start + static_cast<uint32_t>(i), w, false});
			}
		}
	}

	names = {"FID", "IID", "ALLELE_CT", "DENOM", "NAMED_ALLELE_DOSAGE_SUM", "SCORE_SUM", "SCORE_AVG"};
	return_types = {LogicalType::VARCHAR, LogicalType::VARCHAR, LogicalType::INTEGER, LogicalType::INTEGER,
	                LogicalType::DOUBLE,  LogicalType::DOUBLE,  LogicalType::DOUBLE};
	return std::move(bind_data);
}

static unique_ptr<GlobalTableFunctionState> PlinkScoreInitGlobal(ClientContext &context,
                                                                 TableFunctionInitInput &input) {
	auto &bind_data = input.bind_data->Cast<PlinkScoreBindData>();
	auto state = make_uniq<PlinkScoreGlobalState>();
	uint32_t n = bind_data.c.effective_sample_ct;
	state->score_sums.assign(n, 0.0);
	state->named_allele_dosage_sums.assign(n, 0.0);
	state->allele_cts.assign(n, 0);
	state->total_samples = n;
	state->scored_variant_count = static_cast<uint32_t>(bind_data.scored_variants.size());
	state->column_ids = input.column_ids;
	state->db_thread_count = static_cast<uint32_t>(context.db_threads);
	state->max_threads_config = GetPlinkingMaxThreads(context);
	bool need_scores = false;
	for (auto col_id : input.column_ids) {
		if (col_id != COLUMN_IDENTIFIER_ROW_ID && col_id >= COL_ALLELE_CT) {
			need_scores = true;
		}
		state->need_dosage_sum |= col_id == COL_NAMED_ALLELE_DOSAGE_SUM;
	}
	if (need_scores && !bind_data.scored_variants.empty()) {
		state->dataset = DeviceDataset::Acquire(bind_data.c.pgen_path, "plink_score");
		if (bind_data.c.has_sample_subset) {
			state->subset =
			    make_uniq<DeviceSubset>(*state->dataset, bind_data.c.sample_subset->sample_include, "plink_score");
		}
	}
	return std::move(state);
}

static unique_ptr<LocalTableFunctionState> PlinkScoreInitLocal(ExecutionContext &, TableFunctionInitInput &,
                                                               GlobalTableFunctionState *) {
	return make_uniq<PlinkScoreLocalState>();
}

//! Files with explicit dosage tracks: the dosage track is decoded on the host
//! (pgh_get_dosage_f64 = PgrGetD + Dosage16ToDoublesMinus9), so this mirrors the
//! reference loop (src/plink_score.cpp:586-652) variant by variant.
static void ScoreFromDosages(const PlinkScoreBindData &bind_data, PlinkScoreGlobalState &gstate) {
	char errbuf[PGH_ERRBUF_LEN] = {0};
	pgh_reader *reader = nullptr;
	int rc = pgh_reader_create(gstate.dataset->handle, gstate.subset ? gstate.subset->handle : nullptr, &reader, errbuf);
	if (rc != PGH_OK) {
		throw IOException("plink_score: thread init failed: %s", string(errbuf));
	}
	uint32_t sample_ct = bind_data.c.effective_sample_ct;
	vector<double> dosage(sample_ct);
	for (auto &sv : bind_data.scored_variants) {
		if (pgh_get_dosage_f64(reader, sv.variant_idx, dosage.data()) != PGH_OK) {
			string msg = pgh_reader_error(reader);
			pgh_reader_destroy(reader);
			throw IOException("plink_score: PgrGetD failed for variant %u: %s", sv.variant_idx, msg);
		}
		double sum_alt = 0.0;
		uint32_t non_missing_ct = 0;
		for (uint32_t s = 0; s < sample_ct; s++) {
			if (dosage[s] != -9.0) {
				sum_alt += dosage[s];
				non_missing_ct++;
			}
		}
		if (non_missing_ct == 0) {
			continue;
		}
		double mean_alt = sum_alt / static_cast<double>(non_missing_ct);
		if (bind_data.center) {
			double freq = mean_alt / 2.0;
			double sd = std::sqrt(2.0 * freq * (1.0 - freq));
			if (sd == 0.0) {
				continue;
			}
			double mean_scored = sv.flip ? (2.0 - mean_alt) : mean_alt;
			for (uint32_t s = 0; s < sample_ct; s++) {
				if (dosage[s] == -9.0) {
					continue;
				}
				double scored = sv.flip ? (2.0 - dosage[s]) : dosage[s];
				gstate.score_sums[s] += sv.weight * ((scored - mean_scored) / sd);
				gstate.allele_cts[s] += 2;
			}
		} else {
			for (uint32_t s = 0; s < sample_ct; s++) {
				double alt = dosage[s];
				if (alt == -9.0) {
					if (bind_data.no_mean_imputation) {
						continue;
					}
					alt = mean_alt;
				}
				double scored = sv.flip ? (2.0 - alt) : alt;
				gstate.score_sums[s] += sv.weight * scored;
				gstate.named_allele_dosage_sums[s] += scored;
				gstate.allele_cts[s] += 2;
			}
		}
	}
	pgh_reader_destroy(reader);
}

static void PlinkScoreScan(ClientContext &, TableFunctionInput &data_p, DataChunk &output) {
	auto &bind_data = data_p.bind_data->Cast<PlinkScoreBindData>();
	auto &gstate = data_p.global_state->Cast<PlinkScoreGlobalState>();
	{
		// Phase 1: the device call covers every scored variant, so the first thread
		// to arrive runs it while the others wait on the mutex, then all emit rows.
		std::lock_guard<std::mutex> lock(gstate.phase1_mutex);
		if (!gstate.scoring_done) {
			if (gstate.dataset && !bind_data.scored_variants.empty()) {
				if (bind_data.c.file_has_dosage) {
					ScoreFromDosages(bind_data, gstate);
				} else {
					size_t n_scored = bind_data.scored_variants.size();
					vector<uint32_t> vidx(n_scored);
					vector<double> weights(n_scored);
					vector<uint8_t> flip(n_scored);
					for (size_t i = 0; i < n_scored; i++) {
						vidx[i] = bind_data.scored_variants[i].variant_idx;
						weights[i] = bind_data.scored_variants[i].weight;
						flip[i] = bind_data.scored_variants[i].flip;
					}
					int mode = bind_data.center ? PGH_SCORE_CENTER
					                            : (bind_data.no_mean_imputation ? PGH_SCORE_NO_MEAN_IMPUTATION
					                                                            : PGH_SCORE_MEAN_IMPUTE);
					char errbuf[PGH_ERRBUF_LEN] = {0};
					int rc = pgh_score(gstate.dataset->handle, gstate.subset ? gstate.subset->handle : nullptr,
					                   static_cast<uint32_t>(n_scored), vidx.data(), weights.data(), flip.data(), 1, mode,
					                   gstate.score_sums.data(),
					                   gstate.need_dosage_sum ? gstate.named_allele_dosage_sums.data() : nullptr,
					                   gstate.allele_cts.data(), errbuf);
					if (rc != PGH_OK) {
						throw IOException("plink_score: scoring failed: %s", string(errbuf));
					}
				}
			}
			gstate.scoring_done = true;
		}
	}

	// Phase 2: one row per sample
	auto &column_ids = gstate.column_ids;
	bool has_fid = !bind_data.c.sample_info.fids.empty();
	idx_t rows_emitted = 0;
	while (rows_emitted < STANDARD_VECTOR_SIZE) {
		uint32_t sidx = gstate.next_sample_idx.fetch_add(1);
		if (sidx >= gstate.total_samples) {
			break;
		}
		uint32_t orig_idx = bind_data.sample_output_order[sidx];
		uint32_t allele_ct = gstate.allele_cts[sidx];
		double score_sum = gstate.score_sums[sidx];
		double dosage_sum = gstate.named_allele_dosage_sums[sidx];
		double score_avg = allele_ct > 0 ? score_sum / static_cast<double>(allele_ct) : 0.0;
		for (idx_t out_col = 0; out_col < column_ids.size(); out_col++) {
			auto file_col = column_ids[out_col];
			if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
				continue;
			}
			auto &vec = output.data[out_col];
			switch (file_col) {
			case COL_FID:
				if (has_fid) {
					FlatVector::GetData<string_t>(vec)[rows_emitted] =
					    StringVector::AddString(vec, bind_data.c.sample_info.fids[orig_idx]);
				} else {
					FlatVector::SetNull(vec, rows_emitted, true);
				}
				break;
			case COL_IID:
				FlatVector::GetData<string_t>(vec)[rows_emitted] =
				    StringVector::AddString(vec, bind_data.c.sample_info.iids[orig_idx]);
				break;
			case COL_ALLELE_CT:
			case COL_DENOM:
				FlatVector::GetData<int32_t>(vec)[rows_emitted] = static_cast<int32_t>(allele_ct);
				break;
			case COL_NAMED_ALLELE_DOSAGE_SUM:
				FlatVector::GetData<double>(vec)[rows_emitted] = dosage_sum;
				break;
			case COL_SCORE_SUM:
				FlatVector::GetData<double>(vec)[rows_emitted] = score_sum;
				break;
			case COL_SCORE_AVG:
				FlatVector::GetData<double>(vec)[rows_emitted] = score_avg;
				break;
			default:
				break;
			}
		}
		rows_emitted++;
	}
	CompatSetOutputCardinality(output, rows_emitted);
}

void RegisterPlinkScore(ExtensionLoader &loader) {
	TableFunction plink_score("plink_score", {LogicalType::VARCHAR}, PlinkScoreScan, PlinkScoreBind,
	                          PlinkScoreInitGlobal, PlinkScoreInitLocal);
	plink_score.projection_pushdown = true;
	plink_score.named_parameters["pvar"] = LogicalType::VARCHAR;
	plink_score.named_parameters["psam"] = LogicalType::VARCHAR;
	plink_score.named_parameters["weights"] = LogicalType::ANY;
	plink_score.named_parameters["samples"] = LogicalType::ANY;
	plink_score.named_parameters["region"] = LogicalType::VARCHAR;
	plink_score.named_parameters["center"] = LogicalType::BOOLEAN;
	plink_score.named_parameters["no_mean_imputation"] = LogicalType::BOOLEAN;
	loader.RegisterFunction(plink_score);
}

} // namespace duckdb

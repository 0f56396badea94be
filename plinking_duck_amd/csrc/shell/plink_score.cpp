// plink_score.cpp -- plink_score(path, weights, pvar, psam, samples, region, center, no_mean_imputation)
//
// Surface of the reference's src/plink_score.cpp.  Weight resolution (positional
// list / ID-keyed structs, zero weights dropped, sorted by variant index) is bind
// code kept as is; phase 1 (per-thread accumulators + mutex merge,
// src/plink_score.cpp:575-664) is one call into libpgenhip's pgh_score, whose
// kernels tally the scored variants, derive the per-genotype contribution tables
// and accumulate score / dosage sum / allele count per sample.
#include "variant_scan.hpp"

#include <cstring>

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
	vector<uint32_t> scored_vidx; // the same list as the arrays pgh_score takes
	vector<double> scored_weights;
	vector<uint8_t> scored_flip;
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
	bool use_tally_cache = true;
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

//! `weights :=` -> the scored variants, ascending by variant index, zero weights dropped
//! (src/plink_score.cpp:329-427).  Two shapes:
//!   LIST(DOUBLE)                         one weight per variant of the region, positionally
//!   LIST(STRUCT(id, allele, weight))     keyed by variant ID; `allele` picks ALT (as is) or REF
//!                                        (flipped: 2 - dosage); unknown IDs / alleles are skipped
static vector<ScoredVariant> ResolveWeights(const Value &weights, const VariantMetadataIndex &variants,
                                            uint32_t range_start, uint32_t range_end) {
	if (weights.IsNull()) {
		throw InvalidInputException("plink_score: weights must not be NULL");
	}
	if (weights.type().id() != LogicalTypeId::LIST) {
		throw InvalidInputException("plink_score: weights must be a list (LIST(DOUBLE) for positional mode, "
		                            "or LIST(STRUCT(id, allele, weight)) for ID-keyed mode)");
	}
	const auto &items = ListValue::GetChildren(weights);
	if (items.empty()) {
		throw InvalidInputException("plink_score: weights list is empty");
	}
	vector<ScoredVariant> scored;
	const auto &item_type = ListType::GetChildType(weights.type());
	if (item_type.id() != LogicalTypeId::STRUCT) {
		const uint32_t expected = range_end - range_start;
		if (items.size() != expected) {
			throw InvalidInputException("plink_score: weights list length (%llu) must match variant count (%u)",
			                            static_cast<unsigned long long>(items.size()), expected);
		}
		for (uint32_t i = 0; i < expected; i++) {
			const double w = items[i].GetValue<double>();
			if (w != 0.0) {
				scored.push_back({range_start + i, w, false});
			}
		}
		return scored;
	}
	// positions of the three fields inside each struct value
	int field_of[3] = {-1, -1, -1}; // id, allele, weight
	const auto &fields = StructType::GetChildTypes(item_type);
	for (idx_t f = 0; f < fields.size(); f++) {
		static const char *const kNames[3] = {"id", "allele", "weight"};
		for (int k = 0; k < 3; k++) {
			if (fields[f].first == kNames[k]) {
				field_of[k] = static_cast<int>(f);
			}
		}
	}
	if (field_of[0] < 0 || field_of[1] < 0 || field_of[2] < 0) {
		throw InvalidInputException("plink_score: ID-keyed weights must be "
		                            "LIST(STRUCT(id VARCHAR, allele VARCHAR, weight DOUBLE))");
	}
	// IDs of the region; of two variants with one ID the later one is the one that gets scored
	std::unordered_map<string, uint32_t> by_id;
	for (uint32_t v = range_start; v < range_end; v++) {
		if (!variants.ids()[v].empty()) {
			by_id[variants.ids()[v]] = v;
		}
	}
	for (const auto &item : items) {
		const auto &vals = StructValue::GetChildren(item);
		auto hit = by_id.find(vals[field_of[0]].GetValue<string>());
		if (hit == by_id.end()) {
			continue;
		}
		const uint32_t v = hit->second;
		const string allele = vals[field_of[1]].GetValue<string>();
		const bool is_alt = allele == variants.GetAlt(v);
		if (!is_alt && allele != variants.GetRef(v)) {
			continue;
		}
		const double w = vals[field_of[2]].GetValue<double>();
		if (w != 0.0) {
			scored.push_back({v, w, !is_alt});
		}
	}
	std::stable_sort(scored.begin(), scored.end(),
	                 [](const ScoredVariant &x, const ScoredVariant &y) { return x.variant_idx < y.variant_idx; });
	return scored;
}

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
	bind_data->scored_variants = ResolveWeights(weights_it->second, c.variants, c.RangeStart(), c.RangeEnd());
	// the three arrays pgh_score takes, once per bind (a million scored variants: a few milliseconds of loops that
	// the first scan thread would otherwise spend with fifteen others waiting on it)
	const size_t n_scored = bind_data->scored_variants.size();
	bind_data->scored_vidx.resize(n_scored);
	bind_data->scored_weights.resize(n_scored);
	bind_data->scored_flip.resize(n_scored);
	for (size_t i = 0; i < n_scored; i++) {
		bind_data->scored_vidx[i] = bind_data->scored_variants[i].variant_idx;
		bind_data->scored_weights[i] = bind_data->scored_variants[i].weight;
		bind_data->scored_flip[i] = bind_data->scored_variants[i].flip;
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
	state->use_tally_cache = GetPlinkingTallyCache(context);
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

static void PlinkScoreScan(ClientContext &, TableFunctionInput &data_p, DataChunk &output) {
	auto &bind_data = data_p.bind_data->Cast<PlinkScoreBindData>();
	auto &gstate = data_p.global_state->Cast<PlinkScoreGlobalState>();
	{
		// Phase 1: the device call covers every scored variant, so the first thread
		// to arrive runs it while the others wait on the mutex, then all emit rows.
		std::lock_guard<std::mutex> lock(gstate.phase1_mutex);
		if (!gstate.scoring_done) {
			if (gstate.dataset && !bind_data.scored_variants.empty()) {
				{
					// hardcalls and dosage tracks alike: the library scores a dosage-bearing variant from its
					// dosages, as PgrGetD hands them to the reference loop (src/plink_score.cpp:586-652)
					const size_t n_scored = bind_data.scored_variants.size();
					const vector<uint32_t> &vidx = bind_data.scored_vidx;
					const vector<double> &weights = bind_data.scored_weights;
					const vector<uint8_t> &flip = bind_data.scored_flip;
					int mode = bind_data.center ? PGH_SCORE_CENTER
					                            : (bind_data.no_mean_imputation ? PGH_SCORE_NO_MEAN_IMPUTATION
					                                                            : PGH_SCORE_MEAN_IMPUTE);
					// the scored variants' tallies, if a pass over this file and subset already holds them
					// (plink_freq & co. ran before): the per-variant means then cost no read of the rows
					vector<uint32_t> counts;
					if (gstate.use_tally_cache) {
						auto pass = gstate.dataset->FindTally(
						    bind_data.c.has_sample_subset ? &bind_data.c.sample_subset->sample_include : nullptr,
						    vidx.front(), vidx.back() + 1);
						if (pass) {
							pass->Wait(PGH_TALLY_COUNTS, vidx.front(), vidx.back() + 1, "plink_score");
							counts.resize(4 * n_scored);
							for (size_t i = 0; i < n_scored; i++) {
								std::memcpy(&counts[4 * i], pass->Counts(vidx[i]), 16);
							}
						}
					}
					if (gstate.dataset->streamed) {
						// A file beyond the HBM budget: the score is a sum over variants, so the windows' partial sums
						// add (the scored variants are in file order, src/plink_score.cpp:407-408) -- one window of
						// the file resident at a time, as the reference's own scan streams the file.
						const size_t n_out = gstate.score_sums.size();
						vector<double> part_score(n_out), part_dosage(n_out);
						vector<uint32_t> part_ct(n_out);
						gstate.dataset->ForEachWindow(
						    vidx.front(), vidx.back() + 1,
						    bind_data.c.has_sample_subset ? &bind_data.c.sample_subset->sample_include : nullptr, "plink_score",
						    [&](pgh_dataset *win, pgh_subset *ss, uint32_t v0, uint32_t v1) {
							    const size_t lo = std::lower_bound(vidx.begin(), vidx.end(), v0) - vidx.begin();
							    const size_t hi = std::lower_bound(vidx.begin(), vidx.end(), v1) - vidx.begin();
							    if (lo == hi) {
								    return;
							    }
							    char werr[PGH_ERRBUF_LEN] = {0};
							    if (pgh_score_counts(win, ss, static_cast<uint32_t>(hi - lo), vidx.data() + lo, weights.data() + lo,
							                         flip.data() + lo, 1, mode, nullptr, part_score.data(),
							                         gstate.need_dosage_sum ? part_dosage.data() : nullptr, part_ct.data(),
							                         werr) != PGH_OK) {
								    throw IOException("plink_score: scoring variants [%u, %u) failed: %s", v0, v1, string(werr));
							    }
							    for (size_t k = 0; k < n_out; k++) {
								    gstate.score_sums[k] += part_score[k];
								    gstate.allele_cts[k] += part_ct[k];
								    if (gstate.need_dosage_sum) {
									    gstate.named_allele_dosage_sums[k] += part_dosage[k];
								    }
							    }
						    });
					} else {
					char errbuf[PGH_ERRBUF_LEN] = {0};
					int rc = pgh_score_counts(gstate.dataset->Resident("plink_score"), gstate.subset ? gstate.subset->handle : nullptr,
					                   static_cast<uint32_t>(n_scored), vidx.data(), weights.data(), flip.data(), 1, mode,
					                   counts.empty() ? nullptr : reinterpret_cast<const uint32_t(*)[4]>(counts.data()),
					                   gstate.score_sums.data(),
					                   gstate.need_dosage_sum ? gstate.named_allele_dosage_sums.data() : nullptr,
					                   gstate.allele_cts.data(), errbuf);
					if (rc != PGH_OK) {
						throw IOException("plink_score: scoring failed: %s", string(errbuf));
					}
					}
				}
			}
			gstate.scoring_done = true;
		}
	}

	// Phase 2: one row per sample
	// Phase 2: a run of output samples per call, one loop per projected column.  ALLELE_CT and
	// DENOM are the same number; SCORE_AVG = SCORE_SUM / ALLELE_CT (src/plink_score.cpp:686-699).
	const uint32_t first = gstate.next_sample_idx.fetch_add(STANDARD_VECTOR_SIZE);
	const idx_t n_rows =
	    first < gstate.total_samples ? std::min<idx_t>(STANDARD_VECTOR_SIZE, gstate.total_samples - first) : 0;
	for (idx_t out_col = 0; out_col < gstate.column_ids.size(); out_col++) {
		const auto file_col = gstate.column_ids[out_col];
		if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
			continue;
		}
		auto &vec = output.data[out_col];
		if (file_col == COL_FID || file_col == COL_IID) {
			for (idx_t r = 0; r < n_rows; r++) {
				FillSampleIdColumn(bind_data.c.sample_info(), file_col == COL_FID, bind_data.sample_output_order[first + r],
				                   vec, r, false);
			}
		} else if (file_col == COL_ALLELE_CT || file_col == COL_DENOM) {
			auto *dst = FlatVector::GetData<int32_t>(vec);
			for (idx_t r = 0; r < n_rows; r++) {
				dst[r] = static_cast<int32_t>(gstate.allele_cts[first + r]);
			}
		} else {
			auto *dst = FlatVector::GetData<double>(vec);
			for (idx_t r = 0; r < n_rows; r++) {
				const double sum = gstate.score_sums[first + r];
				const uint32_t allele_ct = gstate.allele_cts[first + r];
				dst[r] = file_col == COL_NAMED_ALLELE_DOSAGE_SUM ? gstate.named_allele_dosage_sums[first + r]
				         : file_col == COL_SCORE_SUM             ? sum
				         : allele_ct > 0                         ? sum / static_cast<double>(allele_ct)
				                                                 : 0.0;
			}
		}
	}
	CompatSetOutputCardinality(output, n_rows);
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

// plink_pca.cpp -- plink_pca(path, pvar, psam, mode, n_pcs, samples, region)
//
// Surface of the reference's src/plink_pca.cpp.  Bind keeps the reference's
// allele-frequency prepass (here one batched device tally instead of a
// sequential PgrGetCounts loop, src/plink_pca.cpp:392-416) and its dimension
// checks; the k+2 generation-barrier passes of the scan (src/plink_pca.cpp:961-1080)
// collapse into one pgh_pca call made by the first thread that enters Scan.
#include "variant_scan.hpp"

#include <mutex>
#include <random>

namespace duckdb {

enum class PcaMode { SAMPLES, PCS, BOTH };

static constexpr idx_t SCOL_FID = 0;
static constexpr idx_t SCOL_IID = 1;
static constexpr idx_t SCOL_PC_START = 2;
static constexpr idx_t PCOL_PC = 0;
static constexpr idx_t PCOL_EIGENVALUE = 1;
static constexpr idx_t PCOL_VARIANCE_PROPORTION = 2;
static constexpr idx_t PCOL_CUMULATIVE_VARIANCE = 3;

struct PlinkPcaBindData : public TableFunctionData {
	PgenBindCommon c;
	PcaMode mode = PcaMode::SAMPLES;
	uint32_t n_pcs = 10;
	uint32_t pc_ct_x2 = 0, qq_col_ct = 0;
	vector<uint32_t> effective_variants;
	vector<double> centers, inv_stdevs;
	vector<uint32_t> sample_output_order;
	shared_ptr<DeviceDataset> dataset;
	shared_ptr<DeviceSubset> subset;
};

struct PlinkPcaGlobalState : public GlobalTableFunctionState {
	uint32_t N = 0, M = 0, n_pcs = 0;
	vector<double> eigenvectors, eigenvalues;
	std::mutex algorithm_mutex;
	bool algorithm_done = false;
	std::atomic<uint32_t> next_emit_idx {0};
	vector<column_t> column_ids;
	uint32_t db_thread_count = 1, max_threads_config = 0;
	idx_t MaxThreads() const override {
		if (M < 240) {
			return 1;
		}
		return ApplyMaxThreadsCap(std::min<idx_t>(M / 240 + 1, db_thread_count), max_threads_config);
	}
};

struct PlinkPcaLocalState : public LocalTableFunctionState {};

static PcaMode ParsePcaMode(const string &s) {
	if (s == "samples") {
		return PcaMode::SAMPLES;
	} else if (s == "pcs") {
		return PcaMode::PCS;
	} else if (s == "both") {
		return PcaMode::BOTH;
	}
	throw InvalidInputException("plink_pca: invalid mode '%s' (expected 'samples', 'pcs', or 'both')", s);
}

static unique_ptr<FunctionData> PlinkPcaBind(ClientContext &context, TableFunctionBindInput &input,
                                             vector<LogicalType> &return_types, vector<string> &names) {
	auto bind_data = make_uniq<PlinkPcaBindData>();
	for (auto &kv : input.named_parameters) {
		if (kv.first == "mode") {
			bind_data->mode = ParsePcaMode(kv.second.GetValue<string>());
		} else if (kv.first == "n_pcs") {
			int32_t val = kv.second.GetValue<int32_t>();
			if (val < 1) {
				throw InvalidInputException("plink_pca: n_pcs must be >= 1 (got %d)", val);
			}
			bind_data->n_pcs = static_cast<uint32_t>(val);
		}
	}
	bind_data->pc_ct_x2 = 2 * bind_data->n_pcs;
	bind_data->qq_col_ct = (bind_data->n_pcs + 1) * bind_data->pc_ct_x2;
	auto &c = bind_data->c;
	c.Bind(context, input, "plink_pca", true);
	if (c.has_sample_subset) {
		bind_data->sample_output_order = c.sample_subset->sorted_indices;
	} else {
		bind_data->sample_output_order.resize(c.raw_sample_ct);
		for (uint32_t i = 0; i < c.raw_sample_ct; i++) {
			bind_data->sample_output_order[i] = i;
		}
	}

	// allele-frequency prepass over the range: drop all-missing and monomorphic variants
	uint32_t range_start = c.RangeStart(), range_end = c.RangeEnd();
	if (range_end > range_start) {
		bind_data->dataset = DeviceDataset::Acquire(c.pgen_path, "plink_pca");
		// (a file beyond the HBM budget is walked window by window, once per pass: RunAlgorithm)
		if (c.has_sample_subset) {
			bind_data->subset =
			    make_shared<DeviceSubset>(*bind_data->dataset, c.sample_subset->sample_include, "plink_pca");
		}
		// the range's tally pass: shared with plink_freq & co. and with the next plink_pca on this file and subset
		// (the reference's prepass is a single-threaded PgrGetCounts loop in bind, src/plink_pca.cpp:392-416)
		auto pass = bind_data->dataset->AcquireTally(c.has_sample_subset ? &c.sample_subset->sample_include : nullptr,
		                                             range_start, range_end, PGH_TALLY_COUNTS, false,
		                                             GetPlinkingTallyCache(context), "plink_pca");
		pass->Wait(PGH_TALLY_COUNTS, range_start, range_end, "plink_pca");
		for (uint32_t vidx = range_start; vidx < range_end; vidx++) {
			const uint32_t *gc = pass->Counts(vidx);
			uint32_t obs = gc[0] + gc[1] + gc[2];
			if (obs == 0) {
				continue;
			}
			double alt_freq =
			    (static_cast<double>(gc[1]) + 2.0 * static_cast<double>(gc[2])) / (2.0 * static_cast<double>(obs));
			VariantNorm norm = ComputeVariantNorm(alt_freq);
			if (norm.skip) {
				continue;
			}
			bind_data->effective_variants.push_back(vidx);
			bind_data->centers.push_back(norm.center);
			bind_data->inv_stdevs.push_back(norm.inv_stdev);
		}
	}
	uint32_t effective_variant_ct = static_cast<uint32_t>(bind_data->effective_variants.size());

	if (c.effective_sample_ct < 2) {
		throw InvalidInputException("plink_pca: need at least 2 samples (got %u)", c.effective_sample_ct);
	}
	if (bind_data->n_pcs >= c.effective_sample_ct) {
		throw InvalidInputException("plink_pca: n_pcs (%u) must be less than sample count (%u)", bind_data->n_pcs,
		                            c.effective_sample_ct);
	}
	if (effective_variant_ct <= bind_data->qq_col_ct) {
		throw InvalidInputException(
		    "plink_pca: too few variants (%u) for %u PCs with approx mode (need > %u non-monomorphic variants)",
		    effective_variant_ct, bind_data->n_pcs, bind_data->qq_col_ct);
	}
	if (c.effective_sample_ct <= bind_data->qq_col_ct) {
		throw InvalidInputException("plink_pca: too few samples (%u) for %u PCs with approx mode "
		                            "(need > %u samples; try fewer PCs or more samples)",
		                            c.effective_sample_ct, bind_data->n_pcs, bind_data->qq_col_ct);
	}
	uint64_t qq_elements = static_cast<uint64_t>(effective_variant_ct) * bind_data->qq_col_ct;
	uint64_t max_elements = 16ULL * 1024 * 1024 * 1024;
	Value max_elements_val;
	if (context.TryGetCurrentSetting("plinking_max_matrix_elements", max_elements_val)) {
		auto val = max_elements_val.GetValue<int64_t>();
		max_elements = val > 0 ? static_cast<uint64_t>(val) : 0;
	}
	if (qq_elements > max_elements) {
		throw InvalidInputException(
		    "plink_pca: QQ matrix would require %llu elements (%llu MB), exceeding plinking_max_matrix_elements "
		    "(%lld). Reduce n_pcs or variant count, or increase the limit with SET plinking_max_matrix_elements = "
		    "<value>.",
		    static_cast<unsigned long long>(qq_elements),
		    static_cast<unsigned long long>(qq_elements * 8 / (1024 * 1024)), static_cast<long long>(max_elements));
	}

	if (bind_data->mode == PcaMode::PCS) {
		names = {"PC", "EIGENVALUE", "VARIANCE_PROPORTION", "CUMULATIVE_VARIANCE"};
		return_types = {LogicalType::INTEGER, LogicalType::DOUBLE, LogicalType::DOUBLE, LogicalType::DOUBLE};
	} else if (bind_data->mode == PcaMode::SAMPLES) {
		names = {"FID", "IID"};
		return_types = {LogicalType::VARCHAR, LogicalType::VARCHAR};
		for (uint32_t i = 0; i < bind_data->n_pcs; i++) {
			names.push_back("PC" + std::to_string(i + 1));
			return_types.push_back(LogicalType::DOUBLE);
		}
	} else {
		child_list_t eigenvec_fields;
		eigenvec_fields.push_back({"FID", LogicalType::VARCHAR});
		eigenvec_fields.push_back({"IID", LogicalType::VARCHAR});
		for (uint32_t i = 0; i < bind_data->n_pcs; i++) {
			eigenvec_fields.push_back({"PC" + std::to_string(i + 1), LogicalType::DOUBLE});
		}
		names = {"EIGENVEC", "EIGENVAL"};
		return_types = {LogicalType::LIST(LogicalType::STRUCT(std::move(eigenvec_fields))),
		                LogicalType::LIST(LogicalType::DOUBLE)};
	}
	return std::move(bind_data);
}

static unique_ptr<GlobalTableFunctionState> PlinkPcaInitGlobal(ClientContext &context, TableFunctionInitInput &input) {
	auto &bind_data = input.bind_data->Cast<PlinkPcaBindData>();
	auto state = make_uniq<PlinkPcaGlobalState>();
	state->N = bind_data.c.effective_sample_ct;
	state->M = static_cast<uint32_t>(bind_data.effective_variants.size());
	state->n_pcs = bind_data.n_pcs;
	state->column_ids = input.column_ids;
	state->db_thread_count = static_cast<uint32_t>(context.db_threads);
	state->max_threads_config = GetPlinkingMaxThreads(context);
	state->eigenvectors.assign(static_cast<size_t>(state->N) * state->n_pcs, 0.0);
	state->eigenvalues.assign(state->n_pcs, 0.0);
	return std::move(state);
}

static unique_ptr<LocalTableFunctionState> PlinkPcaInitLocal(ExecutionContext &, TableFunctionInitInput &,
                                                             GlobalTableFunctionState *) {
	return make_uniq<PlinkPcaLocalState>();
}

static void RunAlgorithm(const PlinkPcaBindData &bind_data, PlinkPcaGlobalState &gs) {
	// G1 seed: the exact libstdc++ stream the reference draws (src/plink_pca.cpp:517-523)
	// The seed is fixed, so the matrix depends on its size only: the last one is kept (ten million sequential
	// draws are ~0.1 s of a 0.5 s call at 500,000 samples).
	static std::mutex seed_mutex;
	static shared_ptr<const vector<double>> seed_cache;
	const size_t g1_len = static_cast<size_t>(gs.N) * bind_data.pc_ct_x2;
	shared_ptr<const vector<double>> g1_ptr;
	{
		std::lock_guard<std::mutex> lock(seed_mutex);
		if (!seed_cache || seed_cache->size() != g1_len) {
			auto fresh = make_shared<vector<double>>(g1_len);
			std::mt19937_64 rng(12345);
			std::normal_distribution<double> dist(0.0, 1.0);
			for (auto &val : *fresh) {
				val = dist(rng);
			}
			seed_cache = fresh;
		}
		g1_ptr = seed_cache;
	}
	const vector<double> &g1 = *g1_ptr;
	char errbuf[PGH_ERRBUF_LEN] = {0};
	int rc;
	if (bind_data.dataset->streamed) {
		// the reference's passes stream the file as well, 240 variants at a time (src/plink_pca.cpp:632-676); here a
		// window is half the HBM budget and every pass opens the windows one after the other (n_pcs + 2 reads of the file)
		rc = pgh_pca_streamed(bind_data.dataset->path.c_str(), nullptr,
		                      bind_data.c.has_sample_subset ? bind_data.c.sample_subset->sample_include.data() : nullptr, gs.M,
		                      bind_data.effective_variants.data(), bind_data.centers.data(), bind_data.inv_stdevs.data(),
		                      bind_data.n_pcs, g1.data(), bind_data.dataset->WindowVariants(), gs.eigenvalues.data(),
		                      gs.eigenvectors.data(), errbuf);
	} else {
		rc = pgh_pca(bind_data.dataset->Resident("plink_pca"), bind_data.subset ? bind_data.subset->handle : nullptr, gs.M,
		             bind_data.effective_variants.data(), bind_data.centers.data(), bind_data.inv_stdevs.data(),
		             bind_data.n_pcs, g1.data(), gs.eigenvalues.data(), gs.eigenvectors.data(), errbuf);
	}
	if (rc != PGH_OK) {
		throw IOException("plink_pca: %s", string(errbuf));
	}
}

static void PlinkPcaScan(ClientContext &, TableFunctionInput &data_p, DataChunk &output) {
	auto &bind_data = data_p.bind_data->Cast<PlinkPcaBindData>();
	auto &gs = data_p.global_state->Cast<PlinkPcaGlobalState>();
	{
		std::lock_guard<std::mutex> lock(gs.algorithm_mutex);
		if (!gs.algorithm_done) {
			RunAlgorithm(bind_data, gs);
			gs.algorithm_done = true;
		}
	}
	auto &column_ids = gs.column_ids;
	idx_t rows_emitted = 0;

	if (bind_data.mode == PcaMode::SAMPLES) {
		// a run of output samples per call, one loop per projected column
		const uint32_t first = gs.next_emit_idx.fetch_add(STANDARD_VECTOR_SIZE);
		rows_emitted = first < gs.N ? std::min<idx_t>(STANDARD_VECTOR_SIZE, gs.N - first) : 0;
		for (idx_t out_col = 0; out_col < column_ids.size(); out_col++) {
			const auto file_col = column_ids[out_col];
			if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
				continue;
			}
			auto &vec = output.data[out_col];
			if (file_col < SCOL_PC_START) {
				for (idx_t r = 0; r < rows_emitted; r++) {
					FillSampleIdColumn(bind_data.c.sample_info(), file_col == SCOL_FID,
					                   bind_data.sample_output_order[first + r], vec, r, false);
				}
			} else if (file_col < SCOL_PC_START + bind_data.n_pcs) {
				const size_t pc = file_col - SCOL_PC_START;
				auto *dst = FlatVector::GetData<double>(vec);
				for (idx_t r = 0; r < rows_emitted; r++) {
					dst[r] = gs.eigenvectors[static_cast<size_t>(first + r) * gs.n_pcs + pc];
				}
			}
		}
	} else if (bind_data.mode == PcaMode::PCS) {
		// proportions are over the k reported eigenvalues only (src/plink_pca.cpp:777-796)
		const uint32_t first = gs.next_emit_idx.fetch_add(STANDARD_VECTOR_SIZE);
		rows_emitted = first < gs.n_pcs ? std::min<idx_t>(STANDARD_VECTOR_SIZE, gs.n_pcs - first) : 0;
		double total_variance = 0.0, running = 0.0;
		for (uint32_t i = 0; i < gs.n_pcs; i++) {
			total_variance += gs.eigenvalues[i];
		}
		for (uint32_t i = 0; i < first; i++) {
			running += gs.eigenvalues[i];
		}
		for (idx_t r = 0; r < rows_emitted; r++) {
			const double eigenvalue = gs.eigenvalues[first + r];
			running += eigenvalue;
			for (idx_t out_col = 0; out_col < column_ids.size(); out_col++) {
				const auto file_col = column_ids[out_col];
				if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
					continue;
				}
				auto &vec = output.data[out_col];
				if (file_col == PCOL_PC) {
					FlatVector::GetData<int32_t>(vec)[r] = static_cast<int32_t>(first + r + 1);
				} else {
					const double share = file_col == PCOL_EIGENVALUE ? eigenvalue
					                     : total_variance <= 0.0    ? 0.0
					                     : (file_col == PCOL_VARIANCE_PROPORTION ? eigenvalue : running) / total_variance;
					FlatVector::GetData<double>(vec)[r] = share;
				}
			}
		}
	} else {
		// 'both': one row {EIGENVEC: LIST(STRUCT(FID, IID, PC1..)), EIGENVAL: LIST(DOUBLE)}
		if (gs.next_emit_idx.fetch_add(1) == 0) {
			const bool has_fid = !bind_data.c.sample_info().fids.empty();
			vector<Value> eigenvec_entries;
			for (uint32_t sidx = 0; sidx < gs.N; sidx++) {
				uint32_t orig_idx = bind_data.sample_output_order[sidx];
				vector<std::pair<string, Value>> fields;
				fields.emplace_back("FID", has_fid ? Value::VARCHAR(bind_data.c.sample_info().fids[orig_idx])
				                                   : Value(LogicalType::VARCHAR));
				fields.emplace_back("IID", Value::VARCHAR(bind_data.c.sample_info().iids[orig_idx]));
				for (uint32_t pc = 0; pc < gs.n_pcs; pc++) {
					fields.emplace_back("PC" + std::to_string(pc + 1),
					                    Value::DOUBLE(gs.eigenvectors[static_cast<size_t>(sidx) * gs.n_pcs + pc]));
				}
				eigenvec_entries.push_back(Value::STRUCT(std::move(fields)));
			}
			vector<Value> eigenval_entries;
			for (uint32_t pc = 0; pc < gs.n_pcs; pc++) {
				eigenval_entries.push_back(Value::DOUBLE(gs.eigenvalues[pc]));
			}
			for (idx_t out_col = 0; out_col < column_ids.size(); out_col++) {
				auto file_col = column_ids[out_col];
				if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
					continue;
				}
				auto &vec = output.data[out_col];
				if (file_col == 0) {
					vec.SetValue(0, Value::LIST(*vec.type.child, std::move(eigenvec_entries)));
				} else if (file_col == 1) {
					vec.SetValue(0, Value::LIST(LogicalType::DOUBLE, std::move(eigenval_entries)));
				}
			}
			rows_emitted = 1;
		}
	}
	CompatSetOutputCardinality(output, rows_emitted);
}

void RegisterPlinkPca(ExtensionLoader &loader) {
	TableFunction plink_pca("plink_pca", {LogicalType::VARCHAR}, PlinkPcaScan, PlinkPcaBind, PlinkPcaInitGlobal,
	                        PlinkPcaInitLocal);
	plink_pca.projection_pushdown = true;
	plink_pca.named_parameters["pvar"] = LogicalType::VARCHAR;
	plink_pca.named_parameters["psam"] = LogicalType::VARCHAR;
	plink_pca.named_parameters["mode"] = LogicalType::VARCHAR;
	plink_pca.named_parameters["n_pcs"] = LogicalType::INTEGER;
	plink_pca.named_parameters["samples"] = LogicalType::ANY;
	plink_pca.named_parameters["region"] = LogicalType::VARCHAR;
	loader.RegisterFunction(plink_pca);
}

} // namespace duckdb

// plink_freq.cpp -- plink_freq(path, pvar, psam, samples, region, counts, dosage, build)
//
// Same table-function surface as the reference's src/plink_freq.cpp (bind /
// init_global / init_local / scan, column ids, NULL rules); the per-variant
// PgrGetCounts calls are replaced by batched tallies from libpgenhip.
#include "variant_scan.hpp"

#include <cmath>

namespace duckdb {

// CHROM(0) POS(1) ID(2) REF(3) ALT(4) ALT_FREQ(5) OBS_CT(6)
// counts := true  -> + HOM_REF_CT(7) HET_CT(8) HOM_ALT_CT(9) MISSING_CT(10)
// dosage := true  -> + IMP_R2 (dynamic index)
static constexpr idx_t COL_ALT_FREQ = 5;
static constexpr idx_t COL_OBS_CT = 6;
static constexpr idx_t COL_HOM_REF_CT = 7;
static constexpr idx_t COL_HET_CT = 8;
static constexpr idx_t COL_HOM_ALT_CT = 9;
static constexpr idx_t COL_MISSING_CT = 10;

struct PlinkFreqBindData : public TableFunctionData {
	PgenBindCommon c;
	bool include_counts = false;
	bool include_dosage = false;
	ParBounds par_bounds;
	PloidyMap ploidy; // per CHROM run, built once at bind
	bool have_sex = false;
	idx_t imp_r2_col_idx = 0;
};

struct PlinkFreqGlobalState : public GlobalTableFunctionState {
	VariantScanGlobal scan;
	vector<column_t> column_ids;
	bool need_frequencies = false;
	uint32_t max_threads_config = 0;

	bool dosage_on_device = false; // dosage := true on a file with dosage tracks: moments from pgh_dosage_sums
	idx_t MaxThreads() const override {
		uint32_t range = scan.end_variant_idx - scan.start_variant_idx;
		return ApplyMaxThreadsCap(range / 500 + 1, max_threads_config);
	}
};

struct PlinkFreqLocalState : public LocalTableFunctionState {
	VariantScanLocal scan;
};

static unique_ptr<FunctionData> PlinkFreqBind(ClientContext &context, TableFunctionBindInput &input,
                                              vector<LogicalType> &return_types, vector<string> &names) {
	auto bind_data = make_uniq<PlinkFreqBindData>();
	string build_str = "GRCh38";
	for (auto &kv : input.named_parameters) {
		if (kv.first == "counts") {
			bind_data->include_counts = kv.second.GetValue<bool>();
		} else if (kv.first == "dosage") {
			bind_data->include_dosage = kv.second.GetValue<bool>();
		} else if (kv.first == "build") {
			build_str = kv.second.GetValue<string>();
		}
	}
	bind_data->par_bounds = ResolveParBounds(build_str, "plink_freq");
	bind_data->c.Bind(context, input, "plink_freq", false);
	bind_data->have_sex = bind_data->c.has_sample_info && !bind_data->c.sample_info().sexes.empty();
	bind_data->ploidy = PloidyMap(bind_data->c.variants, bind_data->par_bounds);

	names = {"CHROM", "POS", "ID", "REF", "ALT", "ALT_FREQ", "OBS_CT"};
	return_types = {LogicalType::VARCHAR, LogicalType::INTEGER, LogicalType::VARCHAR, LogicalType::VARCHAR,
	                LogicalType::VARCHAR, LogicalType::DOUBLE,  LogicalType::INTEGER};
	if (bind_data->include_counts) {
		names.insert(names.end(), {"HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "MISSING_CT"});
		return_types.insert(return_types.end(),
		                    {LogicalType::INTEGER, LogicalType::INTEGER, LogicalType::INTEGER, LogicalType::INTEGER});
	}
	if (bind_data->include_dosage) {
		bind_data->imp_r2_col_idx = names.size();
		names.push_back("IMP_R2");
		return_types.push_back(LogicalType::DOUBLE);
	}
	return std::move(bind_data);
}

static unique_ptr<GlobalTableFunctionState> PlinkFreqInitGlobal(ClientContext &context, TableFunctionInitInput &input) {
	auto &bind_data = input.bind_data->Cast<PlinkFreqBindData>();
	auto state = make_uniq<PlinkFreqGlobalState>();
	state->scan.start_variant_idx = bind_data.c.RangeStart();
	state->scan.end_variant_idx = bind_data.c.RangeEnd();
	state->scan.next_variant_idx.store(state->scan.start_variant_idx);
	state->scan.effective_sample_ct = bind_data.c.effective_sample_ct;
	state->column_ids = input.column_ids;
	state->max_threads_config = GetPlinkingMaxThreads(context);
	for (auto col_id : input.column_ids) {
		if (col_id == COLUMN_IDENTIFIER_ROW_ID) {
			continue;
		}
		if ((col_id >= COL_ALT_FREQ && col_id <= COL_MISSING_CT) ||
		    (bind_data.include_dosage && col_id == bind_data.imp_r2_col_idx)) {
			state->need_frequencies = true;
			break;
		}
	}
	if (state->need_frequencies) {
		// one HBM-resident copy of the genotype matrix for all scan threads
		state->scan.dataset = DeviceDataset::Acquire(bind_data.c.pgen_path, "plink_freq");
		if (bind_data.c.has_sample_subset) {
			state->scan.subset =
			    make_uniq<DeviceSubset>(*state->scan.dataset, bind_data.c.sample_subset->sample_include, "plink_freq");
		}
		// the range's tallies start now and land while the scan threads emit (variant_scan.hpp); with no sample
		// subset the same pass tallies missing calls per sample too, so plink_hardy / plink_missing on this file
		// need no pass of their own (the reference scans three times: src/plink_freq.cpp:482,
		// src/plink_hardy.cpp:510, src/plink_missing.cpp:479,593-609)
		state->scan.StartTallies(bind_data.c.sample_subset.get(), bind_data.have_sex ? &bind_data.c.sample_info() : nullptr,
		                         bind_data.c.raw_sample_ct, &bind_data.ploidy,
		                         bind_data.c.has_sample_subset ? 0u : static_cast<uint32_t>(PGH_TALLY_SAMPLE_MISSING), false,
		                         GetPlinkingTallyCache(context), "plink_freq");
		state->dosage_on_device = bind_data.include_dosage && bind_data.c.file_has_dosage;
	}
	return std::move(state);
}

static unique_ptr<LocalTableFunctionState> PlinkFreqInitLocal(ExecutionContext &, TableFunctionInitInput &input,
                                                              GlobalTableFunctionState *global_state) {
	(void)input;
	(void)global_state;
	return make_uniq<PlinkFreqLocalState>();
}

static void PlinkFreqScan(ClientContext &, TableFunctionInput &data_p, DataChunk &output) {
	auto &bind_data = data_p.bind_data->Cast<PlinkFreqBindData>();
	auto &gstate = data_p.global_state->Cast<PlinkFreqGlobalState>();
	auto &lstate = data_p.local_state->Cast<PlinkFreqLocalState>();
	auto &column_ids = gstate.column_ids;
	auto &variants = bind_data.c.variants;

	// Two steps per chunk, the columnar way round: first the derived values of every row the
	// chunk will hold (the tallies are already there, batch by batch), then one tight loop per
	// projected column.
	struct FreqRow {
		uint32_t vidx;
		uint32_t obs_ct = 0;
		double alt_freq = 0.0;
		double imp_r2 = 0.0;
		int32_t counts[4] = {0, 0, 0, 0};
		bool freq_is_null = false, counts_are_null = false, r2_is_null = false;
	};
	static constexpr uint64_t kDosageMid = 16384;
	vector<FreqRow> rows;
	rows.reserve(STANDARD_VECTOR_SIZE);
	vector<uint32_t> dosage_rows, dosage_vidx; // rows whose frequency comes from the dosage tracks
	uint32_t vidx;
	while (rows.size() < STANDARD_VECTOR_SIZE && lstate.scan.Next(gstate.scan, "plink_freq", vidx)) {
		FreqRow row;
		row.vidx = vidx;
		if (!gstate.need_frequencies) {
			rows.push_back(row);
			continue;
		}
		const ChromPloidy ploidy = bind_data.ploidy.At(vidx);
		const uint32_t *gc = lstate.scan.Counts(vidx);
		for (int k = 0; k < 4; k++) {
			row.counts[k] = static_cast<int32_t>(gc[k]);
		}
		const uint32_t observed = gc[0] + gc[1] + gc[2];
		if (ploidy != ChromPloidy::AUTOSOMAL) {
			// chrX / chrY / chrMT: ploidy-aware allele counts from the sex strata
			// (ComputeSexAwareCounts, src/plink_freq.cpp:463-472); IMP_R2 is not defined here
			SexAwareCounts sac = SexAwareFromStrata(ploidy, gc, lstate.scan.MaleCounts(vidx),
			                                        lstate.scan.FemaleCounts(vidx), bind_data.have_sex);
			row.r2_is_null = true;
			row.counts_are_null = sac.sex_unavailable;
			row.freq_is_null = sac.sex_unavailable || sac.obs_allele_ct == 0;
			if (!row.freq_is_null) {
				row.alt_freq = static_cast<double>(sac.alt_allele_ct) / static_cast<double>(sac.obs_allele_ct);
				row.obs_ct = sac.obs_allele_ct;
			}
			row.counts[0] = static_cast<int32_t>(sac.geno_hom_ref);
			row.counts[1] = static_cast<int32_t>(sac.geno_het);
			row.counts[2] = static_cast<int32_t>(sac.geno_hom_alt);
			row.counts[3] = static_cast<int32_t>(sac.geno_missing);
		} else if (bind_data.include_dosage) {
			// dosage-weighted allele sums (PgrGetDCounts, src/plink_freq.cpp:525-535)
			uint64_t alt_sum, ref_sum;
			row.r2_is_null = !bind_data.c.file_has_dosage;
			if (gstate.dosage_on_device) {
				// the moments come from one device call per chunk, below
				dosage_rows.push_back(static_cast<uint32_t>(rows.size()));
				dosage_vidx.push_back(vidx);
				rows.push_back(row);
				continue;
			} else {
				alt_sum = (static_cast<uint64_t>(gc[1]) + 2ull * gc[2]) * kDosageMid;
				ref_sum = 2ull * observed * kDosageMid - alt_sum;
			}
			const uint64_t total = alt_sum + ref_sum;
			row.freq_is_null = total == 0;
			if (total) {
				row.alt_freq = static_cast<double>(alt_sum) / static_cast<double>(total);
				row.obs_ct = static_cast<uint32_t>(total / kDosageMid);
			}
		} else {
			// src/plink_freq.cpp:536-544: OBS_CT counts alleles; no observation -> NULL frequency, OBS_CT 0
			row.freq_is_null = observed == 0;
			if (observed) {
				row.obs_ct = 2 * observed;
				row.alt_freq = (static_cast<double>(gc[1]) + 2.0 * static_cast<double>(gc[2])) /
				               (2.0 * static_cast<double>(observed));
			}
		}
		rows.push_back(row);
	}

	if (!dosage_rows.empty()) {
		// PgrGetDCounts for the chunk's rows at once (src/plink_freq.cpp:475-480, :525-535)
		vector<uint64_t> moments(3 * dosage_rows.size());
		char errbuf[PGH_ERRBUF_LEN] = {0};
		// (a file beyond the HBM budget: the window that holds the chunk's variants -- they are one claim's, ascending)
		RowLease span = LeaseRows(*gstate.scan.dataset, gstate.scan.subset.get(), gstate.scan.row_windows,
		                          bind_data.c.has_sample_subset ? &bind_data.c.sample_subset->sample_include : nullptr,
		                          dosage_vidx.front(), dosage_vidx.back() + 1, bind_data.c.raw_variant_ct, "plink_freq");
		if (pgh_dosage_sums(span.ds, span.ss, 0,
		                    static_cast<uint32_t>(dosage_vidx.size()), dosage_vidx.data(),
		                    reinterpret_cast<uint64_t(*)[3]>(moments.data()), errbuf) != PGH_OK) {
			throw IOException("plink_freq: PgrGetDCounts failed for variants [%u, %u]: %s", dosage_vidx.front(),
			                  dosage_vidx.back(), string(errbuf));
		}
		for (size_t k = 0; k < dosage_rows.size(); k++) {
			FreqRow &row = rows[dosage_rows[k]];
			const uint64_t sum = moments[3 * k], ssq = moments[3 * k + 1], nm = moments[3 * k + 2];
			if (nm) { // MaCH r2 = popvar(d) / (2 p (1 - p)) on the 16384 scale
				const double sumd = static_cast<double>(sum);
				const double avg = sumd / static_cast<double>(nm);
				row.imp_r2 = 2.0 * (static_cast<double>(ssq) - sumd * avg) / (sumd * (32768.0 - avg));
			}
			const uint64_t total = nm * 2 * kDosageMid; // alt + ref dosage
			row.freq_is_null = total == 0;
			if (total) {
				row.alt_freq = static_cast<double>(sum) / static_cast<double>(total);
				row.obs_ct = static_cast<uint32_t>(total / kDosageMid);
			}
		}
	}

	const idx_t n_rows = rows.size();
	for (idx_t out_col = 0; out_col < column_ids.size(); out_col++) {
		const auto file_col = column_ids[out_col];
		if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
			continue;
		}
		auto &vec = output.data[out_col];
		if (bind_data.include_dosage && file_col == bind_data.imp_r2_col_idx) {
			auto *dst = FlatVector::GetData<double>(vec);
			for (idx_t r = 0; r < n_rows; r++) {
				if (rows[r].r2_is_null) {
					FlatVector::SetNull(vec, r, true);
				} else {
					dst[r] = rows[r].imp_r2;
				}
			}
		} else if (file_col < COL_ALT_FREQ) {
			for (idx_t r = 0; r < n_rows; r++) {
				FillVariantMetadataColumn(variants, file_col, rows[r].vidx, vec, r);
			}
		} else if (file_col == COL_ALT_FREQ) {
			auto *dst = FlatVector::GetData<double>(vec);
			for (idx_t r = 0; r < n_rows; r++) {
				if (rows[r].freq_is_null) {
					FlatVector::SetNull(vec, r, true);
				} else {
					dst[r] = rows[r].alt_freq;
				}
			}
		} else if (file_col == COL_OBS_CT) {
			auto *dst = FlatVector::GetData<int32_t>(vec);
			for (idx_t r = 0; r < n_rows; r++) {
				dst[r] = static_cast<int32_t>(rows[r].obs_ct);
			}
		} else if (file_col >= COL_HOM_REF_CT && file_col <= COL_MISSING_CT) {
			const int which = static_cast<int>(file_col - COL_HOM_REF_CT);
			auto *dst = FlatVector::GetData<int32_t>(vec);
			for (idx_t r = 0; r < n_rows; r++) {
				if (rows[r].counts_are_null) {
					FlatVector::SetNull(vec, r, true);
				} else {
					dst[r] = rows[r].counts[which];
				}
			}
		}
	}
	CompatSetOutputCardinality(output, n_rows);
}

void RegisterPlinkFreq(ExtensionLoader &loader) {
	TableFunction plink_freq("plink_freq", {LogicalType::VARCHAR}, PlinkFreqScan, PlinkFreqBind, PlinkFreqInitGlobal,
	                         PlinkFreqInitLocal);
	plink_freq.projection_pushdown = true;
	plink_freq.named_parameters["pvar"] = LogicalType::VARCHAR;
	plink_freq.named_parameters["psam"] = LogicalType::VARCHAR;
	plink_freq.named_parameters["samples"] = LogicalType::ANY;
	plink_freq.named_parameters["region"] = LogicalType::VARCHAR;
	plink_freq.named_parameters["counts"] = LogicalType::BOOLEAN;
	plink_freq.named_parameters["dosage"] = LogicalType::BOOLEAN;
	plink_freq.named_parameters["build"] = LogicalType::VARCHAR;
	loader.RegisterFunction(plink_freq);
}

} // namespace duckdb

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
	bool have_sex = false;
	idx_t imp_r2_col_idx = 0;
};

struct PlinkFreqGlobalState : public GlobalTableFunctionState {
	VariantScanGlobal scan;
	vector<column_t> column_ids;
	bool need_frequencies = false;
	uint32_t max_threads_config = 0;

	idx_t MaxThreads() const override {
		uint32_t range = scan.end_variant_idx - scan.start_variant_idx;
		return ApplyMaxThreadsCap(range / 500 + 1, max_threads_config);
	}
};

struct PlinkFreqLocalState : public LocalTableFunctionState {
	VariantScanLocal scan;
	pgh_reader *reader = nullptr; // per-variant dosage decode (dosage := true on a dosage file)
	vector<double> dosage_doubles;
	~PlinkFreqLocalState() override {
		if (reader) {
			pgh_reader_destroy(reader);
		}
	}
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
	bind_data->have_sex = bind_data->c.has_sample_info && !bind_data->c.sample_info.sexes.empty();

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
		if (bind_data.have_sex) {
			BuildSexStrata(state->scan, bind_data.c.sample_info, bind_data.c.sample_subset.get(),
			               bind_data.c.raw_sample_ct, "plink_freq");
		}
	}
	return std::move(state);
}

static unique_ptr<LocalTableFunctionState> PlinkFreqInitLocal(ExecutionContext &, TableFunctionInitInput &input,
                                                              GlobalTableFunctionState *global_state) {
	auto &bind_data = input.bind_data->Cast<PlinkFreqBindData>();
	auto &gstate = global_state->Cast<PlinkFreqGlobalState>();
	auto state = make_uniq<PlinkFreqLocalState>();
	if (gstate.need_frequencies && bind_data.include_dosage && bind_data.c.file_has_dosage) {
		char errbuf[PGH_ERRBUF_LEN] = {0};
		int rc = pgh_reader_create(gstate.scan.dataset->handle, gstate.scan.subset ? gstate.scan.subset->handle : nullptr,
		                           &state->reader, errbuf);
		if (rc != PGH_OK) {
			throw IOException("plink_freq: thread init failed: %s", string(errbuf));
		}
		state->dosage_doubles.resize(bind_data.c.effective_sample_ct);
	}
	return std::move(state);
}

static void PlinkFreqScan(ClientContext &, TableFunctionInput &data_p, DataChunk &output) {
	auto &bind_data = data_p.bind_data->Cast<PlinkFreqBindData>();
	auto &gstate = data_p.global_state->Cast<PlinkFreqGlobalState>();
	auto &lstate = data_p.local_state->Cast<PlinkFreqLocalState>();
	auto &column_ids = gstate.column_ids;
	auto &variants = bind_data.c.variants;

	auto needs_strata = [&](uint32_t begin, uint32_t end) {
		for (uint32_t v = begin; v < end; v++) {
			if (ClassifyChromPloidy(variants.GetChrom(v), variants.GetPos(v), bind_data.par_bounds) !=
			    ChromPloidy::AUTOSOMAL) {
				return true;
			}
		}
		return false;
	};

	idx_t rows_emitted = 0;
	uint32_t vidx;
	while (rows_emitted < STANDARD_VECTOR_SIZE && lstate.scan.Next(gstate.scan, "plink_freq", needs_strata, vidx)) {
		ChromPloidy ploidy = ChromPloidy::AUTOSOMAL;
		if (gstate.need_frequencies) {
			ploidy = ClassifyChromPloidy(variants.GetChrom(vidx), variants.GetPos(vidx), bind_data.par_bounds);
		}
		const bool sex_aware = ploidy != ChromPloidy::AUTOSOMAL;

		uint32_t genocounts[4] = {0, 0, 0, 0};
		uint64_t all_dosages[2] = {0, 0};
		double imp_r2 = 0.0;
		SexAwareCounts sac;
		if (gstate.need_frequencies) {
			std::memcpy(genocounts, lstate.scan.Counts(vidx), sizeof genocounts);
			if (sex_aware) {
				static const uint32_t zero[4] = {0, 0, 0, 0};
				const bool strata = lstate.scan.have_strata;
				sac = SexAwareFromStrata(ploidy, genocounts, strata ? lstate.scan.MaleCounts(vidx) : zero,
				                         strata ? lstate.scan.FemaleCounts(vidx) : zero, bind_data.have_sex);
			} else if (bind_data.include_dosage) {
				static constexpr uint64_t kDosageMid = 16384;
				if (lstate.reader) {
					// PgrGetDCounts: dosage-weighted sums + MaCH r2, decoded per variant
					if (pgh_get_dosage_f64(lstate.reader, vidx, lstate.dosage_doubles.data()) != PGH_OK) {
						throw IOException("plink_freq: PgrGetDCounts failed for variant %u: %s", vidx,
						                  string(pgh_reader_error(lstate.reader)));
					}
					uint64_t sum = 0, ssq = 0;
					uint32_t nm = 0;
					for (double d : lstate.dosage_doubles) {
						if (d == -9.0) {
							continue;
						}
						uint64_t u = static_cast<uint64_t>(std::llround(d * 16384.0));
						sum += u;
						ssq += u * u;
						nm++;
					}
					all_dosages[1] = sum;
					all_dosages[0] = static_cast<uint64_t>(nm) * 2 * kDosageMid - sum;
					if (nm) {
						double sumd = static_cast<double>(sum);
						double avg = sumd / static_cast<double>(nm);
						double var = static_cast<double>(ssq) - sumd * avg;
						imp_r2 = 2.0 * var / (sumd * (32768.0 - avg));
					}
				} else {
					uint32_t obs = genocounts[0] + genocounts[1] + genocounts[2];
					all_dosages[1] = (static_cast<uint64_t>(genocounts[1]) + 2ull * genocounts[2]) * kDosageMid;
					all_dosages[0] = 2ull * obs * kDosageMid - all_dosages[1];
				}
			}
		}

		static constexpr uint64_t kDosageMid = 16384;
		uint32_t hardcall_obs_sample_ct = genocounts[0] + genocounts[1] + genocounts[2];
		uint32_t obs_ct;
		double alt_freq;
		bool freq_is_null = false;
		int32_t out_hom_ref = static_cast<int32_t>(genocounts[0]);
		int32_t out_het = static_cast<int32_t>(genocounts[1]);
		int32_t out_hom_alt = static_cast<int32_t>(genocounts[2]);
		int32_t out_missing = static_cast<int32_t>(genocounts[3]);
		bool counts_are_null = false;

		if (sex_aware) {
			if (sac.sex_unavailable || sac.obs_allele_ct == 0) {
				freq_is_null = true;
				alt_freq = 0.0;
				obs_ct = 0;
				counts_are_null = sac.sex_unavailable;
			} else {
				alt_freq = static_cast<double>(sac.alt_allele_ct) / static_cast<double>(sac.obs_allele_ct);
				obs_ct = sac.obs_allele_ct;
			}
			out_hom_ref = static_cast<int32_t>(sac.geno_hom_ref);
			out_het = static_cast<int32_t>(sac.geno_het);
			out_hom_alt = static_cast<int32_t>(sac.geno_hom_alt);
			out_missing = static_cast<int32_t>(sac.geno_missing);
		} else if (bind_data.include_dosage) {
			uint64_t total_dosage = all_dosages[0] + all_dosages[1];
			if (total_dosage == 0) {
				freq_is_null = true;
				alt_freq = 0.0;
				obs_ct = 0;
			} else {
				alt_freq = static_cast<double>(all_dosages[1]) / static_cast<double>(total_dosage);
				obs_ct = static_cast<uint32_t>(total_dosage / kDosageMid);
			}
		} else if (hardcall_obs_sample_ct == 0) {
			freq_is_null = true;
			alt_freq = 0.0;
			obs_ct = 0;
		} else {
			obs_ct = 2 * hardcall_obs_sample_ct;
			alt_freq = (static_cast<double>(genocounts[1]) + 2.0 * static_cast<double>(genocounts[2])) /
			           (2.0 * static_cast<double>(hardcall_obs_sample_ct));
		}

		for (idx_t out_col = 0; out_col < column_ids.size(); out_col++) {
			auto file_col = column_ids[out_col];
			if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
				continue;
			}
			auto &vec = output.data[out_col];
			if (bind_data.include_dosage && file_col == bind_data.imp_r2_col_idx) {
				if (sex_aware || !bind_data.c.file_has_dosage) {
					FlatVector::SetNull(vec, rows_emitted, true);
				} else {
					FlatVector::GetData<double>(vec)[rows_emitted] = imp_r2;
				}
				continue;
			}
			if (FillVariantMetadataColumn(variants, file_col, vidx, vec, rows_emitted)) {
				continue;
			}
			auto put_count = [&](int32_t v) {
				if (counts_are_null) {
					FlatVector::SetNull(vec, rows_emitted, true);
				} else {
					FlatVector::GetData<int32_t>(vec)[rows_emitted] = v;
				}
			};
			switch (file_col) {
			case COL_ALT_FREQ:
				if (freq_is_null) {
					FlatVector::SetNull(vec, rows_emitted, true);
				} else {
					FlatVector::GetData<double>(vec)[rows_emitted] = alt_freq;
				}
				break;
			case COL_OBS_CT:
				FlatVector::GetData<int32_t>(vec)[rows_emitted] = static_cast<int32_t>(obs_ct);
				break;
			case COL_HOM_REF_CT:
				put_count(out_hom_ref);
				break;
			case COL_HET_CT:
				put_count(out_het);
				break;
			case COL_HOM_ALT_CT:
				put_count(out_hom_alt);
				break;
			case COL_MISSING_CT:
				put_count(out_missing);
				break;
			default:
				break;
			}
		}
		rows_emitted++;
	}
	CompatSetOutputCardinality(output, rows_emitted);
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

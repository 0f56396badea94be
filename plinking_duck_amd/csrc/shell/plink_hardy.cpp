// plink_hardy.cpp -- plink_hardy(path, pvar, psam, samples, region, midp, build)
//
// Surface of the reference's src/plink_hardy.cpp; counts and the autosomal exact
// tests come from the range's tally pass (the device runs each batch's tests behind
// its tally, variant_scan.hpp), the chrX tests of a device batch from one more
// launch (pgh_hwe_xchr_lnp_batch) -- the replacements for one plink2::HweLnP /
// HweXchrLnP call per variant.
#include "variant_scan.hpp"

#include <cmath>
#include <limits>

namespace duckdb {

// CHROM(0) POS(1) ID(2) REF(3) ALT(4) A1(5) HOM_REF_CT(6) HET_CT(7) HOM_ALT_CT(8) O_HET(9) E_HET(10) P_HWE(11)
static constexpr idx_t COL_A1 = 5;
static constexpr idx_t COL_HOM_REF_CT = 6;
static constexpr idx_t COL_HET_CT = 7;
static constexpr idx_t COL_HOM_ALT_CT = 8;
static constexpr idx_t COL_O_HET = 9;
static constexpr idx_t COL_E_HET = 10;
static constexpr idx_t COL_P_HWE = 11;

//! src/plink_hardy.cpp:52-64
static double LnPToPvalue(double ln_p) {
	if (std::isnan(ln_p)) {
		return 1.0;
	}
	double p = std::exp(ln_p);
	return p < 0.0 ? 0.0 : (p > 1.0 ? 1.0 : p);
}

struct PlinkHardyBindData : public TableFunctionData {
	PgenBindCommon c;
	bool midp = false;
	ParBounds par_bounds;
	PloidyMap ploidy;
	bool have_sex = false;
};

struct PlinkHardyGlobalState : public GlobalTableFunctionState {
	VariantScanGlobal scan;
	vector<column_t> column_ids;
	bool need_genotype_counts = false;
	bool need_p_hwe = false;
	uint32_t max_threads_config = 0;
	idx_t MaxThreads() const override {
		uint32_t range = scan.end_variant_idx - scan.start_variant_idx;
		return ApplyMaxThreadsCap(range / 500 + 1, max_threads_config);
	}
};

struct PlinkHardyLocalState : public LocalTableFunctionState {
	VariantScanLocal scan;
	// ln p of the chrX rows of the claimed device batch, by (variant - batch_begin); NaN elsewhere
	uint32_t x_batch_begin = UINT32_MAX;
	vector<double> x_lnp;
};

static unique_ptr<FunctionData> PlinkHardyBind(ClientContext &context, TableFunctionBindInput &input,
                                               vector<LogicalType> &return_types, vector<string> &names) {
	auto bind_data = make_uniq<PlinkHardyBindData>();
	string build_str = "GRCh38";
	for (auto &kv : input.named_parameters) {
		if (kv.first == "midp") {
			bind_data->midp = kv.second.GetValue<bool>();
		} else if (kv.first == "build") {
			build_str = kv.second.GetValue<string>();
		}
	}
	bind_data->par_bounds = ResolveParBounds(build_str, "plink_hardy");
	bind_data->c.Bind(context, input, "plink_hardy", false);
	bind_data->have_sex = bind_data->c.has_sample_info && !bind_data->c.sample_info().sexes.empty();
	bind_data->ploidy = PloidyMap(bind_data->c.variants, bind_data->par_bounds);
	names = {"CHROM", "POS", "ID", "REF", "ALT", "A1", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "O_HET", "E_HET", "P_HWE"};
	return_types = {LogicalType::VARCHAR, LogicalType::INTEGER, LogicalType::VARCHAR, LogicalType::VARCHAR,
	                LogicalType::VARCHAR, LogicalType::VARCHAR, LogicalType::INTEGER, LogicalType::INTEGER,
	                LogicalType::INTEGER, LogicalType::DOUBLE,  LogicalType::DOUBLE,  LogicalType::DOUBLE};
	return std::move(bind_data);
}

static unique_ptr<GlobalTableFunctionState> PlinkHardyInitGlobal(ClientContext &context,
                                                                 TableFunctionInitInput &input) {
	auto &bind_data = input.bind_data->Cast<PlinkHardyBindData>();
	auto state = make_uniq<PlinkHardyGlobalState>();
	state->scan.start_variant_idx = bind_data.c.RangeStart();
	state->scan.end_variant_idx = bind_data.c.RangeEnd();
	state->scan.next_variant_idx.store(state->scan.start_variant_idx);
	state->scan.effective_sample_ct = bind_data.c.effective_sample_ct;
	state->column_ids = input.column_ids;
	state->max_threads_config = GetPlinkingMaxThreads(context);
	for (auto col_id : input.column_ids) {
		if (col_id != COLUMN_IDENTIFIER_ROW_ID && col_id >= COL_HOM_REF_CT && col_id <= COL_P_HWE) {
			state->need_genotype_counts = true;
		}
		state->need_p_hwe |= col_id == COL_P_HWE;
	}
	if (state->need_genotype_counts) {
		state->scan.dataset = DeviceDataset::Acquire(bind_data.c.pgen_path, "plink_hardy");
		if (bind_data.c.has_sample_subset) {
			state->scan.subset =
			    make_uniq<DeviceSubset>(*state->scan.dataset, bind_data.c.sample_subset->sample_include, "plink_hardy");
		}
		uint32_t products = bind_data.c.has_sample_subset ? 0u : static_cast<uint32_t>(PGH_TALLY_SAMPLE_MISSING);
		if (state->need_p_hwe) {
			products |= bind_data.midp ? PGH_TALLY_HWE_MIDP : PGH_TALLY_HWE;
		}
		state->scan.StartTallies(bind_data.c.sample_subset.get(), bind_data.have_sex ? &bind_data.c.sample_info() : nullptr,
		                         bind_data.c.raw_sample_ct, &bind_data.ploidy, products, false,
		                         GetPlinkingTallyCache(context), "plink_hardy");
	}
	return std::move(state);
}

static unique_ptr<LocalTableFunctionState> PlinkHardyInitLocal(ExecutionContext &, TableFunctionInitInput &,
                                                               GlobalTableFunctionState *) {
	return make_uniq<PlinkHardyLocalState>();
}

//! The chrX exact tests of the batch the thread has just claimed (females' genotypes + males' alleles), in one
//! launch; the autosomal rule's ln p of every row is already in the tally pass.
static void PrepareXchrTests(const PlinkHardyBindData &bind_data, PlinkHardyLocalState &lstate) {
	auto &scan = lstate.scan;
	const uint32_t n = scan.batch_end - scan.batch_begin;
	lstate.x_batch_begin = scan.batch_begin;
	lstate.x_lnp.assign(n, std::numeric_limits<double>::quiet_NaN());
	if (!scan.have_strata) {
		return;
	}
	vector<int32_t> strata;
	vector<uint32_t> where;
	for (uint32_t v = scan.batch_begin; v < scan.batch_end; v++) {
		const ChromPloidy ploidy = bind_data.ploidy.At(v);
		if (ploidy == ChromPloidy::AUTOSOMAL) {
			continue;
		}
		SexAwareCounts sac = SexAwareFromStrata(ploidy, scan.Counts(v), scan.MaleCounts(v), scan.FemaleCounts(v),
		                                        bind_data.have_sex);
		if (sac.sex_unavailable || !sac.hwe_defined) {
			continue;
		}
		// males contribute only to geno_hom_* (het -> missing), females to both
		where.push_back(v - scan.batch_begin);
		strata.push_back(static_cast<int32_t>(sac.hwe_het));
		strata.push_back(static_cast<int32_t>(sac.hwe_hom_ref));
		strata.push_back(static_cast<int32_t>(sac.hwe_hom_alt));
		strata.push_back(static_cast<int32_t>(sac.geno_hom_ref) - static_cast<int32_t>(sac.hwe_hom_ref));
		strata.push_back(static_cast<int32_t>(sac.geno_hom_alt) - static_cast<int32_t>(sac.hwe_hom_alt));
	}
	if (where.empty()) {
		return;
	}
	vector<double> got(where.size());
	char errbuf[PGH_ERRBUF_LEN] = {0};
	if (pgh_hwe_xchr_lnp_batch(reinterpret_cast<const int32_t(*)[5]>(strata.data()), static_cast<uint32_t>(where.size()),
	                           bind_data.midp ? 1u : 0u, got.data(), errbuf) != PGH_OK) {
		throw IOException("plink_hardy: chrX exact tests failed for variants [%u, %u): %s", scan.batch_begin,
		                  scan.batch_end, string(errbuf));
	}
	for (size_t k = 0; k < where.size(); k++) {
		lstate.x_lnp[where[k]] = got[k];
	}
}

static void PlinkHardyScan(ClientContext &, TableFunctionInput &data_p, DataChunk &output) {
	auto &bind_data = data_p.bind_data->Cast<PlinkHardyBindData>();
	auto &gstate = data_p.global_state->Cast<PlinkHardyGlobalState>();
	auto &lstate = data_p.local_state->Cast<PlinkHardyLocalState>();
	auto &column_ids = gstate.column_ids;
	auto &variants = bind_data.c.variants;

	// The chunk's rows first (counts of the HWE-test stratum, heterozygosities, the batch's exact
	// test), then one loop per projected column.
	struct HardyRow {
		uint32_t vidx;
		int32_t counts[3] = {0, 0, 0}; // HOM_REF_CT, HET_CT, HOM_ALT_CT of the tested stratum
		double stats[3] = {0.0, 0.0, 1.0}; // O_HET, E_HET, P_HWE
		bool counts_are_null = false, stats_are_null = true;
	};
	vector<HardyRow> rows;
	rows.reserve(STANDARD_VECTOR_SIZE);
	uint32_t vidx;
	while (rows.size() < STANDARD_VECTOR_SIZE && lstate.scan.Next(gstate.scan, "plink_hardy", vidx)) {
		HardyRow row;
		row.vidx = vidx;
		if (!gstate.need_genotype_counts) {
			rows.push_back(row);
			continue;
		}
		if (gstate.need_p_hwe && lstate.scan.have_strata && lstate.x_batch_begin != lstate.scan.batch_begin) {
			PrepareXchrTests(bind_data, lstate);
		}
		const ChromPloidy ploidy = bind_data.ploidy.At(vidx);
		double ln_p = 0.0;
		if (gstate.need_p_hwe) {
			ln_p = ploidy == ChromPloidy::AUTOSOMAL || !lstate.scan.have_strata
			           ? lstate.scan.LnP(vidx, bind_data.midp)
			           : lstate.x_lnp[vidx - lstate.scan.batch_begin];
		}
		const uint32_t *gc = lstate.scan.Counts(vidx);
		// O_HET / E_HET of a diploid stratum (src/plink_hardy.cpp:575-588)
		auto diploid_stats = [&](uint32_t hom_ref, uint32_t het, uint32_t hom_alt, bool test_ok) {
			const uint32_t obs = hom_ref + het + hom_alt;
			row.counts[0] = static_cast<int32_t>(hom_ref);
			row.counts[1] = static_cast<int32_t>(het);
			row.counts[2] = static_cast<int32_t>(hom_alt);
			row.stats_are_null = obs == 0;
			if (obs) {
				const double p = (2.0 * hom_ref + het) / (2.0 * obs);
				row.stats[0] = static_cast<double>(het) / static_cast<double>(obs);
				row.stats[1] = 2.0 * p * (1.0 - p);
				row.stats[2] = test_ok ? LnPToPvalue(ln_p) : 1.0;
			}
		};
		if (ploidy == ChromPloidy::AUTOSOMAL) {
			diploid_stats(gc[0], gc[1], gc[2], true);
		} else {
			SexAwareCounts sac = SexAwareFromStrata(ploidy, gc, lstate.scan.MaleCounts(vidx),
			                                        lstate.scan.FemaleCounts(vidx), bind_data.have_sex);
			if (sac.sex_unavailable) {
				row.counts_are_null = true;
			} else if (sac.hwe_defined) {
				// chrX: the females are the tested stratum; the males enter the exact test as allele counts
				// (geno_hom_* minus the females'), which must not come out negative
				const bool males_ok = sac.geno_hom_ref >= sac.hwe_hom_ref && sac.geno_hom_alt >= sac.hwe_hom_alt;
				diploid_stats(sac.hwe_hom_ref, sac.hwe_het, sac.hwe_hom_alt, males_ok);
			} else {
				// chrY / chrMT: haploid carriers, HET_CT = 0, no test
				row.counts[0] = static_cast<int32_t>(sac.geno_hom_ref);
				row.counts[1] = static_cast<int32_t>(sac.geno_het);
				row.counts[2] = static_cast<int32_t>(sac.geno_hom_alt);
			}
		}
		rows.push_back(row);
	}

	const idx_t n_rows = rows.size();
	for (idx_t out_col = 0; out_col < column_ids.size(); out_col++) {
		const auto file_col = column_ids[out_col];
		if (file_col == COLUMN_IDENTIFIER_ROW_ID) {
			continue;
		}
		auto &vec = output.data[out_col];
		if (file_col <= COL_A1) {
			const idx_t source = file_col == COL_A1 ? 4 : file_col; // the tested allele A1 is ALT
			for (idx_t r = 0; r < n_rows; r++) {
				FillVariantMetadataColumn(variants, source, rows[r].vidx, vec, r);
			}
		} else if (file_col <= COL_HOM_ALT_CT) {
			auto *dst = FlatVector::GetData<int32_t>(vec);
			for (idx_t r = 0; r < n_rows; r++) {
				if (rows[r].counts_are_null) {
					FlatVector::SetNull(vec, r, true);
				} else {
					dst[r] = rows[r].counts[file_col - COL_HOM_REF_CT];
				}
			}
		} else if (file_col <= COL_P_HWE) {
			auto *dst = FlatVector::GetData<double>(vec);
			for (idx_t r = 0; r < n_rows; r++) {
				if (rows[r].stats_are_null) {
					FlatVector::SetNull(vec, r, true);
				} else {
					dst[r] = rows[r].stats[file_col - COL_O_HET];
				}
			}
		}
	}
	CompatSetOutputCardinality(output, n_rows);
}

void RegisterPlinkHardy(ExtensionLoader &loader) {
	TableFunction plink_hardy("plink_hardy", {LogicalType::VARCHAR}, PlinkHardyScan, PlinkHardyBind,
	                          PlinkHardyInitGlobal, PlinkHardyInitLocal);
	plink_hardy.projection_pushdown = true;
	plink_hardy.named_parameters["pvar"] = LogicalType::VARCHAR;
	plink_hardy.named_parameters["psam"] = LogicalType::VARCHAR;
	plink_hardy.named_parameters["samples"] = LogicalType::ANY;
	plink_hardy.named_parameters["region"] = LogicalType::VARCHAR;
	plink_hardy.named_parameters["midp"] = LogicalType::BOOLEAN;
	plink_hardy.named_parameters["build"] = LogicalType::VARCHAR;
	loader.RegisterFunction(plink_hardy);
}

} // namespace duckdb

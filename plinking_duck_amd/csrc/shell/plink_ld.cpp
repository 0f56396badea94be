// plink_ld.cpp -- plink_ld(path, pvar, psam, variant1, variant2, window_kb, r2_threshold,
//                          region, samples, inter_chr)
//
// Surface of the reference's src/plink_ld.cpp: pairwise mode (one row for variant1 x
// variant2) and windowed mode (every anchor against the later variants of its chromosome
// within window_kb, optionally every later variant of other chromosomes; rows with
// r2 >= r2_threshold).  The reference reads both genovecs with PgrGet and walks the samples
// with doubles (ComputeLdStats, src/plink_ld.cpp:52-134); here a scan thread claims a run
// of anchors, lists their partner pairs with the reference's window walk, and ONE
// pgh_ld_pairs launch returns the exact integer sums of all of them, to which the
// reference's double arithmetic is then applied.
#include "variant_scan.hpp"

#include <cmath>
#include <deque>

namespace duckdb {

static constexpr idx_t COL_CHROM_A = 0;
static constexpr idx_t COL_POS_A = 1;
static constexpr idx_t COL_ID_A = 2;
static constexpr idx_t COL_CHROM_B = 3;
static constexpr idx_t COL_POS_B = 4;
static constexpr idx_t COL_ID_B = 5;
static constexpr idx_t COL_R2 = 6;
static constexpr idx_t COL_D_PRIME = 7;
static constexpr idx_t COL_OBS_CT = 8;

static constexpr uint32_t kAnchorsPerClaim = 16;   // the reference claims one anchor at a time
static constexpr size_t kPairsPerLaunch = 1u << 17; // soft cap; an anchor's list is never split

enum class LdMode : uint8_t { PAIRWISE, WINDOWED };

struct LdResult {
	double r2 = 0;
	double d_prime = 0;
	uint32_t obs_ct = 0;
	bool is_valid = false; // false if monomorphic, < 2 observations
};

//! src/plink_ld.cpp:86-134 on the sums of src/plink_ld.cpp:52-84
static LdResult LdFromSums(const uint32_t s[6]) {
	LdResult result;
	const uint32_t n = s[0];
	result.obs_ct = n;
	if (n < 2) {
		return result;
	}
	const double sum_a = s[1], sum_b = s[2], sum_ab = s[3], sum_a2 = s[4], sum_b2 = s[5];
	const double dn = static_cast<double>(n);
	const double mean_a = sum_a / dn;
	const double mean_b = sum_b / dn;
	const double cov_ab = sum_ab / dn - mean_a * mean_b;
	const double var_a = sum_a2 / dn - mean_a * mean_a;
	const double var_b = sum_b2 / dn - mean_b * mean_b;
	if (var_a < 1e-15 || var_b < 1e-15) {
		return result; // monomorphic: correlation undefined
	}
	result.is_valid = true;
	result.r2 = (cov_ab * cov_ab) / (var_a * var_b);
	// composite estimator (Weir 1979): D = cov / 4, D' = D / D_max by the sign of D
	const double D = cov_ab / 4.0;
	const double p_a = sum_a / (2.0 * dn);
	const double p_b = sum_b / (2.0 * dn);
	double D_max;
	if (D >= 0) {
		D_max = std::min(p_a * (1.0 - p_b), (1.0 - p_a) * p_b);
	} else {
		D_max = std::max(-p_a * p_b, -(1.0 - p_a) * (1.0 - p_b));
	}
	result.d_prime = std::abs(D_max) < 1e-15 ? 0.0 : D / D_max;
	return result;
}

struct PlinkLdBindData : public TableFunctionData {
	PgenBindCommon c;
	LdMode mode = LdMode::WINDOWED;
	uint32_t pairwise_vidx_a = 0;
	uint32_t pairwise_vidx_b = 0;
	int64_t window_bp = 1000000;
	double r2_threshold = 0.2;
	bool inter_chr = false;
};

struct PlinkLdGlobalState : public GlobalTableFunctionState {
	LdMode mode = LdMode::WINDOWED;
	uint32_t start_variant_idx = 0;
	uint32_t end_variant_idx = 0;
	std::atomic<bool> pair_emitted {false};
	std::atomic<uint32_t> next_anchor_idx {0};
	uint32_t max_threads_config = 0;
	shared_ptr<DeviceDataset> dataset;
	unique_ptr<DeviceSubset> subset;
	RowWindows row_windows; // a file beyond the HBM budget: the window that holds a call's pairs (LeaseRows)

	idx_t MaxThreads() const override {
		if (mode == LdMode::PAIRWISE) {
			return 1;
		}
		uint32_t range = end_variant_idx - start_variant_idx;
		return ApplyMaxThreadsCap(range / 50 + 1, max_threads_config);
	}
};

struct PendingRow {
	uint32_t vidx_a, vidx_b;
	LdResult result;
};

struct PlinkLdLocalState : public LocalTableFunctionState {
	std::deque<PendingRow> pending; // rows that passed the threshold, in (anchor, partner) order
	vector<uint32_t> pair_a, pair_b;
	vector<uint32_t> sums;
};

static unique_ptr<FunctionData> PlinkLdBind(ClientContext &context, TableFunctionBindInput &input,
                                            vector<LogicalType> &return_types, vector<string> &names) {
	auto bind_data = make_uniq<PlinkLdBindData>();
	string variant1_id, variant2_id;
	for (auto &kv : input.named_parameters) {
		if (kv.first == "variant1") {
			variant1_id = kv.second.GetValue<string>();
		} else if (kv.first == "variant2") {
			variant2_id = kv.second.GetValue<string>();
		} else if (kv.first == "window_kb") {
			auto kb = kv.second.GetValue<int64_t>();
			if (kb < 0) {
				throw InvalidInputException("plink_ld: window_kb must be non-negative");
			}
			bind_data->window_bp = kb * 1000;
		} else if (kv.first == "r2_threshold") {
			bind_data->r2_threshold = kv.second.GetValue<double>();
			if (bind_data->r2_threshold < 0.0 || bind_data->r2_threshold > 1.0) {
				throw InvalidInputException("plink_ld: r2_threshold must be between 0.0 and 1.0");
			}
		} else if (kv.first == "inter_chr") {
			bind_data->inter_chr = kv.second.GetValue<bool>();
		}
	}
	if (!variant1_id.empty() && !variant2_id.empty()) {
		bind_data->mode = LdMode::PAIRWISE;
	} else if (!variant1_id.empty() || !variant2_id.empty()) {
		throw InvalidInputException("plink_ld: both variant1 and variant2 must be specified for pairwise mode");
	}
	auto &c = bind_data->c;
	c.Bind(context, input, "plink_ld", false);
	if (bind_data->mode == LdMode::PAIRWISE) {
		// ID -> index, the later of two equal IDs winning (src/plink_common.cpp:1598-1612)
		auto find_id = [&](const string &id) {
			for (idx_t v = c.variants.ids().size(); v-- > 0;) {
				if (c.variants.ids()[v] == id) {
					return static_cast<uint32_t>(v);
				}
			}
			throw InvalidInputException("plink_ld: variant '%s' not found in .pvar", id);
		};
		bind_data->pairwise_vidx_a = find_id(variant1_id);
		bind_data->pairwise_vidx_b = find_id(variant2_id);
	}
	names = {"CHROM_A", "POS_A", "ID_A", "CHROM_B", "POS_B", "ID_B", "R2", "D_PRIME", "OBS_CT"};
	return_types = {LogicalType::VARCHAR, LogicalType::INTEGER, LogicalType::VARCHAR,
	                LogicalType::VARCHAR, LogicalType::INTEGER, LogicalType::VARCHAR,
	                LogicalType::DOUBLE,  LogicalType::DOUBLE,  LogicalType::INTEGER};
	return std::move(bind_data);
}

static unique_ptr<GlobalTableFunctionState> PlinkLdInitGlobal(ClientContext &context, TableFunctionInitInput &input) {
	auto &bind_data = input.bind_data->Cast<PlinkLdBindData>();
	auto state = make_uniq<PlinkLdGlobalState>();
	state->mode = bind_data.mode;
	state->max_threads_config = GetPlinkingMaxThreads(context);
	state->start_variant_idx = bind_data.c.RangeStart();
	state->end_variant_idx = bind_data.c.RangeEnd();
	state->next_anchor_idx.store(state->start_variant_idx);
	const bool any_pair = bind_data.mode == LdMode::PAIRWISE || state->end_variant_idx - state->start_variant_idx >= 2;
	if (any_pair) {
		state->dataset = DeviceDataset::Acquire(bind_data.c.pgen_path, "plink_ld");
		if (bind_data.c.has_sample_subset) {
			state->subset =
			    make_uniq<DeviceSubset>(*state->dataset, bind_data.c.sample_subset->sample_include, "plink_ld");
		}
	}
	return std::move(state);
}

static unique_ptr<LocalTableFunctionState> PlinkLdInitLocal(ExecutionContext &, TableFunctionInitInput &,
                                                            GlobalTableFunctionState *) {
	return make_uniq<PlinkLdLocalState>();
}

static void EmitRow(DataChunk &output, idx_t row_idx, const PlinkLdBindData &bind_data, const PendingRow &row) {
	auto &variants = bind_data.c.variants;
	auto put_text = [&](idx_t col, const string &text, bool null_if_empty) {
		if (null_if_empty && text.empty()) {
			FlatVector::SetNull(output.data[col], row_idx, true);
		} else {
			FlatVector::GetData<string_t>(output.data[col])[row_idx] = StringVector::AddString(output.data[col], text);
		}
	};
	put_text(COL_CHROM_A, variants.GetChrom(row.vidx_a), false);
	FlatVector::GetData<int32_t>(output.data[COL_POS_A])[row_idx] = variants.GetPos(row.vidx_a);
	put_text(COL_ID_A, variants.GetId(row.vidx_a), true);
	put_text(COL_CHROM_B, variants.GetChrom(row.vidx_b), false);
	FlatVector::GetData<int32_t>(output.data[COL_POS_B])[row_idx] = variants.GetPos(row.vidx_b);
	put_text(COL_ID_B, variants.GetId(row.vidx_b), true);
	if (row.result.is_valid) {
		FlatVector::GetData<double>(output.data[COL_R2])[row_idx] = row.result.r2;
		FlatVector::GetData<double>(output.data[COL_D_PRIME])[row_idx] = row.result.d_prime;
	} else {
		FlatVector::SetNull(output.data[COL_R2], row_idx, true);
		FlatVector::SetNull(output.data[COL_D_PRIME], row_idx, true);
	}
	FlatVector::GetData<int32_t>(output.data[COL_OBS_CT])[row_idx] = static_cast<int32_t>(row.result.obs_ct);
}

//! The partners of one anchor, in the reference's walk order (src/plink_ld.cpp:617-660).
static void ListPartners(const PlinkLdBindData &bind_data, uint32_t anchor, uint32_t end_idx, vector<uint32_t> &pair_a,
                         vector<uint32_t> &pair_b) {
	auto &variants = bind_data.c.variants;
	const string &anchor_chrom = variants.GetChrom(anchor);
	const int32_t anchor_pos = variants.GetPos(anchor);
	uint32_t j = anchor + 1;
	while (j < end_idx) {
		if (variants.GetChrom(j) == anchor_chrom) {
			int64_t dist = static_cast<int64_t>(variants.GetPos(j)) - static_cast<int64_t>(anchor_pos);
			if (dist > bind_data.window_bp) {
				if (!bind_data.inter_chr) {
					break; // past the window, and other chromosomes are not wanted
				}
				while (j < end_idx && variants.GetChrom(j) == anchor_chrom) {
					j++;
				}
				continue;
			}
		} else if (!bind_data.inter_chr) {
			break; // past the chromosome boundary
		}
		pair_a.push_back(anchor);
		pair_b.push_back(j);
		j++;
	}
}

static void RunPairs(const PlinkLdBindData &bind_data, PlinkLdGlobalState &gstate, PlinkLdLocalState &lstate) {
	const size_t n = lstate.pair_a.size();
	lstate.sums.resize(6 * n);
	if (n == 0) {
		return;
	}
	// Both rows of a pair have to be resident together.  A file beyond the HBM budget keeps one window of its rows
	// resident at a time: the one that holds every variant this call names -- a claim's anchors and their partners
	// inside the LD window, so a few thousand neighbours -- or, when the pairs reach further than a window (inter_chr,
	// a far-apart variant1 / variant2), nothing: that does not fit.
	uint32_t lo = lstate.pair_a[0], hi = lstate.pair_a[0];
	for (size_t i = 0; i < n; i++) {
		lo = std::min({lo, lstate.pair_a[i], lstate.pair_b[i]});
		hi = std::max({hi, lstate.pair_a[i], lstate.pair_b[i]});
	}
	if (gstate.dataset->streamed && static_cast<uint64_t>(hi) + 1 - lo > gstate.dataset->WindowVariants() && n == 1) {
		// variant1 / variant2 far apart in a file beyond the HBM budget: each row through a one-variant window of its
		// own, the two of them side by side in a scratch dataset of two rows
		const uint32_t rec = (bind_data.c.raw_sample_ct + 3) / 4;
		vector<uint8_t> two(2 * static_cast<size_t>(rec));
		const vector<uint64_t> *mask = bind_data.c.has_sample_subset ? &bind_data.c.sample_subset->sample_include : nullptr;
		char errbuf[PGH_ERRBUF_LEN] = {0};
		const uint32_t want[2] = {lstate.pair_a[0], lstate.pair_b[0]};
		for (int k = 0; k < 2; k++) {
			RowLease one = LeaseRows(*gstate.dataset, nullptr, gstate.row_windows, nullptr, want[k], want[k] + 1,
			                         bind_data.c.raw_variant_ct, "plink_ld");
			if (pgh_copy_rows_to_host(one.ds, want[k], want[k] + 1, two.data() + static_cast<size_t>(k) * rec, rec, errbuf) !=
			    PGH_OK) {
				throw IOException("plink_ld: PgrGet failed for variant %u: %s", want[k], string(errbuf));
			}
		}
		pgh_dataset *scratch = nullptr;
		pgh_subset *ss = nullptr;
		int rc = pgh_from_host_rows(two.data(), rec, 2, bind_data.c.raw_sample_ct, &scratch, errbuf);
		if (rc == PGH_OK && mask) {
			rc = pgh_subset_create(scratch, mask->data(), &ss, errbuf);
		}
		const uint32_t a = 0, b = 1;
		if (rc == PGH_OK) {
			rc = pgh_ld_pairs(scratch, ss, 1, &a, &b, reinterpret_cast<uint32_t(*)[6]>(lstate.sums.data()), errbuf);
		}
		pgh_subset_destroy(ss);
		if (scratch) {
			pgh_close(scratch);
		}
		if (rc != PGH_OK) {
			throw IOException("plink_ld: PgrGet failed for variants %u..%u: %s", want[0], want[1], string(errbuf));
		}
		return;
	}
	if (gstate.dataset->streamed && static_cast<uint64_t>(hi) + 1 - lo > gstate.dataset->WindowVariants()) {
		throw IOException("plink_ld: '%s' does not fit the HBM budget, and the pairs over variants %u..%u reach further than "
		                  "one window of it (%llu variants: PLINKING_HBM_CACHE_GB) -- both rows of a pair must be resident "
		                  "together",
		                  gstate.dataset->path, lo, hi, static_cast<unsigned long long>(gstate.dataset->WindowVariants()));
	}
	RowLease rows = LeaseRows(*gstate.dataset, gstate.subset.get(), gstate.row_windows,
	                          bind_data.c.has_sample_subset ? &bind_data.c.sample_subset->sample_include : nullptr, lo, hi + 1,
	                          bind_data.c.raw_variant_ct, "plink_ld");
	char errbuf[PGH_ERRBUF_LEN] = {0};
	int rc = pgh_ld_pairs(rows.ds, rows.ss,
	                      static_cast<uint32_t>(n), lstate.pair_a.data(), lstate.pair_b.data(),
	                      reinterpret_cast<uint32_t(*)[6]>(lstate.sums.data()), errbuf);
	if (rc != PGH_OK) {
		throw IOException("plink_ld: PgrGet failed for variants %u..%u: %s", n ? lstate.pair_a.front() : 0u,
		                  n ? lstate.pair_b.back() : 0u, string(errbuf));
	}
}

static void PlinkLdScan(ClientContext &, TableFunctionInput &data_p, DataChunk &output) {
	auto &bind_data = data_p.bind_data->Cast<PlinkLdBindData>();
	auto &gstate = data_p.global_state->Cast<PlinkLdGlobalState>();
	auto &lstate = data_p.local_state->Cast<PlinkLdLocalState>();

	if (gstate.mode == LdMode::PAIRWISE) {
		if (gstate.pair_emitted.exchange(true)) {
			CompatSetOutputCardinality(output, 0);
			return;
		}
		lstate.pair_a.assign(1, bind_data.pairwise_vidx_a);
		lstate.pair_b.assign(1, bind_data.pairwise_vidx_b);
		RunPairs(bind_data, gstate, lstate);
		EmitRow(output, 0, bind_data,
		        PendingRow {bind_data.pairwise_vidx_a, bind_data.pairwise_vidx_b, LdFromSums(lstate.sums.data())});
		CompatSetOutputCardinality(output, 1);
		return;
	}

	const uint32_t end_idx = gstate.end_variant_idx;
	idx_t rows_emitted = 0;
	while (rows_emitted < STANDARD_VECTOR_SIZE) {
		if (lstate.pending.empty()) {
			// claim anchors until there is a launch's worth of pairs (or the range is drained)
			lstate.pair_a.clear();
			lstate.pair_b.clear();
			bool drained = false;
			while (lstate.pair_a.size() < kPairsPerLaunch) {
				uint32_t first = gstate.next_anchor_idx.fetch_add(kAnchorsPerClaim);
				if (first >= end_idx) {
					drained = true;
					break;
				}
				uint32_t last = std::min<uint64_t>(end_idx, static_cast<uint64_t>(first) + kAnchorsPerClaim);
				for (uint32_t anchor = first; anchor < last; anchor++) {
					ListPartners(bind_data, anchor, end_idx, lstate.pair_a, lstate.pair_b);
				}
			}
			if (lstate.pair_a.empty()) {
				if (drained) {
					break;
				}
				continue;
			}
			RunPairs(bind_data, gstate, lstate);
			for (size_t p = 0; p < lstate.pair_a.size(); p++) {
				LdResult result = LdFromSums(lstate.sums.data() + 6 * p);
				if (result.is_valid && result.r2 >= bind_data.r2_threshold) {
					lstate.pending.push_back(PendingRow {lstate.pair_a[p], lstate.pair_b[p], result});
				}
			}
			continue;
		}
		EmitRow(output, rows_emitted++, bind_data, lstate.pending.front());
		lstate.pending.pop_front();
	}
	CompatSetOutputCardinality(output, rows_emitted);
}

void RegisterPlinkLd(ExtensionLoader &loader) {
	TableFunction plink_ld("plink_ld", {LogicalType::VARCHAR}, PlinkLdScan, PlinkLdBind, PlinkLdInitGlobal,
	                       PlinkLdInitLocal);
	plink_ld.named_parameters["pvar"] = LogicalType::VARCHAR;
	plink_ld.named_parameters["psam"] = LogicalType::VARCHAR;
	plink_ld.named_parameters["variant1"] = LogicalType::VARCHAR;
	plink_ld.named_parameters["variant2"] = LogicalType::VARCHAR;
	plink_ld.named_parameters["window_kb"] = LogicalType::INTEGER;
	plink_ld.named_parameters["r2_threshold"] = LogicalType::DOUBLE;
	plink_ld.named_parameters["region"] = LogicalType::VARCHAR;
	plink_ld.named_parameters["samples"] = LogicalType::ANY;
	plink_ld.named_parameters["inter_chr"] = LogicalType::BOOLEAN;
	loader.RegisterFunction(plink_ld);
}

} // namespace duckdb

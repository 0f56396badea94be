// pgen_reader.hpp -- the variant-oriented genotype scan, shared by read_pgen and by
// read_pfile's orient := 'variant' (pfile_reader.cpp), which differ only in how the three
// file paths are found and in the name that error messages carry.
#pragma once

#include "variant_scan.hpp"

#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace duckdb {

static constexpr idx_t COL_GENOTYPES = 5;
static constexpr uint32_t kUnpackSpan = 2048; // variants unpacked per launch (one output vector)

struct PgenBindData : public TableFunctionData {
	string func = "read_pgen"; // name in error messages
	PgenBindCommon c;
	bool include_dosages = false;
	bool include_phased = false;
	GenotypeMode genotype_mode = GenotypeMode::ARRAY;
	CountFilter count_filter;
	GenotypeRangeFilter genotype_filter;
	uint32_t output_sample_ct = 0;
	bool has_variant_list = false;
	vector<uint32_t> variant_indices; // `variants :=`, caller order
	vector<string> genotype_column_names; // columns / struct modes: IIDs in ascending file order
};

struct PgenGlobalState : public GlobalTableFunctionState {
	VariantScanGlobal scan;
	vector<column_t> column_ids;
	bool need_genotypes = false;
	bool need_counts = false;
	uint32_t max_threads_config = 0;
	idx_t MaxThreads() const override {
		uint32_t total = scan.has_variant_list ? static_cast<uint32_t>(scan.variant_list.size())
		                                       : scan.end_variant_idx - scan.start_variant_idx;
		return ApplyMaxThreadsCap(total / 1000 + 1, max_threads_config);
	}
};

struct RowPlan {
	uint32_t vidx;
	bool geno_range_all_pass;
};

struct PgenLocalState : public LocalTableFunctionState {
	VariantScanLocal scan;
	pgh_reader *reader = nullptr;
	uint64_t reader_window = 0; // streamed files: the window `reader` was made on (RowLease::window_id)
	PinnedBuffer<int8_t> bytes;      // unpacked span [rows][n_out]; page-locked: the device copies straight into it
	PinnedBuffer<uint64_t> validity; // [rows][ceil(n_out/64)]
	// The genotype-list pipeline (plain hardcalls over a variant range): while a Scan call fills its output vector
	// from chunk k, chunk k + 1 -- already planned -- is being unpacked and copied over the host link into the other
	// slot's pinned block (pgh_reader_unpack_start / _wait).
	struct ChunkSlot {
		vector<RowPlan> plan;
		PinnedBuffer<int8_t> bytes;
		PinnedBuffer<uint64_t> validity;
	};
	ChunkSlot slot[2];
	int cur = 0;
	bool primed = false;
	vector<double> dosage_doubles;
	vector<uint64_t> genovec, phasepresent, phaseinfo;
	// PLINKING_TIMING=1: where this thread's scan time went (printed when the thread's state goes)
	double plan_ms = 0.0, unpack_ms = 0.0, fill_ms = 0.0;
	uint64_t chunks = 0;
	~PgenLocalState() override {
		if (reader) {
			pgh_reader_destroy(reader);
		}
		if (chunks && std::getenv("PLINKING_TIMING")) {
			std::fprintf(stderr, "read_pgen thread: %llu chunks, plan %.1f ms, unpack %.1f ms, fill %.1f ms\n",
			             static_cast<unsigned long long>(chunks), plan_ms, unpack_ms, fill_ms);
		}
	}
};

//! `func`: "read_pgen" or "read_pfile"; with_region: accept `region :=` (read_pfile only).
unique_ptr<FunctionData> PgenBindNamed(ClientContext &context, TableFunctionBindInput &input,
                                       vector<LogicalType> &return_types, vector<string> &names, const string &func,
                                       bool with_region);
unique_ptr<GlobalTableFunctionState> PgenInitGlobal(ClientContext &context, TableFunctionInitInput &input);
unique_ptr<LocalTableFunctionState> PgenInitLocal(ExecutionContext &context, TableFunctionInitInput &input,
                                                  GlobalTableFunctionState *global_state);
void PgenScan(ClientContext &context, TableFunctionInput &data_p, DataChunk &output);

} // namespace duckdb

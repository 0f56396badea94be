// phase.hpp -- hardcall phase tracks (vrtype bit 0x10) brought to a resident bit-array form at pgh_open.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pgh {

//! One staged run of records (decode.hpp:DecodeBatch) whose phase tracks are expanded.  All pointers are
//! device pointers.  A phase track lists one bit per HET sample, in sample order; PgrGetP hands it back as
//! two bit arrays over all samples (phasepresent, phaseinfo), and that is the form kept in HBM.
struct PhaseIngest {
	const uint8_t *bytes; // the records' file bytes (16 readable zero bytes follow bytes_len)
	uint64_t bytes_len;
	const uint64_t *rec_begin; // [n + 1]
	const uint8_t *vrtype;     // [n]
	const uint64_t *aux_at;    // [n] first byte after each record's main track
	const int32_t *ph_row;     // [n] row of the record in the arrays below, or -1: no phase track
	const uint8_t *rows;       // the finished 2-bit rows: the track is indexed by the hets
	uint64_t pitch;
	uint32_t row0, variant0, n, sample_ct;
	uint64_t *present; // rows x words: the sample's het call is phased
	uint64_t *info;    // rows x words: 1 = ALT allele first (meaningful where present)
	uint32_t words;    // ceil(sample_ct / 64)
	int *error;        // set (once) to 1 + the variant index of a malformed record
};

//! Largest sample count the device path takes (its two per-word prefix tables live in LDS).
uint32_t PhaseIngestMaxSamples();
hipError_t LaunchPhaseIngest(const PhaseIngest &batch, hipStream_t stream);

} // namespace pgh

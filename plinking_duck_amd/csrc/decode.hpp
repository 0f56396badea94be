// decode.hpp -- device-side expansion of .pgen variant records (definitions in decode.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace pgh {

// One staged run of consecutive records and where their rows go.  All pointers are
// device pointers.
struct DecodeBatch {
	const uint8_t *bytes;      // the records' file bytes, back to back; 16 readable zero bytes follow bytes_len
	uint64_t bytes_len;
	const uint64_t *rec_begin; // [n + 1] offsets into bytes
	const uint8_t *vrtype;     // [n] vrtype bytes from the file's index
	const uint32_t *ld_row;    // [n] row (relative to `rows`) of the LD base for types 2/3, else unused;
	                           //     0xffffffff = no base inside the resident range (flagged as an error)
	uint8_t *rows;             // first resident row
	uint64_t pitch;            // bytes between rows (multiple of 16)
	uint32_t row0;             // row of record 0
	uint32_t variant0;         // file variant index of record 0 (error reporting)
	uint32_t n;                // records in the batch
	uint32_t sample_ct;
	uint32_t id_bytes;         // width of a difflist sample id
	int *error;                // set (once) to 1 + the variant index of a malformed record
	uint64_t *aux_at;          // [n] or NULL: offset into bytes of the first byte after each record's main track
};

// Expands every record of the batch into its row: types 0/1/4/6/7 in one launch, the
// LD-compressed types 2/3 in a second one that reads the finished base rows.
hipError_t LaunchDecodeRecords(const DecodeBatch &batch, bool any_ld, hipStream_t stream);

} // namespace pgh

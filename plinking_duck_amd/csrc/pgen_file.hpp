// pgen_file.hpp -- host side of libpgenhip: .pgen header/index parser and the
// record normaliser that turns any variant record into a plain 2-bit row.
//
// Replaces what the reference gets from pgenlib's PgfiInitPhase1/2 + PgrInit
// (src/plink_freq.cpp:168-208, 344-390) and the record-type handling inside
// PgrGet* (pgenlib_read.cc, absent from the reference tree).  Format rules: the
// public PLINK 2 .pgen specification, as listed in SURVEY.md section 8c.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace pgh {

struct PgenIndex {
	uint32_t variant_ct = 0;
	uint32_t sample_ct = 0;
	uint8_t mode = 0;
	uint8_t ctrl = 0;
	std::vector<uint8_t> vrtype;   // one byte per variant
	std::vector<uint64_t> offset;  // variant_ct + 1 byte offsets into the .pgen body
	uint32_t max_record_bytes = 0;
	uint32_t sample_id_bytes = 1;  // width of sample ids inside difflists
	bool has_dosage = false;
	bool has_phase = false;
	bool has_multiallelic = false;
	// alleles per variant (REF + ALTs) when the header carries ALT allele counts, else empty (every variant has two);
	// needed only to size a multiallelic record's aux track (SkipAux1)
	std::vector<uint32_t> allele_ct;
	uint32_t vrtype_hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};

	uint32_t RecordBytes() const {
		return (sample_ct + 3) / 4;
	}
};

// Parses header + per-variant tables.  Returns false and fills err on failure.
bool ParsePgenIndex(const std::string &pgen_path, const std::string &pgi_path, PgenIndex &out, std::string &err);

// Positional reader over the .pgen body (one per thread; owns its descriptor).
class RecordFile {
public:
	RecordFile() = default;
	~RecordFile();
	RecordFile(const RecordFile &) = delete;
	RecordFile &operator=(const RecordFile &) = delete;
	bool Open(const std::string &path, std::string &err);
	bool ReadAt(uint64_t off, size_t len, uint8_t *dst, std::string &err) const;
	uint64_t Size() const {
		return size_;
	}

private:
	int fd_ = -1;
	uint64_t size_ = 0;
};

// Expands records to plain 2-bit rows (00 hom-ref, 01 het, 10 hom-alt, 11 missing;
// sample s in bits 2*(s%4) of byte s/4; bits past sample_ct zero).
class Normalizer {
public:
	Normalizer(const PgenIndex &index, const RecordFile &file);

	// Expand the records of [v_begin, v_end) into dst (row r at dst + r*pitch; the
	// bytes between RecordBytes() and pitch are zeroed).  Variants must be visited
	// in ascending order within one call; LD bases before v_begin are resolved.
	bool ExpandRange(uint32_t v_begin, uint32_t v_end, uint8_t *dst, size_t pitch, std::string &err);

	// Aux tracks, decoded on the host for one variant (raw sample order).
	// dosage16[s] = 0..32768, or 0xffff when sample s carries no explicit dosage.
	bool DecodeDosage(uint32_t v, std::vector<uint8_t> &row2bit, std::vector<uint16_t> &dosage16, std::string &err);
	// phasepresent/phaseinfo: one byte per raw sample.
	bool DecodePhase(uint32_t v, std::vector<uint8_t> &row2bit, std::vector<uint8_t> &phasepresent,
	                 std::vector<uint8_t> &phaseinfo, std::string &err);

private:
	bool ExpandOne(uint32_t v, const uint8_t *rec, size_t rec_len, uint8_t *row, size_t *main_len, std::string &err);
	bool LoadRecord(uint32_t v, std::vector<uint8_t> &buf, std::string &err) const;
	bool ResolveLdBase(uint32_t v, std::string &err);
	bool ExpandWithAux(uint32_t v, std::vector<uint8_t> &rec, std::vector<uint8_t> &row2bit, size_t &aux_off,
	                   std::string &err);
	// Steps over the multiallelic track (vrtype bit 0x08) that follows the main track.  PgrGet / PgrGetCounts /
	// PgrGetD read a multiallelic variant with its ALT alleles collapsed (0 hom-ref, 1 ref + any ALT, 2 two ALTs --
	// the main track as stored, src/pgen_reader.cpp:727, src/plink_freq.cpp:482), so this track is never
	// decoded here, only measured, to find the phase / dosage tracks behind it.
	bool SkipAux1(uint32_t v, const std::vector<uint8_t> &rec, const std::vector<uint8_t> &row2bit, size_t &aux_off,
	              std::string &err) const;

	const PgenIndex &index_;
	const RecordFile &file_;
	std::vector<uint8_t> ld_base_;  // expanded row of the most recent non-LD variant
	int64_t ld_base_variant_ = -1;
};

} // namespace pgh

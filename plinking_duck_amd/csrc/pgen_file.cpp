// pgen_file.cpp -- see pgen_file.hpp.
#include "pgen_file.hpp"

#include <cerrno>
#include <cstring>
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

namespace pgh {

namespace {

inline uint32_t LoadLe(const uint8_t *p, uint32_t nbytes) {
	uint32_t v = 0;
	for (uint32_t i = 0; i < nbytes; i++) {
		v |= static_cast<uint32_t>(p[i]) << (8 * i);
	}
	return v;
}

inline uint64_t LoadLe64(const uint8_t *p) {
	return static_cast<uint64_t>(LoadLe(p, 4)) | (static_cast<uint64_t>(LoadLe(p + 4, 4)) << 32);
}

bool SlurpFile(const std::string &path, std::vector<uint8_t> &out, uint64_t max_bytes, std::string &err) {
	int fd = ::open(path.c_str(), O_RDONLY);
	if (fd < 0) {
		err = "cannot open '" + path + "': " + std::strerror(errno);
		return false;
	}
	struct stat st;
	if (fstat(fd, &st) != 0) {
		err = "cannot stat '" + path + "'";
		::close(fd);
		return false;
	}
	uint64_t want = static_cast<uint64_t>(st.st_size) < max_bytes ? static_cast<uint64_t>(st.st_size) : max_bytes;
	out.resize(want);
	uint64_t got = 0;
	while (got < want) {
		ssize_t n = ::pread(fd, out.data() + got, want - got, static_cast<off_t>(got));
		if (n <= 0) {
			err = "short read on '" + path + "'";
			::close(fd);
			return false;
		}
		got += static_cast<uint64_t>(n);
	}
	::close(fd);
	return true;
}

// Streaming view over the bytes of one record.
struct ByteCursor {
	const uint8_t *p;
	const uint8_t *end;
	bool ok = true;

	uint8_t Byte() {
		if (p >= end) {
			ok = false;
			return 0;
		}
		return *p++;
	}
	uint32_t Varint() {
		uint32_t v = 0;
		for (uint32_t shift = 0; shift < 35; shift += 7) {
			uint8_t b = Byte();
			v |= static_cast<uint32_t>(b & 0x7f) << shift;
			if (!(b & 0x80)) {
				return v;
			}
		}
		ok = false;
		return v;
	}
	const uint8_t *Take(size_t n) {
		if (static_cast<size_t>(end - p) < n) {
			ok = false;
			return nullptr;
		}
		const uint8_t *r = p;
		p += n;
		return r;
	}
};

// Walks a difflist: 64-entry groups, explicit id for each group's first entry,
// varint gaps for the rest; optional packed 2-bit values.
template <class Fn>
bool WalkDifflist(ByteCursor &cur, uint32_t sample_ct, uint32_t id_bytes, bool has_values, Fn &&fn, uint32_t *len_out) {
	uint32_t len = cur.Varint();
	if (len_out) {
		*len_out = len;
	}
	if (!cur.ok || len > sample_ct) {
		return false;
	}
	if (len == 0) {
		return true;
	}
	uint32_t groups = (len + 63) / 64;
	const uint8_t *group_first = cur.Take(static_cast<size_t>(groups) * id_bytes);
	cur.Take(groups - 1); // byte lengths of the groups' gap sections (random access only)
	const uint8_t *values = has_values ? cur.Take((len + 3) / 4) : nullptr;
	if (!cur.ok) {
		return false;
	}
	uint32_t entry = 0;
	for (uint32_t g = 0; g < groups; g++) {
		uint32_t id = LoadLe(group_first + static_cast<size_t>(g) * id_bytes, id_bytes);
		uint32_t in_group = (len - entry) < 64 ? (len - entry) : 64;
		for (uint32_t j = 0; j < in_group; j++, entry++) {
			if (j != 0) {
				id += cur.Varint();
			}
			if (!cur.ok || id >= sample_ct) {
				return false;
			}
			uint32_t val = has_values ? (values[entry >> 2] >> (2 * (entry & 3))) & 3u : 0u;
			fn(id, val, entry);
		}
	}
	return true;
}

inline void PokeGenotype(uint8_t *row, uint32_t sample, uint32_t val) {
	uint32_t sh = 2 * (sample & 3);
	uint8_t &b = row[sample >> 2];
	b = static_cast<uint8_t>((b & ~(3u << sh)) | (val << sh));
}

// 8 presence bits -> 16 bits with each input bit on an even position.
struct SpreadTable {
	uint16_t t[256];
	SpreadTable() {
		for (uint32_t b = 0; b < 256; b++) {
			uint32_t s = 0;
			for (uint32_t i = 0; i < 8; i++) {
				s |= ((b >> i) & 1u) << (2 * i);
			}
			t[b] = static_cast<uint16_t>(s);
		}
	}
};
const SpreadTable kSpread;

inline void ClearTail(uint8_t *row, uint32_t sample_ct) {
	uint32_t rem = sample_ct & 3;
	if (rem) {
		row[sample_ct >> 2] &= static_cast<uint8_t>((1u << (2 * rem)) - 1);
	}
}

uint32_t CountHets(const uint8_t *row, uint32_t sample_ct) {
	uint32_t n = 0;
	for (uint32_t s = 0; s < sample_ct; s++) {
		n += ((row[s >> 2] >> (2 * (s & 3))) & 3u) == 1u;
	}
	return n;
}

} // namespace

// ---------------------------------------------------------------------------
// Index
// ---------------------------------------------------------------------------

bool ParsePgenIndex(const std::string &pgen_path, const std::string &pgi_path, PgenIndex &out, std::string &err) {
	std::vector<uint8_t> head;
	// Header tables are at most 12 + 8*blocks + 9 bytes per variant; read what is
	// needed in two steps so a 125 GB body is never slurped.
	if (!SlurpFile(pgen_path, head, 12, err)) {
		return false;
	}
	if (head.size() < 3 || head[0] != 0x6c || head[1] != 0x1b) {
		err = "'" + pgen_path + "' is not a .pgen file (bad magic number)";
		return false;
	}
	struct stat st;
	uint64_t body_size = 0;
	if (::stat(pgen_path.c_str(), &st) == 0) {
		body_size = static_cast<uint64_t>(st.st_size);
	}
	uint8_t mode = head[2];
	std::string index_path = pgen_path;
	if (mode == 0x20) {
		index_path = pgi_path.empty() ? pgen_path + ".pgi" : pgi_path;
		if (!SlurpFile(index_path, head, 12, err)) {
			return false;
		}
		if (head.size() < 12 || head[0] != 0x6c || head[1] != 0x1b || head[2] != 0x30) {
			err = "'" + index_path + "' is not a .pgen.pgi index";
			return false;
		}
	}
	if (head.size() < 12) {
		err = "'" + index_path + "': truncated header";
		return false;
	}
	out = PgenIndex();
	out.mode = mode;
	out.variant_ct = LoadLe(head.data() + 3, 4);
	out.sample_ct = LoadLe(head.data() + 7, 4);
	out.ctrl = head[11];
	const uint32_t M = out.variant_ct;
	const uint32_t N = out.sample_ct;
	if (N == 0) {
		err = "'" + pgen_path + "': zero samples";
		return false;
	}
	out.sample_id_bytes = N < (1u << 8) ? 1 : (N < (1u << 16) ? 2 : (N < (1u << 24) ? 3 : 4));
	// The counts come from the file: nothing is sized by them until the file has been shown to be
	// long enough to hold what they imply (a corrupt count must not turn into a 30 GB allocation).
	uint64_t index_size = body_size;
	if (mode == 0x20 && ::stat(index_path.c_str(), &st) == 0) {
		index_size = static_cast<uint64_t>(st.st_size);
	}

	if (mode == 0x02) {
		// fixed-width 2-bit records directly after the 12-byte header
		const uint64_t w = out.RecordBytes();
		if (12 + static_cast<uint64_t>(M) * w > body_size) {
			err = "'" + pgen_path + "': variant records run past the end of the file";
			return false;
		}
		out.vrtype.assign(M, 0);
		out.offset.assign(static_cast<size_t>(M) + 1, 0);
		for (uint64_t v = 0; v <= M; v++) {
			out.offset[v] = 12 + v * w;
		}
	} else if (mode == 0x10 || mode == 0x20) {
		const uint32_t width_code = out.ctrl & 0x0f;
		if (width_code >= 8) {
			err = "'" + index_path + "': unsupported header control byte";
			return false;
		}
		const uint32_t type_bits = width_code < 4 ? 4 : 8;
		const uint32_t len_bytes = (width_code & 3) + 1;
		const uint32_t allele_ct_bytes = (out.ctrl >> 4) & 3;
		const bool nonref_flags = ((out.ctrl >> 6) & 3) == 3;
		const uint32_t blocks = static_cast<uint32_t>((static_cast<uint64_t>(M) + 65535) / 65536); // M + 65535 can wrap
		uint64_t table_bytes = 12 + 8ull * blocks;
		for (uint32_t b = 0; b < blocks; b++) {
			uint32_t cnt = (M - b * 65536u) < 65536u ? (M - b * 65536u) : 65536u;
			table_bytes += (type_bits == 4 ? (cnt + 1) / 2 : cnt) + static_cast<uint64_t>(cnt) * len_bytes +
			               static_cast<uint64_t>(cnt) * allele_ct_bytes + (nonref_flags ? (cnt + 7) / 8 : 0);
		}
		if (table_bytes > index_size) {
			err = "'" + index_path + "': truncated variant-record tables";
			return false;
		}
		std::vector<uint8_t> tab;
		if (!SlurpFile(index_path, tab, table_bytes, err)) {
			return false;
		}
		if (tab.size() < table_bytes) {
			err = "'" + index_path + "': truncated variant-record tables";
			return false;
		}
		out.vrtype.assign(M, 0);
		out.offset.assign(static_cast<size_t>(M) + 1, 0);
		uint64_t pos = 12 + 8ull * blocks;
		uint64_t floor_at = mode == 0x10 ? table_bytes : 0; // records never start inside the tables or go backwards
		for (uint32_t b = 0; b < blocks; b++) {
			const uint32_t v0 = b * 65536u;
			const uint32_t cnt = (M - v0) < 65536u ? (M - v0) : 65536u;
			uint64_t at = LoadLe64(tab.data() + 12 + 8ull * b);
			if (at < floor_at || at > body_size) {
				err = "'" + index_path + "': variant block offset out of range";
				return false;
			}
			const uint8_t *types = tab.data() + pos;
			pos += type_bits == 4 ? (cnt + 1) / 2 : cnt;
			const uint8_t *lens = tab.data() + pos;
			pos += static_cast<uint64_t>(cnt) * len_bytes;
			if (allele_ct_bytes) {
				// ALT allele counts of the block's variants
				if (out.allele_ct.empty()) {
					out.allele_ct.assign(M, 2);
				}
				for (uint32_t i = 0; i < cnt; i++) {
					out.allele_ct[v0 + i] = 1u + LoadLe(tab.data() + pos + static_cast<size_t>(i) * allele_ct_bytes, allele_ct_bytes);
				}
			}
			pos += static_cast<uint64_t>(cnt) * allele_ct_bytes;
			if (nonref_flags) {
				pos += (cnt + 7) / 8;
			}
			for (uint32_t i = 0; i < cnt; i++) {
				out.vrtype[v0 + i] =
				    type_bits == 4 ? static_cast<uint8_t>((types[i >> 1] >> (4 * (i & 1))) & 0x0f) : types[i];
				out.offset[v0 + i] = at;
				at += LoadLe(lens + static_cast<size_t>(i) * len_bytes, len_bytes);
			}
			if (at > body_size) {
				err = "'" + pgen_path + "': variant records run past the end of the file";
				return false;
			}
			out.offset[v0 + cnt] = at;
			floor_at = at;
		}
	} else {
		char buf[64];
		std::snprintf(buf, sizeof buf, "0x%02x", mode);
		err = "'" + pgen_path + "': unsupported .pgen storage mode " + buf;
		return false;
	}
	if (M > 0 && out.offset[M] > body_size) {
		err = "'" + pgen_path + "': variant records run past the end of the file";
		return false;
	}
	for (uint32_t v = 0; v < M; v++) {
		uint8_t t = out.vrtype[v];
		out.vrtype_hist[t & 7]++;
		out.has_dosage |= (t & 0x60) != 0;
		out.has_phase |= (t & 0x10) != 0;
		out.has_multiallelic |= (t & 0x08) != 0;
		uint64_t len = out.offset[v + 1] - out.offset[v];
		if (len > out.max_record_bytes) {
			out.max_record_bytes = static_cast<uint32_t>(len);
		}
	}
	return true;
}

// ---------------------------------------------------------------------------
// RecordFile
// ---------------------------------------------------------------------------

RecordFile::~RecordFile() {
	if (fd_ >= 0) {
		::close(fd_);
	}
}

bool RecordFile::Open(const std::string &path, std::string &err) {
	fd_ = ::open(path.c_str(), O_RDONLY);
	if (fd_ < 0) {
		err = "cannot open '" + path + "': " + std::strerror(errno);
		return false;
	}
	struct stat st;
	if (fstat(fd_, &st) == 0) {
		size_ = static_cast<uint64_t>(st.st_size);
	}
	return true;
}

bool RecordFile::ReadAt(uint64_t off, size_t len, uint8_t *dst, std::string &err) const {
	size_t got = 0;
	while (got < len) {
		ssize_t n = ::pread(fd_, dst + got, len - got, static_cast<off_t>(off + got));
		if (n <= 0) {
			err = "short read in .pgen body";
			return false;
		}
		got += static_cast<size_t>(n);
	}
	return true;
}

// ---------------------------------------------------------------------------
// Normalizer
// ---------------------------------------------------------------------------

Normalizer::Normalizer(const PgenIndex &index, const RecordFile &file) : index_(index), file_(file) {
}

bool Normalizer::LoadRecord(uint32_t v, std::vector<uint8_t> &buf, std::string &err) const {
	uint64_t off = index_.offset[v];
	size_t len = static_cast<size_t>(index_.offset[v + 1] - off);
	buf.resize(len);
	return len == 0 || file_.ReadAt(off, len, buf.data(), err);
}

bool Normalizer::ExpandOne(uint32_t v, const uint8_t *rec, size_t rec_len, uint8_t *row, size_t *main_len,
                           std::string &err) {
	const uint32_t N = index_.sample_ct;
	const uint32_t rb = index_.RecordBytes();
	const uint32_t kind = index_.vrtype[v] & 7;
	ByteCursor cur {rec, rec + rec_len};
	auto patch = [&](uint32_t id, uint32_t val, uint32_t) { PokeGenotype(row, id, val); };
	bool ok = true;
	switch (kind) {
	case 0: {
		const uint8_t *src = cur.Take(rb);
		if (!src) {
			ok = false;
			break;
		}
		std::memcpy(row, src, rb);
		ClearTail(row, N);
		break;
	}
	case 1: {
		// two-valued record: one bit per sample picks between `low` and `high`
		uint8_t code = cur.Byte();
		uint32_t low = code >> 2;
		uint32_t delta = code & 3;
		const uint8_t *bits = cur.Take((N + 7) / 8);
		if (!bits) {
			ok = false;
			break;
		}
		const uint32_t base16 = low * 0x5555u;
		uint32_t full = N / 8;
		for (uint32_t i = 0; i < full; i++) {
			uint32_t w = base16 + kSpread.t[bits[i]] * delta;
			row[2 * i] = static_cast<uint8_t>(w);
			row[2 * i + 1] = static_cast<uint8_t>(w >> 8);
		}
		if (N & 7) {
			uint32_t w = base16 + kSpread.t[bits[full]] * delta;
			row[2 * full] = static_cast<uint8_t>(w);
			if (2 * full + 1 < rb) {
				row[2 * full + 1] = static_cast<uint8_t>(w >> 8);
			}
		}
		ClearTail(row, N);
		ok = WalkDifflist(cur, N, index_.sample_id_bytes, true, patch, nullptr);
		break;
	}
	case 2:
	case 3: {
		if (ld_base_variant_ < 0 || ld_base_.size() != rb) {
			err = "LD-compressed record without a base variant";
			return false;
		}
		std::memcpy(row, ld_base_.data(), rb);
		ok = WalkDifflist(cur, N, index_.sample_id_bytes, true, patch, nullptr);
		if (ok && kind == 3) {
			// swap hom-ref <-> hom-alt after patching: flip the high bit where the low bit is clear
			for (uint32_t i = 0; i < rb; i++) {
				uint8_t x = row[i];
				row[i] = static_cast<uint8_t>(x ^ ((~x & 0x55u) << 1));
			}
			ClearTail(row, N);
		}
		break;
	}
	case 4:
	case 6:
	case 7: {
		std::memset(row, kind == 4 ? 0x00 : (kind == 6 ? 0xaa : 0xff), rb);
		ClearTail(row, N);
		ok = WalkDifflist(cur, N, index_.sample_id_bytes, true, patch, nullptr);
		break;
	}
	default:
		err = "unsupported variant record type " + std::to_string(kind);
		return false;
	}
	if (!ok || !cur.ok) {
		err = "malformed variant record " + std::to_string(v);
		return false;
	}
	if (kind != 2 && kind != 3) {
		ld_base_.assign(row, row + rb);
		ld_base_variant_ = v;
	}
	if (main_len) {
		*main_len = static_cast<size_t>(cur.p - rec);
	}
	return true;
}

bool Normalizer::ResolveLdBase(uint32_t v, std::string &err) {
	// nearest earlier record that is not LD-compressed
	uint32_t b = v;
	while (b > 0) {
		b--;
		uint32_t k = index_.vrtype[b] & 7;
		if (k != 2 && k != 3) {
			if (ld_base_variant_ == static_cast<int64_t>(b)) {
				return true;
			}
			std::vector<uint8_t> rec;
			if (!LoadRecord(b, rec, err)) {
				return false;
			}
			std::vector<uint8_t> row(index_.RecordBytes());
			return ExpandOne(b, rec.data(), rec.size(), row.data(), nullptr, err);
		}
	}
	err = "LD-compressed record without a base variant";
	return false;
}

bool Normalizer::ExpandRange(uint32_t v_begin, uint32_t v_end, uint8_t *dst, size_t pitch, std::string &err) {
	if (v_begin >= v_end) {
		return true;
	}
	const uint32_t rb = index_.RecordBytes();
	const uint64_t byte_begin = index_.offset[v_begin];
	const uint64_t byte_end = index_.offset[v_end];
	std::vector<uint8_t> raw(static_cast<size_t>(byte_end - byte_begin));
	if (!raw.empty() && !file_.ReadAt(byte_begin, raw.size(), raw.data(), err)) {
		return false;
	}
	for (uint32_t v = v_begin; v < v_end; v++) {
		uint32_t kind = index_.vrtype[v] & 7;
		if ((kind == 2 || kind == 3) && v == v_begin) {
			if (!ResolveLdBase(v, err)) {
				return false;
			}
		}
		uint8_t *row = dst + static_cast<size_t>(v - v_begin) * pitch;
		const uint8_t *rec = raw.data() + (index_.offset[v] - byte_begin);
		size_t rec_len = static_cast<size_t>(index_.offset[v + 1] - index_.offset[v]);
		if (!ExpandOne(v, rec, rec_len, row, nullptr, err)) {
			return false;
		}
		if (pitch > rb) {
			std::memset(row + rb, 0, pitch - rb);
		}
	}
	return true;
}

bool Normalizer::ExpandWithAux(uint32_t v, std::vector<uint8_t> &rec, std::vector<uint8_t> &row2bit, size_t &aux_off,
                               std::string &err) {
	if (v >= index_.variant_ct) {
		err = "variant index out of range";
		return false;
	}
	uint32_t kind = index_.vrtype[v] & 7;
	if (kind == 2 || kind == 3) {
		if (!ResolveLdBase(v, err)) {
			return false;
		}
	}
	if (!LoadRecord(v, rec, err)) {
		return false;
	}
	row2bit.assign(index_.RecordBytes(), 0);
	if (!ExpandOne(v, rec.data(), rec.size(), row2bit.data(), &aux_off, err)) {
		return false;
	}
	if (index_.vrtype[v] & 0x08) {
		return SkipAux1(v, rec, row2bit, aux_off, err);
	}
	return true;
}

bool Normalizer::SkipAux1(uint32_t v, const std::vector<uint8_t> &rec, const std::vector<uint8_t> &row2bit, size_t &aux_off,
                          std::string &err) const {
	// Layout (PLINK 2 .pgen specification; no reference fixture holds such a record -- parity unpinned):
	//   1 byte: low nibble = mode of part a (patches of the ref/ALT calls, genotype 1: which of them carry an ALT
	//           other than ALT1), high nibble = mode of part b (patches of the two-ALT calls, genotype 2: which are
	//           not ALT1/ALT1).  Mode 0: a bit per such call; 1: a list of sample ids (difflist layout without
	//           values); 15: no patches.
	//   part a: [bit array | id list], then the patched calls' allele codes: none with 3 alleles, 1 bit each with 4,
	//           2 bits with 5-6, 4 bits with 7-18, a byte beyond;
	//   part b: [bit array | id list], then two codes per patched call: 1 bit per call with 3 alleles, 2 + 2 bits
	//           with 4-5, 4 + 4 with 6-17, 8 + 8 beyond.
	const uint32_t N = index_.sample_ct;
	const uint32_t alleles = index_.allele_ct.empty() ? 2u : index_.allele_ct[v];
	if (alleles < 3) {
		err = "variant " + std::to_string(v) + " has a multiallelic track but fewer than three alleles";
		return false;
	}
	uint32_t n_01 = 0, n_10 = 0;
	for (uint32_t s = 0; s < N; s++) {
		const uint32_t g = (row2bit[s >> 2] >> (2 * (s & 3))) & 3u;
		n_01 += g == 1u;
		n_10 += g == 2u;
	}
	ByteCursor cur {rec.data() + aux_off, rec.data() + rec.size()};
	const uint8_t modes = cur.Byte();
	auto patches = [&](uint32_t mode, uint32_t calls, uint32_t &count) {
		count = 0;
		if (mode == 15) {
			return true;
		}
		if (mode == 0) {
			const uint8_t *bits = cur.Take((calls + 7) / 8);
			for (uint32_t i = 0; bits && i < calls; i++) {
				count += (bits[i >> 3] >> (i & 7)) & 1u;
			}
			return bits != nullptr || calls == 0;
		}
		if (mode == 1) {
			return WalkDifflist(cur, N, index_.sample_id_bytes, false, [](uint32_t, uint32_t, uint32_t) {}, &count);
		}
		return false;
	};
	uint32_t n_a = 0, n_b = 0;
	bool ok = cur.ok && patches(modes & 15u, n_01, n_a);
	if (ok) {
		const uint32_t bits_a = alleles == 3 ? 0u : alleles == 4 ? 1u : alleles <= 6 ? 2u : alleles <= 18 ? 4u : 8u;
		ok = cur.Take((static_cast<uint64_t>(n_a) * bits_a + 7) / 8) != nullptr || n_a * bits_a == 0;
	}
	ok = ok && patches(modes >> 4, n_10, n_b);
	if (ok) {
		const uint32_t bits_b = alleles == 3 ? 1u : alleles <= 5 ? 4u : alleles <= 17 ? 8u : 16u;
		ok = cur.Take((static_cast<uint64_t>(n_b) * bits_b + 7) / 8) != nullptr || n_b == 0;
	}
	if (!ok || !cur.ok) {
		err = "malformed multiallelic track in variant " + std::to_string(v);
		return false;
	}
	aux_off = static_cast<size_t>(cur.p - rec.data());
	return true;
}

bool Normalizer::DecodePhase(uint32_t v, std::vector<uint8_t> &row2bit, std::vector<uint8_t> &phasepresent,
                             std::vector<uint8_t> &phaseinfo, std::string &err) {
	std::vector<uint8_t> rec;
	size_t aux = 0;
	if (!ExpandWithAux(v, rec, row2bit, aux, err)) {
		return false;
	}
	const uint32_t N = index_.sample_ct;
	phasepresent.assign(N, 0);
	phaseinfo.assign(N, 0);
	if (!(index_.vrtype[v] & 0x10)) {
		return true;
	}
	// Track: bit 0 = "explicit phasepresent follows"; then one bit per het.  If the
	// flag is clear every het is phased and those bits are the phase itself;
	// otherwise they say which hets are phased and a second bitarray holds the phase.
	const uint32_t het_ct = CountHets(row2bit.data(), N);
	ByteCursor cur {rec.data() + aux, rec.data() + rec.size()};
	const uint8_t *first = cur.Take((1 + het_ct + 7) / 8);
	if (!first) {
		err = "truncated phase track in variant " + std::to_string(v);
		return false;
	}
	const bool explicit_present = first[0] & 1;
	uint32_t phased_ct = 0;
	if (explicit_present) {
		for (uint32_t i = 0; i < het_ct; i++) {
			phased_ct += (first[(1 + i) >> 3] >> ((1 + i) & 7)) & 1u;
		}
	}
	const uint8_t *info = explicit_present ? cur.Take((phased_ct + 7) / 8) : nullptr;
	if (explicit_present && !info) {
		err = "truncated phase track in variant " + std::to_string(v);
		return false;
	}
	uint32_t het_i = 0, phased_i = 0;
	for (uint32_t s = 0; s < N; s++) {
		if (((row2bit[s >> 2] >> (2 * (s & 3))) & 3u) != 1u) {
			continue;
		}
		uint32_t bit = (first[(1 + het_i) >> 3] >> ((1 + het_i) & 7)) & 1u;
		if (!explicit_present) {
			phasepresent[s] = 1;
			phaseinfo[s] = static_cast<uint8_t>(bit);
		} else if (bit) {
			phasepresent[s] = 1;
			phaseinfo[s] = (info[phased_i >> 3] >> (phased_i & 7)) & 1u;
			phased_i++;
		}
		het_i++;
	}
	return true;
}

bool Normalizer::DecodeDosage(uint32_t v, std::vector<uint8_t> &row2bit, std::vector<uint16_t> &dosage16,
                              std::string &err) {
	std::vector<uint8_t> rec;
	size_t aux = 0;
	if (!ExpandWithAux(v, rec, row2bit, aux, err)) {
		return false;
	}
	const uint32_t N = index_.sample_ct;
	const uint8_t t = index_.vrtype[v];
	dosage16.assign(N, 0xffff);
	// a phased-dosage track (0x80) follows the dosage track; PgrGetD does not read it either
	ByteCursor cur {rec.data() + aux, rec.data() + rec.size()};
	if (t & 0x10) {
		// step over the phase track
		const uint32_t het_ct = CountHets(row2bit.data(), N);
		const uint8_t *first = cur.Take((1 + het_ct + 7) / 8);
		if (!first) {
			err = "truncated phase track in variant " + std::to_string(v);
			return false;
		}
		if (first[0] & 1) {
			uint32_t phased_ct = 0;
			for (uint32_t i = 0; i < het_ct; i++) {
				phased_ct += (first[(1 + i) >> 3] >> ((1 + i) & 7)) & 1u;
			}
			cur.Take((phased_ct + 7) / 8);
		}
	}
	bool ok = true;
	switch (t & 0x60) {
	case 0x00:
		break;
	case 0x20: {
		// sparse: id list (difflist layout without values), then one u16 per entry
		std::vector<uint32_t> ids;
		ok = WalkDifflist(
		    cur, N, index_.sample_id_bytes, false, [&](uint32_t id, uint32_t, uint32_t) { ids.push_back(id); },
		    nullptr);
		const uint8_t *vals = ok ? cur.Take(2 * ids.size()) : nullptr;
		ok = ok && (vals || ids.empty());
		for (size_t i = 0; ok && i < ids.size(); i++) {
			dosage16[ids[i]] = static_cast<uint16_t>(LoadLe(vals + 2 * i, 2));
		}
		break;
	}
	case 0x40: {
		const uint8_t *vals = cur.Take(2 * static_cast<size_t>(N));
		ok = vals != nullptr;
		for (uint32_t s = 0; ok && s < N; s++) {
			dosage16[s] = static_cast<uint16_t>(LoadLe(vals + 2 * static_cast<size_t>(s), 2));
		}
		break;
	}
	case 0x60: {
		const uint8_t *present = cur.Take((N + 7) / 8);
		ok = present != nullptr;
		for (uint32_t s = 0; ok && s < N; s++) {
			if ((present[s >> 3] >> (s & 7)) & 1u) {
				const uint8_t *p = cur.Take(2);
				if (!p) {
					ok = false;
					break;
				}
				dosage16[s] = static_cast<uint16_t>(LoadLe(p, 2));
			}
		}
		break;
	}
	}
	if (!ok || !cur.ok) {
		err = "malformed dosage track in variant " + std::to_string(v);
		return false;
	}
	return true;
}

} // namespace pgh

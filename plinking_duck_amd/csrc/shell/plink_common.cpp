// plink_common.cpp -- see plink_common.hpp.
#include "plink_common.hpp"

#include <future>

#include <algorithm>
#include <cerrno>
#include <condition_variable>
#include <cstring>
#include <cmath>
#include <cstdlib>
#include <fstream>
#include <functional>
#include <limits>
#include <sstream>
#include <thread>
#include <climits>
#include <sys/stat.h>
#include <unistd.h>
#include <unordered_set>

namespace duckdb {

namespace {

string Lower(string s) {
	for (auto &c : s) {
		c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
	}
	return s;
}

string Trim(const string &s) {
	size_t a = 0, b = s.size();
	while (a < b && std::isspace(static_cast<unsigned char>(s[a]))) {
		a++;
	}
	while (b > a && std::isspace(static_cast<unsigned char>(s[b - 1]))) {
		b--;
	}
	return s.substr(a, b - a);
}

string ReplaceExtension(const string &path, const string &ext) {
	auto slash = path.find_last_of('/');
	auto dot = path.find_last_of('.');
	if (dot == string::npos || (slash != string::npos && dot < slash)) {
		return path + ext;
	}
	return path.substr(0, dot) + ext;
}

bool ReadWholeFile(const string &path, string &out) {
	std::ifstream f(path, std::ios::binary);
	if (!f) {
		return false;
	}
	// a regular file: one read of its size (a .pvar of a million variants is tens of megabytes); anything else
	// (a pipe, a file still growing) through the stream buffer
	f.seekg(0, std::ios::end);
	const std::streamoff size = f.tellg();
	if (size > 0) {
		f.seekg(0, std::ios::beg);
		out.resize(static_cast<size_t>(size));
		f.read(&out[0], size);
		if (f.gcount() == size && f.peek() == std::char_traits<char>::eof()) {
			return true;
		}
		f.clear();
		f.seekg(0, std::ios::beg);
	}
	std::ostringstream ss;
	ss << f.rdbuf();
	out = ss.str();
	return true;
}

// split on tabs (.pvar/.psam) or runs of blanks (.bim/.fam)
vector<string> SplitFields(const string &line, bool whitespace) {
	vector<string> out;
	if (!whitespace) {
		size_t start = 0;
		while (true) {
			size_t tab = line.find('\t', start);
			if (tab == string::npos) {
				out.push_back(line.substr(start));
				break;
			}
			out.push_back(line.substr(start, tab - start));
			start = tab + 1;
		}
		return out;
	}
	size_t i = 0;
	while (i < line.size()) {
		while (i < line.size() && (line[i] == ' ' || line[i] == '\t')) {
			i++;
		}
		size_t j = i;
		while (j < line.size() && line[j] != ' ' && line[j] != '\t') {
			j++;
		}
		if (j > i) {
			out.push_back(line.substr(i, j - i));
		}
		i = j;
	}
	return out;
}

vector<string> Lines(const string &content) {
	vector<string> out;
	size_t pos = 0;
	while (pos < content.size()) {
		size_t nl = content.find('\n', pos);
		if (nl == string::npos) {
			nl = content.size();
		}
		size_t end = nl;
		if (end > pos && content[end - 1] == '\r') {
			end--;
		}
		out.push_back(content.substr(pos, end - pos));
		pos = nl + 1;
	}
	return out;
}

} // namespace

// ---------------------------------------------------------------------------
// files
// ---------------------------------------------------------------------------

bool FileExists(const string &path) {
	struct stat st;
	return ::stat(path.c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

bool ParseSynthPath(const string &path, SynthSpec &out) {
	if (path.compare(0, 6, "synth:") != 0) {
		return false;
	}
	// synth:<variants>x<samples>[:<seed>[:<missing rate>]]
	SynthSpec spec;
	const char *p = path.c_str() + 6;
	char *end = nullptr;
	const unsigned long long m = std::strtoull(p, &end, 10);
	if (end == p || *end != 'x') {
		throw InvalidInputException("'%s': expected synth:<variants>x<samples>[:<seed>[:<missing rate>]]", path);
	}
	p = end + 1;
	const unsigned long long n = std::strtoull(p, &end, 10);
	if (end == p || m == 0 || n == 0 || m > 0x7fffffffull || n > 0x7fffffffull) {
		throw InvalidInputException("'%s': expected synth:<variants>x<samples>[:<seed>[:<missing rate>]]", path);
	}
	spec.variants = static_cast<uint32_t>(m);
	spec.samples = static_cast<uint32_t>(n);
	if (*end == ':') {
		p = end + 1;
		spec.seed = std::strtoull(p, &end, 10);
		if (end == p) {
			throw InvalidInputException("'%s': bad seed", path);
		}
		if (*end == ':') {
			p = end + 1;
			spec.missing_rate = std::strtod(p, &end);
			if (end == p || !(spec.missing_rate >= 0.0 && spec.missing_rate <= 1.0)) {
				throw InvalidInputException("'%s': bad missing rate", path);
			}
		}
	}
	if (*end != '\0') {
		throw InvalidInputException("'%s': expected synth:<variants>x<samples>[:<seed>[:<missing rate>]]", path);
	}
	out = spec;
	return true;
}

string FindCompanionFile(const string &pgen_path, const vector<string> &extensions) {
	SynthSpec synth;
	if (ParseSynthPath(pgen_path, synth)) {
		return pgen_path; // the generator stands in for all three files
	}
	for (auto &ext : extensions) {
		auto candidate = ReplaceExtension(pgen_path, ext);
		if (FileExists(candidate)) {
			return candidate;
		}
	}
	return "";
}

// ---------------------------------------------------------------------------
// variant metadata
// ---------------------------------------------------------------------------

namespace {
struct PvarCacheEntry {
	string path;
	int64_t mtime_ns = 0, size = 0;
	VariantMetadataIndex index;
};
std::mutex g_pvar_cache_mutex;
vector<PvarCacheEntry> g_pvar_cache; // most recently used last; a handful of files
constexpr size_t kPvarCacheEntries = 8;
} // namespace

// One pass over the file's bytes: lines and fields are (pointer, length) views into the buffer, the only
// allocations are the column vectors' own (reserved from a newline count) and the strings that do not fit the
// small-string buffer.  (The first form copied every line and every field into a std::string first: 41 ms per
// 200,000 variants; the reference's own loader, src/plink_common.cpp:171-375, works the same way.)
static VariantMetadataIndex ParseVariantMetadata(const string &path, const string &func_name) {
	string content;
	if (!ReadWholeFile(path, content)) {
		throw IOException("%s: cannot open .pvar/.bim file '%s'", func_name, path);
	}
	if (content.empty()) {
		throw InvalidInputException("%s: .pvar/.bim file '%s' is empty", func_name, path);
	}
	VariantMetadataIndex out;
	auto columns = make_shared<VariantColumns>();
	VariantColumns &idx = *columns;
	const char *const base = content.data();
	const char *const stop = base + content.size();
	const char *cur = base;
	size_t line_no = 0; // 1-based number of the line in `line`
	struct View {
		const char *p;
		size_t n;
	};
	View line {nullptr, 0};
	auto next_line = [&]() -> bool {
		if (cur >= stop) {
			return false;
		}
		const char *nl = static_cast<const char *>(std::memchr(cur, '\n', static_cast<size_t>(stop - cur)));
		const char *e = nl ? nl : stop;
		line.p = cur;
		line.n = static_cast<size_t>(e - cur);
		if (line.n && line.p[line.n - 1] == '\r') {
			line.n--;
		}
		cur = nl ? nl + 1 : stop;
		line_no++;
		return true;
	};
	bool have = next_line();
	while (have && (line.n == 0 || (line.n >= 2 && line.p[0] == '#' && line.p[1] == '#'))) {
		have = next_line();
	}
	if (!have) {
		throw InvalidInputException("%s: .pvar/.bim file '%s' contains no header or data", func_name, path);
	}
	constexpr size_t kNone = static_cast<size_t>(-1);
	size_t chrom_f = kNone, pos_f = kNone, id_f = kNone, ref_f = kNone, alt_f = kNone;
	bool line_is_data = true;
	if (line.n >= 6 && std::memcmp(line.p, "#CHROM", 6) == 0) {
		auto fields = SplitFields(string(line.p + 1, line.n - 1), false);
		for (size_t i = 0; i < fields.size(); i++) {
			if (fields[i] == "CHROM") {
				chrom_f = i;
			} else if (fields[i] == "POS") {
				pos_f = i;
			} else if (fields[i] == "ID") {
				id_f = i;
			} else if (fields[i] == "REF") {
				ref_f = i;
			} else if (fields[i] == "ALT") {
				alt_f = i;
			}
		}
		line_is_data = false;
	} else {
		// .bim: CHROM ID CM POS ALT REF
		out.is_bim = true;
		chrom_f = 0;
		id_f = 1;
		pos_f = 3;
		alt_f = 4;
		ref_f = 5;
	}
	if (chrom_f == kNone || pos_f == kNone || id_f == kNone || ref_f == kNone || alt_f == kNone) {
		throw InvalidInputException("%s: .pvar/.bim file '%s' is missing required columns "
		                            "(need CHROM, POS, ID, REF, ALT)",
		                            func_name, path);
	}
	const size_t max_f = std::max({chrom_f, pos_f, id_f, ref_f, alt_f});
	{
		size_t newlines = 0;
		for (const char *q = cur; q < stop;) {
			const char *nl = static_cast<const char *>(std::memchr(q, '\n', static_cast<size_t>(stop - q)));
			if (!nl) {
				break;
			}
			newlines++;
			q = nl + 1;
		}
		const size_t expect = newlines + 2;
		idx.chroms.reserve(expect);
		idx.positions.reserve(expect);
		idx.ids.reserve(expect);
		idx.refs.reserve(expect);
		idx.alts.reserve(expect);
	}
	vector<View> f(max_f + 1);
	const bool blanks = out.is_bim; // .bim: runs of blanks separate fields; .pvar: single tabs
	for (have = line_is_data ? true : next_line(); have; have = next_line()) {
		if (line.n == 0) {
			continue;
		}
		// the first max_f + 1 fields of the line
		size_t got = 0;
		const char *q = line.p, *const le = line.p + line.n;
		if (!blanks) {
			while (got <= max_f) {
				const char *tab = static_cast<const char *>(std::memchr(q, '\t', static_cast<size_t>(le - q)));
				const char *fe = tab ? tab : le;
				f[got++] = View {q, static_cast<size_t>(fe - q)};
				if (!tab) {
					break;
				}
				q = tab + 1;
			}
		} else {
			while (got <= max_f) {
				while (q < le && (*q == ' ' || *q == '\t')) {
					q++;
				}
				const char *fe = q;
				while (fe < le && *fe != ' ' && *fe != '\t') {
					fe++;
				}
				if (fe == q) {
					break;
				}
				f[got++] = View {q, static_cast<size_t>(fe - q)};
				q = fe;
			}
		}
		if (got <= max_f) {
			throw InvalidInputException("%s: .pvar/.bim file '%s' has a line missing required fields (line %llu)",
			                            func_name, path, static_cast<unsigned long long>(line_no));
		}
		// POS: strtol's grammar and its whole-field rule, on a terminated copy
		char num[32];
		const View pf = f[pos_f];
		string long_field; // (a POS field of 32 characters or more: leading zeros, or an error either way)
		const char *text = num;
		if (pf.n < sizeof(num)) {
			std::memcpy(num, pf.p, pf.n);
			num[pf.n] = '\0';
		} else {
			long_field.assign(pf.p, pf.n);
			text = long_field.c_str();
		}
		char *endp;
		errno = 0;
		const long pos_value = std::strtol(text, &endp, 10);
		const bool pos_ok = endp != text && *endp == '\0' && errno == 0;
		if (!pos_ok) {
			throw InvalidInputException("%s: invalid POS value '%s' at line %llu", func_name, string(pf.p, pf.n),
			                            static_cast<unsigned long long>(line_no));
		}
		auto is_dot = [](const View &v) { return v.n == 1 && v.p[0] == '.'; };
		idx.chroms.emplace_back(f[chrom_f].p, f[chrom_f].n);
		idx.positions.push_back(static_cast<int32_t>(pos_value));
		if (is_dot(f[id_f])) {
			idx.ids.emplace_back();
		} else {
			idx.ids.emplace_back(f[id_f].p, f[id_f].n);
		}
		idx.refs.emplace_back(f[ref_f].p, f[ref_f].n);
		if (is_dot(f[alt_f])) {
			idx.alts.emplace_back();
		} else {
			idx.alts.emplace_back(f[alt_f].p, f[alt_f].n);
		}
	}
	out.variant_ct = idx.chroms.size();
	// contiguous chromosome runs (region lookups binary-search POS inside a run)
	idx_t run_start = 0;
	for (idx_t i = 1; i <= out.variant_ct; i++) {
		if (i == out.variant_ct || idx.chroms[i] != idx.chroms[run_start]) {
			if (out.variant_ct == 0) {
				break;
			}
			auto ins = idx.chrom_offsets.emplace(idx.chroms[run_start], std::make_pair(run_start, i));
			if (!ins.second) {
				throw InvalidInputException("%s: chromosome '%s' appears in non-contiguous runs (variants must be "
				                            "sorted by (CHROM, POS) as required by the PLINK spec)",
				                            path, idx.chroms[run_start]);
			}
			run_start = i;
		}
	}
	out.cols = std::move(columns);
	return out;
}

// ---- the on-disk side-cache of the parsed columns ---------------------------------------------------------------
// A second process binding the same big .pvar need not split a million text lines again: the first parse leaves
// the columns behind in binary form -- positions as int32, the string columns as offsets + one blob each, CHROM as
// its runs -- under $PLINKING_PVAR_CACHE_DIR (else $XDG_CACHE_HOME/plinking_duck_amd, else
// ~/.cache/plinking_duck_amd), in a file named after the absolute path's hash and valid for exactly one
// (path, size, mtime).  PLINKING_PVAR_CACHE=0 turns it off; files under 256 KB are not worth a cache entry; every
// failure on the way (no directory, short read, stale or foreign contents) falls back to parsing the text.
namespace {

constexpr char kPvarCacheMagic[8] = {'P', 'G', 'H', 'P', 'V', 'A', 'R', '1'};
constexpr int64_t kPvarCacheMinBytes = 256 << 10;

string PvarCacheDir() {
	const char *off = std::getenv("PLINKING_PVAR_CACHE");
	if (off && off[0] == '0') {
		return "";
	}
	if (const char *dir = std::getenv("PLINKING_PVAR_CACHE_DIR")) {
		return dir;
	}
	if (const char *xdg = std::getenv("XDG_CACHE_HOME")) {
		if (*xdg) {
			return string(xdg) + "/plinking_duck_amd";
		}
	}
	if (const char *home = std::getenv("HOME")) {
		if (*home) {
			return string(home) + "/.cache/plinking_duck_amd";
		}
	}
	return "";
}

string AbsolutePath(const string &path) {
	char buf[PATH_MAX];
	return ::realpath(path.c_str(), buf) ? string(buf) : path;
}

string PvarCacheFile(const string &dir, const string &abs_path) {
	uint64_t h = 1469598103934665603ull; // FNV-1a
	for (unsigned char c : abs_path) {
		h = (h ^ c) * 1099511628211ull;
	}
	char name[40];
	std::snprintf(name, sizeof name, "/%016llx.pvarc", static_cast<unsigned long long>(h));
	return dir + name;
}

struct ByteWriter {
	string out;
	template <class T>
	void Put(const T &v) {
		out.append(reinterpret_cast<const char *>(&v), sizeof v);
	}
	void PutStrings(const vector<string> &col) {
		uint64_t total = 0;
		for (auto &x : col) {
			total += x.size();
		}
		Put(total);
		// offsets and bytes written in place (one append per element made the write twice as slow as the parse)
		const size_t offs_at = out.size(), blob_at = offs_at + 4 * (col.size() + 1);
		out.resize(blob_at + total);
		char *base = &out[0];
		uint32_t at = 0;
		for (size_t i = 0; i < col.size(); i++) {
			std::memcpy(base + offs_at + 4 * i, &at, 4);
			std::memcpy(base + blob_at + at, col[i].data(), col[i].size());
			at += static_cast<uint32_t>(col[i].size());
		}
		std::memcpy(base + offs_at + 4 * col.size(), &at, 4);
	}
};

struct ByteReader {
	const char *p, *end;
	bool ok = true;
	template <class T>
	T Get() {
		T v {};
		if (static_cast<size_t>(end - p) < sizeof v) {
			ok = false;
			return v;
		}
		std::memcpy(&v, p, sizeof v);
		p += sizeof v;
		return v;
	}
	const char *Take(uint64_t n) {
		if (static_cast<uint64_t>(end - p) < n) {
			ok = false;
			return nullptr;
		}
		const char *q = p;
		p += n;
		return q;
	}
	bool GetStrings(uint64_t n, vector<string> &col) {
		const uint64_t total = Get<uint64_t>();
		const char *offs = Take(4 * (n + 1));
		const char *blob = Take(total);
		if (!ok) {
			return false;
		}
		col.clear();
		col.reserve(n);
		uint32_t a, b;
		std::memcpy(&a, offs, 4);
		for (uint64_t i = 0; i < n; i++) {
			std::memcpy(&b, offs + 4 * (i + 1), 4);
			if (b < a || b > total) {
				return false;
			}
			col.emplace_back(blob + a, b - a);
			a = b;
		}
		return true;
	}
};

void WritePvarCache(const string &file, const string &abs_path, int64_t size, int64_t mtime_ns,
                    const VariantMetadataIndex &idx) {
	const VariantColumns &c = *idx.cols;
	const uint64_t n = idx.variant_ct;
	uint64_t blob = 0;
	for (auto *col : {&c.ids, &c.refs, &c.alts}) {
		for (auto &x : *col) {
			blob += x.size();
		}
	}
	if (n > 0xfffffff0ull || blob > 0xfffffff0ull) {
		return; // the 32-bit offsets of the format do not hold it
	}
	ByteWriter w;
	w.out.reserve(64 + abs_path.size() + 4 * n + 3 * 4 * (n + 1) + blob);
	w.out.append(kPvarCacheMagic, sizeof kPvarCacheMagic);
	w.Put(static_cast<int64_t>(size));
	w.Put(static_cast<int64_t>(mtime_ns));
	w.Put(static_cast<uint64_t>(abs_path.size()));
	w.out.append(abs_path);
	w.Put(static_cast<uint8_t>(idx.is_bim ? 1 : 0));
	w.Put(n);
	// CHROM as its runs, in variant order
	vector<std::pair<idx_t, const string *>> runs;
	for (auto &kv : c.chrom_offsets) {
		runs.emplace_back(kv.second.first, &kv.first);
	}
	std::sort(runs.begin(), runs.end());
	w.Put(static_cast<uint32_t>(runs.size()));
	for (auto &r : runs) {
		w.Put(static_cast<uint32_t>(c.chrom_offsets.at(*r.second).first));
		w.Put(static_cast<uint32_t>(c.chrom_offsets.at(*r.second).second));
		w.Put(static_cast<uint32_t>(r.second->size()));
		w.out.append(*r.second);
	}
	w.out.append(reinterpret_cast<const char *>(c.positions.data()), 4 * n);
	w.PutStrings(c.ids);
	w.PutStrings(c.refs);
	w.PutStrings(c.alts);
	const string tmp = file + "." + std::to_string(static_cast<long long>(::getpid())) + ".tmp";
	{
		std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
		if (!f) {
			return;
		}
		f.write(w.out.data(), static_cast<std::streamsize>(w.out.size()));
		if (!f) {
			f.close();
			::unlink(tmp.c_str());
			return;
		}
	}
	if (::rename(tmp.c_str(), file.c_str()) != 0) {
		::unlink(tmp.c_str());
	}
}

bool ReadPvarCache(const string &file, const string &abs_path, int64_t size, int64_t mtime_ns,
                   VariantMetadataIndex &out) {
	string bytes;
	if (!ReadWholeFile(file, bytes) || bytes.size() < sizeof kPvarCacheMagic ||
	    std::memcmp(bytes.data(), kPvarCacheMagic, sizeof kPvarCacheMagic) != 0) {
		return false;
	}
	ByteReader r {bytes.data() + sizeof kPvarCacheMagic, bytes.data() + bytes.size()};
	if (r.Get<int64_t>() != size || r.Get<int64_t>() != mtime_ns) {
		return false;
	}
	const uint64_t path_len = r.Get<uint64_t>();
	const char *path_bytes = r.Take(path_len);
	if (!r.ok || path_len != abs_path.size() || std::memcmp(path_bytes, abs_path.data(), path_len) != 0) {
		return false;
	}
	auto columns = make_shared<VariantColumns>();
	VariantColumns &c = *columns;
	const bool is_bim = r.Get<uint8_t>() != 0;
	const uint64_t n = r.Get<uint64_t>();
	const uint32_t n_runs = r.Get<uint32_t>();
	if (!r.ok || n > 0xfffffff0ull || n_runs > n) {
		return false;
	}
	c.chroms.resize(n);
	uint64_t covered = 0;
	for (uint32_t k = 0; k < n_runs; k++) {
		const uint32_t a = r.Get<uint32_t>(), b = r.Get<uint32_t>(), len = r.Get<uint32_t>();
		const char *name = r.Take(len);
		if (!r.ok || a != covered || b <= a || b > n) {
			return false;
		}
		const string chrom(name, len);
		if (!c.chrom_offsets.emplace(chrom, std::make_pair(static_cast<idx_t>(a), static_cast<idx_t>(b))).second) {
			return false;
		}
		for (uint32_t v = a; v < b; v++) {
			c.chroms[v] = chrom;
		}
		covered = b;
	}
	if (covered != n) {
		return false;
	}
	const char *pos = r.Take(4 * n);
	if (!r.ok) {
		return false;
	}
	c.positions.resize(n);
	std::memcpy(c.positions.data(), pos, 4 * n);
	if (!r.GetStrings(n, c.ids) || !r.GetStrings(n, c.refs) || !r.GetStrings(n, c.alts) || r.p != r.end) {
		return false;
	}
	out.cols = std::move(columns);
	out.variant_ct = n;
	out.is_bim = is_bim;
	return true;
}

} // namespace

static VariantMetadataIndex ParseOrLoadVariantMetadata(const string &path, const string &func_name, bool have_stat,
                                                       int64_t size, int64_t mtime_ns) {
	string dir;
	if (have_stat && size >= kPvarCacheMinBytes) {
		dir = PvarCacheDir();
	}
	if (dir.empty()) {
		return ParseVariantMetadata(path, func_name);
	}
	const string abs_path = AbsolutePath(path);
	const string file = PvarCacheFile(dir, abs_path);
	VariantMetadataIndex cached;
	if (ReadPvarCache(file, abs_path, size, mtime_ns, cached)) {
		return cached;
	}
	VariantMetadataIndex parsed = ParseVariantMetadata(path, func_name);
	::mkdir(dir.c_str(), 0700); // (one level: the parent is the user's cache directory or the one they named)
	WritePvarCache(file, abs_path, size, mtime_ns, parsed);
	return parsed;
}

//! The columns pgh_synth_write_files writes for `variants` variants (api_dataset.cpp:WriteSynthCompanions).
static VariantMetadataIndex SynthVariantMetadata(uint32_t variants) {
	auto columns = make_shared<VariantColumns>();
	VariantColumns &c = *columns;
	c.chroms.reserve(variants);
	c.ids.reserve(variants);
	c.positions.reserve(variants);
	const uint32_t per_chrom = (variants + 21) / 22;
	for (uint32_t v = 0; v < variants; v++) {
		c.chroms.push_back(std::to_string(v / per_chrom + 1));
		c.positions.push_back(static_cast<int32_t>((v % per_chrom + 1) * 100));
		c.ids.push_back("sv" + std::to_string(v));
	}
	c.refs.assign(variants, "A");
	c.alts.assign(variants, "G");
	for (uint32_t a = 0; a < variants; a += per_chrom) {
		c.chrom_offsets.emplace(c.chroms[a], std::make_pair(static_cast<idx_t>(a),
		                                                    static_cast<idx_t>(std::min(variants, a + per_chrom))));
	}
	VariantMetadataIndex out;
	out.cols = std::move(columns);
	out.variant_ct = variants;
	return out;
}

VariantMetadataIndex LoadVariantMetadata(ClientContext &, const string &path, const string &func_name) {
	SynthSpec synth;
	if (ParseSynthPath(path, synth)) {
		std::unique_lock<std::mutex> lock(g_pvar_cache_mutex);
		for (auto &e : g_pvar_cache) {
			if (e.path == path) {
				return e.index;
			}
		}
		lock.unlock();
		VariantMetadataIndex made = SynthVariantMetadata(synth.variants);
		lock.lock();
		g_pvar_cache.push_back(PvarCacheEntry {path, 0, 0, made});
		if (g_pvar_cache.size() > kPvarCacheEntries) {
			g_pvar_cache.erase(g_pvar_cache.begin());
		}
		return made;
	}
	struct stat st;
	const bool have_stat = ::stat(path.c_str(), &st) == 0;
	const int64_t mtime_ns = have_stat ? static_cast<int64_t>(st.st_mtim.tv_sec) * 1000000000LL + st.st_mtim.tv_nsec : 0;
	const int64_t size = have_stat ? static_cast<int64_t>(st.st_size) : -1;
	if (have_stat) {
		std::lock_guard<std::mutex> lock(g_pvar_cache_mutex);
		for (size_t i = 0; i < g_pvar_cache.size(); i++) {
			if (g_pvar_cache[i].path == path && g_pvar_cache[i].mtime_ns == mtime_ns && g_pvar_cache[i].size == size) {
				PvarCacheEntry hit = g_pvar_cache[i];
				g_pvar_cache.erase(g_pvar_cache.begin() + static_cast<std::ptrdiff_t>(i));
				g_pvar_cache.push_back(hit);
				return hit.index; // shares the columns
			}
		}
	}
	// outside the lock: binds of other files go on
	VariantMetadataIndex parsed = ParseOrLoadVariantMetadata(path, func_name, have_stat, size, mtime_ns);
	if (have_stat) {
		std::lock_guard<std::mutex> lock(g_pvar_cache_mutex);
		g_pvar_cache.push_back(PvarCacheEntry {path, mtime_ns, size, parsed});
		if (g_pvar_cache.size() > kPvarCacheEntries) {
			g_pvar_cache.erase(g_pvar_cache.begin());
		}
	}
	return parsed;
}

// ---------------------------------------------------------------------------
// sample metadata
// ---------------------------------------------------------------------------

void SampleInfo::EnsureIidMap(const string &source_label) {
	static std::mutex build_mutex; // the object may be the cache's shared one: one builder, then read-only
	std::lock_guard<std::mutex> lock(build_mutex);
	if (!iid_to_idx.empty() || iids.empty()) {
		return;
	}
	for (idx_t i = 0; i < iids.size(); i++) {
		if (!iid_to_idx.emplace(iids[i], i).second) {
			throw InvalidInputException("%s: duplicate IID '%s' in sample file", source_label, iids[i]);
		}
	}
}

static bool IsMissingValue(const string &s) {
	return s.empty() || s == "NA" || s == "na" || s == "." || s == "-9" || s == "nan" || s == "NaN";
}

static SampleInfo ParseSampleMetadata(const string &path);

namespace {
struct PsamCacheEntry {
	string path;
	int64_t mtime_ns = 0, size = 0;
	shared_ptr<const SampleInfo> info;
};
std::mutex g_psam_cache_mutex;
vector<PsamCacheEntry> g_psam_cache; // most recently used last
} // namespace

shared_ptr<const SampleInfo> LoadSampleMetadata(ClientContext &, const string &path) {
	struct stat st;
	SynthSpec synth;
	const bool is_synth = ParseSynthPath(path, synth);
	const bool have_stat = !is_synth && ::stat(path.c_str(), &st) == 0;
	const int64_t mtime_ns = have_stat ? static_cast<int64_t>(st.st_mtim.tv_sec) * 1000000000LL + st.st_mtim.tv_nsec : 0;
	const int64_t size = have_stat ? static_cast<int64_t>(st.st_size) : -1;
	if (have_stat || is_synth) {
		std::lock_guard<std::mutex> lock(g_psam_cache_mutex);
		for (size_t i = 0; i < g_psam_cache.size(); i++) {
			if (g_psam_cache[i].path == path && g_psam_cache[i].mtime_ns == mtime_ns && g_psam_cache[i].size == size) {
				PsamCacheEntry hit = g_psam_cache[i];
				g_psam_cache.erase(g_psam_cache.begin() + static_cast<std::ptrdiff_t>(i));
				g_psam_cache.push_back(hit);
				return hit.info;
			}
		}
	}
	auto parsed = make_shared<SampleInfo>(ParseSampleMetadata(path)); // outside the lock
	if (have_stat || is_synth) {
		std::lock_guard<std::mutex> lock(g_psam_cache_mutex);
		g_psam_cache.push_back(PsamCacheEntry {path, mtime_ns, size, parsed});
		if (g_psam_cache.size() > kPvarCacheEntries) {
			g_psam_cache.erase(g_psam_cache.begin());
		}
	}
	return parsed;
}

static SampleInfo ParseSampleMetadata(const string &path) {
	SynthSpec synth;
	if (ParseSynthPath(path, synth)) {
		// '#FID IID SEX' rows of WriteSynthCompanions: F<s/4>, S<s>, 1 + (s & 1)
		SampleInfo info;
		info.column_names = {"FID", "IID", "SEX"};
		info.iids.reserve(synth.samples);
		info.fids.reserve(synth.samples);
		info.sexes.reserve(synth.samples);
		info.rows.reserve(synth.samples);
		for (uint32_t s = 0; s < synth.samples; s++) {
			info.fids.push_back("F" + std::to_string(s / 4));
			info.iids.push_back("S" + std::to_string(s));
			info.sexes.push_back(static_cast<uint8_t>(1 + (s & 1)));
			info.rows.push_back({info.fids.back(), info.iids.back(), (s & 1) ? "2" : "1"});
		}
		info.sample_ct = synth.samples;
		return info;
	}
	string content;
	if (!ReadWholeFile(path, content)) {
		throw IOException("cannot open .psam/.fam file '%s'", path);
	}
	SampleInfo info;
	auto lines = Lines(content);
	size_t li = 0;
	while (li < lines.size() && (lines[li].empty() || lines[li].compare(0, 2, "##") == 0)) {
		li++;
	}
	constexpr size_t kNone = static_cast<size_t>(-1);
	size_t iid_f = kNone, fid_f = kNone, sex_f = kNone;
	bool whitespace = false;
	if (li < lines.size() && !lines[li].empty() && lines[li][0] == '#') {
		auto fields = SplitFields(lines[li].substr(1), false);
		if (fields.size() == 1) {
			fields = SplitFields(lines[li].substr(1), true);
			whitespace = fields.size() > 1;
		}
		for (size_t i = 0; i < fields.size(); i++) {
			if (fields[i] == "IID") {
				iid_f = i;
			} else if (fields[i] == "FID") {
				fid_f = i;
			} else if (fields[i] == "SEX") {
				sex_f = i;
			}
		}
		if (iid_f == kNone) {
			throw InvalidInputException("'%s' missing required IID column", path);
		}
		info.column_names = fields;
		li++;
	} else {
		// .fam: FID IID PAT MAT SEX PHENO1
		whitespace = true;
		fid_f = 0;
		iid_f = 1;
		sex_f = 4;
		info.column_names = {"FID", "IID", "PAT", "MAT", "SEX", "PHENO1"};
	}
	for (; li < lines.size(); li++) {
		if (lines[li].empty()) {
			continue;
		}
		auto f = SplitFields(lines[li], whitespace);
		if (f.size() <= iid_f) {
			throw InvalidInputException("'%s': line %llu has too few fields", path,
			                            static_cast<unsigned long long>(li + 1));
		}
		info.iids.push_back(f[iid_f]);
		info.rows.push_back(f);
		if (fid_f != kNone) {
			info.fids.push_back(fid_f < f.size() ? f[fid_f] : "");
		}
		if (sex_f != kNone) {
			uint8_t code = 0;
			if (sex_f < f.size() && !IsMissingValue(f[sex_f])) {
				code = f[sex_f] == "1" ? 1 : (f[sex_f] == "2" ? 2 : 0);
			}
			info.sexes.push_back(code);
		}
	}
	info.sample_ct = info.iids.size();
	return info;
}

// ---------------------------------------------------------------------------
// samples parameter
// ---------------------------------------------------------------------------

// `samples :=` -- file indices of the named samples, in the caller's order.  Behaviour and messages are the
// reference's (src/plink_common.cpp:1161-1216: integers are 0-based file positions, strings are IIDs looked up in
// the .psam, a repeated sample is an error reported after every entry has resolved); the shape is this repo's: one
// resolver per element kind, picked once, and a bitmap over the file's samples for the repeats.
vector<uint32_t> ResolveSampleIndices(const Value &samples_val, uint32_t raw_sample_ct, const SampleInfo *sample_info,
                                      const string &func_name) {
	const bool is_list = !samples_val.IsNull() && samples_val.type().id() == LogicalTypeId::LIST;
	if (samples_val.IsNull() || (is_list && ListValue::GetChildren(samples_val).empty())) {
		throw InvalidInputException("%s: samples list must not be empty", func_name);
	}
	const LogicalTypeId kind = is_list ? ListType::GetChildType(samples_val.type()).id() : LogicalTypeId::SQLNULL;
	std::function<uint32_t(const Value &)> resolve;
	if (kind == LogicalTypeId::INTEGER || kind == LogicalTypeId::BIGINT) {
		resolve = [&](const Value &entry) {
			const int64_t idx = entry.GetValue<int64_t>();
			if (idx < 0 || static_cast<uint64_t>(idx) >= raw_sample_ct) {
				throw InvalidInputException("%s: sample index %lld out of range (sample count: %u)", func_name,
				                            static_cast<long long>(idx), raw_sample_ct);
			}
			return static_cast<uint32_t>(idx);
		};
	} else if (kind == LogicalTypeId::VARCHAR) {
		if (!sample_info) {
			throw InvalidInputException("%s: samples parameter requires LIST(INTEGER) when no .psam "
			                            "is available (no sample IDs to match against)",
			                            func_name);
		}
		const_cast<SampleInfo *>(sample_info)->EnsureIidMap(func_name);
		resolve = [&](const Value &entry) {
			const string iid = entry.GetValue<string>();
			const auto hit = sample_info->iid_to_idx.find(iid);
			if (hit == sample_info->iid_to_idx.end()) {
				throw InvalidInputException("%s: sample '%s' not found in .psam", func_name, iid);
			}
			return static_cast<uint32_t>(hit->second);
		};
	} else {
		throw InvalidInputException("%s: samples parameter must be LIST(VARCHAR) or LIST(INTEGER)", func_name);
	}
	vector<uint32_t> indices;
	for (const Value &entry : ListValue::GetChildren(samples_val)) {
		indices.push_back(resolve(entry));
	}
	vector<bool> taken(raw_sample_ct, false);
	for (const uint32_t idx : indices) {
		if (taken[idx]) {
			throw InvalidInputException("%s: duplicate sample index %u in samples list", func_name, idx);
		}
		taken[idx] = true;
	}
	return indices;
}

SampleSubset BuildSampleSubset(uint32_t raw_sample_ct, const vector<uint32_t> &sample_indices) {
	SampleSubset r;
	r.raw_sample_ct = raw_sample_ct;
	r.subset_sample_ct = static_cast<uint32_t>(sample_indices.size());
	r.sample_include.assign((raw_sample_ct + 63) / 64, 0);
	for (auto idx : sample_indices) {
		r.sample_include[idx >> 6] |= 1ull << (idx & 63);
	}
	r.sorted_indices = sample_indices;
	std::sort(r.sorted_indices.begin(), r.sorted_indices.end());
	return r;
}

// ---------------------------------------------------------------------------
// variants := ...
// ---------------------------------------------------------------------------

namespace {

// One `variants` parameter being resolved: owns the lazily built ID -> index map.
class VariantSelector {
public:
	VariantSelector(const VariantMetadataIndex &variants, uint32_t raw_variant_ct, const string &func_name)
	    : variants_(variants), raw_variant_ct_(raw_variant_ct), func_(func_name) {
	}

	uint32_t FromIndex(int64_t idx) const {
		if (idx < 0 || idx >= static_cast<int64_t>(raw_variant_ct_)) {
			throw InvalidInputException("%s: variant index %lld out of range (variant count: %u)", func_,
			                            static_cast<long long>(idx), raw_variant_ct_);
		}
		return static_cast<uint32_t>(idx);
	}

	// an rsid, or CHROM:POS / CHROM:POS:REF:ALT when the text carries a ':'
	uint32_t FromText(const string &text) {
		if (text.find(':') == string::npos) {
			if (by_id_.empty()) {
				by_id_.reserve(variants_.ids().size());
				for (idx_t v = 0; v < variants_.ids().size(); v++) {
					if (!variants_.ids()[v].empty()) {
						by_id_[variants_.ids()[v]] = static_cast<uint32_t>(v); // duplicates: last one wins
					}
				}
			}
			auto hit = by_id_.find(text);
			if (hit == by_id_.end()) {
				throw InvalidInputException("%s: variant '%s' not found", func_, text);
			}
			return hit->second;
		}
		vector<string> parts;
		size_t from = 0;
		while (true) {
			size_t colon = text.find(':', from);
			parts.push_back(text.substr(from, colon == string::npos ? string::npos : colon - from));
			if (colon == string::npos) {
				break;
			}
			from = colon + 1;
		}
		if (parts.size() != 2 && parts.size() != 4) {
			throw InvalidInputException("%s: invalid CPRA format '%s' (expected CHROM:POS or CHROM:POS:REF:ALT)", func_,
			                            text);
		}
		errno = 0;
		char *tail = nullptr;
		long pos = std::strtol(parts[1].c_str(), &tail, 10);
		if (*tail != '\0' || errno != 0) {
			throw InvalidInputException("%s: invalid position in CPRA '%s'", func_, text);
		}
		const bool alleles = parts.size() == 4;
		return FromLocus(parts[0], static_cast<int32_t>(pos), alleles ? &parts[2] : nullptr,
		                 alleles ? &parts[3] : nullptr, text);
	}

	uint32_t FromLocusStruct(const Value &val) {
		auto &fields = StructType::GetChildTypes(val.type());
		auto &kids = StructValue::GetChildren(val);
		string chrom, ref, alt;
		int32_t pos = 0;
		bool has_ref = false, has_alt = false;
		for (idx_t i = 0; i < fields.size(); i++) {
			const string &name = fields[i].first;
			if (name == "chrom") {
				chrom = kids[i].GetValue<string>();
			} else if (name == "pos") {
				pos = kids[i].GetValue<int32_t>();
			} else if (name == "ref") {
				ref = kids[i].GetValue<string>();
				has_ref = true;
			} else if (name == "alt") {
				alt = kids[i].GetValue<string>();
				has_alt = true;
			}
		}
		const bool alleles = has_ref && has_alt;
		string desc = chrom + ":" + std::to_string(pos);
		if (alleles) {
			desc += ":" + ref + ":" + alt;
		}
		return FromLocus(chrom, pos, alleles ? &ref : nullptr, alleles ? &alt : nullptr, desc);
	}

	vector<uint32_t> FromRangeStruct(const Value &val) {
		auto &fields = StructType::GetChildTypes(val.type());
		auto &kids = StructValue::GetChildren(val);
		const Value *start = nullptr, *stop = nullptr;
		for (idx_t i = 0; i < fields.size(); i++) {
			if (fields[i].first == "start") {
				start = &kids[i];
			} else if (fields[i].first == "stop") {
				stop = &kids[i];
			}
		}
		if (!start || !stop) {
			throw InvalidInputException("%s: range struct must have 'start' and 'stop' fields", func_);
		}
		uint32_t lo, hi;
		auto kind = start->type().id();
		if (kind == LogicalTypeId::INTEGER || kind == LogicalTypeId::BIGINT) {
			lo = FromIndex(start->GetValue<int64_t>());
			hi = FromIndex(stop->GetValue<int64_t>());
		} else if (kind == LogicalTypeId::VARCHAR) {
			lo = FromText(start->GetValue<string>());
			hi = FromText(stop->GetValue<string>());
		} else {
			throw InvalidInputException("%s: range struct start/stop must be INTEGER or VARCHAR", func_);
		}
		if (lo > hi) {
			throw InvalidInputException("%s: variants range start (%u) is after stop (%u)", func_, lo, hi);
		}
		vector<uint32_t> out(hi - lo + 1);
		for (uint32_t v = lo; v <= hi; v++) {
			out[v - lo] = v;
		}
		return out;
	}

private:
	// POS is ascending inside a CHROM run: lower_bound, then walk the ties for the alleles.
	uint32_t FromLocus(const string &chrom, int32_t pos, const string *ref, const string *alt, const string &desc) const {
		auto run = variants_.chrom_offsets().find(chrom);
		if (run != variants_.chrom_offsets().end()) {
			auto first = variants_.positions().begin() + static_cast<std::ptrdiff_t>(run->second.first);
			auto last = variants_.positions().begin() + static_cast<std::ptrdiff_t>(run->second.second);
			for (auto it = std::lower_bound(first, last, pos); it != last && *it == pos; ++it) {
				idx_t v = static_cast<idx_t>(it - variants_.positions().begin());
				if (!ref || (variants_.refs()[v] == *ref && variants_.alts()[v] == *alt)) {
					return static_cast<uint32_t>(v);
				}
			}
		}
		throw InvalidInputException("%s: variant '%s' not found", func_, desc);
	}

	const VariantMetadataIndex &variants_;
	uint32_t raw_variant_ct_;
	const string &func_;
	std::unordered_map<string, uint32_t> by_id_;
};

} // namespace

vector<uint32_t> ResolveVariantsParameter(const Value &val, const VariantMetadataIndex &variants,
                                          uint32_t raw_variant_ct, const string &func_name) {
	if (val.IsNull()) {
		throw InvalidInputException("%s: variants must not be NULL", func_name);
	}
	VariantSelector pick(variants, raw_variant_ct, func_name);
	vector<uint32_t> out;
	auto has_field = [](const LogicalType &t, const char *name) {
		for (auto &f : StructType::GetChildTypes(t)) {
			if (f.first == name) {
				return true;
			}
		}
		return false;
	};
	const auto &type = val.type();
	switch (type.id()) {
	case LogicalTypeId::INTEGER:
	case LogicalTypeId::BIGINT:
		out.push_back(pick.FromIndex(val.GetValue<int64_t>()));
		break;
	case LogicalTypeId::VARCHAR:
		out.push_back(pick.FromText(val.GetValue<string>()));
		break;
	case LogicalTypeId::STRUCT: {
		const bool range = has_field(type, "start"), locus = has_field(type, "chrom");
		if (range && locus) {
			throw InvalidInputException("%s: ambiguous variants struct — has both 'start' and 'chrom' fields. "
			                            "Use {start:, stop:} for a range or {chrom:, pos:} for a CPRA lookup.",
			                            func_name);
		}
		if (range) {
			out = pick.FromRangeStruct(val);
		} else if (locus) {
			out.push_back(pick.FromLocusStruct(val));
		} else {
			throw InvalidInputException(
			    "%s: variants struct must have either 'start'/'stop' (range) or 'chrom'/'pos' (CPRA) fields", func_name);
		}
		break;
	}
	case LogicalTypeId::LIST: {
		auto &kids = ListValue::GetChildren(val);
		if (kids.empty()) {
			throw InvalidInputException("%s: variants list must not be empty", func_name);
		}
		auto &elem = ListType::GetChildType(type);
		for (auto &kid : kids) {
			switch (elem.id()) {
			case LogicalTypeId::INTEGER:
			case LogicalTypeId::BIGINT:
				out.push_back(pick.FromIndex(kid.GetValue<int64_t>()));
				break;
			case LogicalTypeId::VARCHAR:
				out.push_back(pick.FromText(kid.GetValue<string>()));
				break;
			case LogicalTypeId::STRUCT:
				out.push_back(pick.FromLocusStruct(kid));
				break;
			default:
				throw InvalidInputException("%s: variants list elements must be INTEGER, VARCHAR, or STRUCT (got %s)",
				                            func_name, elem.ToString());
			}
		}
		break;
	}
	default:
		throw InvalidInputException("%s: variants parameter must be an integer, string, struct, or list (got %s)",
		                            func_name, type.ToString());
	}
	vector<uint32_t> sorted = out;
	std::sort(sorted.begin(), sorted.end());
	auto dup = std::adjacent_find(sorted.begin(), sorted.end());
	if (dup != sorted.end()) {
		// the reference names the first repeat met in caller order
		std::unordered_set<uint32_t> seen;
		for (auto v : out) {
			if (!seen.insert(v).second) {
				throw InvalidInputException("%s: duplicate variant index %u in variants parameter", func_name, v);
			}
		}
	}
	return out;
}

// ---------------------------------------------------------------------------
// region
// ---------------------------------------------------------------------------

VariantRange ParseRegion(const string &region_str, const VariantMetadataIndex &variants, const string &func_name) {
	auto colon = region_str.find(':');
	if (colon == string::npos || colon == 0) {
		throw InvalidInputException("%s: invalid region format '%s' (expected 'chr:start-end')", func_name, region_str);
	}
	string chrom = region_str.substr(0, colon);
	string range_part = region_str.substr(colon + 1);
	auto dash = range_part.find('-');
	if (dash == string::npos) {
		throw InvalidInputException("%s: invalid region format '%s' (expected 'chr:start-end')", func_name, region_str);
	}
	string start_str = range_part.substr(0, dash);
	string end_str = range_part.substr(dash + 1);
	char *parse_end;
	errno = 0;
	long start_pos = std::strtol(start_str.c_str(), &parse_end, 10);
	if (parse_end == start_str.c_str() || *parse_end != '\0' || errno != 0 || start_pos < 0) {
		throw InvalidInputException("%s: invalid region start position in '%s'", func_name, region_str);
	}
	errno = 0;
	long end_pos = std::strtol(end_str.c_str(), &parse_end, 10);
	if (parse_end == end_str.c_str() || *parse_end != '\0' || errno != 0 || end_pos < 0) {
		throw InvalidInputException("%s: invalid region end position in '%s'", func_name, region_str);
	}
	VariantRange range;
	range.has_filter = true;
	auto it = variants.chrom_offsets().find(chrom); // exact string match, no chr-prefix normalisation
	if (it == variants.chrom_offsets().end()) {
		return range; // empty
	}
	auto first = variants.positions().begin() + static_cast<std::ptrdiff_t>(it->second.first);
	auto last = variants.positions().begin() + static_cast<std::ptrdiff_t>(it->second.second);
	auto lo = std::lower_bound(first, last, static_cast<int32_t>(start_pos));
	auto hi = std::upper_bound(first, last, static_cast<int32_t>(end_pos));
	range.start_idx = static_cast<uint32_t>(lo - variants.positions().begin());
	range.end_idx = static_cast<uint32_t>(hi - variants.positions().begin());
	if (range.end_idx < range.start_idx) {
		range.end_idx = range.start_idx;
	}
	return range;
}

// ---------------------------------------------------------------------------
// read_pgen filters
// ---------------------------------------------------------------------------

void GenotypeRangeFilter::SetFromRange(const RangeFilter &r, bool inc_missing) {
	for (int g = 0; g <= 2; g++) {
		allowed[g] = r.Passes(static_cast<double>(g));
	}
	include_missing = inc_missing;
	active = r.active;
}

// af_range / ac_range / genotype_range: a STRUCT of optional bounds (and, for genotype_range, include_missing).
// Contract and messages are the reference's (src/plink_common.cpp:1340-1391); here the accepted fields are a table
// and the walk over the STRUCT looks each one up.
RangeFilter ParseRangeFilter(const Value &val, const string &param_name, double valid_min, double valid_max,
                             const string &func_name, bool *include_missing_out) {
	if (val.type().id() != LogicalTypeId::STRUCT) {
		throw InvalidInputException("%s: %s must be a STRUCT (e.g. {min: 0.0, max: 0.5})", func_name, param_name);
	}
	RangeFilter result;
	enum class Field { LOWER, UPPER, MISSING_FLAG };
	struct Accepted {
		const char *name;
		Field what;
	};
	static const Accepted kBounds[] = {{"min", Field::LOWER}, {"max", Field::UPPER}, {"include_missing", Field::MISSING_FLAG}};
	const size_t n_accepted = include_missing_out ? 3 : 2; // the flag belongs to genotype_range only
	const auto &names = StructType::GetChildTypes(val.type());
	const auto &values = StructValue::GetChildren(val);
	if (values.empty()) {
		return result;
	}
	for (idx_t i = 0; i < names.size(); i++) {
		const string &field = names[i].first;
		const Accepted *hit = nullptr;
		for (size_t k = 0; k < n_accepted && !hit; k++) {
			hit = field == kBounds[k].name ? &kBounds[k] : nullptr;
		}
		if (!hit) {
			throw InvalidInputException("%s: %s has unknown field '%s' (expected %s)", func_name, param_name, field,
			                            include_missing_out ? "'min', 'max', and/or 'include_missing'"
			                                                : "'min' and/or 'max'");
		}
		if (values[i].IsNull()) {
			continue;
		}
		if (hit->what == Field::MISSING_FLAG) {
			*include_missing_out = values[i].GetValue<bool>();
			continue;
		}
		const double bound = values[i].GetValue<double>();
		if (bound < valid_min || bound > valid_max) {
			throw InvalidInputException("%s: %s.%s value %g is out of range [%g, %g]", func_name, param_name, field, bound,
			                            valid_min, valid_max);
		}
		(hit->what == Field::LOWER ? result.min : result.max) = bound;
	}
	if (result.min > result.max) {
		throw InvalidInputException("%s: %s min (%g) > max (%g)", func_name, param_name, result.min, result.max);
	}
	result.active = true;
	return result;
}

// include_genotypes := ['het', ...] (src/plink_common.cpp:1393-1436): category names, case and blanks ignored
void ParseIncludeGenotypes(const Value &val, GenotypeRangeFilter &out, const string &func_name) {
	if (val.IsNull()) {
		return;
	}
	if (val.type().id() != LogicalTypeId::LIST) {
		throw InvalidInputException("%s: include_genotypes must be a LIST of category names "
		                            "(e.g. ['het', 'hom_alt'])",
		                            func_name);
	}
	static const char *const kCalls[3] = {"hom_ref", "het", "hom_alt"};
	bool any = false;
	for (const Value &entry : ListValue::GetChildren(val)) {
		if (entry.IsNull()) {
			throw InvalidInputException("%s: include_genotypes contains a NULL category name", func_name);
		}
		const string label = Lower(Trim(entry.GetValue<string>()));
		int call = -1;
		for (int g = 0; g < 3 && call < 0; g++) {
			call = label == kCalls[g] ? g : -1;
		}
		if (call >= 0) {
			out.allowed[call] = true;
		} else if (label == "missing") {
			out.include_missing = true;
		} else {
			throw InvalidInputException("%s: include_genotypes has unknown category '%s' "
			                            "(expected 'hom_ref', 'het', 'hom_alt', and/or 'missing')",
			                            func_name, label);
		}
		any = true;
	}
	out.active = out.active || any;
}

PreDecompFilterResult CheckPreDecompFilters(const CountFilter &count_filter, const GenotypeRangeFilter &genotype_filter,
                                            const uint32_t c[4], uint32_t) {
	PreDecompFilterResult result;
	if (count_filter.HasFilter()) {
		uint32_t non_missing = c[0] + c[1] + c[2];
		if (non_missing == 0) {
			result.skip = true;
			return result;
		}
		uint32_t ac = c[1] + 2 * c[2];
		if (count_filter.ac_filter.active && !count_filter.ac_filter.Passes(static_cast<double>(ac))) {
			result.skip = true;
			return result;
		}
		if (count_filter.af_filter.active) {
			double af = static_cast<double>(ac) / (2.0 * static_cast<double>(non_missing));
			if (!count_filter.af_filter.Passes(af)) {
				result.skip = true;
				return result;
			}
		}
	}
	if (genotype_filter.active) {
		bool any_pass = false, all_pass = true;
		for (int g = 0; g <= 2; g++) {
			if (genotype_filter.allowed[g] && c[g] > 0) {
				any_pass = true;
			}
			if (!genotype_filter.allowed[g] && c[g] > 0) {
				all_pass = false;
			}
		}
		if (genotype_filter.include_missing && c[3] > 0) {
			any_pass = true;
		}
		if (!any_pass) {
			result.skip = true;
			return result;
		}
		result.all_pass = all_pass;
	}
	return result;
}

GenotypeMode ResolveGenotypeMode(const string &mode_str, uint32_t sample_ct, const string &func_name) {
	auto mode = Lower(mode_str);
	if (mode == "auto") {
		return sample_ct <= ArrayType::MAX_ARRAY_SIZE ? GenotypeMode::ARRAY : GenotypeMode::LIST;
	} else if (mode == "array") {
		if (sample_ct > ArrayType::MAX_ARRAY_SIZE) {
			throw InvalidInputException("%s: genotypes := 'array' requires sample count (%u) <= %u. "
			                            "Use genotypes := 'list' or genotypes := 'auto' for large cohorts.",
			                            func_name, sample_ct, static_cast<uint32_t>(ArrayType::MAX_ARRAY_SIZE));
		}
		return GenotypeMode::ARRAY;
	} else if (mode == "list") {
		return GenotypeMode::LIST;
	} else if (mode == "columns") {
		return GenotypeMode::COLUMNS;
	} else if (mode == "struct") {
		return GenotypeMode::STRUCT;
	} else if (mode == "counts") {
		return GenotypeMode::COUNTS;
	} else if (mode == "stats") {
		return GenotypeMode::STATS;
	}
	throw InvalidInputException(
	    "%s: invalid genotypes value '%s' (expected 'auto', 'array', 'list', 'columns', 'struct', 'counts', or "
	    "'stats')",
	    func_name, mode_str);
}

LogicalType MakeGenotypeCountsType() {
	return LogicalType::STRUCT({{"hom_ref", LogicalType::UINTEGER},
	                            {"het", LogicalType::UINTEGER},
	                            {"hom_alt", LogicalType::UINTEGER},
	                            {"missing", LogicalType::UINTEGER}});
}

LogicalType MakeGenotypeStatsType() {
	return LogicalType::STRUCT({{"hom_ref", LogicalType::UINTEGER},
	                            {"het", LogicalType::UINTEGER},
	                            {"hom_alt", LogicalType::UINTEGER},
	                            {"missing", LogicalType::UINTEGER},
	                            {"n", LogicalType::UINTEGER},
	                            {"af", LogicalType::DOUBLE},
	                            {"maf", LogicalType::DOUBLE},
	                            {"missing_rate", LogicalType::DOUBLE},
	                            {"carrier_count", LogicalType::UINTEGER},
	                            {"het_rate", LogicalType::DOUBLE}});
}

// ---------------------------------------------------------------------------
// PCA normalisation, threads
// ---------------------------------------------------------------------------

VariantNorm ComputeVariantNorm(double alt_freq) {
	VariantNorm norm;
	if (alt_freq <= 0.0 || alt_freq >= 1.0) {
		return norm;
	}
	norm.center = 2.0 * alt_freq;
	norm.inv_stdev = 1.0 / std::sqrt(2.0 * alt_freq * (1.0 - alt_freq));
	norm.skip = false;
	return norm;
}

uint32_t GetPlinkingMaxThreads(ClientContext &context) {
	Value val;
	if (context.TryGetCurrentSetting("plinking_max_threads", val)) {
		auto v = val.GetValue<int64_t>();
		if (v > 0) {
			return static_cast<uint32_t>(v);
		}
	}
	return 0;
}

idx_t ApplyMaxThreadsCap(idx_t computed, uint32_t config_max_threads) {
	if (config_max_threads > 0) {
		return std::min<idx_t>(computed, config_max_threads);
	}
	return std::min<idx_t>(computed, 16);
}

// ---------------------------------------------------------------------------
// ploidy / sex
// ---------------------------------------------------------------------------

ParBounds ResolveParBounds(const string &build, const string &func_name) {
	string norm;
	for (char c : Lower(build)) {
		if (c != '-' && c != '_' && c != ' ' && c != '.') {
			norm.push_back(c);
		}
	}
	ParBounds pb;
	if (norm.empty() || norm == "none") {
		return pb;
	}
	if (norm == "grch38" || norm == "hg38" || norm == "b38" || norm == "38") {
		pb.par1_end = 2781479;
		pb.par2_start = 155701383;
		pb.par2_end = 156030895;
		pb.active = true;
		return pb;
	}
	if (norm == "grch37" || norm == "hg19" || norm == "b37" || norm == "37") {
		pb.par1_end = 2699520;
		pb.par2_start = 154931044;
		pb.par2_end = 155260560;
		pb.active = true;
		return pb;
	}
	throw InvalidInputException("%s: unrecognized build '%s' (expected 'GRCh38'/'hg38', 'GRCh37'/'hg19', or 'none')",
	                            func_name, build);
}

namespace {
//! What a chromosome NAME says about ploidy (case and a "chr" prefix do not matter; PLINK's numeric codes
//! 23-26 count): X still depends on the position (the PARs are diploid), everything not listed is diploid.
ChromPloidy ChromNameClass(const string &chrom) {
	static const struct {
		const char *name;
		ChromPloidy cls;
	} kNames[] = {{"x", ChromPloidy::CHR_X},   {"23", ChromPloidy::CHR_X},  {"y", ChromPloidy::CHR_Y},
	              {"24", ChromPloidy::CHR_Y},  {"mt", ChromPloidy::CHR_MT}, {"m", ChromPloidy::CHR_MT},
	              {"26", ChromPloidy::CHR_MT}}; // par1 / par2 / xy / 25 and the autosomes: diploid
	size_t from = 0;
	if (chrom.size() >= 3 && (chrom[0] | 0x20) == 'c' && (chrom[1] | 0x20) == 'h' && (chrom[2] | 0x20) == 'r') {
		from = 3;
	}
	const size_t len = chrom.size() - from;
	for (auto &e : kNames) {
		if (std::strlen(e.name) != len) {
			continue;
		}
		bool same = true;
		for (size_t i = 0; i < len && same; i++) {
			const char c = chrom[from + i];
			same = ((c >= 'A' && c <= 'Z') ? static_cast<char>(c | 0x20) : c) == e.name[i];
		}
		if (same) {
			return e.cls;
		}
	}
	return ChromPloidy::AUTOSOMAL;
}

inline bool InPar(int32_t pos, const ParBounds &par) {
	return par.active && ((pos > 0 && pos <= par.par1_end) || (pos >= par.par2_start && pos <= par.par2_end));
}
} // namespace

ChromPloidy ClassifyChromPloidy(const string &chrom, int32_t pos, const ParBounds &par) {
	const ChromPloidy cls = ChromNameClass(chrom);
	return cls == ChromPloidy::CHR_X && InPar(pos, par) ? ChromPloidy::AUTOSOMAL : cls;
}

PloidyMap::PloidyMap(const VariantMetadataIndex &variants, const ParBounds &par) : cols_(variants.cols), par_(par) {
	for (auto &kv : variants.chrom_offsets()) {
		const ChromPloidy cls = ChromNameClass(kv.first);
		if (cls != ChromPloidy::AUTOSOMAL) {
			runs_.push_back(Run {static_cast<uint32_t>(kv.second.first), static_cast<uint32_t>(kv.second.second), cls});
		}
	}
	std::sort(runs_.begin(), runs_.end(), [](const Run &a, const Run &b) { return a.begin < b.begin; });
}

ChromPloidy PloidyMap::At(uint32_t vidx) const {
	for (auto &r : runs_) { // a handful at most (X, Y, MT)
		if (vidx >= r.begin && vidx < r.end) {
			return r.name_class == ChromPloidy::CHR_X && InPar(cols_->positions[vidx], par_) ? ChromPloidy::AUTOSOMAL
			                                                                                  : r.name_class;
		}
	}
	return ChromPloidy::AUTOSOMAL;
}

bool PloidyMap::NonAutosomalSpan(uint32_t begin, uint32_t end, uint32_t &first, uint32_t &last) const {
	bool any = false;
	for (auto &r : runs_) {
		uint32_t a = std::max(begin, r.begin), b = std::min(end, r.end);
		if (r.name_class == ChromPloidy::CHR_X && par_.active) {
			// positions ascend inside a run: trim the PAR rows off both ends
			while (a < b && InPar(cols_->positions[a], par_)) {
				a++;
			}
			while (b > a && InPar(cols_->positions[b - 1], par_)) {
				b--;
			}
		}
		if (a >= b) {
			continue;
		}
		first = any ? std::min(first, a) : a;
		last = any ? std::max(last, b) : b;
		any = true;
	}
	return any;
}

vector<uint8_t> BuildAlignedSex(const SampleInfo &sample_info, const vector<uint32_t> *subset_sorted) {
	if (sample_info.sexes.empty()) {
		return {};
	}
	if (!subset_sorted) {
		return sample_info.sexes;
	}
	vector<uint8_t> aligned;
	aligned.reserve(subset_sorted->size());
	for (uint32_t idx : *subset_sorted) {
		aligned.push_back(idx < sample_info.sexes.size() ? sample_info.sexes[idx] : uint8_t {0});
	}
	return aligned;
}

SexAwareCounts SexAwareFromStrata(ChromPloidy ploidy, const uint32_t total[4], const uint32_t male[4],
                                  const uint32_t female[4], bool have_sex) {
	SexAwareCounts r;
	if ((ploidy == ChromPloidy::CHR_X || ploidy == ChromPloidy::CHR_Y) && !have_sex) {
		r.sex_unavailable = true;
		return r;
	}
	auto haploid = [&](const uint32_t c[4]) {
		// one allele per sample; a heterozygous hardcall is invalid -> missing
		r.obs_allele_ct += c[0] + c[2];
		r.alt_allele_ct += c[2];
		r.geno_hom_ref += c[0];
		r.geno_hom_alt += c[2];
		r.geno_missing += c[1] + c[3];
	};
	auto diploid = [&](const uint32_t c[4]) {
		r.obs_allele_ct += 2 * (c[0] + c[1] + c[2]);
		r.alt_allele_ct += c[1] + 2 * c[2];
		r.hwe_hom_ref += c[0];
		r.hwe_het += c[1];
		r.hwe_hom_alt += c[2];
		r.geno_hom_ref += c[0];
		r.geno_het += c[1];
		r.geno_hom_alt += c[2];
		r.geno_missing += c[3];
	};
	const uint32_t n_total = total[0] + total[1] + total[2] + total[3];
	const uint32_t n_male = male[0] + male[1] + male[2] + male[3];
	const uint32_t n_female = female[0] + female[1] + female[2] + female[3];
	switch (ploidy) {
	case ChromPloidy::CHR_MT:
		haploid(total);
		break;
	case ChromPloidy::CHR_Y:
		haploid(male);
		r.geno_missing += n_total - n_male; // females and unknown sex carry no Y
		break;
	case ChromPloidy::CHR_X:
		diploid(female);
		haploid(male);
		r.geno_missing += n_total - n_male - n_female; // unknown sex: ploidy undetermined
		break;
	case ChromPloidy::AUTOSOMAL:
	default:
		diploid(total);
		break;
	}
	r.hwe_defined = (ploidy == ChromPloidy::CHR_X || ploidy == ChromPloidy::AUTOSOMAL);
	return r;
}

// ---------------------------------------------------------------------------
// libpgenhip wrappers
// ---------------------------------------------------------------------------

void ThrowOnPghError(int rc, const char *errbuf, const string &func_name, const string &what) {
	if (rc == PGH_OK) {
		return;
	}
	if (rc == PGH_ERR_ARG) {
		throw InvalidInputException("%s: %s: %s", func_name, what, string(errbuf));
	}
	throw IOException("%s: %s: %s", func_name, what, string(errbuf));
}

pgh_info ProbePgen(const string &pgen_path, const string &func_name) {
	pgh_info info;
	SynthSpec synth;
	if (ParseSynthPath(pgen_path, synth)) {
		std::memset(&info, 0, sizeof info);
		info.raw_variant_ct = synth.variants;
		info.raw_sample_ct = synth.samples;
		info.variant_end = synth.variants;
		info.record_bytes = (synth.samples + 3) / 4;
		info.max_record_bytes = info.record_bytes;
		info.vrtype_hist[0] = synth.variants;
		info.device = -1;
		return info;
	}
	char errbuf[PGH_ERRBUF_LEN] = {0};
	int rc = pgh_probe(pgen_path.c_str(), nullptr, &info, errbuf);
	if (rc != PGH_OK) {
		throw IOException("%s: failed to open '%s': %s", func_name, pgen_path, string(errbuf));
	}
	return info;
}

DeviceDataset::~DeviceDataset() {
	tallies_.clear(); // passes read the matrix: they go first (queries that still hold one keep the dataset too)
	if (handle) {
		pgh_close(handle);
	}
}

namespace {
uint64_t CacheBudgetBytes();
}

// ---- pinned host blocks ----------------------------------------------------------------------------

namespace {
struct PinnedBlock {
	void *p;
	size_t bytes;
};
std::mutex g_pinned_mutex;
vector<PinnedBlock> g_pinned_free;
size_t g_pinned_free_bytes = 0;
size_t PinnedPoolLimit() {
	const char *env = std::getenv("PLINKING_PINNED_POOL_GB");
	const double gb = env ? std::atof(env) : 24.0;
	return static_cast<size_t>(gb * 1e9);
}
} // namespace

void *PinnedPoolAcquire(size_t bytes, size_t &got_bytes) {
	{
		// the smallest parked block that is large enough, and not more than twice what was asked for
		std::lock_guard<std::mutex> lock(g_pinned_mutex);
		size_t best = g_pinned_free.size();
		for (size_t i = 0; i < g_pinned_free.size(); i++) {
			if (g_pinned_free[i].bytes >= bytes && g_pinned_free[i].bytes <= 2 * bytes + (1u << 20) &&
			    (best == g_pinned_free.size() || g_pinned_free[i].bytes < g_pinned_free[best].bytes)) {
				best = i;
			}
		}
		if (best < g_pinned_free.size()) {
			PinnedBlock b = g_pinned_free[best];
			g_pinned_free.erase(g_pinned_free.begin() + static_cast<std::ptrdiff_t>(best));
			g_pinned_free_bytes -= b.bytes;
			got_bytes = b.bytes;
			return b.p;
		}
	}
	void *q = nullptr;
	char errbuf[PGH_ERRBUF_LEN] = {0};
	if (pgh_host_alloc(bytes, &q, errbuf) != PGH_OK) {
		throw IOException("cannot allocate %llu bytes of pinned host memory: %s", static_cast<unsigned long long>(bytes),
		                  string(errbuf));
	}
	got_bytes = bytes;
	return q;
}

void PinnedPoolRelease(void *p, size_t bytes) {
	if (!p) {
		return;
	}
	{
		std::lock_guard<std::mutex> lock(g_pinned_mutex);
		if (g_pinned_free_bytes + bytes <= PinnedPoolLimit()) {
			g_pinned_free.push_back(PinnedBlock {p, bytes});
			g_pinned_free_bytes += bytes;
			return;
		}
	}
	pgh_host_free(p);
}

// ---- tally passes ------------------------------------------------------------------------------

bool GetPlinkingTallyCache(ClientContext &context) {
	const char *env = std::getenv("PLINKING_TALLY_CACHE");
	if (env && env[0] == '0') {
		return false;
	}
	Value val;
	if (context.TryGetCurrentSetting("plinking_tally_cache", val) && !val.IsNull()) {
		return val.GetValue<bool>();
	}
	return true;
}

namespace {
//! One window of a streamed file, opened (pgh_open of the variant range, the sample subset staged next to it).
struct OpenedWindow {
	pgh_dataset *ds = nullptr;
	pgh_subset *ss = nullptr;
	uint32_t v0 = 0, v1 = 0;
	int rc = PGH_OK;
	string error;
	void Close() {
		if (ss) {
			pgh_subset_destroy(ss);
			ss = nullptr;
		}
		if (ds) {
			pgh_close(ds);
			ds = nullptr;
		}
	}
};

OpenedWindow OpenWindow(const string &path, const vector<uint64_t> *sample_include, uint32_t v0, uint32_t v1) {
	OpenedWindow w;
	w.v0 = v0;
	w.v1 = v1;
	char errbuf[PGH_ERRBUF_LEN] = {0};
	w.rc = pgh_open(path.c_str(), nullptr, v0, v1, &w.ds, errbuf);
	if (w.rc == PGH_OK && sample_include && !sample_include->empty()) {
		w.rc = pgh_subset_create(w.ds, sample_include->data(), &w.ss, errbuf);
	}
	if (w.rc != PGH_OK) {
		w.error = errbuf;
		w.Close();
	}
	return w;
}
} // namespace

struct DeviceTally::Streamed {
	std::thread producer;
	std::mutex m;
	std::condition_variable cv;
	uint32_t done_upto = 0; // variants below this have landed
	bool stop = false, failed = false;
	string error;
	vector<uint32_t> counts;  // [end - begin][4]
	vector<double> lnp[2];    // [end - begin]
	vector<uint32_t> missing; // per included sample, summed over the windows
};

DeviceTally::DeviceTally(DeviceDataset &ds, const vector<uint64_t> *sample_include, uint32_t begin_p, uint32_t end_p,
                         uint32_t products, const string &func_name)
    : begin(begin_p), end(end_p) {
	if (sample_include) {
		mask = *sample_include;
	}
	if (ds.streamed) {
		// every product, always: next to reading the file a second time for a later request they cost nothing
		streamed_ = make_uniq<Streamed>();
		Streamed &st = *streamed_;
		const size_t n = end - begin;
		st.done_upto = begin;
		st.counts.assign(4 * n, 0);
		st.lnp[0].assign(n, 0.0);
		st.lnp[1].assign(n, 0.0);
		uint32_t n_out = ds.info.raw_sample_ct;
		if (sample_include) {
			n_out = 0;
			for (uint64_t w : mask) {
				n_out += static_cast<uint32_t>(__builtin_popcountll(w));
			}
		}
		st.missing.assign(n_out, 0);
		counts_ = reinterpret_cast<const uint32_t(*)[4]>(st.counts.data());
		lnp_[0] = st.lnp[0].data();
		lnp_[1] = st.lnp[1].data();
		// a window: half the HBM budget (the other half is the pass's and the ingest's working room), whole variants
		const uint64_t pitch = std::max<uint64_t>(16, (static_cast<uint64_t>(ds.info.record_bytes) + 127) / 128 * 128);
		const uint64_t window = std::max<uint64_t>(1, CacheBudgetBytes() / 2 / pitch);
		const string path = ds.path, fn = func_name;
		const uint32_t sample_ct = ds.info.raw_sample_ct;
		st.producer = std::thread([this, path, sample_ct, window, fn] { RunStream(path, sample_ct, window, fn); });
		return;
	}
	if (sample_include) {
		subset_ = make_uniq<DeviceSubset>(ds, mask, func_name);
	}
	char errbuf[PGH_ERRBUF_LEN] = {0};
	int rc = pgh_tally_start(ds.handle, subset_ ? subset_->handle : nullptr, begin, end, products, &handle, errbuf);
	if (rc != PGH_OK) {
		throw IOException("%s: PgrGetCounts failed for variants [%u, %u): %s", func_name, begin, end, string(errbuf));
	}
	counts_ = pgh_tally_counts(handle);
	lnp_[0] = pgh_tally_hwe_lnp(handle, 0);
	lnp_[1] = pgh_tally_hwe_lnp(handle, 1);
}

void DeviceTally::RunStream(const string &path, uint32_t sample_ct, uint64_t window_variants, const string &func_name) {
	Streamed &st = *streamed_;
	auto fail = [&](const string &msg) {
		std::lock_guard<std::mutex> lock(st.m);
		st.failed = true;
		st.error = msg;
		st.cv.notify_all();
	};
	const uint32_t all = PGH_TALLY_COUNTS | PGH_TALLY_SAMPLE_MISSING | PGH_TALLY_HWE | PGH_TALLY_HWE_MIDP;
	vector<uint32_t> part(st.missing.size());
	// the next window is read from the file while this one is tallied (two in flight: half a window each)
	const uint64_t window = std::max<uint64_t>(1, window_variants / 2);
	const vector<uint64_t> *mask_ptr = mask.empty() ? nullptr : &mask;
	auto open_from = [&](uint64_t v0) {
		const uint32_t v1 = static_cast<uint32_t>(std::min<uint64_t>(end, v0 + window));
		return std::async(std::launch::async, OpenWindow, path, mask_ptr, static_cast<uint32_t>(v0), v1);
	};
	std::future<OpenedWindow> next;
	if (begin < end) {
		next = open_from(begin);
	}
	for (uint64_t v0 = begin; v0 < end; v0 += window) {
		OpenedWindow w = next.get();
		const bool more = v0 + window < end;
		bool stop;
		{
			std::lock_guard<std::mutex> lock(st.m);
			stop = st.stop;
		}
		if (more && w.rc == PGH_OK && !stop) {
			next = open_from(v0 + window);
		}
		if (stop) {
			w.Close();
			return;
		}
		const uint32_t v1 = w.v1;
		char errbuf[PGH_ERRBUF_LEN] = {0};
		pgh_tally *pass = nullptr;
		int rc = w.rc;
		if (rc != PGH_OK) {
			std::snprintf(errbuf, sizeof errbuf, "%s", w.error.c_str());
		}
		if (rc == PGH_OK) {
			rc = pgh_tally_start(w.ds, w.ss, static_cast<uint32_t>(v0), v1, all, &pass, errbuf);
		}
		if (rc == PGH_OK) {
			rc = pgh_tally_wait(pass, all, static_cast<uint32_t>(v0), v1, errbuf);
		}
		if (rc == PGH_OK) {
			const size_t at = v0 - begin, n = v1 - v0;
			std::memcpy(st.counts.data() + 4 * at, pgh_tally_counts(pass), 16 * n);
			std::memcpy(st.lnp[0].data() + at, pgh_tally_hwe_lnp(pass, 0), 8 * n);
			std::memcpy(st.lnp[1].data() + at, pgh_tally_hwe_lnp(pass, 1), 8 * n);
			rc = pgh_tally_sample_missing(pass, part.data(), errbuf);
			for (size_t k = 0; rc == PGH_OK && k < part.size(); k++) {
				st.missing[k] += part[k];
			}
		}
		pgh_tally_destroy(pass);
		w.Close();
		if (rc != PGH_OK) {
			if (more && next.valid()) {
				next.get().Close();
			}
			fail(func_name + ": streaming variants [" + std::to_string(v0) + ", " + std::to_string(v1) + ") of '" + path +
			     "' failed: " + errbuf);
			return;
		}
		(void)sample_ct;
		std::lock_guard<std::mutex> lock(st.m);
		st.done_upto = v1;
		st.cv.notify_all();
	}
	std::lock_guard<std::mutex> lock(st.m);
	st.done_upto = end;
	st.cv.notify_all();
}

DeviceTally::~DeviceTally() {
	if (streamed_) {
		{
			std::lock_guard<std::mutex> lock(streamed_->m);
			streamed_->stop = true;
		}
		if (streamed_->producer.joinable()) {
			streamed_->producer.join();
		}
	}
	if (handle) {
		pgh_tally_destroy(handle); // drains the pass before the subset below goes
	}
}

void DeviceTally::Request(uint32_t products, const string &func_name) {
	if (streamed_) {
		return; // a streamed pass makes every product
	}
	char errbuf[PGH_ERRBUF_LEN] = {0};
	if (pgh_tally_request(handle, products, errbuf) != PGH_OK) {
		throw IOException("%s: tally pass over variants [%u, %u) failed: %s", func_name, begin, end, string(errbuf));
	}
	lnp_[0] = pgh_tally_hwe_lnp(handle, 0);
	lnp_[1] = pgh_tally_hwe_lnp(handle, 1);
}

void DeviceTally::Wait(uint32_t products, uint32_t v_begin, uint32_t v_end, const string &func_name) {
	if (streamed_) {
		Streamed &st = *streamed_;
		const uint32_t need = (products & PGH_TALLY_SAMPLE_MISSING) ? end : v_end;
		std::unique_lock<std::mutex> lock(st.m);
		st.cv.wait(lock, [&] { return st.failed || st.done_upto >= need; });
		if (st.failed) {
			throw IOException("%s", st.error);
		}
		return;
	}
	char errbuf[PGH_ERRBUF_LEN] = {0};
	if (pgh_tally_wait(handle, products, v_begin, v_end, errbuf) != PGH_OK) {
		throw IOException("%s: PgrGetCounts failed for variants [%u, %u): %s", func_name, v_begin, v_end,
		                  string(errbuf));
	}
}

void DeviceTally::SampleMissing(uint32_t *out, const string &func_name) {
	if (streamed_) {
		Wait(PGH_TALLY_SAMPLE_MISSING, begin, end, func_name);
		std::memcpy(out, streamed_->missing.data(), sizeof(uint32_t) * streamed_->missing.size());
		return;
	}
	char errbuf[PGH_ERRBUF_LEN] = {0};
	if (pgh_tally_sample_missing(handle, out, errbuf) != PGH_OK) {
		throw IOException("%s: PgrGetMissingness failed: %s", func_name, string(errbuf));
	}
}

pgh_dataset *DeviceDataset::Resident(const string &func_name) const {
	if (streamed) {
		throw IOException("%s: '%s' does not fit the HBM budget (%.1f GB of rows, budget %.1f GB: PLINKING_HBM_CACHE_GB); "
		                  "every function streams a file of this size window by window; this call needs rows resident together "
		                  "that lie further apart than a window",
		                  func_name, path,
		                  static_cast<double>(info.raw_variant_ct) * static_cast<double>(info.record_bytes) / 1e9,
		                  static_cast<double>(CacheBudgetBytes()) / 1e9);
	}
	return handle;
}

uint64_t DeviceDataset::WindowVariants() const {
	const uint64_t pitch = std::max<uint64_t>(16, (static_cast<uint64_t>(info.record_bytes) + 127) / 128 * 128);
	return std::max<uint64_t>(1, CacheBudgetBytes() / 2 / pitch);
}

RowWindows::~RowWindows() {
	if (ss) {
		pgh_subset_destroy(ss);
	}
	if (ds) {
		pgh_close(ds);
	}
}

RowLease::~RowLease() {
	if (windows) {
		std::lock_guard<std::mutex> lock(windows->m);
		windows->users--;
		windows->cv.notify_all();
	}
}

RowLease LeaseRows(DeviceDataset &dataset, DeviceSubset *subset, RowWindows &w, const vector<uint64_t> *sample_include,
                   uint32_t span_begin, uint32_t span_end, uint32_t file_end, const string &func_name) {
	RowLease lease;
	if (!dataset.streamed) {
		lease.ds = dataset.Resident(func_name);
		lease.ss = subset ? subset->handle : nullptr;
		return lease;
	}
	std::unique_lock<std::mutex> lock(w.m);
	while (!(w.ds && span_begin >= w.begin && span_end <= w.end)) {
		if (w.users > 0) {
			w.cv.wait(lock); // the window in place is still being read
			continue;
		}
		if (w.ss) {
			pgh_subset_destroy(w.ss);
			w.ss = nullptr;
		}
		if (w.ds) {
			pgh_close(w.ds);
			w.ds = nullptr;
		}
		const uint64_t len = std::max<uint64_t>(dataset.WindowVariants(), span_end - span_begin);
		const uint32_t w_end = static_cast<uint32_t>(std::min<uint64_t>(file_end, span_begin + len));
		char errbuf[PGH_ERRBUF_LEN] = {0};
		int rc = pgh_open(dataset.path.c_str(), nullptr, span_begin, w_end, &w.ds, errbuf);
		if (rc == PGH_OK && sample_include && !sample_include->empty()) {
			rc = pgh_subset_create(w.ds, sample_include->data(), &w.ss, errbuf);
		}
		if (rc != PGH_OK) {
			if (w.ds) {
				pgh_close(w.ds);
				w.ds = nullptr;
			}
			w.cv.notify_all();
			throw IOException("%s: streaming variants [%u, %u) of '%s' failed: %s", func_name, span_begin, w_end, dataset.path,
			                  string(errbuf));
		}
		w.begin = span_begin;
		w.end = w_end;
		w.opened++;
	}
	w.users++;
	lease.ds = w.ds;
	lease.ss = w.ss;
	lease.windows = &w;
	lease.window_id = w.opened;
	return lease;
}


// Two windows are in flight -- the one being worked on and the next one being read from the file by a helper thread
// (the ingest runs at the host link's rate, the work on a window mostly far above it: without the overlap a pass over
// the file took the sum of the two) -- so a window is a quarter of the budget here, half of WindowVariants().
void DeviceDataset::ForEachWindow(uint32_t begin, uint32_t end, const vector<uint64_t> *sample_include,
                                  const string &func_name,
                                  const std::function<void(pgh_dataset *, pgh_subset *, uint32_t, uint32_t)> &fn) const {
	if (begin >= end) {
		return;
	}
	const uint64_t window = std::max<uint64_t>(1, WindowVariants() / 2);
	auto open_from = [&](uint64_t v0) {
		const uint32_t v1 = static_cast<uint32_t>(std::min<uint64_t>(end, v0 + window));
		return std::async(std::launch::async, OpenWindow, path, sample_include, static_cast<uint32_t>(v0), v1);
	};
	std::future<OpenedWindow> next = open_from(begin);
	for (uint64_t v0 = begin; v0 < end; v0 += window) {
		OpenedWindow w = next.get();
		const bool more = v0 + window < end;
		if (more && w.rc == PGH_OK) {
			next = open_from(v0 + window);
		}
		auto drain = [&] { // an error: whatever the helper is reading must not stay open
			if (more && next.valid()) {
				next.get().Close();
			}
		};
		if (w.rc != PGH_OK) {
			drain();
			throw IOException("%s: streaming variants [%u, %u) of '%s' failed: %s", func_name, w.v0, w.v1, path, w.error);
		}
		try {
			fn(w.ds, w.ss, w.v0, w.v1);
		} catch (...) {
			w.Close();
			drain();
			throw;
		}
		w.Close();
	}
}

shared_ptr<DeviceTally> DeviceDataset::FindTally(const vector<uint64_t> *sample_include, uint32_t begin, uint32_t end) {
	static const vector<uint64_t> kAll;
	const vector<uint64_t> &want_mask = sample_include ? *sample_include : kAll;
	std::lock_guard<std::mutex> lock(tally_mutex_);
	for (auto &t : tallies_) {
		if (t->begin <= begin && t->end >= end && t->mask == want_mask) {
			return t;
		}
	}
	return nullptr;
}

shared_ptr<DeviceTally> DeviceDataset::AcquireTally(const vector<uint64_t> *sample_include, uint32_t begin,
                                                    uint32_t end, uint32_t products, bool exact_range, bool use_cache,
                                                    const string &func_name) {
	static const vector<uint64_t> kAll;
	const vector<uint64_t> &want_mask = sample_include ? *sample_include : kAll;
	if (use_cache) {
		std::lock_guard<std::mutex> lock(tally_mutex_);
		for (size_t i = 0; i < tallies_.size(); i++) {
			auto &t = tallies_[i];
			const bool covers = exact_range ? (t->begin == begin && t->end == end) : (t->begin <= begin && t->end >= end);
			if (covers && t->mask == want_mask) {
				auto hit = t;
				tallies_.erase(tallies_.begin() + static_cast<std::ptrdiff_t>(i));
				tallies_.push_back(hit);
				hit->Request(products, func_name);
				return hit;
			}
		}
	}
	// (outside the lock: starting a pass only enqueues, but it allocates)
	auto made = make_shared<DeviceTally>(*this, sample_include, begin, end, products, func_name);
	if (use_cache) {
		std::lock_guard<std::mutex> lock(tally_mutex_);
		tallies_.push_back(made);
		constexpr size_t kTallyEntries = 6; // 40 B per variant of pinned host memory each, at most
		if (tallies_.size() > kTallyEntries) {
			tallies_.erase(tallies_.begin());
		}
	}
	return made;
}

namespace {
std::mutex g_devices_mutex;
vector<int> g_devices;
bool g_devices_from_env = false;

vector<int> ParseDeviceList(const string &spec) {
	vector<int> devices;
	const int have = pgh_device_count();
	if (spec == "all" || spec == "ALL") {
		for (int d = 0; d < have; d++) {
			devices.push_back(d);
		}
		return devices;
	}
	size_t at = 0;
	while (at < spec.size()) {
		size_t comma = spec.find(',', at);
		if (comma == string::npos) {
			comma = spec.size();
		}
		string tok = spec.substr(at, comma - at);
		tok.erase(0, tok.find_first_not_of(" \t"));
		tok.erase(tok.find_last_not_of(" \t") + 1);
		if (tok.empty() || tok.find_first_not_of("0123456789") != string::npos) {
			throw InvalidInputException("plinking_devices must be '', 'all' or a comma-separated list of device "
			                            "ordinals, got '%s'", spec);
		}
		const int d = std::atoi(tok.c_str());
		if (d >= have) {
			throw InvalidInputException("plinking_devices: device %d does not exist (%d visible)", d, have);
		}
		devices.push_back(d);
		at = comma + 1;
	}
	return devices;
}

struct CacheKey {
	string path;
	int64_t mtime_ns;
	int64_t size;
	vector<int> devices;
	bool operator==(const CacheKey &o) const {
		return path == o.path && mtime_ns == o.mtime_ns && size == o.size && devices == o.devices;
	}
};
struct CacheEntry {
	CacheKey key;
	shared_ptr<DeviceDataset> ds; // null while the first caller is still opening the file
	uint64_t bytes;
};
std::mutex g_cache_mutex;
std::condition_variable g_cache_opened; // an entry finished opening (or gave up)
vector<CacheEntry> g_cache;             // most recently used last

uint64_t CacheBudgetBytes() {
	const char *env = std::getenv("PLINKING_HBM_CACHE_GB");
	double gb = env ? std::atof(env) : 160.0;
	return static_cast<uint64_t>(gb * 1e9);
}
} // namespace

void SetPlinkingDevices(const string &spec) {
	vector<int> parsed = ParseDeviceList(spec);
	std::lock_guard<std::mutex> lock(g_devices_mutex);
	g_devices = std::move(parsed);
	g_devices_from_env = true; // an explicit setting, '' included, overrides the environment
}

vector<int> GetPlinkingDevices() {
	std::lock_guard<std::mutex> lock(g_devices_mutex);
	if (!g_devices_from_env) {
		if (const char *env = std::getenv("PLINKING_DEVICES")) {
			g_devices = ParseDeviceList(env); // throws on a malformed list -- every time, not only the first
		}
		g_devices_from_env = true;
	}
	return g_devices;
}

//! pgh_synth_create of the spec: one dataset, or near-equal contiguous ranges on the listed devices as a group.
static int OpenSynth(const SynthSpec &spec, const vector<int> &devices, pgh_dataset **out, char *errbuf) {
	if (devices.empty()) {
		return pgh_synth_create(0, spec.variants, spec.samples, spec.seed, spec.missing_rate, out, errbuf);
	}
	vector<pgh_dataset *> shards;
	int rc = PGH_OK;
	const uint64_t k_total = devices.size();
	for (uint64_t k = 0; k < k_total && rc == PGH_OK; k++) {
		const uint32_t b = static_cast<uint32_t>(spec.variants * k / k_total);
		const uint32_t e = static_cast<uint32_t>(spec.variants * (k + 1) / k_total);
		pgh_dataset *sh = nullptr;
		rc = pgh_set_device(devices[k], errbuf);
		if (rc == PGH_OK) {
			rc = pgh_synth_create(b, e, spec.samples, spec.seed, spec.missing_rate, &sh, errbuf);
		}
		if (rc == PGH_OK) {
			shards.push_back(sh);
		}
	}
	if (rc == PGH_OK) {
		rc = pgh_group_create(shards.data(), static_cast<uint32_t>(shards.size()), out, errbuf);
	}
	if (rc != PGH_OK) {
		for (auto *sh : shards) {
			pgh_close(sh);
		}
	}
	return rc;
}

shared_ptr<DeviceDataset> DeviceDataset::Acquire(const string &pgen_path, const string &func_name) {
	struct stat st;
	std::memset(&st, 0, sizeof st);
	SynthSpec synth;
	const bool is_synth = ParseSynthPath(pgen_path, synth);
	if (!is_synth && ::stat(pgen_path.c_str(), &st) != 0) {
		throw IOException("%s: failed to open '%s': %s", func_name, pgen_path, string(std::strerror(errno)));
	}
	const vector<int> devices = GetPlinkingDevices();
	CacheKey key {pgen_path, static_cast<int64_t>(st.st_mtim.tv_sec) * 1000000000LL + st.st_mtim.tv_nsec,
	              static_cast<int64_t>(st.st_size), devices};
	// The cache lock is NOT held while a file travels to HBM (seconds for a large one): the first caller leaves an
	// entry without a dataset behind, binds of the same file wait for it, binds of other files go on.
	std::unique_lock<std::mutex> lock(g_cache_mutex);
	for (;;) {
		size_t at = g_cache.size();
		for (size_t i = 0; i < g_cache.size(); i++) {
			if (g_cache[i].key == key) {
				at = i;
				break;
			}
		}
		if (at == g_cache.size()) {
			break; // nobody has it: this caller opens it
		}
		if (g_cache[at].ds) {
			auto e = g_cache[at];
			g_cache.erase(g_cache.begin() + static_cast<std::ptrdiff_t>(at));
			g_cache.push_back(e);
			return e.ds;
		}
		g_cache_opened.wait(lock); // being opened by another thread (which removes the entry if it fails)
	}
	g_cache.push_back({key, nullptr, 0});
	lock.unlock();
	// Whatever happens between here and the relock -- an allocation that throws included -- the placeholder must
	// not stay behind without a dataset: binds of this file wait on it.
	struct PlaceholderGuard {
		const CacheKey &key;
		bool armed = true;
		~PlaceholderGuard() {
			if (!armed) {
				return;
			}
			std::lock_guard<std::mutex> relock(g_cache_mutex);
			for (size_t i = 0; i < g_cache.size(); i++) {
				if (g_cache[i].key == key && !g_cache[i].ds) {
					g_cache.erase(g_cache.begin() + static_cast<std::ptrdiff_t>(i));
					break;
				}
			}
			g_cache_opened.notify_all();
		}
	} guard {key};
	auto ds = make_shared<DeviceDataset>();
	ds->path = pgen_path;
	char errbuf[PGH_ERRBUF_LEN] = {0};
	// one resident matrix on the current device, or one contiguous variant shard per listed device behind one
	// handle: every pgh_* call the table functions make accepts either
	int rc;
	if (!is_synth) {
		// a file whose rows exceed the HBM budget is not made resident: its tallies stream (DeviceTally)
		pgh_info probe;
		if (pgh_probe(pgen_path.c_str(), nullptr, &probe, errbuf) == PGH_OK) {
			const uint64_t pitch = (static_cast<uint64_t>(probe.record_bytes) + 127) / 128 * 128;
			if (pitch * probe.raw_variant_ct > CacheBudgetBytes() * std::max<size_t>(1, devices.size())) {
				ds->streamed = true;
				ds->info = probe;
				ds->info.variant_end = probe.raw_variant_ct;
			}
		}
	}
	if (ds->streamed) {
		rc = PGH_OK;
	} else if (is_synth) {
		rc = OpenSynth(synth, devices, &ds->handle, errbuf);
	} else {
		rc = devices.empty() ? pgh_open(pgen_path.c_str(), nullptr, 0, UINT32_MAX, &ds->handle, errbuf)
		                     : pgh_open_sharded(pgen_path.c_str(), nullptr, 0, UINT32_MAX, devices.data(),
		                                        static_cast<uint32_t>(devices.size()), &ds->handle, errbuf);
	}
	lock.lock();
	guard.armed = false;
	size_t mine = g_cache.size();
	for (size_t i = 0; i < g_cache.size(); i++) {
		if (g_cache[i].key == key && !g_cache[i].ds) {
			mine = i;
			break;
		}
	}
	if (rc != PGH_OK) {
		if (mine < g_cache.size()) {
			g_cache.erase(g_cache.begin() + static_cast<std::ptrdiff_t>(mine));
		}
		g_cache_opened.notify_all();
		throw IOException("%s: failed to open '%s': %s", func_name, pgen_path, string(errbuf));
	}
	if (!ds->streamed) {
		pgh_get_info(ds->handle, &ds->info);
	}
	// what the dataset holds in HBM: the 2-bit rows, and per dosage-bearing variant a presence bit and a 4-byte rank
	// per 64 samples, 2 bytes per explicit value and up to 4 more once plink_score has built the entry records of
	// the sparse tracks (phase tracks, two bit rows per phased variant, are not reported by pgh_get_info: not counted)
	const uint64_t words = (static_cast<uint64_t>(ds->info.raw_sample_ct) + 63) / 64;
	const uint64_t bytes = ds->streamed ? 0
	                                    : ds->info.pitch_bytes * (ds->info.variant_end - ds->info.variant_begin) +
	                                          12ull * words * ds->info.dosage_variant_ct + 6ull * ds->info.dosage_value_ct;
	if (mine < g_cache.size()) {
		g_cache[mine].ds = ds;
		g_cache[mine].bytes = bytes;
		// most recently used last
		CacheEntry e = g_cache[mine];
		g_cache.erase(g_cache.begin() + static_cast<std::ptrdiff_t>(mine));
		g_cache.push_back(e);
	}
	g_cache_opened.notify_all();
	// evict least recently used datasets beyond the HBM budget (in-flight queries keep theirs alive; entries
	// still opening hold no bytes yet and stay)
	uint64_t total = 0;
	for (auto &e : g_cache) {
		total += e.bytes;
	}
	for (size_t i = 0; i + 1 < g_cache.size() && total > CacheBudgetBytes();) {
		if (!g_cache[i].ds) {
			i++;
			continue;
		}
		total -= g_cache[i].bytes;
		g_cache.erase(g_cache.begin() + static_cast<std::ptrdiff_t>(i));
	}
	return ds;
}

DeviceSubset::DeviceSubset(const DeviceDataset &ds, const vector<uint64_t> &include, const string &func_name) {
	if (ds.streamed) {
		return; // nothing resident to stage a mask next to: the streamed pass stages it per window
	}
	char errbuf[PGH_ERRBUF_LEN] = {0};
	int rc = pgh_subset_create(ds.handle, include.data(), &handle, errbuf);
	ThrowOnPghError(rc, errbuf, func_name, "sample subset");
}

DeviceSubset::~DeviceSubset() {
	if (handle) {
		pgh_subset_destroy(handle);
	}
}

} // namespace duckdb

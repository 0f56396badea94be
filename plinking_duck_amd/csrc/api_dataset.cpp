// api_dataset.cpp -- library/device entry points, datasets (open = header parse + staged ingest
// with device record decode), synthetic data, sample subsets.
#include "api_internal.hpp"

// ---------------------------------------------------------------------------
// per-thread streams (api_internal.hpp:PghThreadStream)
// ---------------------------------------------------------------------------

namespace {
struct ThreadScratchBlock {
	int device;
	hipStream_t stream;
	void *p;
	size_t bytes;
};
struct ThreadStreams {
	std::vector<std::pair<int, hipStream_t>> by_device;
	std::vector<ThreadScratchBlock> scratch;
	~ThreadStreams() {
		for (auto &b : scratch) {
			DeviceScope scope(b.device);
			(void)hipFree(b.p); // (waits for the device: nothing of this thread's is still running on it afterwards)
		}
		for (auto &e : by_device) {
			DeviceScope scope(e.first);
			(void)hipStreamSynchronize(e.second);
			(void)hipStreamDestroy(e.second);
		}
	}
};
thread_local ThreadStreams t_streams;
} // namespace

hipError_t PghThreadScratch(size_t bytes, hipStream_t st, void **out) {
	int device = 0;
	hipError_t e = hipGetDevice(&device);
	if (e != hipSuccess) {
		return e;
	}
	if (bytes == 0) {
		bytes = 16;
	}
	ThreadScratchBlock *slot = nullptr;
	for (auto &b : t_streams.scratch) {
		if (b.device == device && b.stream == st) {
			slot = &b;
			break;
		}
	}
	if (slot && slot->bytes >= bytes) {
		*out = slot->p;
		return hipSuccess;
	}
	if (slot) {
		e = hipStreamSynchronize(st); // the block may still be read by what this thread enqueued before
		if (e != hipSuccess) {
			return e;
		}
		(void)hipFree(slot->p);
		slot->p = nullptr;
		slot->bytes = 0;
	}
	void *p = nullptr;
	e = PghMalloc(&p, bytes);
	if (e != hipSuccess) {
		return e;
	}
	if (slot) {
		slot->p = p;
		slot->bytes = bytes;
	} else {
		if (t_streams.scratch.size() >= 8) { // a thread that keeps changing streams: drop the oldest block
			ThreadScratchBlock &old = t_streams.scratch.front();
			DeviceScope scope(old.device);
			(void)hipStreamSynchronize(old.stream);
			(void)hipFree(old.p);
			t_streams.scratch.erase(t_streams.scratch.begin());
		}
		t_streams.scratch.push_back(ThreadScratchBlock {device, st, p, bytes});
	}
	*out = p;
	return hipSuccess;
}

// ---- the block cache (api_internal.hpp) ----
namespace {
struct CachedBlock {
	int device;
	void *p;
	size_t bytes;
};
std::mutex g_block_mu;
std::vector<CachedBlock> g_block_free;                 // oldest first
std::vector<CachedBlock> g_block_live;                 // cacheable blocks handed out: device and size
size_t BlockCacheCap() {
	static const size_t cap = [] {
		const char *e = std::getenv("PGH_BLOCK_CACHE_GB");
		const double gb = e && *e ? std::atof(e) : 64.0;
		return gb > 0 ? static_cast<size_t>(gb * 1073741824.0) : size_t(0);
	}();
	return cap;
}
} // namespace

void PghTrimBlockCache() {
	std::vector<CachedBlock> drop;
	{
		std::lock_guard<std::mutex> lk(g_block_mu);
		drop.swap(g_block_free);
	}
	for (auto &b : drop) {
		DeviceScope scope(b.device);
		(void)hipFree(b.p);
	}
}

void PghMakeRoom(size_t bytes) {
	if (bytes < (256ull << 20)) {
		return;
	}
	{
		std::lock_guard<std::mutex> lk(g_block_mu);
		if (g_block_free.empty()) {
			return;
		}
	}
	size_t free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
		(void)hipGetLastError();
		return;
	}
	if (free_b < bytes + (1ull << 30)) {
		PghTrimBlockCache();
	}
}

hipError_t PghBlockAlloc(void **out, size_t bytes) {
	int device = 0;
	hipError_t e = hipGetDevice(&device);
	if (e != hipSuccess) {
		return e;
	}
	if (bytes >= kBlockCacheMin && BlockCacheCap()) {
		std::lock_guard<std::mutex> lk(g_block_mu);
		size_t best = g_block_free.size();
		for (size_t i = 0; i < g_block_free.size(); i++) {
			const CachedBlock &b = g_block_free[i];
			if (b.device == device && b.bytes >= bytes && b.bytes - bytes <= bytes / 4 &&
			    (best == g_block_free.size() || b.bytes < g_block_free[best].bytes)) {
				best = i;
			}
		}
		if (best != g_block_free.size()) {
			*out = g_block_free[best].p;
			g_block_live.push_back(CachedBlock {device, *out, g_block_free[best].bytes});
			g_block_free.erase(g_block_free.begin() + static_cast<ptrdiff_t>(best));
			return hipSuccess;
		}
	}
	PghMakeRoom(bytes);
	e = hipMalloc(out, bytes);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		PghTrimBlockCache(); // what the list holds may be exactly what is missing
		e = hipMalloc(out, bytes);
	}
	if (e == hipSuccess && bytes >= kBlockCacheMin && BlockCacheCap()) {
		std::lock_guard<std::mutex> lk(g_block_mu);
		g_block_live.push_back(CachedBlock {device, *out, bytes});
	}
	return e;
}

void PghBlockFree(void *p) {
	if (!p) {
		return;
	}
	size_t bytes = 0;
	int device = -1;
	{
		std::lock_guard<std::mutex> lk(g_block_mu);
		for (size_t i = 0; i < g_block_live.size(); i++) {
			if (g_block_live[i].p == p) {
				bytes = g_block_live[i].bytes;
				device = g_block_live[i].device;
				g_block_live.erase(g_block_live.begin() + static_cast<ptrdiff_t>(i));
				break;
			}
		}
	}
	if (bytes == 0 || bytes > BlockCacheCap()) {
		(void)hipFree(p);
		return;
	}
	DeviceScope scope(device); // (the block's device, whatever the caller's current one is)
	// hipFree waits for the device before the memory can be handed out again; so does this
	if (hipDeviceSynchronize() != hipSuccess) {
		(void)hipGetLastError();
		(void)hipFree(p);
		return;
	}
	std::vector<CachedBlock> drop;
	{
		std::lock_guard<std::mutex> lk(g_block_mu);
		size_t held = bytes;
		for (const auto &b : g_block_free) {
			held += b.device == device ? b.bytes : 0;
		}
		for (size_t i = 0; held > BlockCacheCap() && i < g_block_free.size();) {
			if (g_block_free[i].device == device) {
				held -= g_block_free[i].bytes;
				drop.push_back(g_block_free[i]);
				g_block_free.erase(g_block_free.begin() + static_cast<ptrdiff_t>(i));
			} else {
				i++;
			}
		}
		g_block_free.push_back(CachedBlock {device, p, bytes});
	}
	for (auto &b : drop) {
		(void)hipFree(b.p);
	}
}

size_t PghDeviceFreeBytes() {
	size_t free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) {
		(void)hipGetLastError();
		return 0;
	}
	int device = 0;
	(void)hipGetDevice(&device);
	std::lock_guard<std::mutex> lk(g_block_mu);
	for (const auto &b : g_block_free) {
		free_b += b.device == device ? b.bytes : 0;
	}
	return free_b;
}

hipStream_t PghThreadStream() {
	int device = 0;
	if (hipGetDevice(&device) != hipSuccess) {
		(void)hipGetLastError();
		return nullptr; // no device: the caller's next HIP call reports it
	}
	for (auto &e : t_streams.by_device) {
		if (e.first == device) {
			return e.second;
		}
	}
	hipStream_t s = nullptr;
	if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
		(void)hipGetLastError();
		return nullptr; // the null stream still orders everything behind it
	}
	t_streams.by_device.emplace_back(device, s);
	return s;
}

// ---------------------------------------------------------------------------
// library / device
// ---------------------------------------------------------------------------

extern "C" const char *pgh_version(void) {
	return "pgenhip 1 gfx950";
}

extern "C" int pgh_device_count(void) {
	int n = 0;
	if (hipGetDeviceCount(&n) != hipSuccess) {
		return 0;
	}
	return n;
}

extern "C" int pgh_set_device(int device, char *errbuf) {
	PGH_HIP(hipSetDevice(device), "hipSetDevice");
	return PGH_OK;
}

// ---------------------------------------------------------------------------
// dataset lifecycle
// ---------------------------------------------------------------------------

static void FillInfo(const PgenIndex &ix, pgh_info *out) {
	std::memset(out, 0, sizeof *out);
	out->raw_variant_ct = ix.variant_ct;
	out->raw_sample_ct = ix.sample_ct;
	out->variant_begin = 0;
	out->variant_end = ix.variant_ct;
	out->has_dosage = ix.has_dosage;
	out->has_phase = ix.has_phase;
	out->max_record_bytes = ix.max_record_bytes;
	out->record_bytes = ix.RecordBytes();
	out->pitch_bytes = ChoosePitch(ix.RecordBytes());
	for (int i = 0; i < 8; i++) {
		out->vrtype_hist[i] = ix.vrtype_hist[i];
	}
	out->device = -1;
}

extern "C" int pgh_probe(const char *pgen_path, const char *pgi_path, pgh_info *out, char *errbuf) {
	if (!pgen_path || !out) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	PgenIndex ix;
	std::string err;
	if (!pgh::ParsePgenIndex(pgen_path, pgi_path ? pgi_path : "", ix, err)) {
		SetErr(errbuf, err);
		return err.find("cannot open") != std::string::npos ? PGH_ERR_OPEN : PGH_ERR_FORMAT;
	}
	FillInfo(ix, out);
	return PGH_OK;
}

// pgh_open's staging buffers (kStages pinned + kStages device, 64 MB each) cost ~40 ms apiece to allocate, more
// than a small file takes to ingest: one set per device is parked here between opens.  Four stages: while
// one is being filled from the page cache, up to three are queued on the copy engine, so neither the readers
// (~90 GB/s with 8 threads on this box) nor the host link (~56 GB/s, tools/ingest_probe.hip) waits for the other.
constexpr int kStages = 4;
struct StageSet {
	uint8_t *pinned[kStages] = {};
	uint8_t *device[kStages] = {};
	uint64_t bytes = 0;
	int device_id = -1;
	void Free() {
		for (int i = 0; i < kStages; i++) {
			if (pinned[i]) {
				(void)hipHostFree(pinned[i]);
			}
			if (device[i]) {
				(void)hipFree(device[i]);
			}
			pinned[i] = device[i] = nullptr;
		}
		bytes = 0;
	}
};
static std::mutex g_stage_mutex;
static StageSet g_parked_stage;

static hipError_t AcquireStage(uint64_t bytes, int device_id, bool want_device, StageSet &out) {
	{
		std::lock_guard<std::mutex> lock(g_stage_mutex);
		if (g_parked_stage.bytes >= bytes && g_parked_stage.device_id == device_id) {
			out = g_parked_stage;
			g_parked_stage = StageSet();
		}
	}
	out.device_id = device_id;
	hipError_t e = hipSuccess;
	for (int i = 0; i < kStages && e == hipSuccess; i++) {
		if (!out.pinned[i]) {
			e = hipHostMalloc(reinterpret_cast<void **>(&out.pinned[i]), bytes, hipHostMallocDefault);
		}
		if (e == hipSuccess && want_device && !out.device[i]) {
			e = PghMalloc(reinterpret_cast<void **>(&out.device[i]), std::max(bytes, out.bytes));
		}
	}
	out.bytes = std::max(bytes, out.bytes);
	if (e != hipSuccess) {
		out.Free();
	}
	return e;
}

static void ReleaseStage(StageSet &set) {
	{
		std::lock_guard<std::mutex> lock(g_stage_mutex);
		if (g_parked_stage.bytes == 0 && set.bytes <= (64ull << 20) + 8192) {
			g_parked_stage = set;
			set = StageSet();
			return;
		}
	}
	set.Free();
}

// pread is the ceiling of the plain-record ingest path (one thread moves ~6 GB/s out of the
// page cache); split a stage across a few threads.
static bool ReadParallel(const pgh::RecordFile &file, uint64_t offset, size_t bytes, uint8_t *dst, std::string &err) {
	constexpr size_t kMinSlice = 4u << 20;
	const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
	static const unsigned want = [] {
		const char *e = std::getenv("PGH_OPEN_THREADS"); // tuning knob: reader threads per stage
		const int v = e ? std::atoi(e) : 0;
		return v > 0 ? static_cast<unsigned>(v) : 8u;
	}();
	const unsigned parts = static_cast<unsigned>(std::min<size_t>(std::min(want, hw), std::max<size_t>(1, bytes / kMinSlice)));
	if (parts <= 1) {
		return file.ReadAt(offset, bytes, dst, err);
	}
	std::vector<std::thread> pool;
	std::vector<std::string> errs(parts);
	std::vector<char> ok(parts, 1);
	const size_t slice = (bytes + parts - 1) / parts;
	for (unsigned t = 0; t < parts; t++) {
		const size_t lo = std::min(bytes, static_cast<size_t>(t) * slice);
		const size_t hi = std::min(bytes, lo + slice);
		pool.emplace_back([&, t, lo, hi] { ok[t] = file.ReadAt(offset + lo, hi - lo, dst + lo, errs[t]) ? 1 : 0; });
	}
	for (auto &th : pool) {
		th.join();
	}
	for (unsigned t = 0; t < parts; t++) {
		if (!ok[t]) {
			err = errs[t];
			return false;
		}
	}
	return true;
}

// Host normalisation of compressed records, split over a few threads.  Each worker owns a
// Normalizer (LD-base scratch) and a contiguous sub-range; a sub-range that starts inside an
// LD run resolves its base by walking back, exactly as a range that starts mid-file does.
static bool ExpandParallel(const pgh::PgenIndex &ix, const pgh::RecordFile &file, pgh::Normalizer &first,
                           uint32_t v_begin, uint32_t v_end, uint8_t *dst, size_t pitch, std::string &err) {
	constexpr uint32_t kMinRows = 64;
	const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
	const uint32_t rows = v_end - v_begin;
	const unsigned parts = std::min<unsigned>(std::min(8u, hw), std::max<uint32_t>(1, rows / kMinRows));
	if (parts <= 1) {
		return first.ExpandRange(v_begin, v_end, dst, pitch, err);
	}
	std::vector<std::thread> pool;
	std::vector<std::string> errs(parts);
	std::vector<char> ok(parts, 1);
	const uint32_t slice = (rows + parts - 1) / parts;
	for (unsigned t = 0; t < parts; t++) {
		const uint32_t lo = std::min<uint64_t>(v_end, static_cast<uint64_t>(v_begin) + static_cast<uint64_t>(t) * slice);
		const uint32_t hi = std::min<uint64_t>(v_end, static_cast<uint64_t>(lo) + slice);
		pool.emplace_back([&, t, lo, hi] {
			if (lo >= hi) {
				return;
			}
			uint8_t *out = dst + static_cast<size_t>(lo - v_begin) * pitch;
			if (t == 0) {
				ok[t] = first.ExpandRange(lo, hi, out, pitch, errs[t]) ? 1 : 0;
			} else {
				pgh::Normalizer mine(ix, file);
				ok[t] = mine.ExpandRange(lo, hi, out, pitch, errs[t]) ? 1 : 0;
			}
		});
	}
	for (auto &th : pool) {
		th.join();
	}
	for (unsigned t = 0; t < parts; t++) {
		if (!ok[t]) {
			err = errs[t];
			return false;
		}
	}
	return true;
}

// Dosage tracks -> the resident bit-array form (dosage.hpp:DosageView).
//
// PrepareDosage sizes and allocates the arrays before the body is streamed: a row per dosage-bearing
// variant of the range, and room for as many values as the records have bytes for.  Records that go
// through the device decode have their tracks extracted there (LaunchDosageIngest).  The few that are
// expanded on the host (an LD run whose base lies before the range, PGH_HOST_NORMALIZE=1) are parsed
// on the host afterwards and appended (AppendDosageTracksHost); value runs need not be in row order.
struct DosageStaging {
	uint64_t capacity = 0;
	uint64_t *d_total = nullptr;       // running count of stored values
	std::vector<uint32_t> host_parsed; // variants whose track the host parses
	~DosageStaging() {
		if (d_total) {
			(void)hipFree(d_total);
		}
	}
};

static int PrepareDosage(pgh_dataset *ds, DosageStaging &stg, char *errbuf) {
	const PgenIndex &ix = ds->index;
	const uint32_t range = ds->v_end - ds->v_begin;
	const uint32_t N = ds->sample_ct;
	const uint32_t words = (N + 63) / 64;
	ds->dos_row_of.assign(range, -1);
	uint32_t rows = 0;
	for (uint32_t i = 0; i < range; i++) {
		const uint32_t v = ds->v_begin + i;
		if (ix.vrtype[v] & 0x60) {
			ds->dos_row_of[i] = static_cast<int32_t>(rows++);
			stg.capacity += std::min<uint64_t>(N, (ix.offset[v + 1] - ix.offset[v]) / 2); // a value is two record bytes
		}
	}
	if (rows == 0) {
		ds->dos_row_of.clear();
		return PGH_OK;
	}
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&ds->d_dos_row_of), sizeof(int32_t) * range), "hipMalloc(dosage)");
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&ds->d_dos_present), 8ull * rows * words), "hipMalloc(dosage)");
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&ds->d_dos_rank), 4ull * rows * words), "hipMalloc(dosage)");
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&ds->d_dos_val_off), 8ull * (rows + 1)), "hipMalloc(dosage)");
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&ds->d_dos_values), 2 * stg.capacity + 32), "hipMalloc(dosage)");
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&stg.d_total), 8), "hipMalloc(dosage)");
	PGH_HIP(hipMemset(ds->d_dos_present, 0, 8ull * rows * words), "dosage memset");
	PGH_HIP(hipMemset(ds->d_dos_rank, 0, 4ull * rows * words), "dosage memset");
	PGH_HIP(hipMemset(ds->d_dos_val_off, 0, 8ull * (rows + 1)), "dosage memset");
	PGH_HIP(hipMemset(ds->d_dos_values, 0, 2 * stg.capacity + 32), "dosage memset");
	PGH_HIP(hipMemset(stg.d_total, 0, 8), "dosage memset");
	PGH_HIP(hipMemcpy(ds->d_dos_row_of, ds->dos_row_of.data(), sizeof(int32_t) * range, hipMemcpyHostToDevice),
	        "dosage upload");
	ds->dos_rows = rows;
	return PGH_OK;
}

// The host twin of LaunchDosageIngest for `variants` (ascending): a few threads over chunks, each worker
// with its own Normalizer (finding a track means walking the record's main and phase tracks first).
// Dense 0x40 tracks drop their 65535 "no dosage" entries, so downstream kernels never meet that sentinel.
static int AppendDosageTracksHost(pgh_dataset *ds, const pgh::RecordFile &file, const std::vector<uint32_t> &variants,
                                  uint64_t capacity, uint64_t &filled, char *errbuf) {
	const PgenIndex &ix = ds->index;
	const uint32_t N = ds->sample_ct;
	const uint32_t words = (N + 63) / 64;
	const uint32_t total = static_cast<uint32_t>(variants.size());
	const uint32_t chunk_rows = std::max<uint32_t>(8, static_cast<uint32_t>((64ull << 20) / (2ull * N + 8ull * words)));
	const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
	std::vector<uint64_t> h_present;
	std::vector<std::vector<uint16_t>> h_values;
	for (uint32_t r0 = 0; r0 < total; r0 += chunk_rows) {
		const uint32_t cnt = std::min(total - r0, chunk_rows);
		h_present.assign(static_cast<size_t>(cnt) * words, 0);
		h_values.assign(cnt, {});
		const unsigned parts = std::min<unsigned>(std::min(8u, hw), std::max<uint32_t>(1, cnt / 8));
		std::vector<std::thread> pool;
		std::vector<std::string> errs(parts);
		const uint32_t slice = (cnt + parts - 1) / parts;
		for (unsigned t = 0; t < parts; t++) {
			pool.emplace_back([&, t] {
				pgh::Normalizer norm(ix, file);
				std::vector<uint8_t> row2bit;
				std::vector<uint16_t> dos16;
				const uint32_t lo = std::min(cnt, t * slice), hi = std::min(cnt, lo + slice);
				for (uint32_t k = lo; k < hi; k++) {
					if (!norm.DecodeDosage(variants[r0 + k], row2bit, dos16, errs[t])) {
						return;
					}
					uint64_t *bits = h_present.data() + static_cast<size_t>(k) * words;
					for (uint32_t s = 0; s < N; s++) {
						if (dos16[s] != 0xffff) {
							if (dos16[s] > 32768) {
								errs[t] = "dosage above 2.0 in variant " + std::to_string(variants[r0 + k]);
								return;
							}
							bits[s >> 6] |= 1ull << (s & 63);
							h_values[k].push_back(dos16[s]);
						}
					}
				}
			});
		}
		for (auto &th : pool) {
			th.join();
		}
		for (auto &e : errs) {
			if (!e.empty()) {
				SetErr(errbuf, e);
				return PGH_ERR_FORMAT;
			}
		}
		for (uint32_t k = 0; k < cnt; k++) {
			const uint32_t row = static_cast<uint32_t>(ds->dos_row_of[variants[r0 + k] - ds->v_begin]);
			if (filled + h_values[k].size() > capacity) {
				SetErr(errbuf, "dosage tracks hold more values than their records have bytes for");
				return PGH_ERR_FORMAT;
			}
			PGH_HIP(hipMemcpy(ds->d_dos_present + static_cast<uint64_t>(row) * words,
			                  h_present.data() + static_cast<size_t>(k) * words, 8ull * words, hipMemcpyHostToDevice),
			        "dosage upload");
			PGH_HIP(hipMemcpy(ds->d_dos_val_off + row, &filled, 8, hipMemcpyHostToDevice), "dosage upload");
			if (!h_values[k].empty()) {
				PGH_HIP(hipMemcpy(ds->d_dos_values + filled, h_values[k].data(), 2 * h_values[k].size(),
				                  hipMemcpyHostToDevice),
				        "dosage upload");
			}
			filled += h_values[k].size();
			PGH_HIP(pgh::LaunchDosageRank(ds->d_dos_present + static_cast<uint64_t>(row) * words, 1, words,
			                              ds->d_dos_rank + static_cast<uint64_t>(row) * words, PghThreadStream()),
			        "dosage rank kernel");
		}
		PGH_HIP(hipStreamSynchronize(PghThreadStream()), "dosage rank sync");
	}
	return PGH_OK;
}

//! Explicit dosages per row, to the host (the score plan picks a kernel per variant by density).
static int FetchDosageRowCounts(pgh_dataset *ds, char *errbuf) {
	const uint32_t rows = ds->dos_rows, words = (ds->sample_ct + 63) / 64;
	DevBuf d_tot;
	PGH_HIP(d_tot.Alloc(8ull * rows), "hipMalloc(dosage totals)");
	PGH_HIP(pgh::LaunchDosageRowTotals(ds->d_dos_present, ds->d_dos_rank, rows, words, d_tot.As<uint64_t>(), PghThreadStream()),
	        "dosage totals kernel");
	std::vector<uint64_t> tot(rows);
	PGH_HIP(hipMemcpyAsync(tot.data(), d_tot.p, 8ull * rows, hipMemcpyDeviceToHost, PghThreadStream()), "dosage totals copy");
	PGH_HIP(hipStreamSynchronize(PghThreadStream()), "dosage totals sync");
	ds->dos_row_count.assign(tot.begin(), tot.end());
	return PGH_OK;
}

// Phase tracks -> two resident bit rows per phased variant (phase.hpp).  Records that go through the device
// decode are expanded there (LaunchPhaseIngest); host-expanded ones (and files wider than the kernel's LDS
// tables) through the host parser below.
static int PreparePhase(pgh_dataset *ds, char *errbuf) {
	const PgenIndex &ix = ds->index;
	const uint32_t range = ds->v_end - ds->v_begin;
	const uint32_t words = (ds->sample_ct + 63) / 64;
	ds->ph_row_of.assign(range, -1);
	uint32_t rows = 0;
	for (uint32_t i = 0; i < range; i++) {
		if (ix.vrtype[ds->v_begin + i] & 0x10) {
			ds->ph_row_of[i] = static_cast<int32_t>(rows++);
		}
	}
	if (rows == 0) {
		ds->ph_row_of.clear();
		return PGH_OK;
	}
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&ds->d_ph_present), 8ull * rows * words), "hipMalloc(phase)");
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&ds->d_ph_info), 8ull * rows * words), "hipMalloc(phase)");
	PGH_HIP(hipMemset(ds->d_ph_present, 0, 8ull * rows * words), "phase memset");
	PGH_HIP(hipMemset(ds->d_ph_info, 0, 8ull * rows * words), "phase memset");
	ds->ph_rows = rows;
	return PGH_OK;
}

static int AppendPhaseTracksHost(pgh_dataset *ds, const pgh::RecordFile &file, const std::vector<uint32_t> &variants,
                                 char *errbuf) {
	const uint32_t N = ds->sample_ct;
	const uint32_t words = (N + 63) / 64;
	pgh::Normalizer norm(ds->index, file);
	std::vector<uint8_t> row, pp, pi;
	std::vector<uint64_t> bits(2ull * words);
	std::string err;
	for (uint32_t v : variants) {
		if (!norm.DecodePhase(v, row, pp, pi, err)) {
			SetErr(errbuf, err);
			return PGH_ERR_FORMAT;
		}
		std::fill(bits.begin(), bits.end(), 0ull);
		for (uint32_t s = 0; s < N; s++) {
			if (pp[s]) {
				bits[s >> 6] |= 1ull << (s & 63);
				if (pi[s]) {
					bits[words + (s >> 6)] |= 1ull << (s & 63);
				}
			}
		}
		const uint64_t at = static_cast<uint64_t>(ds->ph_row_of[v - ds->v_begin]) * words;
		PGH_HIP(hipMemcpy(ds->d_ph_present + at, bits.data(), 8ull * words, hipMemcpyHostToDevice), "phase upload");
		PGH_HIP(hipMemcpy(ds->d_ph_info + at, bits.data() + words, 8ull * words, hipMemcpyHostToDevice), "phase upload");
	}
	return PGH_OK;
}

extern "C" int pgh_open(const char *pgen_path, const char *pgi_path, uint32_t variant_begin, uint32_t variant_end,
                        pgh_dataset **out, char *errbuf) {
	if (!pgen_path || !out) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	*out = nullptr;
	const char *trace_env = std::getenv("PGH_TRACE_OPEN");
	const bool trace = trace_env && *trace_env && *trace_env != '0';
	const auto t_start = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (trace) {
			std::fprintf(stderr, "pgh_open: %-14s +%.2f ms\n", what,
			             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
		}
	};
	std::unique_ptr<pgh_dataset> ds(new pgh_dataset());
	std::string err;
	if (!pgh::ParsePgenIndex(pgen_path, pgi_path ? pgi_path : "", ds->index, err)) {
		SetErr(errbuf, err);
		return err.find("cannot open") != std::string::npos ? PGH_ERR_OPEN : PGH_ERR_FORMAT;
	}
	const PgenIndex &ix = ds->index;
	if (variant_end == UINT32_MAX) {
		variant_end = ix.variant_ct;
	}
	if (variant_begin > variant_end || variant_end > ix.variant_ct) {
		SetErr(errbuf, "variant range out of bounds");
		return PGH_ERR_ARG;
	}
	ds->has_file = true;
	ds->pgen_path = pgen_path;
	ds->raw_variant_ct = ix.variant_ct;
	ds->sample_ct = ix.sample_ct;
	ds->record_bytes = ix.RecordBytes();
	ds->pitch = ChoosePitch(ds->record_bytes);
	ds->v_begin = variant_begin;
	ds->v_end = variant_end;
	PGH_HIP(hipGetDevice(&ds->device), "hipGetDevice");
	lap("index parsed");
	int rc = AllocRows(ds.get(), errbuf);
	lap("rows allocated");
	DosageStaging dosage;
	if (rc == PGH_OK && ix.has_dosage) {
		rc = PrepareDosage(ds.get(), dosage, errbuf);
	}
	if (rc == PGH_OK && ix.has_phase) {
		rc = PreparePhase(ds.get(), errbuf);
	}
	std::vector<uint32_t> host_phase; // phased variants whose track the host parses
	const bool phase_on_device = ds->sample_ct <= pgh::PhaseIngestMaxSamples();
	if (rc != PGH_OK) {
		pgh_close(ds.release());
		return rc;
	}
	lap("dosage arrays");
	auto has_track = [&](uint32_t r) { return ds->dos_rows != 0 && ds->dos_row_of[r - variant_begin] >= 0; };
	auto has_phase = [&](uint32_t r) { return ds->ph_rows != 0 && ds->ph_row_of[r - variant_begin] >= 0; };

	// Stream the body through two pinned staging buffers, three ways per run of records:
	//   plain   a long run of literal 2-bit records is already the row image: pread into the
	//           pinned buffer, re-pitch on the copy engine;
	//   device  anything else: the records' file bytes go up as they are and
	//           k_decode_records expands them in HBM (decode.hip);
	//   host    an LD run whose base lies before the opened range (or PGH_HOST_NORMALIZE=1):
	//           the host normaliser expands rows, which are then copied.
	pgh::RecordFile file;
	if (!file.Open(pgen_path, err)) {
		SetErr(errbuf, err);
		pgh_close(ds.release());
		return PGH_ERR_OPEN;
	}
	pgh::Normalizer norm(ix, file);
	const char *force_host = std::getenv("PGH_HOST_NORMALIZE");
	const bool host_only = force_host && *force_host && *force_host != '0';
	const uint32_t rb = ds->record_bytes;
	auto is_ld = [&](uint32_t r) { return (ix.vrtype[r] & 6u) == 2u; }; // types 2 and 3
	auto is_plain = [&](uint32_t r) { return ix.vrtype[r] == 0 && ix.offset[r + 1] - ix.offset[r] == rb; };
	// plain_run[i]: length of the run of plain records starting at variant_begin + i
	const uint32_t range = variant_end - variant_begin;
	std::vector<uint32_t> plain_run(static_cast<size_t>(range) + 1, 0);
	bool any_encoded = false;
	for (uint32_t i = range; i-- > 0;) {
		plain_run[i] = is_plain(variant_begin + i) ? plain_run[i + 1] + 1 : 0;
		any_encoded |= plain_run[i] == 0;
	}
	constexpr uint32_t kMinPlainRun = 256; // shorter plain runs ride along with their encoded neighbours
	const uint64_t stage_bytes = std::max<uint64_t>(64ull << 20, ds->pitch + 4096);
	const uint32_t rows_per_stage = static_cast<uint32_t>(std::max<uint64_t>(1, stage_bytes / ds->pitch));
	StageSet staging;
	int *d_error = nullptr;
	hipEvent_t done[kStages] = {};
	hipStream_t stream = nullptr;
	auto cleanup = [&]() {
		ReleaseStage(staging);
		for (int i = 0; i < kStages; i++) {
			if (done[i]) {
				(void)hipEventDestroy(done[i]);
			}
		}
		if (d_error) {
			(void)hipFree(d_error);
		}
		if (stream) {
			(void)hipStreamDestroy(stream);
		}
	};
	const bool device_decode = any_encoded && !host_only;
	hipError_t e = hipStreamCreateWithFlags(&stream, hipStreamNonBlocking);
	if (e == hipSuccess) {
		e = AcquireStage(stage_bytes, ds->device, device_decode, staging);
	}
	for (int i = 0; i < kStages && e == hipSuccess; i++) {
		e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
	}
	uint8_t *const *stage = staging.pinned;
	uint8_t *const *d_stage = staging.device;
	if (e == hipSuccess && device_decode) {
		e = PghMalloc(reinterpret_cast<void **>(&d_error), sizeof(int));
		if (e == hipSuccess) {
			e = hipMemsetAsync(d_error, 0, sizeof(int), stream);
		}
	}
	if (e != hipSuccess) {
		cleanup();
		pgh_close(ds.release());
		return DeviceFail(errbuf, "staging setup", e);
	}
	lap("staging ready");
	auto fail = [&](int code, const std::string &msg) {
		(void)hipStreamSynchronize(stream);
		cleanup();
		SetErr(errbuf, msg);
		pgh_close(ds.release());
		return code;
	};
	int which = 0;
	bool used[kStages] = {};
	int64_t last_base = -1; // most recent non-LD variant inside the opened range
	uint32_t v = variant_begin;
	while (v < variant_end) {
		if (used[which]) {
			e = hipEventSynchronize(done[which]);
			if (e != hipSuccess) {
				break;
			}
		}
		uint8_t *d_dst = ds->d_rows + static_cast<uint64_t>(v - variant_begin) * ds->pitch;
		const uint32_t room = std::min<uint64_t>(rows_per_stage, variant_end - v);
		const uint32_t run = plain_run[v - variant_begin];
		uint32_t stop = v;
		if (!host_only && run > 0 && (run >= kMinPlainRun || run >= variant_end - v)) {
			stop = v + std::min(run, room);
			if (!ReadParallel(file, ix.offset[v], static_cast<size_t>(stop - v) * rb, stage[which], err)) {
				return fail(PGH_ERR_OPEN, err);
			}
			e = hipMemsetAsync(d_dst, 0, static_cast<size_t>(stop - v) * ds->pitch, stream);
			if (e == hipSuccess) {
				e = hipMemcpy2DAsync(d_dst, ds->pitch, stage[which], rb, rb, stop - v, hipMemcpyHostToDevice, stream);
			}
			if (e == hipSuccess) {
				e = pgh::LaunchSanitizeTail(d_dst, ds->pitch, ds->sample_ct, stop - v, stream);
			}
			last_base = static_cast<int64_t>(stop) - 1;
		} else {
			// how many records fit as raw bytes + their tables?
			uint32_t n = 0;
			uint64_t raw = 0;
			// A multiallelic record (0x08) keeps its ALT patches between the main track and the phase / dosage tracks.
			// The device kernels do not measure that track, the host parser does (Normalizer::SkipAux1): a record that
			// has both goes through the host-rows path by itself (a multiallelic record WITHOUT further tracks is
			// expanded on the device like any other -- its main track is PgrGet's answer).
			auto host_tracks = [&](uint32_t r) { return (ix.vrtype[r] & 0x08) != 0 && (has_track(r) || has_phase(r)); };
			if (!host_only && !(is_ld(v) && last_base < 0) && !host_tracks(v)) {
				// the stage holds file bytes here, not rows: only its byte budget limits the run
				const uint32_t left = variant_end - v;
				while (n < left) {
					const uint32_t r = v + n;
					if (n > 0 && (plain_run[r - variant_begin] >= kMinPlainRun || host_tracks(r))) {
						break;
					}
					const uint64_t len = ix.offset[r + 1] - ix.offset[r];
					if (raw + len + 64 + 41ull * (n + 2) > stage_bytes) {
						break;
					}
					raw += len;
					n++;
				}
			}
			if (n == 0) {
				// host rows: everything under PGH_HOST_NORMALIZE, an LD run without a resident base,
				// or one record too large to stage
				stop = v + 1;
				if (host_only) {
					stop = v + room;
				} else {
					while (stop < v + room && is_ld(stop) && last_base < 0) {
						stop++;
					}
				}
				if (!ExpandParallel(ix, file, norm, v, stop, stage[which], ds->pitch, err)) {
					return fail(PGH_ERR_FORMAT, err);
				}
				e = hipMemcpyAsync(d_dst, stage[which], static_cast<size_t>(stop - v) * ds->pitch,
				                   hipMemcpyHostToDevice, stream);
				for (uint32_t r = v; r < stop; r++) {
					if (!is_ld(r)) {
						last_base = r;
					}
					if (has_track(r)) {
						dosage.host_parsed.push_back(r);
					}
					if (has_phase(r)) {
						host_phase.push_back(r);
					}
				}
			} else {
				stop = v + n;
				uint8_t *h = stage[which];
				if (!ReadParallel(file, ix.offset[v], raw, h, err)) {
					return fail(PGH_ERR_OPEN, err);
				}
				const uint64_t tables = (raw + 16 + 15) & ~15ull; // 16 zero bytes the kernel may read past the end
				std::memset(h + raw, 0, tables - raw);
				// tables behind the bytes: rec_begin u64[n+1] | aux_at u64[n] | track u64[n] | ld_row u32[n] |
				// dos_row i32[n] | count u32[n] | ph_row i32[n] | vrtype u8[n]  (aux_at / track / count are device scratch)
				uint64_t *rec_begin = reinterpret_cast<uint64_t *>(h + tables);
				uint32_t *ld_row = reinterpret_cast<uint32_t *>(rec_begin + (n + 1) + 2ull * n);
				int32_t *dos_row = reinterpret_cast<int32_t *>(ld_row + n);
				int32_t *ph_row = dos_row + 2ull * n;
				uint8_t *vrtype = reinterpret_cast<uint8_t *>(ph_row + n);
				bool any_ld = false;
				int64_t first_track = -1;
				uint32_t n_tracks = 0, n_phased = 0;
				for (uint32_t i = 0; i < n; i++) {
					const uint32_t r = v + i;
					rec_begin[i] = ix.offset[r] - ix.offset[v];
					vrtype[i] = ix.vrtype[r];
					dos_row[i] = has_track(r) ? ds->dos_row_of[r - variant_begin] : -1;
					if (dos_row[i] >= 0) {
						first_track = first_track < 0 ? dos_row[i] : first_track;
						n_tracks++;
					}
					ph_row[i] = -1;
					if (has_phase(r)) {
						if (phase_on_device) {
							ph_row[i] = ds->ph_row_of[r - variant_begin];
							n_phased++;
						} else {
							host_phase.push_back(r);
						}
					}
					if (is_ld(r)) {
						any_ld = true;
						ld_row[i] = last_base < 0 ? 0xffffffffu : static_cast<uint32_t>(last_base - variant_begin);
					} else {
						ld_row[i] = 0;
						last_base = r;
					}
				}
				rec_begin[n] = raw;
				const uint64_t used_bytes = tables + 8ull * (n + 1) + 16ull * n + 16ull * n + n;
				e = hipMemcpyAsync(d_stage[which], h, used_bytes, hipMemcpyHostToDevice, stream);
				if (e == hipSuccess) {
					pgh::DecodeBatch batch;
					batch.bytes = d_stage[which];
					batch.bytes_len = raw;
					batch.rec_begin = reinterpret_cast<const uint64_t *>(d_stage[which] + tables);
					uint64_t *d_aux = const_cast<uint64_t *>(batch.rec_begin) + (n + 1);
					uint64_t *d_track = d_aux + n;
					batch.ld_row = reinterpret_cast<const uint32_t *>(d_track + n);
					const int32_t *d_dos_row = reinterpret_cast<const int32_t *>(batch.ld_row + n);
					uint32_t *d_count = reinterpret_cast<uint32_t *>(const_cast<int32_t *>(d_dos_row) + n);
					const int32_t *d_ph_row = reinterpret_cast<const int32_t *>(d_count + n);
					batch.vrtype = reinterpret_cast<const uint8_t *>(d_ph_row + n);
					batch.rows = ds->d_rows;
					batch.pitch = ds->pitch;
					batch.row0 = v - variant_begin;
					batch.variant0 = v;
					batch.n = n;
					batch.sample_ct = ds->sample_ct;
					batch.id_bytes = ix.sample_id_bytes;
					batch.error = d_error;
					batch.aux_at = (n_tracks || n_phased) ? d_aux : nullptr;
					e = pgh::LaunchDecodeRecords(batch, any_ld, stream);
					if (e == hipSuccess && n_phased) {
						pgh::PhaseIngest ph;
						ph.bytes = batch.bytes;
						ph.bytes_len = raw;
						ph.rec_begin = batch.rec_begin;
						ph.vrtype = batch.vrtype;
						ph.aux_at = d_aux;
						ph.ph_row = d_ph_row;
						ph.rows = ds->d_rows;
						ph.pitch = ds->pitch;
						ph.row0 = batch.row0;
						ph.variant0 = v;
						ph.n = n;
						ph.sample_ct = ds->sample_ct;
						ph.present = ds->d_ph_present;
						ph.info = ds->d_ph_info;
						ph.words = (ds->sample_ct + 63) / 64;
						ph.error = d_error;
						e = pgh::LaunchPhaseIngest(ph, stream);
					}
					if (e == hipSuccess && n_tracks) {
						pgh::DosageIngest in;
						in.bytes = batch.bytes;
						in.bytes_len = raw;
						in.rec_begin = batch.rec_begin;
						in.vrtype = batch.vrtype;
						in.aux_at = d_aux;
						in.dos_row = d_dos_row;
						in.rows = ds->d_rows;
						in.pitch = ds->pitch;
						in.row0 = batch.row0;
						in.variant0 = v;
						in.n = n;
						in.sample_ct = ds->sample_ct;
						in.id_bytes = ix.sample_id_bytes;
						in.present = ds->d_dos_present;
						in.rank = ds->d_dos_rank;
						in.words = (ds->sample_ct + 63) / 64;
						in.val_off = ds->d_dos_val_off;
						in.values = ds->d_dos_values;
						in.capacity = dosage.capacity;
						in.total = dosage.d_total;
						in.count = d_count;
						in.track = d_track;
						in.error = d_error;
						e = pgh::LaunchDosageIngest(in, static_cast<uint32_t>(first_track), n_tracks, stream);
					}
				}
			}
		}
		if (e == hipSuccess) {
			e = hipEventRecord(done[which], stream);
		}
		if (e != hipSuccess) {
			break;
		}
		used[which] = true;
		which = (which + 1) % kStages;
		v = stop;
	}
	lap("runs enqueued");
	int bad_variant = 0;
	if (e == hipSuccess && d_error) {
		e = hipMemcpyAsync(&bad_variant, d_error, sizeof(int), hipMemcpyDeviceToHost, stream);
	}
	if (e == hipSuccess) {
		e = hipStreamSynchronize(stream);
	}
	lap("stream drained");
	cleanup();
	lap("staging freed");
	if (e != hipSuccess) {
		pgh_close(ds.release());
		return DeviceFail(errbuf, "genotype upload", e);
	}
	if (bad_variant != 0) {
		SetErr(errbuf, "malformed variant record " + std::to_string(bad_variant - 1));
		pgh_close(ds.release());
		return PGH_ERR_FORMAT;
	}
	if (!host_phase.empty()) {
		rc = AppendPhaseTracksHost(ds.get(), file, host_phase, errbuf);
		if (rc != PGH_OK) {
			pgh_close(ds.release());
			return rc;
		}
	}
	if (ds->dos_rows) {
		uint64_t filled = 0;
		hipError_t de = hipMemcpy(&filled, dosage.d_total, 8, hipMemcpyDeviceToHost);
		if (de != hipSuccess) {
			pgh_close(ds.release());
			return DeviceFail(errbuf, "dosage total", de);
		}
		if (!dosage.host_parsed.empty()) {
			rc = AppendDosageTracksHost(ds.get(), file, dosage.host_parsed, dosage.capacity, filled, errbuf);
			if (rc != PGH_OK) {
				pgh_close(ds.release());
				return rc;
			}
		}
		ds->dos_values = filled;
		rc = FetchDosageRowCounts(ds.get(), errbuf);
		if (rc != PGH_OK) {
			pgh_close(ds.release());
			return rc;
		}
		lap("dosage tracks");
	}
	*out = ds.release();
	return PGH_OK;
}

extern "C" int pgh_normalize_range_host(const char *pgen_path, const char *pgi_path, uint32_t variant_begin,
                                        uint32_t variant_end, uint8_t *rows, size_t row_stride, char *errbuf) {
	if (!pgen_path || (!rows && variant_end > variant_begin)) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	PgenIndex ix;
	std::string err;
	if (!pgh::ParsePgenIndex(pgen_path, pgi_path ? pgi_path : "", ix, err)) {
		SetErr(errbuf, err);
		return err.find("cannot open") != std::string::npos ? PGH_ERR_OPEN : PGH_ERR_FORMAT;
	}
	if (variant_end == UINT32_MAX) {
		variant_end = ix.variant_ct;
	}
	if (variant_begin > variant_end || variant_end > ix.variant_ct || row_stride < ix.RecordBytes()) {
		SetErr(errbuf, "variant range or row_stride out of bounds");
		return PGH_ERR_ARG;
	}
	pgh::RecordFile file;
	if (!file.Open(pgen_path, err)) {
		SetErr(errbuf, err);
		return PGH_ERR_OPEN;
	}
	pgh::Normalizer norm(ix, file);
	if (!norm.ExpandRange(variant_begin, variant_end, rows, row_stride, err)) {
		SetErr(errbuf, err);
		return PGH_ERR_FORMAT;
	}
	return PGH_OK;
}

extern "C" int pgh_from_host_rows(const uint8_t *rows, size_t row_stride, uint32_t variant_ct, uint32_t sample_ct,
                                  pgh_dataset **out, char *errbuf) {
	if (!out || (!rows && variant_ct) || sample_ct == 0) {
		SetErr(errbuf, "bad argument");
		return PGH_ERR_ARG;
	}
	*out = nullptr;
	std::unique_ptr<pgh_dataset> ds(new pgh_dataset());
	ds->raw_variant_ct = variant_ct;
	ds->sample_ct = sample_ct;
	ds->record_bytes = (sample_ct + 3) / 4;
	ds->pitch = ChoosePitch(ds->record_bytes);
	ds->v_begin = 0;
	ds->v_end = variant_ct;
	if (row_stride < ds->record_bytes) {
		SetErr(errbuf, "row_stride smaller than ceil(N/4)");
		return PGH_ERR_ARG;
	}
	PGH_HIP(hipGetDevice(&ds->device), "hipGetDevice");
	int rc = AllocRows(ds.get(), errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	if (variant_ct) {
		hipError_t e = hipMemset(ds->d_rows, 0, static_cast<size_t>(variant_ct) * ds->pitch);
		if (e == hipSuccess) {
			e = hipMemcpy2D(ds->d_rows, ds->pitch, rows, row_stride, ds->record_bytes, variant_ct,
			                hipMemcpyHostToDevice);
		}
		if (e == hipSuccess) {
			e = pgh::LaunchSanitizeTail(ds->d_rows, ds->pitch, sample_ct, variant_ct, nullptr);
		}
		if (e == hipSuccess) {
			e = hipDeviceSynchronize();
		}
		if (e != hipSuccess) {
			pgh_close(ds.release());
			return DeviceFail(errbuf, "row upload", e);
		}
	}
	*out = ds.release();
	return PGH_OK;
}

extern "C" int pgh_synth_create(uint32_t variant_begin, uint32_t variant_end, uint32_t sample_ct, uint64_t seed,
                                double missing_rate, pgh_dataset **out, char *errbuf) {
	if (!out || variant_begin > variant_end || sample_ct == 0) {
		SetErr(errbuf, "bad argument");
		return PGH_ERR_ARG;
	}
	*out = nullptr;
	std::unique_ptr<pgh_dataset> ds(new pgh_dataset());
	ds->raw_variant_ct = variant_end;
	ds->sample_ct = sample_ct;
	ds->record_bytes = (sample_ct + 3) / 4;
	ds->pitch = ChoosePitch(ds->record_bytes);
	ds->v_begin = variant_begin;
	ds->v_end = variant_end;
	PGH_HIP(hipGetDevice(&ds->device), "hipGetDevice");
	int rc = AllocRows(ds.get(), errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	hipError_t e = pgh::LaunchSynthFill(ds->d_rows, ds->pitch, sample_ct, variant_begin, variant_end - variant_begin,
	                                    seed, pgh::SynthMissThreshold(missing_rate), nullptr);
	if (e == hipSuccess) {
		e = hipDeviceSynchronize();
	}
	if (e != hipSuccess) {
		pgh_close(ds.release());
		return DeviceFail(errbuf, "synthetic fill", e);
	}
	*out = ds.release();
	return PGH_OK;
}

extern "C" int pgh_synth_add_dosage(pgh_dataset *ds, double rate, uint64_t seed, char *errbuf) {
	if (!ds || ds->dos_rows || !(rate >= 0.0 && rate <= 1.0)) {
		SetErr(errbuf, "bad argument (null dataset, tracks already present, or rate outside [0, 1])");
		return PGH_ERR_ARG;
	}
	PGH_ONE_DEVICE(ds);
	PGH_ENTER(ds);
	const uint32_t rows = ds->v_end - ds->v_begin;
	const uint32_t words = (ds->sample_ct + 63) / 64;
	if (rows == 0) {
		return PGH_OK;
	}
	hipStream_t st = PghThreadStream();
	std::vector<uint64_t> off(rows + 1);
	HostSourceFence fence(st); // `off` feeds an asynchronous upload (dos_row_of lives in the handle)
	ds->dos_row_of.resize(rows);
	for (uint32_t i = 0; i < rows; i++) {
		ds->dos_row_of[i] = static_cast<int32_t>(i);
	}
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&ds->d_dos_row_of), sizeof(int32_t) * rows), "hipMalloc(dosage)");
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&ds->d_dos_present), 8ull * rows * words), "hipMalloc(dosage)");
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&ds->d_dos_rank), 4ull * rows * words), "hipMalloc(dosage)");
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&ds->d_dos_val_off), 8ull * (rows + 1)), "hipMalloc(dosage)");
	PGH_HIP(hipMemcpyAsync(ds->d_dos_row_of, ds->dos_row_of.data(), sizeof(int32_t) * rows, hipMemcpyHostToDevice, st),
	        "dosage upload");
	PGH_HIP(pgh::LaunchSynthDosageBits(ds->d_dos_present, rows, words, ds->sample_ct, ds->v_begin, seed, rate, st),
	        "synthetic dosage bits");
	PGH_HIP(pgh::LaunchDosageRank(ds->d_dos_present, rows, words, ds->d_dos_rank, st), "dosage rank kernel");
	PGH_HIP(pgh::LaunchDosageRowTotals(ds->d_dos_present, ds->d_dos_rank, rows, words, ds->d_dos_val_off, st),
	        "dosage totals kernel");
	PGH_HIP(hipMemcpyAsync(off.data(), ds->d_dos_val_off, 8ull * rows, hipMemcpyDeviceToHost, st), "dosage totals copy");
	PGH_HIP(hipStreamSynchronize(st), "dosage totals sync");
	uint64_t total = 0;
	for (uint32_t r = 0; r < rows; r++) {
		const uint64_t c = off[r];
		off[r] = total;
		total += c;
	}
	off[rows] = total;
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&ds->d_dos_values), 2 * total + 32), "hipMalloc(dosage)");
	PGH_HIP(hipMemsetAsync(ds->d_dos_values, 0, 2 * total + 32, st), "dosage memset");
	PGH_HIP(hipMemcpyAsync(ds->d_dos_val_off, off.data(), 8ull * (rows + 1), hipMemcpyHostToDevice, st), "dosage upload");
	PGH_HIP(pgh::LaunchSynthDosageValues(ds->d_dos_present, ds->d_dos_rank, ds->d_dos_val_off, ds->d_dos_values, rows, words,
	                                     ds->sample_ct, ds->v_begin, seed, st),
	        "synthetic dosage values");
	PGH_HIP(hipStreamSynchronize(st), "synthetic dosage sync");
	ds->dos_rows = rows;
	ds->dos_values = total;
	return FetchDosageRowCounts(ds, errbuf);
}

extern "C" int pgh_synth_record_host(uint32_t v, uint32_t sample_ct, uint64_t seed, double missing_rate,
                                     uint8_t *out) {
	if (!out || sample_ct == 0) {
		return PGH_ERR_ARG;
	}
	const uint32_t thr = pgh::SynthMissThreshold(missing_rate);
	const pgh::SynthVariant sv = pgh::SynthVariantParams(seed, v);
	std::memset(out, 0, (static_cast<size_t>(sample_ct) + 3) / 4);
	for (uint32_t s = 0; s < sample_ct; s++) {
		out[s >> 2] |= static_cast<uint8_t>(pgh::SynthGenotype(sv, s, thr) << (2 * (s & 3)));
	}
	return PGH_OK;
}

static int WriteSynthCompanions(const std::string &base, uint32_t variant_ct, uint32_t sample_ct, char *errbuf) {
	FILE *f = std::fopen((base + ".pvar").c_str(), "w");
	if (!f) {
		SetErr(errbuf, "cannot create '" + base + ".pvar'");
		return PGH_ERR_OPEN;
	}
	std::fprintf(f, "#CHROM\tPOS\tID\tREF\tALT\n");
	for (uint32_t v = 0; v < variant_ct; v++) {
		// 22 autosomes, equal-sized runs, ascending positions
		const uint32_t per_chrom = (variant_ct + 21) / 22;
		std::fprintf(f, "%u\t%u\tsv%u\tA\tG\n", v / per_chrom + 1, (v % per_chrom + 1) * 100, v);
	}
	std::fclose(f);
	f = std::fopen((base + ".psam").c_str(), "w");
	if (!f) {
		SetErr(errbuf, "cannot create '" + base + ".psam'");
		return PGH_ERR_OPEN;
	}
	std::fprintf(f, "#FID\tIID\tSEX\n");
	for (uint32_t s = 0; s < sample_ct; s++) {
		std::fprintf(f, "F%u\tS%u\t%u\n", s / 4, s, 1 + (s & 1));
	}
	std::fclose(f);
	return PGH_OK;
}

// A synthetic .pgen whose records carry a 0x60 dosage track behind the 2-bit bytes pgh_synth_write_files
// would write: presence bits Bernoulli(dosage_rate) per sample, values uniform on 0..32768.  Records vary
// in length, so the header tables are written after the body.
static int WriteSynthDosagePgen(const std::string &base, uint32_t variant_ct, uint32_t sample_ct, uint64_t seed,
                                   double missing_rate, double dosage_rate, char *errbuf) {
	const uint32_t rb = (sample_ct + 3) / 4, pb = (sample_ct + 7) / 8;
	FILE *f = std::fopen((base + ".pgen").c_str(), "wb");
	if (!f) {
		SetErr(errbuf, "cannot create '" + base + ".pgen'");
		return PGH_ERR_OPEN;
	}
	const uint32_t blocks = (variant_ct + 65535) / 65536;
	const uint64_t table_bytes = 12 + 8ull * blocks + 5ull * variant_ct; // 8-bit vrtypes, 4-byte record lengths
	std::vector<uint8_t> pad(table_bytes, 0);
	bool ok = std::fwrite(pad.data(), 1, pad.size(), f) == pad.size();
	std::vector<uint32_t> lens(variant_ct);
	const uint32_t threshold = pgh::SynthMissThreshold(dosage_rate);
	const uint32_t workers = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
	const uint32_t block_rows = std::max<uint32_t>(workers, static_cast<uint32_t>((64ull << 20) / (rb + pb + 2ull * sample_ct)));
	std::vector<std::vector<uint8_t>> recs(block_rows);
	for (uint32_t v0 = 0; ok && v0 < variant_ct; v0 += block_rows) {
		const uint32_t n = std::min(block_rows, variant_ct - v0);
		std::vector<std::thread> pool;
		for (uint32_t t = 0; t < workers; t++) {
			pool.emplace_back([&, t] {
				for (uint32_t i = t; i < n; i += workers) {
					const uint32_t v = v0 + i;
					std::vector<uint8_t> &rec = recs[i];
					rec.assign(static_cast<size_t>(rb) + pb, 0);
					pgh_synth_record_host(v, sample_ct, seed, missing_rate, rec.data());
					const uint64_t key = pgh::Mix64(pgh::Mix64(seed) ^ (static_cast<uint64_t>(v) << 32) ^ 0x5851f42d4c957f2dULL);
					for (uint32_t s = 0; s < sample_ct; s++) {
						const uint64_t h = pgh::Mix64(key ^ s);
						if (static_cast<uint32_t>(h) < threshold) {
							rec[rb + (s >> 3)] |= static_cast<uint8_t>(1u << (s & 7));
							const uint32_t val = static_cast<uint32_t>(((h >> 32) * 32769ull) >> 32);
							rec.push_back(static_cast<uint8_t>(val));
							rec.push_back(static_cast<uint8_t>(val >> 8));
						}
					}
					lens[v] = static_cast<uint32_t>(rec.size());
				}
			});
		}
		for (auto &th : pool) {
			th.join();
		}
		for (uint32_t i = 0; ok && i < n; i++) {
			ok = std::fwrite(recs[i].data(), 1, recs[i].size(), f) == recs[i].size();
		}
	}
	std::vector<uint8_t> head = {0x6c, 0x1b, 0x10};
	auto put = [&](uint64_t v, int n) {
		for (int i = 0; i < n; i++) {
			head.push_back(static_cast<uint8_t>(v >> (8 * i)));
		}
	};
	put(variant_ct, 4);
	put(sample_ct, 4);
	head.push_back(0x40 | 4 | 3);
	uint64_t at = table_bytes;
	for (uint32_t b = 0; b < blocks; b++) {
		put(at, 8);
		const uint32_t lo = b * 65536u, hi = std::min<uint64_t>(variant_ct, (b + 1ull) * 65536u);
		for (uint32_t v = lo; v < hi; v++) {
			at += lens[v];
		}
	}
	for (uint32_t b = 0; b < blocks; b++) {
		const uint32_t lo = b * 65536u, hi = std::min<uint64_t>(variant_ct, (b + 1ull) * 65536u);
		head.insert(head.end(), hi - lo, 0x60);
		for (uint32_t v = lo; v < hi; v++) {
			put(lens[v], 4);
		}
	}
	ok = ok && head.size() == table_bytes && std::fseek(f, 0, SEEK_SET) == 0 &&
	     std::fwrite(head.data(), 1, head.size(), f) == head.size();
	ok = (std::fclose(f) == 0) && ok;
	if (!ok) {
		SetErr(errbuf, "write failed on '" + base + ".pgen'");
		return PGH_ERR_OPEN;
	}
	return PGH_OK;
}

extern "C" int pgh_synth_write_files(const char *prefix, uint32_t variant_ct, uint32_t sample_ct, uint64_t seed,
                                     double missing_rate, char *errbuf) {
	if (!prefix || sample_ct == 0) {
		SetErr(errbuf, "bad argument");
		return PGH_ERR_ARG;
	}
	const std::string base(prefix);
	const uint32_t rb = (sample_ct + 3) / 4;
	if (rb > 0xffffffu) {
		SetErr(errbuf, "sample count too large for 3-byte record lengths");
		return PGH_ERR_ARG;
	}
	FILE *f = std::fopen((base + ".pgen").c_str(), "wb");
	if (!f) {
		SetErr(errbuf, "cannot create '" + base + ".pgen'");
		return PGH_ERR_OPEN;
	}
	// mode 0x10; ctrl: 4-bit vrtypes + the narrowest record-length width, nonref mode 1
	const uint32_t len_bytes = rb < 0x100 ? 1 : (rb < 0x10000 ? 2 : 3);
	const uint8_t ctrl = static_cast<uint8_t>(0x40 | (len_bytes - 1));
	std::vector<uint8_t> head = {0x6c, 0x1b, 0x10};
	auto put = [&](uint64_t v, int n) {
		for (int i = 0; i < n; i++) {
			head.push_back(static_cast<uint8_t>(v >> (8 * i)));
		}
	};
	put(variant_ct, 4);
	put(sample_ct, 4);
	head.push_back(ctrl);
	const uint32_t blocks = (variant_ct + 65535) / 65536;
	uint64_t table_bytes = 12 + 8ull * blocks;
	for (uint32_t b = 0; b < blocks; b++) {
		const uint32_t cnt = std::min<uint32_t>(65536, variant_ct - b * 65536u);
		table_bytes += (cnt + 1) / 2 + static_cast<uint64_t>(cnt) * len_bytes;
	}
	for (uint32_t b = 0; b < blocks; b++) {
		put(table_bytes + static_cast<uint64_t>(b) * 65536ull * rb, 8);
	}
	for (uint32_t b = 0; b < blocks; b++) {
		const uint32_t cnt = std::min<uint32_t>(65536, variant_ct - b * 65536u);
		head.insert(head.end(), (cnt + 1) / 2, 0); // vrtype 0
		for (uint32_t i = 0; i < cnt; i++) {
			put(rb, static_cast<int>(len_bytes));
		}
	}
	bool ok = std::fwrite(head.data(), 1, head.size(), f) == head.size();
	// records are generated on the host (no GPU needed to make a fixture), a block of rows at a
	// time over a few threads
	const uint32_t workers = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
	const uint32_t block_rows = static_cast<uint32_t>(std::max<uint64_t>(workers, (32ull << 20) / rb));
	std::vector<uint8_t> block(static_cast<size_t>(block_rows) * rb);
	for (uint32_t v0 = 0; ok && v0 < variant_ct; v0 += block_rows) {
		const uint32_t n = std::min(block_rows, variant_ct - v0);
		std::vector<std::thread> pool;
		for (uint32_t t = 0; t < workers; t++) {
			pool.emplace_back([&, t] {
				for (uint32_t i = t; i < n; i += workers) {
					pgh_synth_record_host(v0 + i, sample_ct, seed, missing_rate, block.data() + static_cast<size_t>(i) * rb);
				}
			});
		}
		for (auto &th : pool) {
			th.join();
		}
		ok = std::fwrite(block.data(), 1, static_cast<size_t>(n) * rb, f) == static_cast<size_t>(n) * rb;
	}
	ok = (std::fclose(f) == 0) && ok;
	if (!ok) {
		SetErr(errbuf, "write failed on '" + base + ".pgen'");
		return PGH_ERR_OPEN;
	}
	return WriteSynthCompanions(base, variant_ct, sample_ct, errbuf);
}

extern "C" int pgh_synth_write_dosage_files(const char *prefix, uint32_t variant_ct, uint32_t sample_ct, uint64_t seed,
                                            double missing_rate, double dosage_rate, char *errbuf) {
	if (!prefix || sample_ct == 0 || !(dosage_rate >= 0.0 && dosage_rate <= 1.0)) {
		SetErr(errbuf, "bad argument");
		return PGH_ERR_ARG;
	}
	const std::string base(prefix);
	int rc = WriteSynthDosagePgen(base, variant_ct, sample_ct, seed, missing_rate, dosage_rate, errbuf);
	return rc == PGH_OK ? WriteSynthCompanions(base, variant_ct, sample_ct, errbuf) : rc;
}

extern "C" int pgh_copy_rows_to_host(const pgh_dataset *ds, uint32_t v_begin, uint32_t v_end, uint8_t *rows,
                                     size_t row_stride, char *errbuf) {
	int rc = CheckRange(ds, v_begin, v_end, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	if (v_end == v_begin) {
		return PGH_OK;
	}
	if (!rows || row_stride < ds->record_bytes) {
		SetErr(errbuf, "bad destination");
		return PGH_ERR_ARG;
	}
	if (ds->IsGroup()) {
		return pgh_group::CopyRowsToHost(ds, v_begin, v_end, rows, row_stride, errbuf);
	}
	PGH_ENTER(ds);
	PGH_HIP(hipMemcpy2D(rows, row_stride, ds->d_rows + static_cast<uint64_t>(v_begin - ds->v_begin) * ds->pitch,
	                    ds->pitch, ds->record_bytes, v_end - v_begin, hipMemcpyDeviceToHost),
	        "row download");
	return PGH_OK;
}

extern "C" int pgh_get_info(const pgh_dataset *ds, pgh_info *out) {
	if (!ds || !out) {
		return PGH_ERR_ARG;
	}
	if (ds->IsGroup()) {
		return pgh_group::GetInfo(ds, out);
	}
	if (ds->has_file) {
		FillInfo(ds->index, out);
	} else {
		std::memset(out, 0, sizeof *out);
		out->raw_variant_ct = ds->raw_variant_ct;
		out->raw_sample_ct = ds->sample_ct;
		out->record_bytes = ds->record_bytes;
		out->max_record_bytes = ds->record_bytes;
		out->vrtype_hist[0] = ds->v_end - ds->v_begin;
	}
	out->variant_begin = ds->v_begin;
	out->variant_end = ds->v_end;
	out->pitch_bytes = ds->pitch;
	out->device = ds->device;
	out->dosage_variant_ct = ds->dos_rows;
	out->dosage_value_ct = ds->dos_values;
	return PGH_OK;
}

extern "C" const void *pgh_device_rows(const pgh_dataset *ds) {
	return ds ? ds->d_rows : nullptr; // NULL for a shard group: ask its shards
}

extern "C" void pgh_close(pgh_dataset *ds) {
	if (!ds) {
		return;
	}
	if (ds->IsGroup()) {
		pgh_group::Close(ds);
		return;
	}
	PGH_ENTER(ds);
	for (void *p : {static_cast<void *>(ds->d_rows), static_cast<void *>(ds->d_dos_row_of),
	                static_cast<void *>(ds->d_dos_present), static_cast<void *>(ds->d_dos_rank),
	                static_cast<void *>(ds->d_dos_val_off), static_cast<void *>(ds->d_dos_values),
	                static_cast<void *>(ds->d_dos_rec), static_cast<void *>(ds->d_dos_rec_off),
	                static_cast<void *>(ds->d_ph_present), static_cast<void *>(ds->d_ph_info)}) {
		if (p) {
			(void)hipFree(p);
		}
	}
	delete ds;
	PghTrimBlockCache(); // cached work blocks were shaped by this dataset's calls
}

extern "C" void pgh_trim_device_cache(void) {
	PghTrimBlockCache();
}

// ---------------------------------------------------------------------------
// sample subsets
// ---------------------------------------------------------------------------

extern "C" int pgh_subset_create(const pgh_dataset *ds, const uint64_t *sample_include, pgh_subset **out,
                                 char *errbuf) {
	if (!ds || !sample_include || !out) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	*out = nullptr;
	if (ds->IsGroup()) {
		return pgh_group::SubsetCreate(ds, sample_include, out, errbuf);
	}
	PGH_ENTER(ds);
	std::unique_ptr<pgh_subset> ss(new pgh_subset());
	ss->ds = ds;
	ss->device = ds->device;
	const uint32_t N = ds->sample_ct;
	ss->include.assign(sample_include, sample_include + (N + 63) / 64);
	std::vector<uint8_t> mask2(ds->pitch, 0);
	for (uint32_t s = 0; s < N; s++) {
		if ((ss->include[s >> 6] >> (s & 63)) & 1ull) {
			ss->sel.push_back(s);
			mask2[s >> 2] |= static_cast<uint8_t>(1u << (2 * (s & 3)));
		}
	}
	ss->n_out = static_cast<uint32_t>(ss->sel.size());
	hipError_t e = PghMalloc(reinterpret_cast<void **>(&ss->d_mask2), ds->pitch);
	if (e == hipSuccess) {
		e = hipMemcpy(ss->d_mask2, mask2.data(), ds->pitch, hipMemcpyHostToDevice);
	}
	if (e == hipSuccess) {
		e = PghMalloc(reinterpret_cast<void **>(&ss->d_sel), sizeof(uint32_t) * std::max<uint32_t>(1, ss->n_out));
	}
	if (e == hipSuccess && ss->n_out) {
		e = hipMemcpy(ss->d_sel, ss->sel.data(), sizeof(uint32_t) * ss->n_out, hipMemcpyHostToDevice);
	}
	if (e == hipSuccess) {
		e = PghMalloc(reinterpret_cast<void **>(&ss->d_include), 8 * std::max<size_t>(1, ss->include.size()));
	}
	if (e == hipSuccess && !ss->include.empty()) {
		e = hipMemcpy(ss->d_include, ss->include.data(), 8 * ss->include.size(), hipMemcpyHostToDevice);
	}
	if (e != hipSuccess) {
		pgh_subset_destroy(ss.release());
		return DeviceFail(errbuf, "subset upload", e);
	}
	*out = ss.release();
	return PGH_OK;
}

extern "C" uint32_t pgh_subset_size(const pgh_subset *ss) {
	return ss ? ss->n_out : 0;
}

extern "C" void pgh_subset_destroy(pgh_subset *ss) {
	if (!ss) {
		return;
	}
	for (pgh_subset *part : ss->parts) {
		pgh_subset_destroy(part);
	}
	DeviceScope scope(ss->device);
	if (ss->d_mask2) {
		(void)hipFree(ss->d_mask2);
	}
	if (ss->d_sel) {
		(void)hipFree(ss->d_sel);
	}
	if (ss->d_include) {
		(void)hipFree(ss->d_include);
	}
	delete ss;
}

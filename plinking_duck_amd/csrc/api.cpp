// api.cpp -- the C ABI of libpgenhip (include/pgenhip.h): handles, HBM residency,
// launches.  Host code only; the kernels live in the *.hip files next to it.
#include "../../include/pgenhip.h"

#include "hwe_core.hpp"
#include "decode.hpp"
#include "kernels.hpp"
#include "ld.hpp"
#include "linalg.hpp"
#include "pgen_file.hpp"
#include "synth.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

using pgh::PgenIndex;
using pgh::RowView;

// ---------------------------------------------------------------------------
// handle types
// ---------------------------------------------------------------------------

struct pgh_dataset {
	int device = 0;
	bool has_file = false;
	std::string pgen_path;
	PgenIndex index; // valid when has_file
	uint32_t raw_variant_ct = 0;
	uint32_t sample_ct = 0;
	uint32_t record_bytes = 0;
	uint32_t v_begin = 0; // resident range
	uint32_t v_end = 0;
	uint64_t pitch = 0;
	uint8_t *d_rows = nullptr;

	RowView View() const {
		return RowView {d_rows, pitch, sample_ct, record_bytes};
	}
};

struct pgh_subset {
	const pgh_dataset *ds = nullptr;
	uint32_t n_out = 0;
	std::vector<uint64_t> include; // ceil(N/64) words
	std::vector<uint32_t> sel;     // raw index of each included sample, ascending
	uint8_t *d_mask2 = nullptr;    // one pitched row of 01 slots
	uint32_t *d_sel = nullptr;
};

struct pgh_reader {
	const pgh_dataset *ds = nullptr;
	const pgh_subset *subset = nullptr;
	hipStream_t stream = nullptr;
	// counts window: one launch serves the next kWindow per-variant calls
	static constexpr uint32_t kWindow = 128; // the reference's claim batch (src/plink_freq.cpp:413)
	uint32_t win_begin = 0, win_end = 0;
	uint32_t *d_counts = nullptr;
	uint32_t *h_counts = nullptr; // pinned
	uint8_t *h_row = nullptr;     // pinned, pitch bytes
	std::unique_ptr<pgh::RecordFile> file;
	std::unique_ptr<pgh::Normalizer> norm;
	std::string err;
};

namespace {

void SetErr(char *errbuf, const std::string &msg) {
	if (errbuf) {
		std::snprintf(errbuf, PGH_ERRBUF_LEN, "%s", msg.c_str());
	}
}

int DeviceFail(char *errbuf, const char *what, hipError_t e) {
	SetErr(errbuf, std::string(what) + ": " + hipGetErrorString(e));
	return PGH_ERR_DEVICE;
}

#define PGH_HIP(call, what)                                                                                            \
	do {                                                                                                               \
		hipError_t e_ = (call);                                                                                        \
		if (e_ != hipSuccess) {                                                                                        \
			return DeviceFail(errbuf, what, e_);                                                                       \
		}                                                                                                              \
	} while (0)

uint64_t ChoosePitch(uint32_t record_bytes) {
	// whole 16-byte lanes always; 128-byte (cache line) aligned rows once rows are long
	const uint64_t align = record_bytes >= 512 ? 128 : 16;
	uint64_t p = (static_cast<uint64_t>(record_bytes) + align - 1) / align * align;
	return p ? p : align;
}

// RAII device buffer for the host-output entry points
struct DevBuf {
	void *p = nullptr;
	~DevBuf() {
		if (p) {
			(void)hipFree(p);
		}
	}
	hipError_t Alloc(size_t bytes) {
		return hipMalloc(&p, bytes ? bytes : 16);
	}
	template <class T>
	T *As() {
		return static_cast<T *>(p);
	}
};

int CheckRange(const pgh_dataset *ds, uint32_t v_begin, uint32_t v_end, char *errbuf) {
	if (!ds) {
		SetErr(errbuf, "null dataset");
		return PGH_ERR_ARG;
	}
	if (v_begin > v_end || v_begin < ds->v_begin || v_end > ds->v_end) {
		char msg[160];
		std::snprintf(msg, sizeof msg, "variant range [%u, %u) is outside the resident range [%u, %u)", v_begin, v_end,
		              ds->v_begin, ds->v_end);
		SetErr(errbuf, msg);
		return PGH_ERR_ARG;
	}
	return PGH_OK;
}

int CheckSubset(const pgh_dataset *ds, const pgh_subset *ss, char *errbuf) {
	if (ss && ss->ds != ds) {
		SetErr(errbuf, "sample subset belongs to a different dataset");
		return PGH_ERR_ARG;
	}
	return PGH_OK;
}

// compact `raw` (one value per raw sample) to the included samples
template <class T>
void Compact(const pgh_subset *ss, const T *raw, size_t stride, T *out, uint32_t n_raw) {
	if (!ss) {
		for (uint32_t s = 0; s < n_raw; s++) {
			std::memcpy(out + static_cast<size_t>(s) * stride, raw + static_cast<size_t>(s) * stride,
			            sizeof(T) * stride);
		}
		return;
	}
	for (uint32_t k = 0; k < ss->n_out; k++) {
		std::memcpy(out + static_cast<size_t>(k) * stride, raw + static_cast<size_t>(ss->sel[k]) * stride,
		            sizeof(T) * stride);
	}
}

int AllocRows(pgh_dataset *ds, char *errbuf) {
	const uint64_t rows = ds->v_end - ds->v_begin;
	const uint64_t bytes = rows * ds->pitch;
	PGH_HIP(hipMalloc(reinterpret_cast<void **>(&ds->d_rows), bytes ? bytes : 16), "hipMalloc(genotype rows)");
	return PGH_OK;
}

} // namespace

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

// pgh_open's staging buffers (2 pinned + 2 device, 64 MB each) cost ~40 ms to allocate, more
// than a small file takes to ingest: one set per device is parked here between opens.
struct StageSet {
	uint8_t *pinned[2] = {nullptr, nullptr};
	uint8_t *device[2] = {nullptr, nullptr};
	uint64_t bytes = 0;
	int device_id = -1;
	void Free() {
		for (int i = 0; i < 2; i++) {
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
	for (int i = 0; i < 2 && e == hipSuccess; i++) {
		if (!out.pinned[i]) {
			e = hipHostMalloc(reinterpret_cast<void **>(&out.pinned[i]), bytes, hipHostMallocDefault);
		}
		if (e == hipSuccess && want_device && !out.device[i]) {
			e = hipMalloc(reinterpret_cast<void **>(&out.device[i]), std::max(bytes, out.bytes));
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
	const unsigned parts = static_cast<unsigned>(std::min<size_t>(std::min(8u, hw), std::max<size_t>(1, bytes / kMinSlice)));
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

extern "C" int pgh_open(const char *pgen_path, const char *pgi_path, uint32_t variant_begin, uint32_t variant_end,
                        pgh_dataset **out, char *errbuf) {
	if (!pgen_path || !out) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	*out = nullptr;
	std::unique_ptr<pgh_dataset> ds(new pgh_dataset());
	std::string err;
	if (!pgh::ParsePgenIndex(pgen_path, pgi_path ? pgi_path : "", ds->index, err)) {
		SetErr(errbuf, err);
		return err.find("cannot open") != std::string::npos ? PGH_ERR_OPEN : PGH_ERR_FORMAT;
	}
	const PgenIndex &ix = ds->index;
	if (ix.has_multiallelic) {
		SetErr(errbuf, "multiallelic hardcall tracks are not supported");
		return PGH_ERR_UNSUPPORTED;
	}
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
	int rc = AllocRows(ds.get(), errbuf);
	if (rc != PGH_OK) {
		return rc;
	}

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
	const char *trace_env = std::getenv("PGH_TRACE_OPEN");
	const bool trace = trace_env && *trace_env && *trace_env != '0';
	const auto t_start = std::chrono::steady_clock::now();
	auto lap = [&](const char *what) {
		if (trace) {
			std::fprintf(stderr, "pgh_open: %-14s +%.2f ms\n", what,
			             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count());
		}
	};
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
	hipEvent_t done[2] = {nullptr, nullptr};
	hipStream_t stream = nullptr;
	auto cleanup = [&]() {
		ReleaseStage(staging);
		for (int i = 0; i < 2; i++) {
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
	for (int i = 0; i < 2 && e == hipSuccess; i++) {
		e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
	}
	uint8_t *const *stage = staging.pinned;
	uint8_t *const *d_stage = staging.device;
	if (e == hipSuccess && device_decode) {
		e = hipMalloc(reinterpret_cast<void **>(&d_error), sizeof(int));
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
	bool used[2] = {false, false};
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
			if (!host_only && !(is_ld(v) && last_base < 0)) {
				// the stage holds file bytes here, not rows: only its byte budget limits the run
				const uint32_t left = variant_end - v;
				while (n < left) {
					const uint32_t r = v + n;
					if (n > 0 && plain_run[r - variant_begin] >= kMinPlainRun) {
						break;
					}
					const uint64_t len = ix.offset[r + 1] - ix.offset[r];
					if (raw + len + 64 + 13ull * (n + 2) > stage_bytes) {
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
				}
			} else {
				stop = v + n;
				uint8_t *h = stage[which];
				if (!ReadParallel(file, ix.offset[v], raw, h, err)) {
					return fail(PGH_ERR_OPEN, err);
				}
				const uint64_t tables = (raw + 16 + 15) & ~15ull; // 16 zero bytes the kernel may read past the end
				std::memset(h + raw, 0, tables - raw);
				uint64_t *rec_begin = reinterpret_cast<uint64_t *>(h + tables);
				uint32_t *ld_row = reinterpret_cast<uint32_t *>(rec_begin + (n + 1));
				uint8_t *vrtype = reinterpret_cast<uint8_t *>(ld_row + n);
				bool any_ld = false;
				for (uint32_t i = 0; i < n; i++) {
					const uint32_t r = v + i;
					rec_begin[i] = ix.offset[r] - ix.offset[v];
					vrtype[i] = ix.vrtype[r];
					if (is_ld(r)) {
						any_ld = true;
						ld_row[i] = last_base < 0 ? 0xffffffffu : static_cast<uint32_t>(last_base - variant_begin);
					} else {
						ld_row[i] = 0;
						last_base = r;
					}
				}
				rec_begin[n] = raw;
				const uint64_t used_bytes = tables + 8ull * (n + 1) + 4ull * n + n;
				e = hipMemcpyAsync(d_stage[which], h, used_bytes, hipMemcpyHostToDevice, stream);
				if (e == hipSuccess) {
					pgh::DecodeBatch batch;
					batch.bytes = d_stage[which];
					batch.bytes_len = raw;
					batch.rec_begin = reinterpret_cast<const uint64_t *>(d_stage[which] + tables);
					batch.ld_row = reinterpret_cast<const uint32_t *>(batch.rec_begin + (n + 1));
					batch.vrtype = reinterpret_cast<const uint8_t *>(batch.ld_row + n);
					batch.rows = ds->d_rows;
					batch.pitch = ds->pitch;
					batch.row0 = v - variant_begin;
					batch.variant0 = v;
					batch.n = n;
					batch.sample_ct = ds->sample_ct;
					batch.id_bytes = ix.sample_id_bytes;
					batch.error = d_error;
					e = pgh::LaunchDecodeRecords(batch, any_ld, stream);
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
		which ^= 1;
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
	f = std::fopen((base + ".pvar").c_str(), "w");
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
	PGH_HIP(hipMemcpy2D(rows, row_stride, ds->d_rows + static_cast<uint64_t>(v_begin - ds->v_begin) * ds->pitch,
	                    ds->pitch, ds->record_bytes, v_end - v_begin, hipMemcpyDeviceToHost),
	        "row download");
	return PGH_OK;
}

extern "C" int pgh_get_info(const pgh_dataset *ds, pgh_info *out) {
	if (!ds || !out) {
		return PGH_ERR_ARG;
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
	return PGH_OK;
}

extern "C" const void *pgh_device_rows(const pgh_dataset *ds) {
	return ds ? ds->d_rows : nullptr;
}

extern "C" void pgh_close(pgh_dataset *ds) {
	if (!ds) {
		return;
	}
	if (ds->d_rows) {
		(void)hipFree(ds->d_rows);
	}
	delete ds;
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
	std::unique_ptr<pgh_subset> ss(new pgh_subset());
	ss->ds = ds;
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
	hipError_t e = hipMalloc(reinterpret_cast<void **>(&ss->d_mask2), ds->pitch);
	if (e == hipSuccess) {
		e = hipMemcpy(ss->d_mask2, mask2.data(), ds->pitch, hipMemcpyHostToDevice);
	}
	if (e == hipSuccess) {
		e = hipMalloc(reinterpret_cast<void **>(&ss->d_sel), sizeof(uint32_t) * std::max<uint32_t>(1, ss->n_out));
	}
	if (e == hipSuccess && ss->n_out) {
		e = hipMemcpy(ss->d_sel, ss->sel.data(), sizeof(uint32_t) * ss->n_out, hipMemcpyHostToDevice);
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
	if (ss->d_mask2) {
		(void)hipFree(ss->d_mask2);
	}
	if (ss->d_sel) {
		(void)hipFree(ss->d_sel);
	}
	delete ss;
}

// ---------------------------------------------------------------------------
// batched device calls
// ---------------------------------------------------------------------------

extern "C" int pgh_counts_range_dev(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin, uint32_t v_end,
                                    void *d_out, void *stream, char *errbuf) {
	int rc = CheckRange(ds, v_begin, v_end, errbuf);
	if (rc == PGH_OK) {
		rc = CheckSubset(ds, subset, errbuf);
	}
	if (rc != PGH_OK) {
		return rc;
	}
	PGH_HIP(pgh::LaunchCounts(ds->View(), v_begin - ds->v_begin, nullptr, v_end - v_begin,
	                          subset ? subset->d_mask2 : nullptr, subset ? subset->n_out : ds->sample_ct,
	                          static_cast<uint32_t *>(d_out), static_cast<hipStream_t>(stream)),
	        "counts kernel");
	return PGH_OK;
}

extern "C" int pgh_counts_range(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin, uint32_t v_end,
                                uint32_t (*out)[4], char *errbuf) {
	int rc = CheckRange(ds, v_begin, v_end, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	const size_t n = v_end - v_begin;
	if (n == 0) {
		return PGH_OK;
	}
	DevBuf buf;
	PGH_HIP(buf.Alloc(n * 16), "hipMalloc(counts)");
	rc = pgh_counts_range_dev(ds, subset, v_begin, v_end, buf.p, hipStreamPerThread, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	PGH_HIP(hipMemcpyAsync(out, buf.p, n * 16, hipMemcpyDeviceToHost, hipStreamPerThread), "counts copy");
	PGH_HIP(hipStreamSynchronize(hipStreamPerThread), "counts sync");
	return PGH_OK;
}

extern "C" int pgh_freq_from_counts_dev(const void *d_counts, uint32_t n, void *d_alt_freq, void *d_obs_ct,
                                        void *stream, char *errbuf) {
	if (n && (!d_counts || !d_alt_freq || !d_obs_ct)) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	PGH_HIP(pgh::LaunchFreqFromCounts(static_cast<const uint32_t *>(d_counts), n, static_cast<double *>(d_alt_freq),
	                                  static_cast<int32_t *>(d_obs_ct), static_cast<hipStream_t>(stream)),
	        "freq kernel");
	return PGH_OK;
}

extern "C" int pgh_missing_per_sample_dev(const pgh_dataset *ds, uint32_t v_begin, uint32_t v_end, void *d_out,
                                          void *stream, char *errbuf) {
	int rc = CheckRange(ds, v_begin, v_end, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	hipStream_t st = static_cast<hipStream_t>(stream);
	// per-slice partial rows: a few MB, stream-ordered so the call stays enqueue-only
	const size_t scratch_bytes = pgh::MissingPerSampleScratchBytes(ds->record_bytes, v_end - v_begin);
	void *scratch = nullptr;
	PGH_HIP(hipMallocAsync(&scratch, scratch_bytes ? scratch_bytes : 16, st), "missing scratch");
	hipError_t e = pgh::LaunchMissingPerSample(ds->View(), v_begin - ds->v_begin, nullptr, v_end - v_begin, nullptr,
	                                           static_cast<uint32_t *>(scratch), static_cast<uint32_t *>(d_out), st);
	(void)hipFreeAsync(scratch, st);
	PGH_HIP(e, "missing-per-sample kernel");
	return PGH_OK;
}

extern "C" int pgh_fused_tally_dev(const pgh_dataset *ds, uint32_t v_begin, uint32_t v_end, void *d_counts,
                                   void *d_missing, void *stream, char *errbuf) {
	int rc = CheckRange(ds, v_begin, v_end, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	if (!d_counts || !d_missing) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	hipStream_t st = static_cast<hipStream_t>(stream);
	const size_t scratch_bytes = pgh::MissingPerSampleScratchBytes(ds->record_bytes, v_end - v_begin);
	void *scratch = nullptr;
	PGH_HIP(hipMallocAsync(&scratch, scratch_bytes ? scratch_bytes : 16, st), "fused scratch");
	hipError_t e = pgh::LaunchFusedTally(ds->View(), v_begin - ds->v_begin, v_end - v_begin,
	                                     static_cast<uint32_t *>(scratch), static_cast<uint32_t *>(d_counts),
	                                     static_cast<uint32_t *>(d_missing), st);
	(void)hipFreeAsync(scratch, st);
	PGH_HIP(e, "fused tally kernel");
	return PGH_OK;
}

extern "C" int pgh_missing_per_sample(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin,
                                      uint32_t v_end, uint32_t *out, char *errbuf) {
	int rc = CheckRange(ds, v_begin, v_end, errbuf);
	if (rc == PGH_OK) {
		rc = CheckSubset(ds, subset, errbuf);
	}
	if (rc != PGH_OK) {
		return rc;
	}
	const uint32_t N = ds->sample_ct;
	const uint32_t padded = (N + 63) / 64 * 64;
	DevBuf buf;
	PGH_HIP(buf.Alloc(sizeof(uint32_t) * padded), "hipMalloc(missing)");
	rc = pgh_missing_per_sample_dev(ds, v_begin, v_end, buf.p, hipStreamPerThread, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	std::vector<uint32_t> raw(N);
	PGH_HIP(hipMemcpyAsync(raw.data(), buf.p, sizeof(uint32_t) * N, hipMemcpyDeviceToHost, hipStreamPerThread),
	        "missing copy");
	PGH_HIP(hipStreamSynchronize(hipStreamPerThread), "missing sync");
	Compact<uint32_t>(subset, raw.data(), 1, out, N);
	return PGH_OK;
}

extern "C" int pgh_sample_counts(const pgh_dataset *ds, const pgh_subset *subset, uint32_t variant_begin,
                                 uint32_t n_var, const uint32_t *vidx, uint32_t (*counts)[4], char *errbuf) {
	if (!ds || !counts) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	int rc = CheckSubset(ds, subset, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	std::vector<uint32_t> local;
	if (vidx) {
		local.resize(n_var);
		for (uint32_t i = 0; i < n_var; i++) {
			if (vidx[i] < ds->v_begin || vidx[i] >= ds->v_end) {
				SetErr(errbuf, "variant index outside the resident range");
				return PGH_ERR_ARG;
			}
			local[i] = vidx[i] - ds->v_begin;
		}
	} else {
		rc = CheckRange(ds, variant_begin, variant_begin + n_var, errbuf);
		if (rc != PGH_OK) {
			return rc;
		}
	}
	const uint32_t N = ds->sample_ct;
	const uint32_t n_out = subset ? subset->n_out : N;
	const uint32_t padded = (N + 63) / 64 * 64;
	hipStream_t st = hipStreamPerThread;
	DevBuf d_cls, d_list, d_scratch;
	PGH_HIP(d_cls.Alloc(sizeof(uint32_t) * 3ull * padded), "hipMalloc(sample counts)");
	const size_t scratch_bytes = pgh::MissingPerSampleScratchBytes(ds->record_bytes, n_var);
	PGH_HIP(d_scratch.Alloc(scratch_bytes ? scratch_bytes : 16), "hipMalloc(sample counts)");
	if (vidx && n_var) {
		PGH_HIP(d_list.Alloc(sizeof(uint32_t) * n_var), "hipMalloc(sample counts)");
		PGH_HIP(hipMemcpyAsync(d_list.p, local.data(), sizeof(uint32_t) * n_var, hipMemcpyHostToDevice, st),
		        "sample counts upload");
	}
	// one column-tally pass per class (het, hom-alt, missing); hom-ref is what is left
	for (int cls = 1; cls <= 3; cls++) {
		PGH_HIP(pgh::LaunchClassPerSample(ds->View(), cls, vidx ? 0 : variant_begin - ds->v_begin,
		                                  vidx ? d_list.As<uint32_t>() : nullptr, n_var, nullptr,
		                                  d_scratch.As<uint32_t>(), d_cls.As<uint32_t>() + (cls - 1) * padded, st),
		        "sample counts kernel");
	}
	std::vector<uint32_t> raw(3ull * padded);
	PGH_HIP(hipMemcpyAsync(raw.data(), d_cls.p, sizeof(uint32_t) * raw.size(), hipMemcpyDeviceToHost, st),
	        "sample counts copy");
	PGH_HIP(hipStreamSynchronize(st), "sample counts sync");
	for (uint32_t k = 0; k < n_out; k++) {
		const uint32_t s = subset ? subset->sel[k] : k;
		const uint32_t het = raw[s], alt = raw[padded + s], miss = raw[2ull * padded + s];
		counts[k][0] = n_var - het - alt - miss;
		counts[k][1] = het;
		counts[k][2] = alt;
		counts[k][3] = miss;
	}
	return PGH_OK;
}

extern "C" int pgh_unpack_range_dev(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin, uint32_t v_end,
                                    void *d_out, size_t out_pitch, void *d_validity, int missing_code, void *stream,
                                    char *errbuf) {
	int rc = CheckRange(ds, v_begin, v_end, errbuf);
	if (rc == PGH_OK) {
		rc = CheckSubset(ds, subset, errbuf);
	}
	if (rc != PGH_OK) {
		return rc;
	}
	const uint32_t n_out = subset ? subset->n_out : ds->sample_ct;
	if (d_out && (out_pitch % 16 != 0 || out_pitch < (static_cast<size_t>(n_out) + 15) / 16 * 16)) {
		SetErr(errbuf, "out_pitch must be a multiple of 16 covering the row");
		return PGH_ERR_ARG;
	}
	hipStream_t st = static_cast<hipStream_t>(stream);
	if (subset) {
		PGH_HIP(pgh::LaunchUnpackSubset(ds->View(), v_begin - ds->v_begin, v_end - v_begin, subset->d_sel, n_out,
		                                static_cast<int8_t *>(d_out), out_pitch, static_cast<uint64_t *>(d_validity),
		                                static_cast<int8_t>(missing_code), st),
		        "unpack kernel");
	} else {
		PGH_HIP(pgh::LaunchUnpack(ds->View(), v_begin - ds->v_begin, v_end - v_begin, static_cast<int8_t *>(d_out),
		                          out_pitch, static_cast<uint64_t *>(d_validity), static_cast<int8_t>(missing_code),
		                          st),
		        "unpack kernel");
	}
	return PGH_OK;
}

extern "C" int pgh_unpack_range(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin, uint32_t v_end,
                                int8_t *out, uint64_t *validity, int missing_code, char *errbuf) {
	int rc = CheckRange(ds, v_begin, v_end, errbuf);
	if (rc == PGH_OK) {
		rc = CheckSubset(ds, subset, errbuf);
	}
	if (rc != PGH_OK) {
		return rc;
	}
	const uint32_t n_out = subset ? subset->n_out : ds->sample_ct;
	const size_t rows = v_end - v_begin;
	if (rows == 0 || n_out == 0) {
		return PGH_OK;
	}
	const size_t out_pitch = (static_cast<size_t>(n_out) + 15) / 16 * 16;
	const size_t val_words = (n_out + 63) / 64;
	// chunk the range so the device staging stays bounded (output is 4x the input)
	const size_t max_rows = std::max<size_t>(1, (512ull << 20) / out_pitch);
	DevBuf d_out, d_val;
	const size_t chunk_rows = std::min(rows, max_rows);
	if (out) {
		PGH_HIP(d_out.Alloc(chunk_rows * out_pitch), "hipMalloc(unpack)");
	}
	if (validity) {
		PGH_HIP(d_val.Alloc(chunk_rows * val_words * 8), "hipMalloc(validity)");
	}
	for (size_t r0 = 0; r0 < rows; r0 += chunk_rows) {
		const size_t r1 = std::min(rows, r0 + chunk_rows);
		rc = pgh_unpack_range_dev(ds, subset, v_begin + static_cast<uint32_t>(r0), v_begin + static_cast<uint32_t>(r1),
		                          d_out.p, out_pitch, d_val.p, missing_code, hipStreamPerThread, errbuf);
		if (rc != PGH_OK) {
			return rc;
		}
		if (out) {
			PGH_HIP(hipMemcpy2DAsync(out + r0 * n_out, n_out, d_out.p, out_pitch, n_out, r1 - r0,
			                         hipMemcpyDeviceToHost, hipStreamPerThread),
			        "unpack copy");
		}
		if (validity) {
			PGH_HIP(hipMemcpyAsync(validity + r0 * val_words, d_val.p, (r1 - r0) * val_words * 8,
			                       hipMemcpyDeviceToHost, hipStreamPerThread),
			        "validity copy");
		}
		PGH_HIP(hipStreamSynchronize(hipStreamPerThread), "unpack sync");
	}
	return PGH_OK;
}

// ---------------------------------------------------------------------------
// plink_score
// ---------------------------------------------------------------------------

struct pgh_score_plan {
	const pgh_dataset *ds = nullptr;
	uint32_t n_scored = 0, n_cols = 0;
	int mode = 0;
	void *d_vlist = nullptr, *d_weights = nullptr, *d_flip = nullptr, *d_counts = nullptr, *d_ts = nullptr,
	     *d_td = nullptr, *d_ac = nullptr;
};

extern "C" void pgh_score_plan_destroy(pgh_score_plan *plan) {
	if (!plan) {
		return;
	}
	for (void *p : {plan->d_vlist, plan->d_weights, plan->d_flip, plan->d_counts, plan->d_ts, plan->d_td, plan->d_ac}) {
		if (p) {
			(void)hipFree(p);
		}
	}
	delete plan;
}

extern "C" int pgh_score_plan_create(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_scored,
                                     const uint32_t *vidx, const double *weights, const uint8_t *flip, uint32_t n_cols,
                                     int mode, pgh_score_plan **out, char *errbuf) {
	if (!ds || !out || (n_scored && (!vidx || !weights))) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	*out = nullptr;
	if (mode < 0 || mode > 2) {
		SetErr(errbuf, "unknown score mode");
		return PGH_ERR_ARG;
	}
	if (n_cols == 0 || n_cols > 4096) {
		SetErr(errbuf, "n_cols must be between 1 and 4096");
		return PGH_ERR_ARG;
	}
	int rc = CheckSubset(ds, subset, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	std::vector<uint32_t> local(n_scored);
	for (uint32_t i = 0; i < n_scored; i++) {
		if (vidx[i] < ds->v_begin || vidx[i] >= ds->v_end) {
			SetErr(errbuf, "scored variant index outside the resident range");
			return PGH_ERR_ARG;
		}
		local[i] = vidx[i] - ds->v_begin;
	}
	std::unique_ptr<pgh_score_plan, void (*)(pgh_score_plan *)> plan(new pgh_score_plan(), pgh_score_plan_destroy);
	plan->ds = ds;
	plan->n_scored = n_scored;
	plan->n_cols = n_cols;
	plan->mode = mode;
	if (n_scored) {
		const uint32_t N = ds->sample_ct;
		hipStream_t st = hipStreamPerThread;
		PGH_HIP(hipMalloc(&plan->d_vlist, sizeof(uint32_t) * n_scored), "hipMalloc(score)");
		PGH_HIP(hipMalloc(&plan->d_weights, sizeof(double) * n_scored * n_cols), "hipMalloc(score)");
		PGH_HIP(hipMalloc(&plan->d_counts, 16ull * n_scored), "hipMalloc(score)");
		PGH_HIP(hipMalloc(&plan->d_ts, 32ull * n_scored), "hipMalloc(score)");
		PGH_HIP(hipMalloc(&plan->d_td, 32ull * n_scored), "hipMalloc(score)");
		PGH_HIP(hipMalloc(&plan->d_ac, 4ull * n_scored), "hipMalloc(score)");
		PGH_HIP(hipMemcpyAsync(plan->d_vlist, local.data(), sizeof(uint32_t) * n_scored, hipMemcpyHostToDevice, st),
		        "score upload");
		PGH_HIP(hipMemcpyAsync(plan->d_weights, weights, sizeof(double) * n_scored * n_cols, hipMemcpyHostToDevice, st),
		        "score upload");
		if (flip) {
			PGH_HIP(hipMalloc(&plan->d_flip, n_scored), "hipMalloc(score)");
			PGH_HIP(hipMemcpyAsync(plan->d_flip, flip, n_scored, hipMemcpyHostToDevice, st), "score upload");
		}
		// per-variant statistics and contribution tables depend on the data only: once per plan
		PGH_HIP(pgh::LaunchCounts(ds->View(), 0, static_cast<uint32_t *>(plan->d_vlist), n_scored,
		                          subset ? subset->d_mask2 : nullptr, subset ? subset->n_out : N,
		                          static_cast<uint32_t *>(plan->d_counts), st),
		        "score counts kernel");
		PGH_HIP(pgh::LaunchScoreTables(static_cast<uint32_t *>(plan->d_counts), static_cast<uint8_t *>(plan->d_flip),
		                               n_scored, mode, static_cast<double *>(plan->d_ts),
		                               static_cast<double *>(plan->d_td), static_cast<uint32_t *>(plan->d_ac), st),
		        "score table kernel");
		PGH_HIP(hipStreamSynchronize(st), "score plan sync"); // host staging vectors die with this frame
	}
	*out = plan.release();
	return PGH_OK;
}

extern "C" int pgh_score_run_dev(const pgh_score_plan *plan, void *d_score_sum, void *d_dosage_sum, void *d_allele_ct,
                                 void *stream, char *errbuf) {
	if (!plan || !d_score_sum || !d_dosage_sum || !d_allele_ct) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	const pgh_dataset *ds = plan->ds;
	const uint32_t N = ds->sample_ct;
	hipStream_t st = static_cast<hipStream_t>(stream);
	PGH_HIP(hipMemsetAsync(d_score_sum, 0, sizeof(double) * N * plan->n_cols, st), "score memset");
	PGH_HIP(hipMemsetAsync(d_dosage_sum, 0, sizeof(double) * N, st), "score memset");
	if (plan->n_scored == 0) {
		PGH_HIP(hipMemsetAsync(d_allele_ct, 0, sizeof(uint32_t) * N, st), "score memset");
		return PGH_OK;
	}
	PGH_HIP(pgh::LaunchScoreAccumulate(ds->View(), static_cast<uint32_t *>(plan->d_vlist), plan->n_scored,
	                                   static_cast<double *>(plan->d_weights), plan->n_cols,
	                                   static_cast<double *>(plan->d_ts), static_cast<double *>(plan->d_td),
	                                   static_cast<uint32_t *>(plan->d_ac), plan->mode != PGH_SCORE_CENTER,
	                                   static_cast<double *>(d_score_sum), static_cast<double *>(d_dosage_sum),
	                                   static_cast<uint32_t *>(d_allele_ct), st),
	        "score accumulate kernel");
	// ALLELE_CT is integer bookkeeping: 2 per scored, non-skipped variant, minus 2 per such
	// variant at which the sample is missing unless missing calls are mean-imputed
	// (src/plink_score.cpp:632-651).
	if (plan->mode == PGH_SCORE_MEAN_IMPUTE) {
		PGH_HIP(pgh::LaunchAlleleCt(static_cast<uint32_t *>(plan->d_ac), plan->n_scored, nullptr, N,
		                            static_cast<uint32_t *>(d_allele_ct), st),
		        "allele count kernel");
	} else {
		const size_t scratch_bytes = pgh::MissingPerSampleScratchBytes(ds->record_bytes, plan->n_scored);
		void *scratch = nullptr, *miss = nullptr;
		PGH_HIP(hipMallocAsync(&scratch, scratch_bytes ? scratch_bytes : 16, st), "score scratch");
		PGH_HIP(hipMallocAsync(&miss, sizeof(uint32_t) * ((N + 63) / 64 * 64), st), "score scratch");
		hipError_t e = pgh::LaunchMissingPerSample(ds->View(), 0, static_cast<uint32_t *>(plan->d_vlist),
		                                           plan->n_scored, static_cast<uint32_t *>(plan->d_ac),
		                                           static_cast<uint32_t *>(scratch), static_cast<uint32_t *>(miss), st);
		if (e == hipSuccess) {
			e = pgh::LaunchAlleleCt(static_cast<uint32_t *>(plan->d_ac), plan->n_scored, static_cast<uint32_t *>(miss),
			                        N, static_cast<uint32_t *>(d_allele_ct), st);
		}
		(void)hipFreeAsync(scratch, st);
		(void)hipFreeAsync(miss, st);
		PGH_HIP(e, "allele count kernels");
	}
	return PGH_OK;
}

extern "C" int pgh_score_dev(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_scored, const uint32_t *vidx,
                             const double *weights, const uint8_t *flip, uint32_t n_cols, int mode, void *d_score_sum,
                             void *d_dosage_sum, void *d_allele_ct, void *stream, char *errbuf) {
	pgh_score_plan *plan = nullptr;
	int rc = pgh_score_plan_create(ds, subset, n_scored, vidx, weights, flip, n_cols, mode, &plan, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	rc = pgh_score_run_dev(plan, d_score_sum, d_dosage_sum, d_allele_ct, stream, errbuf);
	if (rc == PGH_OK) {
		hipError_t e = hipStreamSynchronize(static_cast<hipStream_t>(stream)); // the plan's buffers are freed next
		if (e != hipSuccess) {
			rc = DeviceFail(errbuf, "score sync", e);
		}
	}
	pgh_score_plan_destroy(plan);
	return rc;
}

extern "C" int pgh_score(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_scored, const uint32_t *vidx,
                         const double *weights, const uint8_t *flip, uint32_t n_cols, int mode, double *score_sum,
                         double *dosage_sum, uint32_t *allele_ct, char *errbuf) {
	if (!ds) {
		SetErr(errbuf, "null dataset");
		return PGH_ERR_ARG;
	}
	const uint32_t N = ds->sample_ct;
	DevBuf d_score, d_dos, d_ac;
	PGH_HIP(d_score.Alloc(sizeof(double) * N * std::max<uint32_t>(1, n_cols)), "hipMalloc(score out)");
	PGH_HIP(d_dos.Alloc(sizeof(double) * N), "hipMalloc(score out)");
	PGH_HIP(d_ac.Alloc(sizeof(uint32_t) * N), "hipMalloc(score out)");
	int rc = pgh_score_dev(ds, subset, n_scored, vidx, weights, flip, n_cols, mode, d_score.p, d_dos.p, d_ac.p,
	                       hipStreamPerThread, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	std::vector<double> h_score(static_cast<size_t>(N) * n_cols), h_dos(N);
	std::vector<uint32_t> h_ac(N);
	PGH_HIP(hipMemcpy(h_score.data(), d_score.p, sizeof(double) * N * n_cols, hipMemcpyDeviceToHost), "score copy");
	PGH_HIP(hipMemcpy(h_dos.data(), d_dos.p, sizeof(double) * N, hipMemcpyDeviceToHost), "score copy");
	PGH_HIP(hipMemcpy(h_ac.data(), d_ac.p, sizeof(uint32_t) * N, hipMemcpyDeviceToHost), "score copy");
	Compact<double>(subset, h_score.data(), n_cols, score_sum, N);
	Compact<double>(subset, h_dos.data(), 1, dosage_sum, N);
	Compact<uint32_t>(subset, h_ac.data(), 1, allele_ct, N);
	return PGH_OK;
}

// ---------------------------------------------------------------------------
// plink_pca
// ---------------------------------------------------------------------------

extern "C" int pgh_pca(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_var, const uint32_t *vidx,
                       const double *center, const double *inv_stdev, uint32_t n_pcs, const double *g1_init,
                       double *eigenvalues, double *eigenvectors, char *errbuf) {
	return pgh_pca_sharded(ds, subset, n_var, vidx, center, inv_stdev, n_var, n_pcs, g1_init, nullptr, nullptr,
	                       eigenvalues, eigenvectors, errbuf);
}

extern "C" int pgh_pca_sharded(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_var, const uint32_t *vidx,
                               const double *center, const double *inv_stdev, uint64_t n_var_total, uint32_t n_pcs,
                               const double *g1_init, pgh_allreduce_fn allreduce, void *allreduce_ctx,
                               double *eigenvalues, double *eigenvectors, char *errbuf) {
	if (!ds || !g1_init || !eigenvalues || !eigenvectors || n_pcs == 0 ||
	    (n_var && (!vidx || !center || !inv_stdev))) {
		SetErr(errbuf, "null or empty argument");
		return PGH_ERR_ARG;
	}
	if (n_var_total < n_var || (!allreduce && n_var_total != n_var)) {
		SetErr(errbuf, "n_var_total must cover this shard's variants (and equal them without an all-reduce)");
		return PGH_ERR_ARG;
	}
	int rc = CheckSubset(ds, subset, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	const uint32_t N = ds->sample_ct;
	const uint32_t n_out = subset ? subset->n_out : N;
	const uint32_t M = n_var; // this shard's rows of X; every 1/M and the eigenvalue divisor use the total
	const double m_total = static_cast<double>(n_var_total);
	const uint32_t k2 = 2 * n_pcs;
	const uint32_t qq = (n_pcs + 1) * k2;
	if (n_var_total < qq || n_out < qq) {
		SetErr(errbuf, "too few variants or samples for the requested number of PCs");
		return PGH_ERR_ARG;
	}
	std::vector<uint32_t> local(M);
	for (uint32_t i = 0; i < M; i++) {
		if (vidx[i] < ds->v_begin || vidx[i] >= ds->v_end) {
			SetErr(errbuf, "effective variant index outside the resident range");
			return PGH_ERR_ARG;
		}
		local[i] = vidx[i] - ds->v_begin;
	}
	hipStream_t st = hipStreamPerThread;
	// Sum a device buffer over the variant shards (X is split by rows, so every X^T(...)
	// product and every Gram matrix of a tall factor is a sum of per-shard terms).
	auto all_sum = [&](double *buf, uint64_t count) -> int {
		if (!allreduce) {
			return PGH_OK;
		}
		if (allreduce(allreduce_ctx, buf, count, st) != 0) {
			SetErr(errbuf, "pca: the all-reduce callback failed");
			return PGH_ERR_DEVICE;
		}
		return PGH_OK;
	};
#define PGH_SUM(buf, count)                                                                                            \
	do {                                                                                                               \
		int rc_sum_ = all_sum((buf), (count));                                                                         \
		if (rc_sum_ != PGH_OK) {                                                                                       \
			return rc_sum_;                                                                                            \
		}                                                                                                              \
	} while (0)
	const size_t m_alloc = std::max<uint32_t>(M, 1);
	DevBuf d_vlist, d_center, d_inv, d_ts, d_g1, d_g2, d_qq, d_bb;
	PGH_HIP(d_vlist.Alloc(sizeof(uint32_t) * m_alloc), "hipMalloc(pca)");
	PGH_HIP(d_center.Alloc(sizeof(double) * m_alloc), "hipMalloc(pca)");
	PGH_HIP(d_inv.Alloc(sizeof(double) * m_alloc), "hipMalloc(pca)");
	PGH_HIP(d_ts.Alloc(32ull * m_alloc), "hipMalloc(pca)");
	PGH_HIP(d_g1.Alloc(sizeof(double) * N * k2), "hipMalloc(pca)");
	PGH_HIP(d_g2.Alloc(sizeof(double) * N * k2), "hipMalloc(pca)");
	PGH_HIP(d_qq.Alloc(sizeof(double) * m_alloc * qq), "hipMalloc(pca)");
	if (M) {
		PGH_HIP(hipMemcpy(d_vlist.p, local.data(), sizeof(uint32_t) * M, hipMemcpyHostToDevice), "pca upload");
		PGH_HIP(hipMemcpy(d_center.p, center, sizeof(double) * M, hipMemcpyHostToDevice), "pca upload");
		PGH_HIP(hipMemcpy(d_inv.p, inv_stdev, sizeof(double) * M, hipMemcpyHostToDevice), "pca upload");
	}
	{
		// start matrix in raw-sample rows (excluded samples stay zero)
		std::vector<double> g1_raw(static_cast<size_t>(N) * k2, 0.0);
		for (uint32_t k = 0; k < n_out; k++) {
			const uint32_t s = subset ? subset->sel[k] : k;
			std::memcpy(&g1_raw[static_cast<size_t>(s) * k2], g1_init + static_cast<size_t>(k) * k2, sizeof(double) * k2);
		}
		PGH_HIP(hipMemcpy(d_g1.p, g1_raw.data(), sizeof(double) * g1_raw.size(), hipMemcpyHostToDevice), "pca upload");
	}
	if (M) {
		PGH_HIP(pgh::LaunchNormTables(d_center.As<double>(), d_inv.As<double>(), M, d_ts.As<double>(), st), "pca tables");
	}
	const RowView view = ds->View();
	double *g1 = d_g1.As<double>();
	double *g2 = d_g2.As<double>();
	const uint8_t *mask2 = subset ? subset->d_mask2 : nullptr;
	for (uint32_t pass = 0; pass <= n_pcs; pass++) {
		double *y = d_qq.As<double>() + static_cast<size_t>(pass) * k2;
		// Step A: QQ[:, pass*2k : (pass+1)*2k] = X * G1   (rows of this shard only)
		if (M) {
			PGH_HIP(pgh::LaunchVariantReduce(view, d_vlist.As<uint32_t>(), M, d_ts.As<double>(), g1, k2, k2, y, qq, st),
			        "pca step A");
		}
		if (pass < n_pcs) {
			// Step B + merge: G1 = X^T Y / M, summed over shards
			PGH_HIP(hipMemsetAsync(g2, 0, sizeof(double) * N * k2, st), "pca memset");
			if (M) {
				PGH_HIP(pgh::LaunchTableAccumulate(view, d_vlist.As<uint32_t>(), M, y, qq, k2, d_ts.As<double>(),
				                                   nullptr, nullptr, false, g2, k2, nullptr, nullptr, st),
				        "pca step B");
			}
			PGH_SUM(g2, static_cast<uint64_t>(N) * k2);
			PGH_HIP(pgh::LaunchMaskRows(g2, N, k2, k2, mask2, st), "pca mask");
			PGH_HIP(pgh::LaunchScale(g2, static_cast<uint64_t>(N) * k2, 1.0 / m_total, st), "pca scale");
			std::swap(g1, g2);
		}
	}
	// Orthonormal basis of the Krylov block's column space, on the device.  The
	// reference takes the left singular vectors of QQ (src/plink_pca.cpp:683-697); the
	// only thing phase 3 uses of them is that they are an orthonormal basis of that
	// space (singular values of B = X^T U do not change under a rotation of U), so a
	// block Gram-Schmidt does the same job without an M x qq SVD on the host:
	// per block of 2k columns, project out the finished blocks twice (BCGS2), then
	// orthonormalise inside the block twice through its 2k x 2k Gram matrix.
	{
		DevBuf d_small, d_tmp;
		PGH_HIP(d_small.Alloc(sizeof(double) * static_cast<size_t>(qq) * k2), "hipMalloc(pca)");
		PGH_HIP(d_tmp.Alloc(sizeof(double) * m_alloc * k2), "hipMalloc(pca)");
		double *q = d_qq.As<double>();
		std::vector<double> g(static_cast<size_t>(k2) * k2), lam, vec, t(static_cast<size_t>(k2) * k2);
		for (uint32_t p = 0; p <= n_pcs; p++) {
			double *bp = q + static_cast<size_t>(p) * k2;
			const uint32_t prev = p * k2;
			for (int rep = 0; rep < 2 && prev > 0; rep++) {
				PGH_HIP(hipMemsetAsync(d_small.p, 0, sizeof(double) * prev * k2, st), "pca memset");
				if (M) {
					PGH_HIP(pgh::LaunchTallGram(q, qq, prev, bp, qq, k2, M, d_small.As<double>(), k2, st), "pca gram");
				}
				PGH_SUM(d_small.As<double>(), static_cast<uint64_t>(prev) * k2);
				if (M) {
					PGH_HIP(pgh::LaunchTallTimesSmall(q, qq, prev, d_small.As<double>(), k2, k2, -1.0, 1.0, bp, qq, bp, qq,
					                                  M, st),
					        "pca project");
				}
			}
			for (int rep = 0; rep < 2; rep++) {
				PGH_HIP(hipMemsetAsync(d_small.p, 0, sizeof(double) * k2 * k2, st), "pca memset");
				if (M) {
					PGH_HIP(pgh::LaunchTallGram(bp, qq, k2, bp, qq, k2, M, d_small.As<double>(), k2, st), "pca gram");
				}
				PGH_SUM(d_small.As<double>(), static_cast<uint64_t>(k2) * k2);
				PGH_HIP(hipMemcpyAsync(g.data(), d_small.p, sizeof(double) * k2 * k2, hipMemcpyDeviceToHost, st),
				        "pca download");
				PGH_HIP(hipStreamSynchronize(st), "pca sync");
				pgh::SymmetricEigen(g, k2, lam, vec);
				const double floor = lam[0] * 1e-13; // below this a direction is rounding noise
				for (uint32_t i = 0; i < k2; i++) {
					for (uint32_t j = 0; j < k2; j++) {
						t[static_cast<size_t>(i) * k2 + j] =
						    lam[j] > floor && lam[j] > 0.0 ? vec[static_cast<size_t>(i) * k2 + j] / std::sqrt(lam[j]) : 0.0;
					}
				}
				PGH_HIP(hipMemcpyAsync(d_small.p, t.data(), sizeof(double) * k2 * k2, hipMemcpyHostToDevice, st),
				        "pca upload");
				if (M) {
					PGH_HIP(pgh::LaunchTallTimesSmall(bp, qq, k2, d_small.As<double>(), k2, k2, 1.0, 0.0, nullptr, 0,
					                                  d_tmp.As<double>(), k2, M, st),
					        "pca orthonormalise");
					PGH_HIP(pgh::LaunchCopyCols(d_tmp.As<double>(), k2, bp, qq, k2, M, st), "pca copy");
				}
				PGH_HIP(hipStreamSynchronize(st), "pca sync"); // t is reused by the next repetition
			}
		}
	}
	// Phase 3: BB = X^T U   (src/plink_pca.cpp:664-676)
	PGH_HIP(d_bb.Alloc(sizeof(double) * static_cast<size_t>(N) * qq), "hipMalloc(pca)");
	PGH_HIP(hipMemsetAsync(d_bb.p, 0, sizeof(double) * static_cast<size_t>(N) * qq, st), "pca memset");
	if (M) {
		PGH_HIP(pgh::LaunchTableAccumulate(view, d_vlist.As<uint32_t>(), M, d_qq.As<double>(), qq, qq,
		                                   d_ts.As<double>(), nullptr, nullptr, false, d_bb.As<double>(), qq, nullptr,
		                                   nullptr, st),
		        "pca phase 3");
	}
	PGH_SUM(d_bb.As<double>(), static_cast<uint64_t>(N) * qq);
	PGH_HIP(pgh::LaunchMaskRows(d_bb.As<double>(), N, qq, qq, mask2, st), "pca mask");
	// Final SVD of BB (src/plink_pca.cpp:700-720) through its qq x qq Gram matrix:
	// BB^T BB = V S^2 V^T gives the eigenvalues S^2 / M directly and U_k = BB V_k S_k^-1.
	{
		DevBuf d_g, d_vk, d_uk;
		PGH_HIP(d_g.Alloc(sizeof(double) * static_cast<size_t>(qq) * qq), "hipMalloc(pca)");
		PGH_HIP(d_vk.Alloc(sizeof(double) * static_cast<size_t>(qq) * n_pcs), "hipMalloc(pca)");
		PGH_HIP(d_uk.Alloc(sizeof(double) * static_cast<size_t>(N) * n_pcs), "hipMalloc(pca)");
		PGH_HIP(hipMemsetAsync(d_g.p, 0, sizeof(double) * static_cast<size_t>(qq) * qq, st), "pca memset");
		PGH_HIP(pgh::LaunchTallGram(d_bb.As<double>(), qq, qq, d_bb.As<double>(), qq, qq, N, d_g.As<double>(), qq, st),
		        "pca gram");
		std::vector<double> g(static_cast<size_t>(qq) * qq), lam, vec;
		PGH_HIP(hipMemcpyAsync(g.data(), d_g.p, sizeof(double) * g.size(), hipMemcpyDeviceToHost, st), "pca download");
		PGH_HIP(hipStreamSynchronize(st), "pca sync");
		for (uint32_t i = 0; i < qq; i++) { // symmetrise away the atomics' rounding asymmetry
			for (uint32_t j = i + 1; j < qq; j++) {
				const double avg = 0.5 * (g[static_cast<size_t>(i) * qq + j] + g[static_cast<size_t>(j) * qq + i]);
				g[static_cast<size_t>(i) * qq + j] = g[static_cast<size_t>(j) * qq + i] = avg;
			}
		}
		pgh::SymmetricEigen(g, qq, lam, vec);
		std::vector<double> vk(static_cast<size_t>(qq) * n_pcs);
		for (uint32_t pc = 0; pc < n_pcs; pc++) {
			const double l = lam[pc] > 0.0 ? lam[pc] : 0.0;
			eigenvalues[pc] = l / m_total;
			const double inv_s = l > 0.0 ? 1.0 / std::sqrt(l) : 0.0;
			for (uint32_t i = 0; i < qq; i++) {
				vk[static_cast<size_t>(i) * n_pcs + pc] = vec[static_cast<size_t>(i) * qq + pc] * inv_s;
			}
		}
		PGH_HIP(hipMemcpyAsync(d_vk.p, vk.data(), sizeof(double) * vk.size(), hipMemcpyHostToDevice, st), "pca upload");
		PGH_HIP(pgh::LaunchTallTimesSmall(d_bb.As<double>(), qq, qq, d_vk.As<double>(), n_pcs, n_pcs, 1.0, 0.0, nullptr, 0,
		                                  d_uk.As<double>(), n_pcs, N, st),
		        "pca eigenvectors");
		std::vector<double> uk_raw(static_cast<size_t>(N) * n_pcs);
		PGH_HIP(hipMemcpyAsync(uk_raw.data(), d_uk.p, sizeof(double) * uk_raw.size(), hipMemcpyDeviceToHost, st),
		        "pca download");
		PGH_HIP(hipStreamSynchronize(st), "pca sync");
		Compact<double>(subset, uk_raw.data(), n_pcs, eigenvectors, N);
	}
	return PGH_OK;
#undef PGH_SUM
}

// ---------------------------------------------------------------------------
// plink_ld
// ---------------------------------------------------------------------------

extern "C" int pgh_ld_pairs(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_pairs, const uint32_t *vidx_a,
                            const uint32_t *vidx_b, uint32_t (*sums)[6], char *errbuf) {
	if (!ds || (n_pairs && (!vidx_a || !vidx_b || !sums))) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	int rc = CheckSubset(ds, subset, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	if (n_pairs == 0) {
		return PGH_OK;
	}
	// runs of pairs that share the anchor and step through consecutive partners become one
	// task of up to four partners (the windowed scan produces exactly such runs)
	std::vector<pgh::LdTask> tasks;
	tasks.reserve(n_pairs / 2 + 1);
	for (uint32_t p = 0; p < n_pairs; p++) {
		const uint32_t a = vidx_a[p], b = vidx_b[p];
		if (a < ds->v_begin || a >= ds->v_end || b < ds->v_begin || b >= ds->v_end) {
			SetErr(errbuf, "variant index outside the resident range");
			return PGH_ERR_ARG;
		}
		if (!tasks.empty()) {
			pgh::LdTask &last = tasks.back();
			if (last.n_b < 4 && last.a_row == a - ds->v_begin && last.b_row + last.n_b == b - ds->v_begin) {
				last.n_b++;
				continue;
			}
		}
		tasks.push_back(pgh::LdTask {a - ds->v_begin, b - ds->v_begin, 1u, p});
	}
	hipStream_t st = hipStreamPerThread;
	void *d_tasks = nullptr, *d_out = nullptr;
	PGH_HIP(hipMallocAsync(&d_tasks, sizeof(pgh::LdTask) * tasks.size(), st), "ld scratch");
	hipError_t e = hipMallocAsync(&d_out, 24ull * n_pairs, st);
	if (e == hipSuccess) {
		e = hipMemcpyAsync(d_tasks, tasks.data(), sizeof(pgh::LdTask) * tasks.size(), hipMemcpyHostToDevice, st);
	}
	if (e == hipSuccess) {
		e = pgh::LaunchLdPairs(ds->View(), static_cast<const pgh::LdTask *>(d_tasks),
		                       static_cast<uint32_t>(tasks.size()), subset ? subset->d_mask2 : nullptr,
		                       static_cast<uint32_t(*)[6]>(d_out), st);
	}
	if (e == hipSuccess) {
		e = hipMemcpyAsync(sums, d_out, 24ull * n_pairs, hipMemcpyDeviceToHost, st);
	}
	if (e == hipSuccess) {
		e = hipStreamSynchronize(st); // `tasks` and `sums` are host memory of this frame / the caller
	}
	(void)hipFreeAsync(d_tasks, st);
	if (d_out) {
		(void)hipFreeAsync(d_out, st);
	}
	PGH_HIP(e, "ld pair kernel");
	return PGH_OK;
}

// ---------------------------------------------------------------------------
// per-variant reader
// ---------------------------------------------------------------------------

extern "C" int pgh_reader_create(const pgh_dataset *ds, const pgh_subset *subset, pgh_reader **out, char *errbuf) {
	if (!ds || !out) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	*out = nullptr;
	int rc = CheckSubset(ds, subset, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	std::unique_ptr<pgh_reader> rd(new pgh_reader());
	rd->ds = ds;
	rd->subset = subset;
	hipError_t e = hipStreamCreateWithFlags(&rd->stream, hipStreamNonBlocking);
	if (e == hipSuccess) {
		e = hipMalloc(reinterpret_cast<void **>(&rd->d_counts), 16 * pgh_reader::kWindow);
	}
	if (e == hipSuccess) {
		e = hipHostMalloc(reinterpret_cast<void **>(&rd->h_counts), 16 * pgh_reader::kWindow, hipHostMallocDefault);
	}
	if (e == hipSuccess) {
		e = hipHostMalloc(reinterpret_cast<void **>(&rd->h_row), ds->pitch, hipHostMallocDefault);
	}
	if (e != hipSuccess) {
		pgh_reader_destroy(rd.release());
		return DeviceFail(errbuf, "reader setup", e);
	}
	if (ds->has_file && (ds->index.has_dosage || ds->index.has_phase)) {
		rd->file.reset(new pgh::RecordFile());
		std::string err;
		if (!rd->file->Open(ds->pgen_path, err)) {
			SetErr(errbuf, err);
			pgh_reader_destroy(rd.release());
			return PGH_ERR_OPEN;
		}
		rd->norm.reset(new pgh::Normalizer(ds->index, *rd->file));
	}
	*out = rd.release();
	return PGH_OK;
}

extern "C" void pgh_reader_destroy(pgh_reader *rd) {
	if (!rd) {
		return;
	}
	if (rd->stream) {
		(void)hipStreamSynchronize(rd->stream);
	}
	if (rd->d_counts) {
		(void)hipFree(rd->d_counts);
	}
	if (rd->h_counts) {
		(void)hipHostFree(rd->h_counts);
	}
	if (rd->h_row) {
		(void)hipHostFree(rd->h_row);
	}
	if (rd->stream) {
		(void)hipStreamDestroy(rd->stream);
	}
	delete rd;
}

extern "C" const char *pgh_reader_error(const pgh_reader *rd) {
	return rd ? rd->err.c_str() : "null reader";
}

namespace {

int ReaderFail(pgh_reader *rd, int code, const std::string &msg) {
	rd->err = msg;
	return code;
}

int ReaderCheck(pgh_reader *rd, uint32_t vidx) {
	if (!rd) {
		return PGH_ERR_ARG;
	}
	if (vidx < rd->ds->v_begin || vidx >= rd->ds->v_end) {
		return ReaderFail(rd, PGH_ERR_ARG, "variant index " + std::to_string(vidx) + " outside the resident range");
	}
	return PGH_OK;
}

// raw 2-bit row of one variant -> pinned host buffer
int FetchRow(pgh_reader *rd, uint32_t vidx) {
	const pgh_dataset *ds = rd->ds;
	hipError_t e = hipMemcpyAsync(rd->h_row, ds->d_rows + static_cast<uint64_t>(vidx - ds->v_begin) * ds->pitch,
	                              ds->record_bytes, hipMemcpyDeviceToHost, rd->stream);
	if (e == hipSuccess) {
		e = hipStreamSynchronize(rd->stream);
	}
	if (e != hipSuccess) {
		return ReaderFail(rd, PGH_ERR_DEVICE, std::string("row fetch: ") + hipGetErrorString(e));
	}
	return PGH_OK;
}

inline uint32_t RowCode(const uint8_t *row, uint32_t s) {
	return (row[s >> 2] >> (2 * (s & 3))) & 3u;
}

template <class Fn>
void ForEachIncluded(const pgh_reader *rd, Fn &&fn) {
	if (rd->subset) {
		for (uint32_t k = 0; k < rd->subset->n_out; k++) {
			fn(k, rd->subset->sel[k]);
		}
	} else {
		for (uint32_t s = 0; s < rd->ds->sample_ct; s++) {
			fn(s, s);
		}
	}
}

} // namespace

extern "C" int pgh_get_counts(pgh_reader *rd, uint32_t vidx, uint32_t out[4]) {
	int rc = ReaderCheck(rd, vidx);
	if (rc != PGH_OK) {
		return rc;
	}
	if (vidx < rd->win_begin || vidx >= rd->win_end) {
		const pgh_dataset *ds = rd->ds;
		const uint32_t stop = std::min<uint64_t>(ds->v_end, static_cast<uint64_t>(vidx) + pgh_reader::kWindow);
		char errbuf[PGH_ERRBUF_LEN];
		rc = pgh_counts_range_dev(ds, rd->subset, vidx, stop, rd->d_counts, rd->stream, errbuf);
		if (rc != PGH_OK) {
			return ReaderFail(rd, rc, errbuf);
		}
		hipError_t e = hipMemcpyAsync(rd->h_counts, rd->d_counts, 16ull * (stop - vidx), hipMemcpyDeviceToHost,
		                              rd->stream);
		if (e == hipSuccess) {
			e = hipStreamSynchronize(rd->stream);
		}
		if (e != hipSuccess) {
			rd->win_begin = rd->win_end = 0;
			return ReaderFail(rd, PGH_ERR_DEVICE, std::string("counts fetch: ") + hipGetErrorString(e));
		}
		rd->win_begin = vidx;
		rd->win_end = stop;
	}
	std::memcpy(out, rd->h_counts + 4 * static_cast<size_t>(vidx - rd->win_begin), 16);
	return PGH_OK;
}

extern "C" int pgh_get_2bit(pgh_reader *rd, uint32_t vidx, uint64_t *genovec) {
	int rc = ReaderCheck(rd, vidx);
	if (rc == PGH_OK) {
		rc = FetchRow(rd, vidx);
	}
	if (rc != PGH_OK) {
		return rc;
	}
	const uint32_t n_out = rd->subset ? rd->subset->n_out : rd->ds->sample_ct;
	std::memset(genovec, 0, sizeof(uint64_t) * ((n_out + 31) / 32));
	ForEachIncluded(rd, [&](uint32_t k, uint32_t s) {
		genovec[k >> 5] |= static_cast<uint64_t>(RowCode(rd->h_row, s)) << (2 * (k & 31));
	});
	return PGH_OK;
}

extern "C" int pgh_get_missingness(pgh_reader *rd, uint32_t vidx, uint64_t *bits) {
	int rc = ReaderCheck(rd, vidx);
	if (rc == PGH_OK) {
		rc = FetchRow(rd, vidx);
	}
	if (rc != PGH_OK) {
		return rc;
	}
	const uint32_t n_out = rd->subset ? rd->subset->n_out : rd->ds->sample_ct;
	std::memset(bits, 0, sizeof(uint64_t) * ((n_out + 63) / 64));
	ForEachIncluded(rd, [&](uint32_t k, uint32_t s) {
		if (RowCode(rd->h_row, s) == 3u) {
			bits[k >> 6] |= 1ull << (k & 63);
		}
	});
	return PGH_OK;
}

extern "C" int pgh_get_int8(pgh_reader *rd, uint32_t vidx, int8_t *out) {
	int rc = ReaderCheck(rd, vidx);
	if (rc == PGH_OK) {
		rc = FetchRow(rd, vidx);
	}
	if (rc != PGH_OK) {
		return rc;
	}
	ForEachIncluded(rd, [&](uint32_t k, uint32_t s) {
		const uint32_t c = RowCode(rd->h_row, s);
		out[k] = c == 3u ? static_cast<int8_t>(-9) : static_cast<int8_t>(c);
	});
	return PGH_OK;
}

extern "C" int pgh_get_phased(pgh_reader *rd, uint32_t vidx, uint64_t *genovec, uint64_t *phasepresent,
                              uint64_t *phaseinfo) {
	int rc = ReaderCheck(rd, vidx);
	if (rc != PGH_OK) {
		return rc;
	}
	const pgh_dataset *ds = rd->ds;
	const uint32_t n_out = rd->subset ? rd->subset->n_out : ds->sample_ct;
	std::memset(phasepresent, 0, sizeof(uint64_t) * ((n_out + 63) / 64));
	std::memset(phaseinfo, 0, sizeof(uint64_t) * ((n_out + 63) / 64));
	if (!(rd->norm && (ds->index.vrtype[vidx] & 0x10))) {
		return pgh_get_2bit(rd, vidx, genovec);
	}
	std::vector<uint8_t> row, pp, pi;
	std::string err;
	if (!rd->norm->DecodePhase(vidx, row, pp, pi, err)) {
		return ReaderFail(rd, PGH_ERR_FORMAT, err);
	}
	std::memset(genovec, 0, sizeof(uint64_t) * ((n_out + 31) / 32));
	ForEachIncluded(rd, [&](uint32_t k, uint32_t s) {
		genovec[k >> 5] |= static_cast<uint64_t>(RowCode(row.data(), s)) << (2 * (k & 31));
		if (pp[s]) {
			phasepresent[k >> 6] |= 1ull << (k & 63);
		}
		if (pi[s]) {
			phaseinfo[k >> 6] |= 1ull << (k & 63);
		}
	});
	return PGH_OK;
}

extern "C" int pgh_get_dosage_f64(pgh_reader *rd, uint32_t vidx, double *out) {
	int rc = ReaderCheck(rd, vidx);
	if (rc != PGH_OK) {
		return rc;
	}
	const pgh_dataset *ds = rd->ds;
	if (rd->norm && (ds->index.vrtype[vidx] & 0x60)) {
		// explicit dosage track: decoded on the host from the record
		std::vector<uint8_t> row;
		std::vector<uint16_t> dos;
		std::string err;
		if (!rd->norm->DecodeDosage(vidx, row, dos, err)) {
			return ReaderFail(rd, PGH_ERR_FORMAT, err);
		}
		ForEachIncluded(rd, [&](uint32_t k, uint32_t s) {
			if (dos[s] != 0xffff) {
				out[k] = static_cast<double>(dos[s]) / 16384.0;
			} else {
				const uint32_t c = RowCode(row.data(), s);
				out[k] = c == 3u ? -9.0 : static_cast<double>(c);
			}
		});
		return PGH_OK;
	}
	rc = FetchRow(rd, vidx);
	if (rc != PGH_OK) {
		return rc;
	}
	ForEachIncluded(rd, [&](uint32_t k, uint32_t s) {
		const uint32_t c = RowCode(rd->h_row, s);
		out[k] = c == 3u ? -9.0 : static_cast<double>(c);
	});
	return PGH_OK;
}

// ---------------------------------------------------------------------------
// HWE
// ---------------------------------------------------------------------------

extern "C" double pgh_hwe_lnp(int32_t obs_hets, int32_t obs_hom1, int32_t obs_hom2, uint32_t midp) {
	return pgh::HweLnP(obs_hets, obs_hom1, obs_hom2, midp);
}

extern "C" double pgh_hwe_xchr_lnp(int32_t female_hets, int32_t female_hom1, int32_t female_hom2, int32_t male1,
                                   int32_t male2, uint32_t midp) {
	return pgh::HweXchrLnP(female_hets, female_hom1, female_hom2, male1, male2, midp);
}

extern "C" int pgh_hwe_lnp_batch_dev(const void *d_counts, uint32_t n, uint32_t midp, void *d_ln_p, void *stream,
                                     char *errbuf) {
	if (n && (!d_counts || !d_ln_p)) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	PGH_HIP(pgh::LaunchHweBatch(static_cast<const uint32_t *>(d_counts), n, midp, static_cast<double *>(d_ln_p),
	                            static_cast<hipStream_t>(stream)),
	        "hwe kernel");
	return PGH_OK;
}

extern "C" int pgh_hwe_lnp_batch(const uint32_t (*counts)[4], uint32_t n, uint32_t midp, double *ln_p, char *errbuf) {
	if (n == 0) {
		return PGH_OK;
	}
	if (!counts || !ln_p) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	DevBuf d_counts, d_lnp;
	PGH_HIP(d_counts.Alloc(16ull * n), "hipMalloc(hwe)");
	PGH_HIP(d_lnp.Alloc(8ull * n), "hipMalloc(hwe)");
	PGH_HIP(hipMemcpyAsync(d_counts.p, counts, 16ull * n, hipMemcpyHostToDevice, hipStreamPerThread), "hwe upload");
	PGH_HIP(pgh::LaunchHweBatch(d_counts.As<uint32_t>(), n, midp, d_lnp.As<double>(), hipStreamPerThread),
	        "hwe kernel");
	PGH_HIP(hipMemcpyAsync(ln_p, d_lnp.p, 8ull * n, hipMemcpyDeviceToHost, hipStreamPerThread), "hwe copy");
	PGH_HIP(hipStreamSynchronize(hipStreamPerThread), "hwe sync");
	return PGH_OK;
}

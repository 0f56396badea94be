// api_internal.hpp -- handle types and small helpers shared by the api_*.cpp files (the C ABI
// of include/pgenhip.h).  Host code only; the kernels live in the *.hip files.
#pragma once

#include "../../include/pgenhip.h"

#include "hwe_core.hpp"
#include "decode.hpp"
#include "dosage.hpp"
#include "phase.hpp"
#include "kernels.hpp"
#include "score_i8.hpp"
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
	// explicit dosages of the resident range (dosage.hpp:DosageView); dos_rows == 0: hardcalls only
	uint32_t dos_rows = 0;
	uint64_t dos_values = 0; // explicit dosages held
	std::vector<int32_t> dos_row_of;
	std::vector<uint32_t> dos_row_count; // explicit dosages per dosage row (the score plan sorts rows by density)
	int32_t *d_dos_row_of = nullptr;
	uint64_t *d_dos_present = nullptr;
	uint32_t *d_dos_rank = nullptr;
	uint64_t *d_dos_val_off = nullptr;
	uint16_t *d_dos_values = nullptr;
	// Entry records of the sparse tracks (dosage.hpp), built by the first plink_score plan that meets such a track
	// and resident from then on: 4 bytes per explicit dosage.  rec_state: 0 not tried, 1 resident, -1 did not fit
	// (the plans then keep to the bit-walking kernel).
	mutable std::mutex dos_rec_mutex;
	mutable int dos_rec_state = 0;
	mutable uint32_t *d_dos_rec = nullptr;
	mutable uint64_t *d_dos_rec_off = nullptr;
	mutable uint64_t dos_rec_ct = 0;
	// phase tracks of the resident range (phase.hpp): two bit rows per phased variant; ph_rows == 0: none
	uint32_t ph_rows = 0;
	std::vector<int32_t> ph_row_of;
	uint64_t *d_ph_present = nullptr;
	uint64_t *d_ph_info = nullptr;
	// A shard GROUP (pgh_open_sharded / pgh_group_create): contiguous, ascending variant ranges of one file, one
	// resident dataset per entry, each on its own device (api_sharded.cpp).  The group handle itself holds no
	// rows (device == -1); [v_begin, v_end) is the union of its shards' ranges.
	std::vector<pgh_dataset *> shards;
	// the group's RCCL communicators, one per shard (api_sharded.cpp: created at the first collective when the
	// shards sit on distinct devices; an opaque pointer so that rccl.h stays out of this header)
	mutable void *group_comms = nullptr;
	mutable void *group_workers = nullptr; // the group's persistent per-shard worker threads (api_sharded.cpp)
	bool IsGroup() const {
		return !shards.empty();
	}

	RowView View() const {
		return RowView {d_rows, pitch, sample_ct, record_bytes};
	}
	pgh::DosageView Dosage() const {
		pgh::DosageView d;
		d.row_of = d_dos_row_of;
		d.present = d_dos_present;
		d.rank = d_dos_rank;
		d.val_off = d_dos_val_off;
		d.values = d_dos_values;
		d.rec = d_dos_rec;
		d.rec_off = d_dos_rec_off;
		d.words = (sample_ct + 63) / 64;
		return d;
	}
};

struct pgh_subset {
	const pgh_dataset *ds = nullptr;
	int device = -1; // where the staged copies live (kept here: the dataset may be closed before the subset)
	uint32_t n_out = 0;
	std::vector<uint64_t> include; // ceil(N/64) words
	std::vector<uint32_t> sel;     // raw index of each included sample, ascending
	uint8_t *d_mask2 = nullptr;    // one pitched row of 01 slots
	uint64_t *d_include = nullptr; // the include words, for the kernels that walk samples bit by bit
	uint32_t *d_sel = nullptr;
	std::vector<pgh_subset *> parts; // subset of a shard group: one staged copy per shard (device)
};

struct pgh_reader {
	const pgh_dataset *ds = nullptr;
	int device = -1; // of the stream and staging buffers (-1: a shard group's reader owns none)
	const pgh_subset *subset = nullptr;
	hipStream_t stream = nullptr;
	// counts window: one launch serves the next kWindow per-variant calls
	static constexpr uint32_t kWindow = 128; // the reference's claim batch (src/plink_freq.cpp:413)
	uint32_t win_begin = 0, win_end = 0;
	uint32_t *d_counts = nullptr;
	uint32_t *h_counts = nullptr; // pinned
	uint8_t *h_row = nullptr;     // pinned, pitch bytes
	double *d_dosage = nullptr;   // one dosage row, allocated by the first pgh_get_dosage_f64
	double *h_dosage = nullptr;   // pinned
	uint64_t *h_phase = nullptr;  // pinned: phasepresent + phaseinfo words of one variant
	// pgh_reader_unpack_start / _wait: two staging blocks on the device, an event each
	void *d_unpack[2] = {nullptr, nullptr};
	size_t unpack_bytes[2] = {0, 0};
	hipEvent_t unpack_done[2] = {nullptr, nullptr};
	bool unpack_pending[2] = {false, false};
	std::string err;
	std::vector<pgh_reader *> parts; // reader of a shard group: one per shard, created on first use
};

// Every entry point runs on the device that holds the dataset it was handed, whatever device the calling thread
// had current (one process may hold shards on several devices); the previous device is restored on return.
struct DeviceScope {
	int prev = -1;
	bool changed = false;
	explicit DeviceScope(int device) {
		if (device >= 0 && hipGetDevice(&prev) == hipSuccess && prev != device) {
			changed = hipSetDevice(device) == hipSuccess;
			if (!changed) {
				(void)hipGetLastError(); // do not leave the failure for an unrelated launch check to find
			}
		}
	}
	DeviceScope(const DeviceScope &) = delete;
	DeviceScope &operator=(const DeviceScope &) = delete;
	~DeviceScope() {
		if (changed) {
			(void)hipSetDevice(prev);
		}
	}
};
#define PGH_ENTER(ds_) DeviceScope pgh_scope_((ds_) ? (ds_)->device : -1)
// entry points that hand out or take raw device pointers work on one device's dataset only
#define PGH_ONE_DEVICE(ds_)                                                                                            \
	do {                                                                                                               \
		if ((ds_) && (ds_)->IsGroup()) {                                                                               \
			SetErr(errbuf, "this entry point takes one device's dataset: pass pgh_shard(group, k)");                   \
			return PGH_ERR_ARG;                                                                                        \
		}                                                                                                              \
	} while (0)

// shard-group forms of the host-buffer entry points (api_sharded.cpp)
namespace pgh_group {
int CountsRange(const pgh_dataset *g, const pgh_subset *ss, uint32_t v_begin, uint32_t v_end, uint32_t (*out)[4],
                char *errbuf);
int MissingPerSample(const pgh_dataset *g, const pgh_subset *ss, uint32_t v_begin, uint32_t v_end, uint32_t *out,
                     char *errbuf);
int SampleCounts(const pgh_dataset *g, const pgh_subset *ss, uint32_t variant_begin, uint32_t n_var,
                 const uint32_t *vidx, uint32_t (*counts)[4], char *errbuf);
int UnpackRange(const pgh_dataset *g, const pgh_subset *ss, uint32_t v_begin, uint32_t v_end, int8_t *out,
                uint64_t *validity, int missing_code, char *errbuf);
int DosageSums(const pgh_dataset *g, const pgh_subset *ss, uint32_t variant_begin, uint32_t n_variants,
               const uint32_t *vidx, uint64_t (*sums)[3], char *errbuf);
int DosageUnpack(const pgh_dataset *g, const pgh_subset *ss, uint32_t variant_begin, uint32_t n_variants,
                 const uint32_t *vidx, double *out, char *errbuf);
int UnpackSamples(const pgh_dataset *g, const pgh_subset *ss, uint32_t n_variants, const uint32_t *vidx, int8_t *out,
                  int missing_code, char *errbuf);
int DosageUnpackSamples(const pgh_dataset *g, const pgh_subset *ss, uint32_t n_variants, const uint32_t *vidx,
                        double *out, char *errbuf);
int Score(const pgh_dataset *g, const pgh_subset *ss, uint32_t n_scored, const uint32_t *vidx, const double *weights,
          const uint8_t *flip, uint32_t n_cols, int mode, const uint32_t (*counts)[4], double *score_sum,
          double *dosage_sum, uint32_t *allele_ct, char *errbuf);
int Pca(const pgh_dataset *g, const pgh_subset *ss, uint32_t n_var, const uint32_t *vidx, const double *center,
        const double *inv_stdev, uint32_t n_pcs, const double *g1_init, double *eigenvalues, double *eigenvectors,
        char *errbuf);
int LdPairs(const pgh_dataset *g, const pgh_subset *ss, uint32_t n_pairs, const uint32_t *vidx_a, const uint32_t *vidx_b,
            uint32_t (*sums)[6], char *errbuf);
int CopyRowsToHost(const pgh_dataset *g, uint32_t v_begin, uint32_t v_end, uint8_t *rows, size_t row_stride,
                   char *errbuf);
int SubsetCreate(const pgh_dataset *g, const uint64_t *sample_include, pgh_subset **out, char *errbuf);
int GetInfo(const pgh_dataset *g, pgh_info *out);
void Close(pgh_dataset *g);
//! The reader of the shard that holds vidx (created on first use), or nullptr with rd->err set.
pgh_reader *ReaderFor(pgh_reader *rd, uint32_t vidx);
} // namespace pgh_group

// pgh_score_dev with the scored variants' class tallies supplied (api_analysis.cpp; counts may be NULL)
int PghScoreDevCounts(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_scored, const uint32_t *vidx,
                      const double *weights, const uint8_t *flip, uint32_t n_cols, int mode, const uint32_t (*counts)[4],
                      void *d_score_sum, void *d_dosage_sum, void *d_allele_ct, void *stream, char *errbuf);

// The calling thread's own stream on the current device: what every host-output entry point enqueues on.
// A real stream handle created by the library (on first use per thread and device, destroyed when the thread ends)
// rather than the special hipStreamPerThread value: PghThreadScratch below keys its blocks by stream, and the
// handle can be handed to callbacks that need a concrete stream.  (hipStreamPerThread was the first suspect of the
// lost-scratch defect described below; a library-owned stream failed the same way -- the allocator was the cause.)
hipStream_t PghThreadStream();

// Device scratch for an enqueue-only entry point: at least `bytes`, valid for the work the caller enqueues on `st`
// right now.  One block per (calling thread, device, stream), kept and re-used (work on one stream is ordered, so the
// next call's kernels cannot start before this call's are done with it), grown with hipMalloc when a call needs more
// (after draining `st`), freed when the thread ends.
// NOT hipMallocAsync / hipFreeAsync: with scratch from the stream-ordered pool pgh_missing_per_sample returned sums
// 9 % short about once in thirty calls -- whole slices of the first kernel's output were gone when the second kernel
// read them -- and 60 of 60 calls were right with hipMalloc'ed scratch and nothing else changed
// (profiles/r02_async_pool_ab.txt).  Round 1's plink_ld task list that "arrived all zeros" sat in a block of the same
// pool.
hipError_t PghThreadScratch(size_t bytes, hipStream_t st, void **out);

// Call-scoped device blocks (DevBuf).  hipMalloc / hipFree of multi-gigabyte blocks is not cheap on this runtime and
// gets dearer with use: plink_pca's three 12.5 GB work matrices (100 k variants x 500 k samples) cost 40 ms of
// allocation in the first call of a process and 0.7 - 3.4 s in every second or third call after that (eight
// back-to-back calls: 400 ms each for the kernels, 1.1 - 3.8 s for the whole call; profiles/r03_pca_alloc_stalls.txt).
// So blocks of kBlockCacheMin bytes and more go back to a per-device free list instead of the driver and the next
// call of the same shape takes them from there (a block up to a quarter larger than asked for is accepted).
// The list holds at most PGH_BLOCK_CACHE_GB (default 64) per device, oldest out first; it is emptied when a
// dataset closes (pgh_close), when any allocation of the library fails (then retried once) and by
// pgh_trim_device_cache().  PghBlockFree waits for the device like hipFree does before the block can be handed on.
// PghDeviceFreeBytes: hipMemGetInfo's free bytes plus what the list holds on the current device -- what the
// "is there room for a tile-major copy" decisions read.
constexpr size_t kBlockCacheMin = 64ull << 20;
hipError_t PghBlockAlloc(void **out, size_t bytes);
void PghBlockFree(void *p);
void PghTrimBlockCache();
size_t PghDeviceFreeBytes();
// hipMalloc for blocks that live longer than a call (dataset rows, plans, readers): a failure empties the block
// cache and tries once more.
// (a failing hipMalloc of a hundred gigabytes takes seconds to say so: a large request that the driver's free memory
// cannot cover has the list emptied BEFORE it is made)
void PghMakeRoom(size_t bytes);
template <class T>
hipError_t PghMalloc(T **out, size_t bytes) {
	PghMakeRoom(bytes);
	hipError_t e = hipMalloc(reinterpret_cast<void **>(out), bytes);
	if (e != hipSuccess) {
		(void)hipGetLastError();
		PghTrimBlockCache();
		e = hipMalloc(reinterpret_cast<void **>(out), bytes);
	}
	return e;
}


namespace {

[[maybe_unused]] void SetErr(char *errbuf, const std::string &msg) {
	if (errbuf) {
		std::snprintf(errbuf, PGH_ERRBUF_LEN, "%s", msg.c_str());
	}
}

[[maybe_unused]] int DeviceFail(char *errbuf, const char *what, hipError_t e) {
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

[[maybe_unused]] uint64_t ChoosePitch(uint32_t record_bytes) {
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
			PghBlockFree(p);
		}
	}
	hipError_t Alloc(size_t bytes) {
		return PghBlockAlloc(&p, bytes ? bytes : 16);
	}
	template <class T>
	T *As() {
		return static_cast<T *>(p);
	}
};

// HIP does not promise that hipMemcpyAsync has finished with a PAGEABLE source when it returns (this runtime
// happens to stage it first, tools/pageable_async_probe.hip), so a frame-local container that feeds one must
// outlive the copy on EVERY exit path, the PGH_HIP early returns included.  Declare one of these right after
// the container(s): it is destroyed before them and drains the stream first.
struct HostSourceFence {
	hipStream_t st;
	explicit HostSourceFence(hipStream_t s) : st(s) {
	}
	HostSourceFence(const HostSourceFence &) = delete;
	HostSourceFence &operator=(const HostSourceFence &) = delete;
	~HostSourceFence() {
		(void)hipStreamSynchronize(st);
	}
};

[[maybe_unused]] int CheckRange(const pgh_dataset *ds, uint32_t v_begin, uint32_t v_end, char *errbuf) {
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

[[maybe_unused]] int CheckSubset(const pgh_dataset *ds, const pgh_subset *ss, char *errbuf) {
	if (ss && ss->ds != ds) {
		SetErr(errbuf, "sample subset belongs to a different dataset");
		return PGH_ERR_ARG;
	}
	return PGH_OK;
}

// compact `raw` (one value per raw sample) to the included samples
template <class T>
[[maybe_unused]] void Compact(const pgh_subset *ss, const T *raw, size_t stride, T *out, uint32_t n_raw) {
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

[[maybe_unused]] int AllocRows(pgh_dataset *ds, char *errbuf) {
	const uint64_t rows = ds->v_end - ds->v_begin;
	const uint64_t bytes = rows * ds->pitch;
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&ds->d_rows), bytes ? bytes : 16), "hipMalloc(genotype rows)");
	return PGH_OK;
}

} // namespace


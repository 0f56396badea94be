// api_sharded.cpp -- shard groups: ONE process holding contiguous variant ranges of a file on several devices.
//
// The reference parallelises inside one process (one DuckDB, T scan threads: src/plink_freq.cpp:434-443) and
// merges per-thread partials under a mutex (src/plink_score.cpp:657-664, src/plink_missing.cpp:614-619,
// src/plink_pca.cpp:940-954).  The same shape here, with devices in the place of threads: a group handle is a
// pgh_dataset whose `shards` are ordinary datasets on their own devices, and every host-buffer entry point of
// include/pgenhip.h accepts it --
//   per-variant outputs (counts, unpacked calls, dosages): each shard fills its slice of the caller's buffer,
//     no exchange;
//   per-sample outputs (missing tallies, sample counts, plink_score, plink_pca's G2 / BB): each shard reduces
//     its variants on its device, the partials travel device to device (hipMemcpyPeerAsync: xGMI between the
//     GPUs of a node) and a kernel adds them on the first shard's device (8e-2, 8e-3).
// Shards work concurrently, one host thread per shard for the duration of a call.
#include "api_internal.hpp"

#include <rccl/rccl.h> // types and enums only: the library is bound at run time (below)

#include <array>
#include <atomic>
#include <condition_variable>
#include <dlfcn.h>
#include <functional>
#include <set>

namespace {

using Shards = std::vector<pgh_dataset *>;

// ---- RCCL inside a shard group ---------------------------------------------------------------------------------
// The per-sample merges the reference runs under a mutex (src/plink_score.cpp:657-664: global += local;
// src/plink_pca.cpp:921-959: MergePass) are, between the devices of a group, RCCL collectives over xGMI: a reduce
// to the first shard for plink_score's partials, an all-reduce per pass for plink_pca -- enqueued on each shard's
// own stream, no host rendezvous and no stream drain in between.  One communicator per shard (ncclCommInitAll over
// the group's device list) is made at the first collective and kept until the group is closed.  librccl is bound
// with dlopen at that moment rather than linked: a process that already holds one (PyTorch-ROCm bundles its own
// next to its HIP runtime) keeps using that one, and a host without it keeps every other entry point.  A group
// whose shards share a device (the one-GPU test shape) has no communicator -- RCCL wants distinct devices -- and
// uses the device-to-device copies further down; PGH_GROUP_RCCL=0 forces that path everywhere (an A/B switch).
struct RcclApi {
	void *handle = nullptr;
	ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
	ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
	ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
	ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
	ncclResult_t (*GroupStart)() = nullptr;
	ncclResult_t (*GroupEnd)() = nullptr;
	const char *(*GetErrorString)(ncclResult_t) = nullptr;
	bool ok = false;
};

const RcclApi &Rccl() {
	static const RcclApi api = [] {
		RcclApi a;
		const char *off = std::getenv("PGH_GROUP_RCCL");
		if (off && *off == '0') {
			return a;
		}
		for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
			a.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
			if (a.handle) {
				break;
			}
		}
		if (!a.handle) {
			return a;
		}
#define PGH_RCCL_SYM(field, sym) a.field = reinterpret_cast<decltype(a.field)>(dlsym(a.handle, sym))
		PGH_RCCL_SYM(CommInitAll, "ncclCommInitAll");
		PGH_RCCL_SYM(CommDestroy, "ncclCommDestroy");
		PGH_RCCL_SYM(CommAbort, "ncclCommAbort");
		PGH_RCCL_SYM(AllReduce, "ncclAllReduce");
		PGH_RCCL_SYM(Reduce, "ncclReduce");
		PGH_RCCL_SYM(GroupStart, "ncclGroupStart");
		PGH_RCCL_SYM(GroupEnd, "ncclGroupEnd");
		PGH_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef PGH_RCCL_SYM
		a.ok = a.CommInitAll && a.CommDestroy && a.CommAbort && a.AllReduce && a.Reduce && a.GroupStart && a.GroupEnd;
		return a;
	}();
	return api;
}

struct GroupComms {
	std::mutex mu;
	bool tried = false;
	std::vector<ncclComm_t> comms; // empty: no RCCL for this group
};

GroupComms *CommsOf(const pgh_dataset *g) {
	static std::mutex create_mu;
	std::lock_guard<std::mutex> lock(create_mu);
	if (!g->group_comms) {
		g->group_comms = new GroupComms();
	}
	return static_cast<GroupComms *>(g->group_comms);
}

//! The group's communicators (made on first use), or null when the group has none.
const std::vector<ncclComm_t> *EnsureComms(const pgh_dataset *g) {
	GroupComms *gc = CommsOf(g);
	std::lock_guard<std::mutex> lock(gc->mu);
	if (!gc->tried) {
		gc->tried = true;
		const RcclApi &api = Rccl();
		std::vector<int> devices;
		std::set<int> distinct;
		for (const pgh_dataset *s : g->shards) {
			devices.push_back(s->device);
			distinct.insert(s->device);
		}
		if (api.ok && distinct.size() == devices.size()) {
			std::vector<ncclComm_t> comms(devices.size(), nullptr);
			if (api.CommInitAll(comms.data(), static_cast<int>(devices.size()), devices.data()) == ncclSuccess) {
				gc->comms = std::move(comms);
			}
			(void)hipGetLastError();
		}
	}
	return gc->comms.empty() ? nullptr : &gc->comms;
}

//! A rank failed while the others may be inside a collective: abort every communicator (their kernels end, the
//! waiting streams drain) and forget them; the next collective makes new ones.
void AbortComms(const pgh_dataset *g) {
	GroupComms *gc = CommsOf(g);
	std::lock_guard<std::mutex> lock(gc->mu);
	for (ncclComm_t c : gc->comms) {
		if (c) {
			(void)Rccl().CommAbort(c);
		}
	}
	gc->comms.clear();
	gc->tried = false;
}

void DestroyComms(pgh_dataset *g) {
	auto *gc = static_cast<GroupComms *>(g->group_comms);
	if (!gc) {
		return;
	}
	for (size_t k = 0; k < gc->comms.size(); k++) {
		if (gc->comms[k]) {
			DeviceScope scope(g->shards.size() > k ? g->shards[k]->device : -1);
			(void)Rccl().CommDestroy(gc->comms[k]);
		}
	}
	delete gc;
	g->group_comms = nullptr;
}

const pgh_subset *PartOf(const pgh_subset *ss, size_t k) {
	return ss ? ss->parts[k] : nullptr;
}

// One persistent worker thread per shard, kept in the group handle: a worker sets its shard's device once and then
// runs what ForShards hands it, so the per-thread stream, scratch block and pinned task buffers the entry points
// keep (api_internal.hpp: PghThreadStream / PghThreadScratch) live across calls.  (Round 2 started K fresh threads
// per call and tore their streams and scratch down at its end: milliseconds per call, paid by every window of a
// plink_ld scan over a group.)  Calls on one group are serialised through the pool -- the shards' devices are
// busy with one call's work anyway.
class ShardWorkers {
public:
	explicit ShardWorkers(const std::vector<int> &devices) : tasks_(devices.size()) {
		for (size_t k = 0; k < devices.size(); k++) {
			threads_.emplace_back([this, k, dev = devices[k]] { Loop(k, dev); });
		}
	}
	~ShardWorkers() {
		{
			std::lock_guard<std::mutex> lock(m_);
			stop_ = true;
		}
		cv_.notify_all();
		for (auto &t : threads_) {
			t.join();
		}
	}
	//! fn(k) on worker k for every k; returns when all are done
	void Run(const std::function<void(size_t)> &fn) {
		std::lock_guard<std::mutex> one_call(call_m_);
		std::unique_lock<std::mutex> lock(m_);
		fn_ = &fn;
		pending_ = tasks_.size();
		tasks_.assign(tasks_.size(), 1);
		cv_.notify_all();
		done_cv_.wait(lock, [&] { return pending_ == 0; });
		fn_ = nullptr;
	}

private:
	void Loop(size_t k, int device) {
		(void)hipSetDevice(device);
		std::unique_lock<std::mutex> lock(m_);
		for (;;) {
			cv_.wait(lock, [&] { return stop_ || tasks_[k]; });
			if (stop_) {
				return;
			}
			tasks_[k] = 0;
			const std::function<void(size_t)> *fn = fn_;
			lock.unlock();
			(*fn)(k);
			lock.lock();
			if (--pending_ == 0) {
				done_cv_.notify_all();
			}
		}
	}
	std::mutex m_, call_m_;
	std::condition_variable cv_, done_cv_;
	std::vector<char> tasks_;
	std::vector<std::thread> threads_;
	const std::function<void(size_t)> *fn_ = nullptr;
	size_t pending_ = 0;
	bool stop_ = false;
};

ShardWorkers *WorkersOf(const pgh_dataset *g) {
	static std::mutex create_mu;
	std::lock_guard<std::mutex> lock(create_mu);
	if (!g->group_workers) {
		std::vector<int> devices;
		for (const pgh_dataset *s : g->shards) {
			devices.push_back(s->device);
		}
		g->group_workers = new ShardWorkers(devices);
	}
	return static_cast<ShardWorkers *>(g->group_workers);
}

//! fn(k, errbuf_k) for every shard, each on its shard's worker thread and device; the first failure is reported.
int ForShards(const pgh_dataset *g, char *errbuf, const std::function<int(size_t, char *)> &fn) {
	const size_t K = g->shards.size();
	std::vector<int> rc(K, PGH_OK);
	std::vector<std::array<char, PGH_ERRBUF_LEN>> eb(K);
	for (size_t k = 0; k < K; k++) {
		eb[k][0] = 0;
	}
	WorkersOf(g)->Run([&](size_t k) { rc[k] = fn(k, eb[k].data()); });
	for (size_t k = 0; k < K; k++) {
		if (rc[k] != PGH_OK) {
			SetErr(errbuf, eb[k].data());
			return rc[k];
		}
	}
	return PGH_OK;
}

size_t ShardOf(const pgh_dataset *g, uint32_t v) {
	size_t lo = 0, hi = g->shards.size();
	while (hi - lo > 1) {
		const size_t mid = (lo + hi) / 2;
		if (v >= g->shards[mid]->v_begin) {
			lo = mid;
		} else {
			hi = mid;
		}
	}
	return lo;
}

//! A variant list cut by shard: positions (in the caller's list) and indices of every shard's entries.
struct ListCut {
	std::vector<std::vector<uint32_t>> pos, idx;
};

int CutList(const pgh_dataset *g, uint32_t n, const uint32_t *vidx, ListCut &cut, char *errbuf) {
	cut.pos.assign(g->shards.size(), {});
	cut.idx.assign(g->shards.size(), {});
	for (uint32_t i = 0; i < n; i++) {
		if (vidx[i] < g->v_begin || vidx[i] >= g->v_end) {
			SetErr(errbuf, "variant index outside the resident range");
			return PGH_ERR_ARG;
		}
		const size_t k = ShardOf(g, vidx[i]);
		cut.pos[k].push_back(i);
		cut.idx[k].push_back(vidx[i]);
	}
	return PGH_OK;
}

//! A reusable barrier for the K shard threads of one call.
class Rendezvous {
public:
	explicit Rendezvous(size_t n) : n_(n) {
	}
	//! false when the meeting was called off (Abort): a rank that will never arrive must not leave the others
	//! waiting for it
	bool Wait() {
		std::unique_lock<std::mutex> lock(m_);
		if (aborted_) {
			return false;
		}
		const uint64_t gen = gen_;
		if (++arrived_ == n_) {
			arrived_ = 0;
			gen_++;
			cv_.notify_all();
		} else {
			cv_.wait(lock, [&] { return gen_ != gen || aborted_; });
		}
		return !aborted_;
	}
	void Abort() {
		std::lock_guard<std::mutex> lock(m_);
		aborted_ = true;
		cv_.notify_all();
	}

private:
	std::mutex m_;
	std::condition_variable cv_;
	size_t n_, arrived_ = 0;
	uint64_t gen_ = 0;
	bool aborted_ = false;
};

//! Sum device buffers of K shards into the first one: peers send to staging blocks on the root device over the
//! device-to-device path, a kernel adds.  The calling thread may have any device current.
//! ncclReduce of the shards' buffers onto the first shard's, one grouped call from this thread (a buffer-less shard
//! sends zeros).  false: the group has no communicators (or the collective failed to start) -- use the copies.
template <class T>
bool ReduceToRootRccl(const pgh_dataset *g, const std::vector<T *> &bufs, uint64_t count, char *errbuf, int &rc_out) {
	const std::vector<ncclComm_t> *comms = EnsureComms(g);
	if (!comms) {
		return false;
	}
	const RcclApi &api = Rccl();
	const Shards &shards = g->shards;
	const size_t K = shards.size();
	std::vector<DevBuf> zeros(K);
	std::vector<hipStream_t> streams(K, nullptr);
	std::vector<const void *> send(K, nullptr);
	for (size_t k = 0; k < K; k++) {
		DeviceScope scope(shards[k]->device);
		streams[k] = PghThreadStream();
		send[k] = bufs[k];
		if (!bufs[k]) {
			if (zeros[k].Alloc(sizeof(T) * count) != hipSuccess ||
			    hipMemsetAsync(zeros[k].p, 0, sizeof(T) * count, streams[k]) != hipSuccess) {
				rc_out = DeviceFail(errbuf, "hipMalloc(shard partial)", hipGetLastError());
				return true;
			}
			send[k] = zeros[k].p;
		}
	}
	ncclResult_t r = api.GroupStart();
	for (size_t k = 0; k < K && r == ncclSuccess; k++) {
		DeviceScope scope(shards[k]->device);
		r = api.Reduce(send[k], k == 0 ? static_cast<void *>(bufs[0]) : nullptr, count,
		               sizeof(T) == 8 ? ncclDouble : ncclUint32, ncclSum, 0, (*comms)[k], streams[k]);
	}
	const ncclResult_t r_end = api.GroupEnd();
	if (r == ncclSuccess) {
		r = r_end;
	}
	hipError_t e = hipSuccess;
	for (size_t k = 0; k < K; k++) {
		DeviceScope scope(shards[k]->device);
		const hipError_t ek = hipStreamSynchronize(streams[k]);
		if (e == hipSuccess) {
			e = ek;
		}
	}
	if (r != ncclSuccess || e != hipSuccess) {
		AbortComms(g);
		SetErr(errbuf, std::string("RCCL reduce of the shard partials failed: ") +
		                   (r != ncclSuccess && api.GetErrorString ? api.GetErrorString(r) : hipGetErrorString(e)));
		rc_out = PGH_ERR_DEVICE;
		return true;
	}
	rc_out = PGH_OK;
	return true;
}

template <class T>
int SumToRoot(const pgh_dataset *g, const std::vector<T *> &bufs, uint64_t count, char *errbuf) {
	if (!bufs[0]) {
		SetErr(errbuf, "internal: the first shard holds no partial");
		return PGH_ERR_ARG;
	}
	int rc_rccl = PGH_OK;
	if (ReduceToRootRccl<T>(g, bufs, count, errbuf, rc_rccl)) {
		return rc_rccl;
	}
	const Shards &shards = g->shards;
	const int root = shards[0]->device;
	DeviceScope scope(root);
	const size_t bytes = sizeof(T) * count;
	std::vector<DevBuf> stage(shards.size());
	for (size_t k = 1; k < shards.size(); k++) {
		if (!bufs[k]) {
			continue;
		}
		PGH_HIP(stage[k].Alloc(bytes), "hipMalloc(shard partial)");
		PGH_HIP(hipMemcpyPeerAsync(stage[k].p, root, bufs[k], shards[k]->device, bytes, PghThreadStream()),
		        "device-to-device copy of a shard partial");
		if (sizeof(T) == 8) {
			PGH_HIP(pgh::LaunchAddF64(reinterpret_cast<double *>(bufs[0]), stage[k].template As<double>(), count,
			                          PghThreadStream()),
			        "partial sum kernel");
		} else {
			PGH_HIP(pgh::LaunchAddU32(reinterpret_cast<uint32_t *>(bufs[0]), stage[k].template As<uint32_t>(), count,
			                          PghThreadStream()),
			        "partial sum kernel");
		}
	}
	PGH_HIP(hipStreamSynchronize(PghThreadStream()), "partial sum sync");
	return PGH_OK;
}

int CheckGroupSubset(const pgh_dataset *g, const pgh_subset *ss, char *errbuf) {
	if (ss && (ss->ds != g || ss->parts.size() != g->shards.size())) {
		SetErr(errbuf, "sample subset belongs to a different dataset");
		return PGH_ERR_ARG;
	}
	return PGH_OK;
}

} // namespace

// ---------------------------------------------------------------------------
// lifecycle
// ---------------------------------------------------------------------------

extern "C" int pgh_group_create(pgh_dataset *const *shards, uint32_t n_shards, pgh_dataset **out, char *errbuf) {
	if (!shards || !out || n_shards == 0) {
		SetErr(errbuf, "null or empty argument");
		return PGH_ERR_ARG;
	}
	*out = nullptr;
	for (uint32_t k = 0; k < n_shards; k++) {
		const pgh_dataset *s = shards[k];
		if (!s || s->IsGroup()) {
			SetErr(errbuf, "a shard must be a plain dataset");
			return PGH_ERR_ARG;
		}
		if (s->sample_ct != shards[0]->sample_ct) {
			SetErr(errbuf, "shards differ in sample count");
			return PGH_ERR_ARG;
		}
		if (k && s->v_begin != shards[k - 1]->v_end) {
			SetErr(errbuf, "shards must hold contiguous, ascending variant ranges");
			return PGH_ERR_ARG;
		}
	}
	std::unique_ptr<pgh_dataset> g(new pgh_dataset());
	g->device = -1;
	g->has_file = shards[0]->has_file;
	g->pgen_path = shards[0]->pgen_path;
	g->raw_variant_ct = shards[0]->raw_variant_ct;
	for (uint32_t k = 1; k < n_shards; k++) {
		g->raw_variant_ct = std::max(g->raw_variant_ct, shards[k]->raw_variant_ct);
	}
	g->sample_ct = shards[0]->sample_ct;
	g->record_bytes = shards[0]->record_bytes;
	g->pitch = shards[0]->pitch;
	g->v_begin = shards[0]->v_begin;
	g->v_end = shards[n_shards - 1]->v_end;
	g->shards.assign(shards, shards + n_shards);
	// direct device-to-device copies for the partial sums (an already-enabled pair reports an error of no interest)
	for (uint32_t a = 0; a < n_shards; a++) {
		for (uint32_t b = 0; b < n_shards; b++) {
			const int da = shards[a]->device, db = shards[b]->device;
			int can = 0;
			if (da != db && hipDeviceCanAccessPeer(&can, da, db) == hipSuccess && can) {
				DeviceScope scope(da);
				(void)hipDeviceEnablePeerAccess(db, 0);
				(void)hipGetLastError();
			}
		}
	}
	*out = g.release();
	return PGH_OK;
}

extern "C" int pgh_open_sharded(const char *pgen_path, const char *pgi_path, uint32_t variant_begin,
                                uint32_t variant_end, const int *devices, uint32_t n_devices, pgh_dataset **out,
                                char *errbuf) {
	if (!pgen_path || !out || !devices || n_devices == 0) {
		SetErr(errbuf, "null or empty argument");
		return PGH_ERR_ARG;
	}
	*out = nullptr;
	pgh_info info;
	int rc = pgh_probe(pgen_path, pgi_path, &info, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	if (variant_end == UINT32_MAX) {
		variant_end = info.raw_variant_ct;
	}
	if (variant_begin > variant_end || variant_end > info.raw_variant_ct) {
		SetErr(errbuf, "variant range outside the file");
		return PGH_ERR_ARG;
	}
	const int have = pgh_device_count();
	for (uint32_t k = 0; k < n_devices; k++) {
		if (devices[k] < 0 || devices[k] >= have) {
			SetErr(errbuf, "device ordinal " + std::to_string(devices[k]) + " does not exist (" + std::to_string(have) +
			                   " visible)");
			return PGH_ERR_ARG;
		}
	}
	// near-equal contiguous ranges, one per device, ingested concurrently (each device has its own host link)
	const uint32_t total = variant_end - variant_begin;
	const uint32_t per = (total + n_devices - 1) / n_devices;
	std::vector<pgh_dataset *> shards(n_devices, nullptr);
	std::vector<int> rcs(n_devices, PGH_OK);
	std::vector<std::array<char, PGH_ERRBUF_LEN>> eb(n_devices);
	std::vector<std::thread> threads;
	for (uint32_t k = 0; k < n_devices; k++) {
		eb[k][0] = 0;
		threads.emplace_back([&, k] {
			DeviceScope scope(devices[k]);
			const uint32_t b = variant_begin + std::min(total, k * per), e = variant_begin + std::min(total, (k + 1) * per);
			rcs[k] = pgh_open(pgen_path, pgi_path, b, e, &shards[k], eb[k].data());
		});
	}
	for (auto &t : threads) {
		t.join();
	}
	for (uint32_t k = 0; k < n_devices && rc == PGH_OK; k++) {
		if (rcs[k] != PGH_OK) {
			SetErr(errbuf, eb[k].data());
			rc = rcs[k];
		}
	}
	if (rc == PGH_OK) {
		rc = pgh_group_create(shards.data(), n_devices, out, errbuf);
	}
	if (rc != PGH_OK) {
		for (pgh_dataset *s : shards) {
			pgh_close(s);
		}
	}
	return rc;
}

extern "C" uint32_t pgh_shard_count(const pgh_dataset *ds) {
	return ds ? static_cast<uint32_t>(ds->shards.size()) : 0u;
}

extern "C" const pgh_dataset *pgh_shard(const pgh_dataset *ds, uint32_t k) {
	return (ds && k < ds->shards.size()) ? ds->shards[k] : nullptr;
}

extern "C" int pgh_group_uses_rccl(const pgh_dataset *ds) {
	return ds && ds->IsGroup() && EnsureComms(ds) != nullptr ? 1 : 0;
}

namespace pgh_group {

void Close(pgh_dataset *g) {
	delete static_cast<ShardWorkers *>(g->group_workers); // (joins the workers: their streams and scratch go with them)
	g->group_workers = nullptr;
	DestroyComms(g);
	for (pgh_dataset *s : g->shards) {
		pgh_close(s);
	}
	g->shards.clear();
	delete g;
}

int GetInfo(const pgh_dataset *g, pgh_info *out) {
	int rc = pgh_get_info(g->shards[0], out);
	if (rc != PGH_OK) {
		return rc;
	}
	out->variant_begin = g->v_begin;
	out->variant_end = g->v_end;
	out->device = -1;
	for (size_t k = 1; k < g->shards.size(); k++) {
		pgh_info part;
		(void)pgh_get_info(g->shards[k], &part);
		out->dosage_variant_ct += part.dosage_variant_ct;
		out->dosage_value_ct += part.dosage_value_ct;
		if (!g->has_file) {
			for (int i = 0; i < 8; i++) {
				out->vrtype_hist[i] += part.vrtype_hist[i];
			}
		}
	}
	return PGH_OK;
}

int SubsetCreate(const pgh_dataset *g, const uint64_t *sample_include, pgh_subset **out, char *errbuf) {
	std::unique_ptr<pgh_subset> ss(new pgh_subset());
	ss->ds = g;
	const uint32_t N = g->sample_ct;
	ss->include.assign(sample_include, sample_include + (N + 63) / 64);
	for (uint32_t s = 0; s < N; s++) {
		if ((ss->include[s >> 6] >> (s & 63)) & 1ull) {
			ss->sel.push_back(s);
		}
	}
	ss->n_out = static_cast<uint32_t>(ss->sel.size());
	ss->parts.assign(g->shards.size(), nullptr);
	pgh_subset *raw = ss.get();
	int rc = ForShards(g, errbuf, [&](size_t k, char *eb) {
		return pgh_subset_create(g->shards[k], sample_include, &raw->parts[k], eb);
	});
	if (rc != PGH_OK) {
		pgh_subset_destroy(ss.release());
		return rc;
	}
	*out = ss.release();
	return PGH_OK;
}

int CopyRowsToHost(const pgh_dataset *g, uint32_t v_begin, uint32_t v_end, uint8_t *rows, size_t row_stride,
                   char *errbuf) {
	return ForShards(g, errbuf, [&](size_t k, char *eb) {
		const pgh_dataset *s = g->shards[k];
		const uint32_t lo = std::max(v_begin, s->v_begin), hi = std::min(v_end, s->v_end);
		return lo < hi ? pgh_copy_rows_to_host(s, lo, hi, rows + static_cast<size_t>(lo - v_begin) * row_stride, row_stride, eb)
		               : PGH_OK;
	});
}

// ---------------------------------------------------------------------------
// per-variant outputs: every shard fills its slice, nothing is exchanged
// ---------------------------------------------------------------------------

int CountsRange(const pgh_dataset *g, const pgh_subset *ss, uint32_t v_begin, uint32_t v_end, uint32_t (*out)[4],
                char *errbuf) {
	int rc = CheckGroupSubset(g, ss, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	return ForShards(g, errbuf, [&](size_t k, char *eb) {
		const pgh_dataset *s = g->shards[k];
		const uint32_t lo = std::max(v_begin, s->v_begin), hi = std::min(v_end, s->v_end);
		return lo < hi ? pgh_counts_range(s, PartOf(ss, k), lo, hi, out + (lo - v_begin), eb) : PGH_OK;
	});
}

int UnpackRange(const pgh_dataset *g, const pgh_subset *ss, uint32_t v_begin, uint32_t v_end, int8_t *out,
                uint64_t *validity, int missing_code, char *errbuf) {
	int rc = CheckGroupSubset(g, ss, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	const size_t n_out = ss ? ss->n_out : g->sample_ct;
	const size_t val_words = (n_out + 63) / 64;
	return ForShards(g, errbuf, [&](size_t k, char *eb) {
		const pgh_dataset *s = g->shards[k];
		const uint32_t lo = std::max(v_begin, s->v_begin), hi = std::min(v_end, s->v_end);
		if (lo >= hi) {
			return static_cast<int>(PGH_OK);
		}
		const size_t r0 = lo - v_begin;
		return pgh_unpack_range(s, PartOf(ss, k), lo, hi, out ? out + r0 * n_out : nullptr,
		                        validity ? validity + r0 * val_words : nullptr, missing_code, eb);
	});
}

//! Range-or-list calls whose output has one row of `row_elems` elements per requested variant.
template <class T, class Call>
static int PerVariantRows(const pgh_dataset *g, uint32_t variant_begin, uint32_t n_variants, const uint32_t *vidx,
                          T *out, size_t row_elems, char *errbuf, Call call) {
	if (!vidx) {
		int rc = CheckRange(g, variant_begin, variant_begin + n_variants, errbuf);
		if (rc != PGH_OK) {
			return rc;
		}
		const uint32_t v_end = variant_begin + n_variants;
		return ForShards(g, errbuf, [&](size_t k, char *eb) {
			const pgh_dataset *s = g->shards[k];
			const uint32_t lo = std::max(variant_begin, s->v_begin), hi = std::min(v_end, s->v_end);
			return lo < hi ? call(k, lo, hi - lo, nullptr, out + static_cast<size_t>(lo - variant_begin) * row_elems, eb)
			               : static_cast<int>(PGH_OK);
		});
	}
	ListCut cut;
	int rc = CutList(g, n_variants, vidx, cut, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	return ForShards(g, errbuf, [&](size_t k, char *eb) {
		const uint32_t n_k = static_cast<uint32_t>(cut.idx[k].size());
		if (n_k == 0) {
			return static_cast<int>(PGH_OK);
		}
		std::vector<T> tmp(static_cast<size_t>(n_k) * row_elems);
		int rck = call(k, 0, n_k, cut.idx[k].data(), tmp.data(), eb);
		if (rck != PGH_OK) {
			return rck;
		}
		for (uint32_t i = 0; i < n_k; i++) {
			std::memcpy(out + static_cast<size_t>(cut.pos[k][i]) * row_elems, &tmp[static_cast<size_t>(i) * row_elems],
			            sizeof(T) * row_elems);
		}
		return static_cast<int>(PGH_OK);
	});
}

int DosageSums(const pgh_dataset *g, const pgh_subset *ss, uint32_t variant_begin, uint32_t n_variants,
               const uint32_t *vidx, uint64_t (*sums)[3], char *errbuf) {
	int rc = CheckGroupSubset(g, ss, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	return PerVariantRows<uint64_t>(g, variant_begin, n_variants, vidx, &sums[0][0], 3, errbuf,
	                                [&](size_t k, uint32_t vb, uint32_t n, const uint32_t *list, uint64_t *dst, char *eb) {
		                                return pgh_dosage_sums(g->shards[k], PartOf(ss, k), vb, n, list,
		                                                       reinterpret_cast<uint64_t(*)[3]>(dst), eb);
	                                });
}

int DosageUnpack(const pgh_dataset *g, const pgh_subset *ss, uint32_t variant_begin, uint32_t n_variants,
                 const uint32_t *vidx, double *out, char *errbuf) {
	int rc = CheckGroupSubset(g, ss, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	const size_t n_out = ss ? ss->n_out : g->sample_ct;
	return PerVariantRows<double>(g, variant_begin, n_variants, vidx, out, n_out, errbuf,
	                              [&](size_t k, uint32_t vb, uint32_t n, const uint32_t *list, double *dst, char *eb) {
		                              return pgh_dosage_unpack(g->shards[k], PartOf(ss, k), vb, n, list, dst, eb);
	                              });
}

//! Sample-major matrices (one column per listed variant): every shard unpacks its variants' columns.
template <class T, class Call>
static int SampleMajor(const pgh_dataset *g, const pgh_subset *ss, uint32_t n_variants, const uint32_t *vidx, T *out,
                       char *errbuf, Call call) {
	int rc = CheckGroupSubset(g, ss, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	ListCut cut;
	rc = CutList(g, n_variants, vidx, cut, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	const size_t n_out = ss ? ss->n_out : g->sample_ct;
	return ForShards(g, errbuf, [&](size_t k, char *eb) {
		const uint32_t n_k = static_cast<uint32_t>(cut.idx[k].size());
		if (n_k == 0) {
			return static_cast<int>(PGH_OK);
		}
		std::vector<T> tmp(n_out * n_k);
		int rck = call(k, n_k, cut.idx[k].data(), tmp.data(), eb);
		if (rck != PGH_OK) {
			return rck;
		}
		for (size_t s = 0; s < n_out; s++) {
			T *dst = out + s * n_variants;
			const T *src = &tmp[s * n_k];
			for (uint32_t i = 0; i < n_k; i++) {
				dst[cut.pos[k][i]] = src[i];
			}
		}
		return static_cast<int>(PGH_OK);
	});
}

int UnpackSamples(const pgh_dataset *g, const pgh_subset *ss, uint32_t n_variants, const uint32_t *vidx, int8_t *out,
                  int missing_code, char *errbuf) {
	return SampleMajor<int8_t>(g, ss, n_variants, vidx, out, errbuf,
	                           [&](size_t k, uint32_t n, const uint32_t *list, int8_t *dst, char *eb) {
		                           return pgh_unpack_samples(g->shards[k], PartOf(ss, k), n, list, dst, missing_code, eb);
	                           });
}

int DosageUnpackSamples(const pgh_dataset *g, const pgh_subset *ss, uint32_t n_variants, const uint32_t *vidx,
                        double *out, char *errbuf) {
	return SampleMajor<double>(g, ss, n_variants, vidx, out, errbuf,
	                           [&](size_t k, uint32_t n, const uint32_t *list, double *dst, char *eb) {
		                           return pgh_dosage_unpack_samples(g->shards[k], PartOf(ss, k), n, list, dst, eb);
	                           });
}

// ---------------------------------------------------------------------------
// per-sample outputs: partials per shard, summed
// ---------------------------------------------------------------------------

int MissingPerSample(const pgh_dataset *g, const pgh_subset *ss, uint32_t v_begin, uint32_t v_end, uint32_t *out,
                     char *errbuf) {
	int rc = CheckGroupSubset(g, ss, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	const size_t n_out = ss ? ss->n_out : g->sample_ct;
	std::vector<std::vector<uint32_t>> part(g->shards.size());
	rc = ForShards(g, errbuf, [&](size_t k, char *eb) {
		const pgh_dataset *s = g->shards[k];
		const uint32_t lo = std::max(v_begin, s->v_begin), hi = std::min(v_end, s->v_end);
		if (lo >= hi) {
			return static_cast<int>(PGH_OK);
		}
		part[k].resize(n_out);
		return pgh_missing_per_sample(s, PartOf(ss, k), lo, hi, part[k].data(), eb);
	});
	if (rc != PGH_OK) {
		return rc;
	}
	std::fill(out, out + n_out, 0u);
	for (const auto &p : part) {
		for (size_t i = 0; i < p.size(); i++) {
			out[i] += p[i];
		}
	}
	return PGH_OK;
}

int SampleCounts(const pgh_dataset *g, const pgh_subset *ss, uint32_t variant_begin, uint32_t n_var,
                 const uint32_t *vidx, uint32_t (*counts)[4], char *errbuf) {
	int rc = CheckGroupSubset(g, ss, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	const size_t n_out = ss ? ss->n_out : g->sample_ct;
	ListCut cut;
	if (vidx) {
		rc = CutList(g, n_var, vidx, cut, errbuf);
	} else {
		rc = CheckRange(g, variant_begin, variant_begin + n_var, errbuf);
	}
	if (rc != PGH_OK) {
		return rc;
	}
	std::vector<std::vector<uint32_t>> part(g->shards.size());
	rc = ForShards(g, errbuf, [&](size_t k, char *eb) {
		const pgh_dataset *s = g->shards[k];
		uint32_t vb = 0, n_k = 0;
		const uint32_t *list = nullptr;
		if (vidx) {
			n_k = static_cast<uint32_t>(cut.idx[k].size());
			list = cut.idx[k].data();
		} else {
			const uint32_t lo = std::max(variant_begin, s->v_begin), hi = std::min(variant_begin + n_var, s->v_end);
			vb = lo;
			n_k = lo < hi ? hi - lo : 0;
		}
		if (n_k == 0) {
			return static_cast<int>(PGH_OK);
		}
		part[k].resize(4 * n_out);
		return pgh_sample_counts(s, PartOf(ss, k), vb, n_k, list, reinterpret_cast<uint32_t(*)[4]>(part[k].data()), eb);
	});
	if (rc != PGH_OK) {
		return rc;
	}
	std::fill(&counts[0][0], &counts[0][0] + 4 * n_out, 0u);
	for (const auto &p : part) {
		for (size_t i = 0; i < p.size(); i++) {
			(&counts[0][0])[i] += p[i];
		}
	}
	return PGH_OK;
}

int Score(const pgh_dataset *g, const pgh_subset *ss, uint32_t n_scored, const uint32_t *vidx, const double *weights,
          const uint8_t *flip, uint32_t n_cols, int mode, const uint32_t (*counts)[4], double *score_sum,
          double *dosage_sum, uint32_t *allele_ct, char *errbuf) {
	int rc = CheckGroupSubset(g, ss, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	if (n_scored && (!vidx || !weights)) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	if (n_cols == 0 || n_cols > 4096) {
		SetErr(errbuf, "n_cols must be between 1 and 4096");
		return PGH_ERR_ARG;
	}
	ListCut cut;
	rc = CutList(g, n_scored, vidx, cut, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	const size_t K = g->shards.size();
	const uint32_t N = g->sample_ct;
	// every shard scores its variants into buffers on its own device (the reference's per-thread partial sums,
	// src/plink_score.cpp:575-654) ...
	std::vector<DevBuf> d_score(K), d_dos(K), d_ac(K);
	std::vector<double *> p_score(K, nullptr), p_dos(K, nullptr);
	std::vector<uint32_t *> p_ac(K, nullptr);
	rc = ForShards(g, errbuf, [&](size_t k, char *eb) {
		const uint32_t n_k = static_cast<uint32_t>(cut.idx[k].size());
		if (n_k == 0 && k != 0) {
			return static_cast<int>(PGH_OK); // nothing to add; shard 0 always provides the (zero) base
		}
		char *errbuf = eb; // PGH_HIP reports here
		PGH_HIP(d_score[k].Alloc(sizeof(double) * N * n_cols), "hipMalloc(score out)");
		PGH_HIP(d_ac[k].Alloc(sizeof(uint32_t) * N), "hipMalloc(score out)");
		if (dosage_sum) {
			PGH_HIP(d_dos[k].Alloc(sizeof(double) * N), "hipMalloc(score out)");
		}
		std::vector<double> w_k(static_cast<size_t>(n_k) * n_cols);
		std::vector<uint8_t> f_k(flip ? n_k : 0);
		std::vector<uint32_t> c_k(counts ? 4ull * n_k : 0);
		for (uint32_t i = 0; i < n_k; i++) {
			std::memcpy(&w_k[static_cast<size_t>(i) * n_cols], weights + static_cast<size_t>(cut.pos[k][i]) * n_cols,
			            sizeof(double) * n_cols);
			if (flip) {
				f_k[i] = flip[cut.pos[k][i]];
			}
			if (counts) {
				std::memcpy(&c_k[4ull * i], counts[cut.pos[k][i]], 16);
			}
		}
		int rck = PghScoreDevCounts(g->shards[k], PartOf(ss, k), n_k, cut.idx[k].data(), w_k.data(),
		                            flip ? f_k.data() : nullptr, n_cols, mode,
		                            counts ? reinterpret_cast<const uint32_t(*)[4]>(c_k.data()) : nullptr, d_score[k].p,
		                            d_dos[k].p, d_ac[k].p, PghThreadStream(), eb);
		if (rck != PGH_OK) {
			return rck;
		}
		PGH_HIP(hipStreamSynchronize(PghThreadStream()), "score sync");
		p_score[k] = d_score[k].As<double>();
		p_dos[k] = d_dos[k].As<double>();
		p_ac[k] = d_ac[k].As<uint32_t>();
		return static_cast<int>(PGH_OK);
	});
	if (rc != PGH_OK) {
		return rc;
	}
	// ... and the merge of src/plink_score.cpp:657-664 runs on the first shard's device
	rc = SumToRoot<double>(g, p_score, static_cast<uint64_t>(N) * n_cols, errbuf);
	if (rc == PGH_OK && dosage_sum) {
		rc = SumToRoot<double>(g, p_dos, N, errbuf);
	}
	if (rc == PGH_OK) {
		rc = SumToRoot<uint32_t>(g, p_ac, N, errbuf);
	}
	if (rc != PGH_OK) {
		return rc;
	}
	DeviceScope scope(g->shards[0]->device);
	std::vector<double> h_score(static_cast<size_t>(N) * n_cols), h_dos(dosage_sum ? N : 0);
	std::vector<uint32_t> h_ac(N);
	PGH_HIP(hipMemcpy(h_score.data(), p_score[0], sizeof(double) * h_score.size(), hipMemcpyDeviceToHost), "score copy");
	if (dosage_sum) {
		PGH_HIP(hipMemcpy(h_dos.data(), p_dos[0], sizeof(double) * N, hipMemcpyDeviceToHost), "score copy");
	}
	PGH_HIP(hipMemcpy(h_ac.data(), p_ac[0], sizeof(uint32_t) * N, hipMemcpyDeviceToHost), "score copy");
	Compact<double>(ss, h_score.data(), n_cols, score_sum, N);
	if (dosage_sum) {
		Compact<double>(ss, h_dos.data(), 1, dosage_sum, N);
	}
	Compact<uint32_t>(ss, h_ac.data(), 1, allele_ct, N);
	return PGH_OK;
}

namespace {
// The all-reduce pgh_pca_sharded asks its host for, between the shard threads of one process.  With communicators
// (shards on distinct devices) it is ncclAllReduce on the calling shard's own stream: nothing to meet for on the
// host and no stream drain -- RCCL orders the exchange on the device.  Without them every thread brings its device
// buffer, the first shard's thread gathers the others' over the device-to-device path, adds, and sends the sum back.
struct InProcessAllReduce {
	const pgh_dataset *group = nullptr;
	const Shards *shards = nullptr;
	const std::vector<ncclComm_t> *comms = nullptr;
	Rendezvous *meet = nullptr;
	std::vector<void *> bufs;
	DevBuf stage; // on the root device
	size_t stage_bytes = 0;
	std::atomic<int> failed {0};
};
struct AllReduceRank {
	InProcessAllReduce *shared;
	size_t k;
};

int AllReduceCallback(void *ctx, void *d_buf, uint64_t count, void *stream) {
	auto *rank = static_cast<AllReduceRank *>(ctx);
	InProcessAllReduce *ar = rank->shared;
	const Shards &shards = *ar->shards;
	if (ar->failed.load()) {
		return 1;
	}
	if (ar->comms) {
		const ncclResult_t r = Rccl().AllReduce(d_buf, d_buf, count, ncclDouble, ncclSum, (*ar->comms)[rank->k],
		                                        static_cast<hipStream_t>(stream));
		if (r != ncclSuccess) {
			ar->failed.store(1);
		}
		return ar->failed.load();
	}
	if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) {
		ar->failed.store(1);
	}
	ar->bufs[rank->k] = d_buf;
	if (!ar->meet->Wait()) {
		return 1;
	}
	if (rank->k == 0 && !ar->failed.load()) {
		const size_t bytes = sizeof(double) * count;
		const int root = shards[0]->device;
		hipError_t e = hipSuccess;
		if (ar->stage_bytes < bytes) {
			if (ar->stage.p) {
				(void)hipFree(ar->stage.p);
				ar->stage.p = nullptr;
			}
			e = ar->stage.Alloc(bytes);
			ar->stage_bytes = e == hipSuccess ? bytes : 0;
		}
		for (size_t j = 1; j < shards.size() && e == hipSuccess; j++) {
			e = hipMemcpyPeerAsync(ar->stage.p, root, ar->bufs[j], shards[j]->device, bytes, PghThreadStream());
			if (e == hipSuccess) {
				e = pgh::LaunchAddF64(static_cast<double *>(d_buf), ar->stage.As<double>(), count, PghThreadStream());
			}
		}
		for (size_t j = 1; j < shards.size() && e == hipSuccess; j++) {
			e = hipMemcpyPeerAsync(ar->bufs[j], shards[j]->device, d_buf, root, bytes, PghThreadStream());
		}
		if (e == hipSuccess) {
			e = hipStreamSynchronize(PghThreadStream());
		}
		if (e != hipSuccess) {
			ar->failed.store(1);
		}
	}
	if (!ar->meet->Wait()) {
		return 1;
	}
	return ar->failed.load();
}
} // namespace

int Pca(const pgh_dataset *g, const pgh_subset *ss, uint32_t n_var, const uint32_t *vidx, const double *center,
        const double *inv_stdev, uint32_t n_pcs, const double *g1_init, double *eigenvalues, double *eigenvectors,
        char *errbuf) {
	int rc = CheckGroupSubset(g, ss, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	if (!g1_init || !eigenvalues || !eigenvectors || n_pcs == 0 || (n_var && (!vidx || !center || !inv_stdev))) {
		SetErr(errbuf, "null or empty argument");
		return PGH_ERR_ARG;
	}
	ListCut cut;
	rc = CutList(g, n_var, vidx, cut, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	const size_t K = g->shards.size();
	const size_t n_out = ss ? ss->n_out : g->sample_ct;
	Rendezvous meet(K);
	InProcessAllReduce ar;
	ar.group = g;
	ar.shards = &g->shards;
	ar.comms = EnsureComms(g);
	ar.meet = &meet;
	ar.bufs.assign(K, nullptr);
	std::vector<AllReduceRank> ranks(K);
	std::vector<std::vector<double>> ev(K), vec(K);
	// Every shard thread must make the same sequence of all-reduce calls, so argument errors that pgh_pca_sharded
	// would report before its first exchange are checked here, once, for all of them.
	const uint64_t qq = static_cast<uint64_t>(n_pcs + 1) * 2u * n_pcs;
	if (n_var < qq || n_out < qq) {
		SetErr(errbuf, "too few variants or samples for the requested number of PCs");
		return PGH_ERR_ARG;
	}
	// PGH_TEST_PCA_FAIL_SHARD=k: shard k gives up before its first exchange (tests: the others must not wait for it)
	const char *fail_env = std::getenv("PGH_TEST_PCA_FAIL_SHARD");
	const long fail_shard = fail_env ? std::atol(fail_env) : -1;
	rc = ForShards(g, errbuf, [&](size_t k, char *eb) {
		const uint32_t n_k = static_cast<uint32_t>(cut.idx[k].size());
		std::vector<double> c_k(n_k), i_k(n_k);
		for (uint32_t i = 0; i < n_k; i++) {
			c_k[i] = center[cut.pos[k][i]];
			i_k[i] = inv_stdev[cut.pos[k][i]];
		}
		ranks[k] = AllReduceRank {&ar, k};
		ev[k].resize(n_pcs);
		vec[k].resize(n_out * n_pcs);
		int rck;
		if (fail_shard == static_cast<long>(k)) {
			SetErr(eb, "shard failure injected by PGH_TEST_PCA_FAIL_SHARD");
			rck = PGH_ERR_DEVICE;
		} else {
			rck = pgh_pca_sharded(g->shards[k], PartOf(ss, k), n_k, cut.idx[k].data(), c_k.data(), i_k.data(), n_var,
			                      n_pcs, g1_init, AllReduceCallback, &ranks[k], ev[k].data(), vec[k].data(), eb);
		}
		if (rck != PGH_OK && !ar.failed.exchange(1)) {
			// This rank is out -- an allocation that failed on a fuller device, a launch error, anything that made
			// pgh_pca_sharded return between two exchanges -- and will not come to the next one: call the meeting
			// off (the copy path), or abort the collectives the others may already be inside (RCCL).
			meet.Abort();
			if (ar.comms) {
				AbortComms(g);
			}
		}
		return rck;
	});
	{
		DeviceScope scope(g->shards[0]->device);
		if (ar.stage.p) {
			(void)hipFree(ar.stage.p);
			ar.stage.p = nullptr;
		}
	}
	if (rc != PGH_OK) {
		return rc;
	}
	std::memcpy(eigenvalues, ev[0].data(), sizeof(double) * n_pcs);
	std::memcpy(eigenvectors, vec[0].data(), sizeof(double) * vec[0].size());
	return PGH_OK;
}

int LdPairs(const pgh_dataset *g, const pgh_subset *ss, uint32_t n_pairs, const uint32_t *vidx_a, const uint32_t *vidx_b,
            uint32_t (*sums)[6], char *errbuf) {
	int rc = CheckGroupSubset(g, ss, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	const size_t K = g->shards.size();
	// pairs inside one shard go to that shard; pairs that straddle two (a window across a shard boundary) are
	// computed on a small scratch dataset assembled from the rows they name
	std::vector<std::vector<uint32_t>> pos(K), a(K), b(K);
	std::vector<uint32_t> x_pos, x_a, x_b;
	for (uint32_t p = 0; p < n_pairs; p++) {
		const uint32_t va = vidx_a[p], vb = vidx_b[p];
		if (va < g->v_begin || va >= g->v_end || vb < g->v_begin || vb >= g->v_end) {
			SetErr(errbuf, "variant index outside the resident range");
			return PGH_ERR_ARG;
		}
		const size_t ka = ShardOf(g, va), kb = ShardOf(g, vb);
		if (ka == kb) {
			pos[ka].push_back(p);
			a[ka].push_back(va);
			b[ka].push_back(vb);
		} else {
			x_pos.push_back(p);
			x_a.push_back(va);
			x_b.push_back(vb);
		}
	}
	rc = ForShards(g, errbuf, [&](size_t k, char *eb) {
		const uint32_t n_k = static_cast<uint32_t>(pos[k].size());
		if (n_k == 0) {
			return static_cast<int>(PGH_OK);
		}
		std::vector<uint32_t> tmp(6ull * n_k);
		int rck = pgh_ld_pairs(g->shards[k], PartOf(ss, k), n_k, a[k].data(), b[k].data(),
		                       reinterpret_cast<uint32_t(*)[6]>(tmp.data()), eb);
		if (rck != PGH_OK) {
			return rck;
		}
		for (uint32_t i = 0; i < n_k; i++) {
			std::memcpy(sums[pos[k][i]], &tmp[6ull * i], sizeof(uint32_t) * 6);
		}
		return static_cast<int>(PGH_OK);
	});
	if (rc != PGH_OK || x_pos.empty()) {
		return rc;
	}
	std::vector<uint32_t> need(x_a);
	need.insert(need.end(), x_b.begin(), x_b.end());
	std::sort(need.begin(), need.end());
	need.erase(std::unique(need.begin(), need.end()), need.end());
	const size_t rb = g->record_bytes;
	std::vector<uint8_t> host(need.size() * rb);
	for (size_t i = 0; i < need.size();) {
		// (runs of consecutive rows -- a window's anchors and partners mostly are -- in one copy each)
		size_t j = i + 1;
		while (j < need.size() && need[j] == need[j - 1] + 1 && ShardOf(g, need[j]) == ShardOf(g, need[i])) {
			j++;
		}
		rc = CopyRowsToHost(g, need[i], need[j - 1] + 1, host.data() + i * rb, rb, errbuf);
		if (rc != PGH_OK) {
			return rc;
		}
		i = j;
	}
	DeviceScope scope(g->shards[0]->device);
	pgh_dataset *scratch = nullptr;
	rc = pgh_from_host_rows(host.data(), rb, static_cast<uint32_t>(need.size()), g->sample_ct, &scratch, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	pgh_subset *scratch_ss = nullptr;
	if (ss) {
		rc = pgh_subset_create(scratch, ss->include.data(), &scratch_ss, errbuf);
	}
	if (rc == PGH_OK) {
		auto local = [&](uint32_t v) {
			return static_cast<uint32_t>(std::lower_bound(need.begin(), need.end(), v) - need.begin());
		};
		std::vector<uint32_t> la(x_a.size()), lb(x_b.size()), tmp(6 * x_a.size());
		for (size_t i = 0; i < x_a.size(); i++) {
			la[i] = local(x_a[i]);
			lb[i] = local(x_b[i]);
		}
		rc = pgh_ld_pairs(scratch, scratch_ss, static_cast<uint32_t>(la.size()), la.data(), lb.data(),
		                  reinterpret_cast<uint32_t(*)[6]>(tmp.data()), errbuf);
		for (size_t i = 0; rc == PGH_OK && i < x_pos.size(); i++) {
			std::memcpy(sums[x_pos[i]], &tmp[6 * i], sizeof(uint32_t) * 6);
		}
	}
	pgh_subset_destroy(scratch_ss);
	pgh_close(scratch);
	return rc;
}

pgh_reader *ReaderFor(pgh_reader *rd, uint32_t vidx) {
	const pgh_dataset *g = rd->ds;
	if (vidx < g->v_begin || vidx >= g->v_end) {
		rd->err = "variant index " + std::to_string(vidx) + " outside the resident range";
		return nullptr;
	}
	const size_t k = ShardOf(g, vidx);
	if (rd->parts.size() != g->shards.size()) {
		rd->parts.assign(g->shards.size(), nullptr);
	}
	if (!rd->parts[k]) {
		char eb[PGH_ERRBUF_LEN] = {0};
		if (pgh_reader_create(g->shards[k], PartOf(rd->subset, k), &rd->parts[k], eb) != PGH_OK) {
			rd->err = eb;
			return nullptr;
		}
	}
	return rd->parts[k];
}

} // namespace pgh_group

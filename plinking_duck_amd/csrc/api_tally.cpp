// api_tally.cpp -- the tally pass (include/pgenhip.h, "tally pass"): one asynchronous walk of the resident
// matrix whose per-variant counts, per-sample missing counts and exact-test ln p land in pinned host memory
// batch by batch, for any number of scan threads and any number of later table-function calls to read.
//
// Shape of one part (= one device's share of the pass):
//
//   main stream:  tally(b0) D2H(b0) ev  tally(b1) D2H(b1) ev  ...  [sum of the per-sample partials] D2H ev
//   side stream:            wait ev(b0) hwe(b0) D2H ev   wait ev(b1) hwe(b1) D2H ev ...
//
// Everything is enqueued by pgh_tally_start / pgh_tally_request; nobody blocks until a scan thread asks for
// rows (pgh_tally_wait = hipEventSynchronize on the last batch it needs).
#include "api_internal.hpp"

#include <atomic>

namespace {

enum { kCounts = 0, kHwe = 1, kHweMidp = 2, kProducts = 3 };

struct TallyPart {
	const pgh_dataset *ds = nullptr; // one device's dataset
	const pgh_subset *subset = nullptr;
	int device = 0;
	uint32_t v_begin = 0, v_end = 0; // this part's variants
	uint32_t batch = 0;              // variants per batch
	uint32_t n_batches = 0;
	hipStream_t main = nullptr, side = nullptr;
	std::vector<hipEvent_t> ev[kProducts]; // per batch; empty until the product is enqueued
	hipEvent_t ev_missing = nullptr;
	uint32_t *d_counts = nullptr; // [v_end - v_begin][4]
	double *d_lnp[2] = {nullptr, nullptr};
	uint32_t *d_missing = nullptr; // uint32[padded N], raw sample order
	void *d_scratch = nullptr;     // per-slice partial rows of the column tally
	void *d_hwe_order = nullptr;   // the exact tests' ordering scratch (side stream; kernels.hpp: LaunchHweBatch)
	size_t scratch_bytes = 0;
	bool fused = false; // the per-sample tally rides the counts kernel
	bool missing_enqueued = false;
};

} // namespace

struct pgh_tally {
	const pgh_dataset *ds = nullptr;
	const pgh_subset *subset = nullptr;
	uint32_t v_begin = 0, v_end = 0;
	uint32_t sample_ct = 0, n_out = 0;
	uint32_t (*h_counts)[4] = nullptr; // pinned, [v_end - v_begin]
	double *h_lnp[2] = {nullptr, nullptr};
	uint32_t *h_missing = nullptr; // pinned, parts x padded N (raw order), summed on the host
	uint32_t padded = 0;
	std::mutex mu; // pgh_tally_request / the first pgh_tally_sample_missing
	std::atomic<uint32_t> products {0};
	bool missing_summed = false;
	std::vector<uint32_t> missing_out; // compacted to the included samples
	std::vector<TallyPart> parts;
};

namespace {

//! Variants per batch: about 16 GB of rows (2.5 ms of HBM time).  The first rows land that long after the start;
//! shorter launches cost the column tally more than they buy (a launch ends with every workgroup writing its
//! slice's planes and a tail of idle CUs: 32,768-variant launches ran the 1 M x 500 k pass in 23.4 ms, 131,072-
//! variant ones in 20.6, one launch in 20.7).  A multiple of 4096, never beyond the part.
uint32_t ChooseBatch(uint64_t pitch, uint32_t variants) {
	uint64_t b = (16ull << 30) / (pitch ? pitch : 1);
	if (const char *env = std::getenv("PGH_TALLY_BATCH")) { // tests: many batches on a small matrix
		const long v = std::atol(env);
		if (v > 0) {
			b = static_cast<uint64_t>(v);
		}
	}
	b = b / 4096 * 4096;
	if (b < 4096) {
		b = 4096;
	}
	if (b > variants) {
		b = variants;
	}
	return static_cast<uint32_t>(b ? b : 1);
}

void ReleasePart(TallyPart &p) {
	DeviceScope scope(p.device);
	if (p.main) {
		(void)hipStreamSynchronize(p.main);
	}
	if (p.side) {
		(void)hipStreamSynchronize(p.side);
	}
	for (auto &list : p.ev) {
		for (auto e : list) {
			(void)hipEventDestroy(e);
		}
		list.clear();
	}
	if (p.ev_missing) {
		(void)hipEventDestroy(p.ev_missing);
	}
	if (p.main) {
		(void)hipStreamDestroy(p.main);
	}
	if (p.side) {
		(void)hipStreamDestroy(p.side);
	}
	(void)hipFree(p.d_counts);
	(void)hipFree(p.d_lnp[0]);
	(void)hipFree(p.d_lnp[1]);
	(void)hipFree(p.d_missing);
	(void)hipFree(p.d_scratch);
	(void)hipFree(p.d_hwe_order);
	p = TallyPart();
}

int NewEvents(std::vector<hipEvent_t> &list, uint32_t n, char *errbuf) {
	list.reserve(n);
	for (uint32_t i = 0; i < n; i++) {
		hipEvent_t e = nullptr;
		PGH_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate(tally)");
		list.push_back(e);
	}
	return PGH_OK;
}

//! The exact tests of every batch, each behind its batch's counts, on the side stream.  k_hwe_batch gives a
//! variant to a lane, so a launch wants some hundred thousand variants to fill the chip: batches are grouped into
//! launches of >= kHweLaunch variants where their counts have already landed (a product asked of a finished pass:
//! 31 launches of one 32,768-variant batch each took 22 ms at 1 M x 500 k, the kernel's time at full width is 5),
//! and go one by one while the tallies are still coming (they hide under the next batch's tally).
int EnqueueHwe(pgh_tally *t, TallyPart &p, int which, char *errbuf) {
	constexpr uint32_t kHweLaunch = 262144;
	DeviceScope scope(p.device);
	const uint32_t midp = which == kHweMidp ? 1u : 0u;
	const uint32_t n = p.v_end - p.v_begin;
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&p.d_lnp[midp]), sizeof(double) * (n ? n : 1)), "hipMalloc(tally ln p)");
	if (!p.d_hwe_order) { // (every launch of this part runs on p.side, one after the other: one block serves them all)
		const size_t bytes = pgh::HweOrderScratchBytes(std::max<uint32_t>(n, pgh::kHweOrderMin));
		PGH_HIP(PghMalloc(&p.d_hwe_order, bytes), "hipMalloc(tally exact-test order)");
	}
	int rc = NewEvents(p.ev[which], p.n_batches, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	for (uint32_t b = 0; b < p.n_batches;) {
		uint32_t last = b; // the launch covers batches [b, last]
		while (last + 1 < p.n_batches && (last + 1 - b) * p.batch < kHweLaunch &&
		       hipEventQuery(p.ev[kCounts][last + 1]) == hipSuccess) {
			last++;
		}
		(void)hipGetLastError(); // hipErrorNotReady of the query above is not an error
		const uint32_t r0 = b * p.batch, r1 = std::min<uint64_t>(n, static_cast<uint64_t>(last + 1) * p.batch);
		PGH_HIP(hipStreamWaitEvent(p.side, p.ev[kCounts][last], 0), "tally stream wait");
		PGH_HIP(pgh::LaunchHweBatch(p.d_counts + 4ull * r0, r1 - r0, midp, p.d_lnp[midp] + r0, p.side, p.d_hwe_order),
		        "exact-test kernel");
		PGH_HIP(hipMemcpyAsync(t->h_lnp[midp] + (p.v_begin - t->v_begin) + r0, p.d_lnp[midp] + r0,
		                       sizeof(double) * (r1 - r0), hipMemcpyDeviceToHost, p.side),
		        "tally ln p copy");
		for (uint32_t k = b; k <= last; k++) {
			PGH_HIP(hipEventRecord(p.ev[which][k], p.side), "hipEventRecord(tally)");
		}
		b = last + 1;
	}
	return PGH_OK;
}

//! The per-sample missing tally as a sweep of its own (a pass with a subset, or a product asked for later).
int EnqueueMissingSweep(pgh_tally *t, TallyPart &p, size_t part_idx, char *errbuf) {
	DeviceScope scope(p.device);
	const uint32_t n = p.v_end - p.v_begin;
	if (!p.d_missing) {
		PGH_HIP(PghMalloc(reinterpret_cast<void **>(&p.d_missing), sizeof(uint32_t) * t->padded), "hipMalloc(tally missing)");
	}
	const size_t need = pgh::MissingPerSampleScratchBytes(p.ds->record_bytes, n);
	if (need > p.scratch_bytes) {
		// the main stream may still be reading the block (the fused batches of a running pass never get here:
		// their per-sample product is enqueued with them)
		PGH_HIP(hipStreamSynchronize(p.main), "tally sync");
		PGH_HIP(hipStreamSynchronize(p.side), "tally sync");
		(void)hipFree(p.d_scratch);
		p.d_scratch = nullptr;
		p.scratch_bytes = 0;
		PGH_HIP(PghMalloc(&p.d_scratch, need), "hipMalloc(tally scratch)");
		p.scratch_bytes = need;
	}
	// behind everything the main stream holds, so the scratch block has one user at a time
	PGH_HIP(pgh::LaunchMissingPerSample(p.ds->View(), p.v_begin - p.ds->v_begin, nullptr, n, nullptr,
	                                    static_cast<uint32_t *>(p.d_scratch), p.d_missing, p.main),
	        "missing-per-sample kernel");
	PGH_HIP(hipMemcpyAsync(t->h_missing + part_idx * t->padded, p.d_missing, sizeof(uint32_t) * t->sample_ct,
	                       hipMemcpyDeviceToHost, p.main),
	        "tally missing copy");
	if (!p.ev_missing) {
		PGH_HIP(hipEventCreateWithFlags(&p.ev_missing, hipEventDisableTiming), "hipEventCreate(tally)");
	}
	PGH_HIP(hipEventRecord(p.ev_missing, p.main), "hipEventRecord(tally)");
	p.missing_enqueued = true;
	return PGH_OK;
}

int StartPart(pgh_tally *t, TallyPart &p, size_t part_idx, uint32_t products, char *errbuf) {
	DeviceScope scope(p.device);
	const uint32_t n = p.v_end - p.v_begin;
	p.batch = ChooseBatch(p.ds->pitch, n);
	p.n_batches = n ? (n + p.batch - 1) / p.batch : 0;
	// (stream priorities -- tallies high, exact tests low -- were tried: no measurable difference at 1 M x 500 k)
	PGH_HIP(hipStreamCreateWithFlags(&p.main, hipStreamNonBlocking), "hipStreamCreate(tally)");
	PGH_HIP(hipStreamCreateWithFlags(&p.side, hipStreamNonBlocking), "hipStreamCreate(tally)");
	PGH_HIP(PghMalloc(reinterpret_cast<void **>(&p.d_counts), 16ull * (n ? n : 1)), "hipMalloc(tally counts)");
	p.fused = !p.subset && (products & PGH_TALLY_SAMPLE_MISSING) != 0;
	if (p.fused) {
		p.scratch_bytes = pgh::MissingPerSampleScratchBytes(p.ds->record_bytes, p.batch);
		PGH_HIP(PghMalloc(&p.d_scratch, p.scratch_bytes), "hipMalloc(tally scratch)");
		PGH_HIP(PghMalloc(reinterpret_cast<void **>(&p.d_missing), sizeof(uint32_t) * t->padded), "hipMalloc(tally missing)");
		PGH_HIP(hipMemsetAsync(p.d_missing, 0, sizeof(uint32_t) * t->padded, p.main), "tally memset");
	}
	int rc = NewEvents(p.ev[kCounts], p.n_batches, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	const pgh::RowView view = p.ds->View();
	for (uint32_t b = 0; b < p.n_batches; b++) {
		const uint32_t r0 = b * p.batch, r1 = std::min(n, r0 + p.batch);
		const uint32_t local = p.v_begin - p.ds->v_begin + r0;
		if (p.fused) {
			PGH_HIP(pgh::LaunchFusedTally(view, local, r1 - r0, static_cast<uint32_t *>(p.d_scratch), p.d_counts + 4ull * r0,
			                              p.d_missing, p.main, true),
			        "fused tally kernel");
		} else {
			PGH_HIP(pgh::LaunchCounts(view, local, nullptr, r1 - r0, p.subset ? p.subset->d_mask2 : nullptr,
			                          p.subset ? p.subset->n_out : p.ds->sample_ct, p.d_counts + 4ull * r0, p.main),
			        "counts kernel");
		}
		PGH_HIP(hipMemcpyAsync(t->h_counts + (p.v_begin - t->v_begin) + r0, p.d_counts + 4ull * r0, 16ull * (r1 - r0),
		                       hipMemcpyDeviceToHost, p.main),
		        "tally counts copy");
		PGH_HIP(hipEventRecord(p.ev[kCounts][b], p.main), "hipEventRecord(tally)");
	}
	if (p.fused) {
		PGH_HIP(hipMemcpyAsync(t->h_missing + part_idx * t->padded, p.d_missing, sizeof(uint32_t) * t->sample_ct,
		                       hipMemcpyDeviceToHost, p.main),
		        "tally missing copy");
		PGH_HIP(hipEventCreateWithFlags(&p.ev_missing, hipEventDisableTiming), "hipEventCreate(tally)");
		PGH_HIP(hipEventRecord(p.ev_missing, p.main), "hipEventRecord(tally)");
		p.missing_enqueued = true;
	}
	return PGH_OK;
}

int HostAlloc(void **out, size_t bytes, char *errbuf) {
	PGH_HIP(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocPortable), "hipHostMalloc(tally)");
	return PGH_OK;
}

//! Adds the products of `want` the pass does not hold yet.  Caller holds t->mu (or is pgh_tally_start).
int AddProducts(pgh_tally *t, uint32_t want, char *errbuf) {
	const uint32_t have = t->products.load();
	const uint32_t fresh = want & ~have;
	const uint32_t n = t->v_end - t->v_begin;
	int rc = PGH_OK;
	for (int which : {kHwe, kHweMidp}) {
		const uint32_t bit = which == kHwe ? PGH_TALLY_HWE : PGH_TALLY_HWE_MIDP;
		if (!(fresh & bit)) {
			continue;
		}
		const uint32_t midp = which == kHweMidp ? 1u : 0u;
		rc = HostAlloc(reinterpret_cast<void **>(&t->h_lnp[midp]), sizeof(double) * n, errbuf);
		for (size_t k = 0; rc == PGH_OK && k < t->parts.size(); k++) {
			rc = EnqueueHwe(t, t->parts[k], which, errbuf);
		}
		if (rc != PGH_OK) {
			return rc;
		}
	}
	if (fresh & PGH_TALLY_SAMPLE_MISSING) {
		for (size_t k = 0; k < t->parts.size(); k++) {
			if (!t->parts[k].missing_enqueued) {
				rc = EnqueueMissingSweep(t, t->parts[k], k, errbuf);
				if (rc != PGH_OK) {
					return rc;
				}
			}
		}
	}
	t->products.fetch_or(fresh);
	return PGH_OK;
}

} // namespace

extern "C" int pgh_host_alloc(size_t bytes, void **out, char *errbuf) {
	if (!out) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	*out = nullptr;
	PGH_HIP(hipHostMalloc(out, bytes ? bytes : 16, hipHostMallocPortable), "hipHostMalloc");
	return PGH_OK;
}

extern "C" void pgh_host_free(void *p) {
	if (p) {
		(void)hipHostFree(p);
	}
}

extern "C" void pgh_tally_destroy(pgh_tally *t) {
	if (!t) {
		return;
	}
	for (auto &p : t->parts) {
		ReleasePart(p);
	}
	(void)hipHostFree(t->h_counts);
	(void)hipHostFree(t->h_lnp[0]);
	(void)hipHostFree(t->h_lnp[1]);
	(void)hipHostFree(t->h_missing);
	delete t;
}

static std::atomic<uint64_t> g_passes_started {0};

extern "C" uint64_t pgh_tally_passes_started(void) {
	return g_passes_started.load();
}

extern "C" int pgh_tally_start(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin, uint32_t v_end,
                               uint32_t products, pgh_tally **out, char *errbuf) {
	if (!out) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	*out = nullptr;
	int rc = CheckRange(ds, v_begin, v_end, errbuf);
	if (rc == PGH_OK) {
		rc = CheckSubset(ds, subset, errbuf);
	}
	if (rc != PGH_OK) {
		return rc;
	}
	if (products & ~static_cast<uint32_t>(PGH_TALLY_COUNTS | PGH_TALLY_SAMPLE_MISSING | PGH_TALLY_HWE | PGH_TALLY_HWE_MIDP)) {
		SetErr(errbuf, "unknown tally product");
		return PGH_ERR_ARG;
	}
	products |= PGH_TALLY_COUNTS;
	auto *t = new pgh_tally();
	t->ds = ds;
	t->subset = subset;
	t->v_begin = v_begin;
	t->v_end = v_end;
	t->sample_ct = ds->sample_ct;
	t->n_out = subset ? subset->n_out : ds->sample_ct;
	t->padded = (ds->sample_ct + 63) / 64 * 64;
	if (ds->IsGroup()) {
		for (size_t k = 0; k < ds->shards.size(); k++) {
			const pgh_dataset *sh = ds->shards[k];
			const uint32_t b = std::max(v_begin, sh->v_begin), e = std::min(v_end, sh->v_end);
			if (b >= e && !(v_begin == v_end && k == 0)) {
				continue;
			}
			TallyPart p;
			p.ds = sh;
			p.subset = subset ? subset->parts[k] : nullptr;
			p.device = sh->device;
			p.v_begin = b < e ? b : v_begin;
			p.v_end = b < e ? e : v_begin;
			t->parts.push_back(p);
		}
	} else {
		TallyPart p;
		p.ds = ds;
		p.subset = subset;
		p.device = ds->device;
		p.v_begin = v_begin;
		p.v_end = v_end;
		t->parts.push_back(p);
	}
	const uint32_t n = v_end - v_begin;
	{
		DeviceScope scope(t->parts.front().device);
		rc = HostAlloc(reinterpret_cast<void **>(&t->h_counts), 16ull * n, errbuf);
		if (rc == PGH_OK) {
			rc = HostAlloc(reinterpret_cast<void **>(&t->h_missing), sizeof(uint32_t) * t->padded * t->parts.size(), errbuf);
		}
	}
	for (size_t k = 0; rc == PGH_OK && k < t->parts.size(); k++) {
		rc = StartPart(t, t->parts[k], k, products, errbuf);
	}
	if (rc == PGH_OK) {
		t->products.store(PGH_TALLY_COUNTS |
		                  (t->parts.front().missing_enqueued ? static_cast<uint32_t>(PGH_TALLY_SAMPLE_MISSING) : 0u));
		rc = AddProducts(t, products, errbuf);
	}
	if (rc != PGH_OK) {
		pgh_tally_destroy(t);
		return rc;
	}
	g_passes_started.fetch_add(1);
	*out = t;
	return PGH_OK;
}

extern "C" int pgh_tally_request(pgh_tally *t, uint32_t products, char *errbuf) {
	if (!t) {
		SetErr(errbuf, "null tally pass");
		return PGH_ERR_ARG;
	}
	if ((products & ~t->products.load()) == 0) {
		return PGH_OK;
	}
	std::lock_guard<std::mutex> lock(t->mu);
	return AddProducts(t, products, errbuf);
}

extern "C" int pgh_tally_wait(pgh_tally *t, uint32_t products, uint32_t v_begin, uint32_t v_end, char *errbuf) {
	if (!t) {
		SetErr(errbuf, "null tally pass");
		return PGH_ERR_ARG;
	}
	if (v_begin > v_end || v_begin < t->v_begin || v_end > t->v_end) {
		SetErr(errbuf, "variant range is outside the tally pass");
		return PGH_ERR_ARG;
	}
	products |= PGH_TALLY_COUNTS;
	if (products & ~t->products.load()) {
		SetErr(errbuf, "tally product was not requested (pgh_tally_request)");
		return PGH_ERR_ARG;
	}
	for (auto &p : t->parts) {
		DeviceScope scope(p.device);
		if (products & PGH_TALLY_SAMPLE_MISSING) {
			PGH_HIP(hipEventSynchronize(p.ev_missing), "tally wait");
		}
		const uint32_t b = std::max(v_begin, p.v_begin), e = std::min(v_end, p.v_end);
		if (b >= e) {
			continue;
		}
		// events of one stream complete in order: the last batch of the range is enough
		const uint32_t last = (e - 1 - p.v_begin) / p.batch;
		static const uint32_t bits[kProducts] = {PGH_TALLY_COUNTS, PGH_TALLY_HWE, PGH_TALLY_HWE_MIDP};
		for (int which = 0; which < kProducts; which++) {
			if (products & bits[which]) {
				PGH_HIP(hipEventSynchronize(p.ev[which][last]), "tally wait");
			}
		}
	}
	return PGH_OK;
}

extern "C" const uint32_t (*pgh_tally_counts(const pgh_tally *t))[4] {
	return t ? t->h_counts : nullptr;
}

extern "C" const double *pgh_tally_hwe_lnp(const pgh_tally *t, uint32_t midp) {
	return t ? t->h_lnp[midp ? 1 : 0] : nullptr;
}

extern "C" int pgh_tally_sample_missing(pgh_tally *t, uint32_t *out, char *errbuf) {
	if (!t || !out) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	int rc = pgh_tally_request(t, PGH_TALLY_SAMPLE_MISSING, errbuf);
	if (rc == PGH_OK) {
		rc = pgh_tally_wait(t, PGH_TALLY_SAMPLE_MISSING, t->v_begin, t->v_begin, errbuf);
	}
	if (rc != PGH_OK) {
		return rc;
	}
	std::lock_guard<std::mutex> lock(t->mu);
	if (!t->missing_summed) {
		// the shards' partial tallies meet on the host: N x 4 bytes per shard
		std::vector<uint32_t> raw(t->h_missing, t->h_missing + t->sample_ct);
		for (size_t k = 1; k < t->parts.size(); k++) {
			const uint32_t *src = t->h_missing + k * t->padded;
			for (uint32_t s = 0; s < t->sample_ct; s++) {
				raw[s] += src[s];
			}
		}
		t->missing_out.resize(t->n_out);
		Compact<uint32_t>(t->subset, raw.data(), 1, t->missing_out.data(), t->sample_ct);
		t->missing_summed = true;
	}
	std::memcpy(out, t->missing_out.data(), sizeof(uint32_t) * t->n_out);
	return PGH_OK;
}

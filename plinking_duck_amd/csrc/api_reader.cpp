// api_reader.cpp -- the per-variant reader calls that mirror pgenlib (PgrGetCounts, PgrGet, ...)
// and the exact-test entry points.
#include "api_internal.hpp"

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
	if (ds->IsGroup()) {
		// per-shard readers come into being on the first call that lands in their shard
		rd->parts.assign(ds->shards.size(), nullptr);
		*out = rd.release();
		return PGH_OK;
	}
	PGH_ENTER(ds);
	rd->device = ds->device;
	hipError_t e = hipStreamCreateWithFlags(&rd->stream, hipStreamNonBlocking);
	if (e == hipSuccess) {
		e = PghMalloc(reinterpret_cast<void **>(&rd->d_counts), 16 * pgh_reader::kWindow);
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
	*out = rd.release();
	return PGH_OK;
}

extern "C" void pgh_reader_destroy(pgh_reader *rd) {
	if (!rd) {
		return;
	}
	for (pgh_reader *part : rd->parts) {
		pgh_reader_destroy(part);
	}
	DeviceScope scope(rd->device);
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
	for (int k = 0; k < 2; k++) {
		if (rd->d_unpack[k]) {
			(void)hipFree(rd->d_unpack[k]);
		}
		if (rd->unpack_done[k]) {
			(void)hipEventDestroy(rd->unpack_done[k]);
		}
	}
	if (rd->d_dosage) {
		(void)hipFree(rd->d_dosage);
	}
	if (rd->h_dosage) {
		(void)hipHostFree(rd->h_dosage);
	}
	if (rd->h_phase) {
		(void)hipHostFree(rd->h_phase);
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
	if (rd && rd->ds->IsGroup()) {
		pgh_reader *part = pgh_group::ReaderFor(rd, vidx);
		if (!part) {
			return PGH_ERR_ARG;
		}
		const int rc_part = pgh_get_counts(part, vidx, out);
		if (rc_part != PGH_OK) {
			rd->err = part->err;
		}
		return rc_part;
	}
	PGH_ENTER(rd ? rd->ds : nullptr);
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
	if (rd && rd->ds->IsGroup()) {
		pgh_reader *part = pgh_group::ReaderFor(rd, vidx);
		if (!part) {
			return PGH_ERR_ARG;
		}
		const int rc_part = pgh_get_2bit(part, vidx, genovec);
		if (rc_part != PGH_OK) {
			rd->err = part->err;
		}
		return rc_part;
	}
	PGH_ENTER(rd ? rd->ds : nullptr);
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
	if (rd && rd->ds->IsGroup()) {
		pgh_reader *part = pgh_group::ReaderFor(rd, vidx);
		if (!part) {
			return PGH_ERR_ARG;
		}
		const int rc_part = pgh_get_missingness(part, vidx, bits);
		if (rc_part != PGH_OK) {
			rd->err = part->err;
		}
		return rc_part;
	}
	PGH_ENTER(rd ? rd->ds : nullptr);
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
	if (rd && rd->ds->IsGroup()) {
		pgh_reader *part = pgh_group::ReaderFor(rd, vidx);
		if (!part) {
			return PGH_ERR_ARG;
		}
		const int rc_part = pgh_get_int8(part, vidx, out);
		if (rc_part != PGH_OK) {
			rd->err = part->err;
		}
		return rc_part;
	}
	PGH_ENTER(rd ? rd->ds : nullptr);
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
	if (rd && rd->ds->IsGroup()) {
		pgh_reader *part = pgh_group::ReaderFor(rd, vidx);
		if (!part) {
			return PGH_ERR_ARG;
		}
		const int rc_part = pgh_get_phased(part, vidx, genovec, phasepresent, phaseinfo);
		if (rc_part != PGH_OK) {
			rd->err = part->err;
		}
		return rc_part;
	}
	PGH_ENTER(rd ? rd->ds : nullptr);
	int rc = ReaderCheck(rd, vidx);
	if (rc != PGH_OK) {
		return rc;
	}
	const pgh_dataset *ds = rd->ds;
	const uint32_t n_out = rd->subset ? rd->subset->n_out : ds->sample_ct;
	std::memset(phasepresent, 0, sizeof(uint64_t) * ((n_out + 63) / 64));
	std::memset(phaseinfo, 0, sizeof(uint64_t) * ((n_out + 63) / 64));
	const int32_t pr = ds->ph_rows ? ds->ph_row_of[vidx - ds->v_begin] : -1;
	if (pr < 0) {
		return pgh_get_2bit(rd, vidx, genovec);
	}
	// PgrGetP: the call row and the variant's two resident bit rows, compacted to the included samples
	const uint32_t words = (ds->sample_ct + 63) / 64;
	hipError_t e = hipSuccess;
	if (!rd->h_phase) {
		e = hipHostMalloc(reinterpret_cast<void **>(&rd->h_phase), 16ull * words, hipHostMallocDefault);
	}
	if (e == hipSuccess) {
		e = hipMemcpyAsync(rd->h_phase, ds->d_ph_present + static_cast<uint64_t>(pr) * words, 8ull * words,
		                   hipMemcpyDeviceToHost, rd->stream);
	}
	if (e == hipSuccess) {
		e = hipMemcpyAsync(rd->h_phase + words, ds->d_ph_info + static_cast<uint64_t>(pr) * words, 8ull * words,
		                   hipMemcpyDeviceToHost, rd->stream);
	}
	if (e != hipSuccess) {
		return ReaderFail(rd, PGH_ERR_DEVICE, std::string("phase rows: ") + hipGetErrorString(e));
	}
	rc = FetchRow(rd, vidx); // synchronises the stream
	if (rc != PGH_OK) {
		return rc;
	}
	const uint64_t *pp = rd->h_phase, *pi = rd->h_phase + words;
	std::memset(genovec, 0, sizeof(uint64_t) * ((n_out + 31) / 32));
	ForEachIncluded(rd, [&](uint32_t k, uint32_t s) {
		genovec[k >> 5] |= static_cast<uint64_t>(RowCode(rd->h_row, s)) << (2 * (k & 31));
		if ((pp[s >> 6] >> (s & 63)) & 1ull) {
			phasepresent[k >> 6] |= 1ull << (k & 63);
			if ((pi[s >> 6] >> (s & 63)) & 1ull) {
				phaseinfo[k >> 6] |= 1ull << (k & 63);
			}
		}
	});
	return PGH_OK;
}

extern "C" int pgh_get_dosage_f64(pgh_reader *rd, uint32_t vidx, double *out) {
	if (rd && rd->ds->IsGroup()) {
		pgh_reader *part = pgh_group::ReaderFor(rd, vidx);
		if (!part) {
			return PGH_ERR_ARG;
		}
		const int rc_part = pgh_get_dosage_f64(part, vidx, out);
		if (rc_part != PGH_OK) {
			rd->err = part->err;
		}
		return rc_part;
	}
	PGH_ENTER(rd ? rd->ds : nullptr);
	int rc = ReaderCheck(rd, vidx);
	if (rc != PGH_OK) {
		return rc;
	}
	// PgrGetD + Dosage16ToDoublesMinus9: one row of the dosage unpack kernel through the reader's buffers
	const pgh_dataset *ds = rd->ds;
	const uint32_t n_out = rd->subset ? rd->subset->n_out : ds->sample_ct;
	if (n_out == 0) {
		return PGH_OK;
	}
	hipError_t e = hipSuccess;
	if (!rd->d_dosage) {
		e = PghMalloc(reinterpret_cast<void **>(&rd->d_dosage), sizeof(double) * n_out);
		if (e == hipSuccess) {
			e = hipHostMalloc(reinterpret_cast<void **>(&rd->h_dosage), sizeof(double) * n_out, hipHostMallocDefault);
		}
	}
	if (e == hipSuccess) {
		e = pgh::LaunchDosageUnpack(ds->View(), ds->Dosage(), vidx - ds->v_begin, nullptr, 1,
		                            rd->subset ? rd->subset->d_sel : nullptr, n_out, rd->d_dosage, n_out, rd->stream);
	}
	if (e == hipSuccess) {
		e = hipMemcpyAsync(rd->h_dosage, rd->d_dosage, sizeof(double) * n_out, hipMemcpyDeviceToHost, rd->stream);
	}
	if (e == hipSuccess) {
		e = hipStreamSynchronize(rd->stream);
	}
	if (e != hipSuccess) {
		return ReaderFail(rd, PGH_ERR_DEVICE, std::string("dosage row: ") + hipGetErrorString(e));
	}
	std::memcpy(out, rd->h_dosage, sizeof(double) * n_out);
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
	void *order = nullptr;
	if (const size_t bytes = pgh::HweOrderScratchBytes(n)) {
		PGH_HIP(PghThreadScratch(bytes, static_cast<hipStream_t>(stream), &order), "hipMalloc(hwe order)");
	}
	PGH_HIP(pgh::LaunchHweBatch(static_cast<const uint32_t *>(d_counts), n, midp, static_cast<double *>(d_ln_p),
	                            static_cast<hipStream_t>(stream), order),
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
	DevBuf d_counts, d_lnp, d_order;
	HostSourceFence fence(PghThreadStream()); // the caller's `counts` / `ln_p` are in flight until the stream drains
	PGH_HIP(d_counts.Alloc(16ull * n), "hipMalloc(hwe)");
	PGH_HIP(d_lnp.Alloc(8ull * n), "hipMalloc(hwe)");
	if (const size_t bytes = pgh::HweOrderScratchBytes(n)) {
		PGH_HIP(d_order.Alloc(bytes), "hipMalloc(hwe)");
	}
	PGH_HIP(hipMemcpyAsync(d_counts.p, counts, 16ull * n, hipMemcpyHostToDevice, PghThreadStream()), "hwe upload");
	PGH_HIP(pgh::LaunchHweBatch(d_counts.As<uint32_t>(), n, midp, d_lnp.As<double>(), PghThreadStream(), d_order.p),
	        "hwe kernel");
	PGH_HIP(hipMemcpyAsync(ln_p, d_lnp.p, 8ull * n, hipMemcpyDeviceToHost, PghThreadStream()), "hwe copy");
	PGH_HIP(hipStreamSynchronize(PghThreadStream()), "hwe sync");
	return PGH_OK;
}

extern "C" int pgh_hwe_xchr_lnp_batch(const int32_t (*strata)[5], uint32_t n, uint32_t midp, double *ln_p,
                                      char *errbuf) {
	if (n == 0) {
		return PGH_OK;
	}
	if (!strata || !ln_p) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	DevBuf d_strata, d_lnp;
	HostSourceFence fence(PghThreadStream()); // the caller's `strata` / `ln_p` are in flight until the stream drains
	PGH_HIP(d_strata.Alloc(20ull * n), "hipMalloc(hwe)");
	PGH_HIP(d_lnp.Alloc(8ull * n), "hipMalloc(hwe)");
	PGH_HIP(hipMemcpyAsync(d_strata.p, strata, 20ull * n, hipMemcpyHostToDevice, PghThreadStream()), "hwe upload");
	PGH_HIP(pgh::LaunchHweXchrBatch(d_strata.As<int32_t>(), n, midp, d_lnp.As<double>(), PghThreadStream()),
	        "chrX hwe kernel");
	PGH_HIP(hipMemcpyAsync(ln_p, d_lnp.p, 8ull * n, hipMemcpyDeviceToHost, PghThreadStream()), "hwe copy");
	PGH_HIP(hipStreamSynchronize(PghThreadStream()), "hwe sync");
	return PGH_OK;
}

// ---------------------------------------------------------------------------
// PgrGet over a range, enqueue-and-return (read_pgen's chunk pipeline)
// ---------------------------------------------------------------------------

extern "C" int pgh_reader_unpack_start(pgh_reader *rd, int slot, uint32_t v_begin, uint32_t v_end, int8_t *out,
                                       uint64_t *validity, int missing_code) {
	if (!rd || slot < 0 || slot > 1 || !out) {
		if (rd) {
			rd->err = "bad argument";
		}
		return PGH_ERR_ARG;
	}
	char errbuf[PGH_ERRBUF_LEN] = {0};
	const pgh_dataset *ds = rd->ds;
	rd->unpack_pending[slot] = false;
	if (ds->IsGroup()) {
		// a shard group fills the caller's buffer shard by shard on its own threads: nothing left to wait for
		int rc = pgh_unpack_range(ds, rd->subset, v_begin, v_end, out, validity, missing_code, errbuf);
		rd->err = errbuf;
		return rc;
	}
	int rc = CheckRange(ds, v_begin, v_end, errbuf);
	if (rc != PGH_OK) {
		rd->err = errbuf;
		return rc;
	}
	const uint32_t n_out = rd->subset ? rd->subset->n_out : ds->sample_ct;
	const size_t rows = v_end - v_begin;
	if (rows == 0 || n_out == 0) {
		return PGH_OK;
	}
	DeviceScope scope(rd->device);
	const size_t out_pitch = (static_cast<size_t>(n_out) + 15) / 16 * 16;
	const size_t val_words = (n_out + 63) / 64;
	const size_t out_bytes = (rows * out_pitch + 255) / 256 * 256;
	const size_t need = out_bytes + (validity ? rows * val_words * 8 : 0);
	auto fail = [&](const char *what, hipError_t e) {
		rd->err = std::string(what) + ": " + hipGetErrorString(e);
		return static_cast<int>(PGH_ERR_DEVICE);
	};
	hipError_t e = hipSuccess;
	if (!rd->unpack_done[slot]) {
		e = hipEventCreateWithFlags(&rd->unpack_done[slot], hipEventDisableTiming);
		if (e != hipSuccess) {
			return fail("hipEventCreate(unpack)", e);
		}
	}
	if (rd->unpack_bytes[slot] < need) {
		// (the slot's previous launch has been waited for: the caller consumed that chunk before reusing the slot)
		(void)hipFree(rd->d_unpack[slot]);
		rd->d_unpack[slot] = nullptr;
		rd->unpack_bytes[slot] = 0;
		e = PghMalloc(&rd->d_unpack[slot], need);
		if (e != hipSuccess) {
			return fail("hipMalloc(unpack staging)", e);
		}
		rd->unpack_bytes[slot] = need;
	}
	int8_t *d_out = static_cast<int8_t *>(rd->d_unpack[slot]);
	uint64_t *d_val = validity ? reinterpret_cast<uint64_t *>(static_cast<char *>(rd->d_unpack[slot]) + out_bytes) : nullptr;
	rc = pgh_unpack_range_dev(ds, rd->subset, v_begin, v_end, d_out, out_pitch, d_val, missing_code, rd->stream, errbuf);
	if (rc != PGH_OK) {
		rd->err = errbuf;
		return rc;
	}
	if (out_pitch == n_out) {
		e = hipMemcpyAsync(out, d_out, rows * static_cast<size_t>(n_out), hipMemcpyDeviceToHost, rd->stream);
	} else {
		e = hipMemcpy2DAsync(out, n_out, d_out, out_pitch, n_out, rows, hipMemcpyDeviceToHost, rd->stream);
	}
	if (e == hipSuccess && validity) {
		e = hipMemcpyAsync(validity, d_val, rows * val_words * 8, hipMemcpyDeviceToHost, rd->stream);
	}
	if (e == hipSuccess) {
		e = hipEventRecord(rd->unpack_done[slot], rd->stream);
	}
	if (e != hipSuccess) {
		return fail("unpack copy", e);
	}
	rd->unpack_pending[slot] = true;
	return PGH_OK;
}

extern "C" int pgh_reader_unpack_wait(pgh_reader *rd, int slot) {
	if (!rd || slot < 0 || slot > 1) {
		return PGH_ERR_ARG;
	}
	if (!rd->unpack_pending[slot]) {
		return PGH_OK;
	}
	DeviceScope scope(rd->device);
	const hipError_t e = hipEventSynchronize(rd->unpack_done[slot]);
	rd->unpack_pending[slot] = false;
	if (e != hipSuccess) {
		rd->err = std::string("unpack wait: ") + hipGetErrorString(e);
		return PGH_ERR_DEVICE;
	}
	return PGH_OK;
}

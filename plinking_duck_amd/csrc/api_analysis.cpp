// api_analysis.cpp -- the batched analysis entry points: tallies, unpack, plink_score,
// plink_pca (single GPU and sharded), plink_ld.
#include "api_internal.hpp"

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
	PGH_ONE_DEVICE(ds);
	PGH_ENTER(ds);
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
	if (ds->IsGroup()) {
		return pgh_group::CountsRange(ds, subset, v_begin, v_end, out, errbuf);
	}
	PGH_ENTER(ds);
	DevBuf buf;
	PGH_HIP(buf.Alloc(n * 16), "hipMalloc(counts)");
	rc = pgh_counts_range_dev(ds, subset, v_begin, v_end, buf.p, PghThreadStream(), errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	PGH_HIP(hipMemcpyAsync(out, buf.p, n * 16, hipMemcpyDeviceToHost, PghThreadStream()), "counts copy");
	PGH_HIP(hipStreamSynchronize(PghThreadStream()), "counts sync");
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
	PGH_ONE_DEVICE(ds);
	PGH_ENTER(ds);
	hipStream_t st = static_cast<hipStream_t>(stream);
	// per-slice partial rows: a few MB from this thread's scratch block for `st`, so the call stays enqueue-only
	const size_t scratch_bytes = pgh::MissingPerSampleScratchBytes(ds->record_bytes, v_end - v_begin);
	void *scratch = nullptr;
	PGH_HIP(PghThreadScratch(scratch_bytes, st, &scratch), "missing scratch");
	hipError_t e = pgh::LaunchMissingPerSample(ds->View(), v_begin - ds->v_begin, nullptr, v_end - v_begin, nullptr,
	                                           static_cast<uint32_t *>(scratch), static_cast<uint32_t *>(d_out), st);
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
	PGH_ONE_DEVICE(ds);
	PGH_ENTER(ds);
	hipStream_t st = static_cast<hipStream_t>(stream);
	const size_t scratch_bytes = pgh::MissingPerSampleScratchBytes(ds->record_bytes, v_end - v_begin);
	void *scratch = nullptr;
	PGH_HIP(PghThreadScratch(scratch_bytes, st, &scratch), "fused scratch");
	hipError_t e = pgh::LaunchFusedTally(ds->View(), v_begin - ds->v_begin, v_end - v_begin,
	                                     static_cast<uint32_t *>(scratch), static_cast<uint32_t *>(d_counts),
	                                     static_cast<uint32_t *>(d_missing), st);
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
	if (ds->IsGroup()) {
		return pgh_group::MissingPerSample(ds, subset, v_begin, v_end, out, errbuf);
	}
	PGH_ENTER(ds);
	const uint32_t N = ds->sample_ct;
	const uint32_t padded = (N + 63) / 64 * 64;
	DevBuf buf;
	PGH_HIP(buf.Alloc(sizeof(uint32_t) * padded), "hipMalloc(missing)");
	hipStream_t st = PghThreadStream();
	rc = pgh_missing_per_sample_dev(ds, v_begin, v_end, buf.p, st, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	std::vector<uint32_t> raw(N);
	PGH_HIP(hipMemcpyAsync(raw.data(), buf.p, sizeof(uint32_t) * N, hipMemcpyDeviceToHost, st), "missing copy");
	PGH_HIP(hipStreamSynchronize(st), "missing sync");
	Compact<uint32_t>(subset, raw.data(), 1, out, N);
	return PGH_OK;
}

extern "C" int pgh_sample_counts_dev(const pgh_dataset *ds, uint32_t v_begin, uint32_t v_end, void *d_classes,
                                     void *stream, char *errbuf) {
	int rc = CheckRange(ds, v_begin, v_end, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	if (!d_classes) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	PGH_ONE_DEVICE(ds);
	PGH_ENTER(ds);
	const uint32_t padded = (ds->sample_ct + 63) / 64 * 64;
	hipStream_t st = static_cast<hipStream_t>(stream);
	const size_t scratch_bytes = pgh::ClassCounts3ScratchBytes(ds->record_bytes);
	void *scratch = nullptr;
	PGH_HIP(PghThreadScratch(scratch_bytes, st, &scratch), "sample counts scratch");
	hipError_t e = pgh::LaunchClassCounts3(ds->View(), v_begin - ds->v_begin, nullptr, v_end - v_begin,
	                                       static_cast<uint8_t *>(scratch), static_cast<uint32_t *>(d_classes), padded, st);
	PGH_HIP(e, "sample counts kernel");
	return PGH_OK;
}

extern "C" int pgh_sample_counts(const pgh_dataset *ds, const pgh_subset *subset, uint32_t variant_begin,
                                 uint32_t n_var, const uint32_t *vidx, uint32_t (*counts)[4], char *errbuf) {
	if (!ds || !counts) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	if (ds->IsGroup()) {
		return pgh_group::SampleCounts(ds, subset, variant_begin, n_var, vidx, counts, errbuf);
	}
	PGH_ENTER(ds);
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
	hipStream_t st = PghThreadStream();
	DevBuf d_cls, d_list, d_scratch;
	HostSourceFence fence(st); // `local` feeds an asynchronous upload
	PGH_HIP(d_cls.Alloc(sizeof(uint32_t) * 3ull * padded), "hipMalloc(sample counts)");
	PGH_HIP(d_scratch.Alloc(pgh::ClassCounts3ScratchBytes(ds->record_bytes)), "hipMalloc(sample counts)");
	if (vidx && n_var) {
		PGH_HIP(d_list.Alloc(sizeof(uint32_t) * n_var), "hipMalloc(sample counts)");
		PGH_HIP(hipMemcpyAsync(d_list.p, local.data(), sizeof(uint32_t) * n_var, hipMemcpyHostToDevice, st),
		        "sample counts upload");
	}
	// het, hom-alt and missing column tallies in one pass; hom-ref is what is left
	PGH_HIP(pgh::LaunchClassCounts3(ds->View(), vidx ? 0 : variant_begin - ds->v_begin,
	                                vidx ? d_list.As<uint32_t>() : nullptr, n_var, d_scratch.As<uint8_t>(),
	                                d_cls.As<uint32_t>(), padded, st),
	        "sample counts kernel");
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
	PGH_ONE_DEVICE(ds);
	PGH_ENTER(ds);
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

extern "C" int pgh_probe_unpack_shape_dev(const void *d_src, size_t n_vec, void *d_dst, void *d_val, void *stream,
                                          char *errbuf) {
	if (n_vec && (!d_src || !d_dst || !d_val)) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	PGH_HIP(pgh::LaunchUnpackShapeProbe(d_src, n_vec, d_dst, d_val, static_cast<hipStream_t>(stream)), "store probe");
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
	if (ds->IsGroup()) {
		return pgh_group::UnpackRange(ds, subset, v_begin, v_end, out, validity, missing_code, errbuf);
	}
	PGH_ENTER(ds);
	const size_t out_pitch = (static_cast<size_t>(n_out) + 15) / 16 * 16;
	const size_t val_words = (n_out + 63) / 64;
	// chunk the range so the device staging stays bounded (output is 4x the input); the staging block is the
	// calling thread's kept scratch (a scan thread unpacks chunk after chunk: no allocation per call), and a
	// pinned `out` (pgh_host_alloc) takes the copy at the link's rate
	const size_t max_rows = std::max<size_t>(1, (1024ull << 20) / out_pitch);
	const size_t chunk_rows = std::min(rows, max_rows);
	hipStream_t st = PghThreadStream();
	const size_t out_bytes = out ? (chunk_rows * out_pitch + 255) / 256 * 256 : 0;
	const size_t val_bytes = validity ? chunk_rows * val_words * 8 : 0;
	void *stage = nullptr;
	PGH_HIP(PghThreadScratch(out_bytes + val_bytes, st, &stage), "unpack staging");
	int8_t *d_out = out ? static_cast<int8_t *>(stage) : nullptr;
	uint64_t *d_val = validity ? reinterpret_cast<uint64_t *>(static_cast<char *>(stage) + out_bytes) : nullptr;
	for (size_t r0 = 0; r0 < rows; r0 += chunk_rows) {
		const size_t r1 = std::min(rows, r0 + chunk_rows);
		rc = pgh_unpack_range_dev(ds, subset, v_begin + static_cast<uint32_t>(r0), v_begin + static_cast<uint32_t>(r1),
		                          d_out, out_pitch, d_val, missing_code, st, errbuf);
		if (rc != PGH_OK) {
			return rc;
		}
		if (out) {
			if (out_pitch == n_out) {
				PGH_HIP(hipMemcpyAsync(out + r0 * n_out, d_out, (r1 - r0) * static_cast<size_t>(n_out),
				                       hipMemcpyDeviceToHost, st),
				        "unpack copy");
			} else {
				PGH_HIP(hipMemcpy2DAsync(out + r0 * n_out, n_out, d_out, out_pitch, n_out, r1 - r0, hipMemcpyDeviceToHost,
				                         st),
				        "unpack copy");
			}
		}
		if (validity) {
			PGH_HIP(hipMemcpyAsync(validity + r0 * val_words, d_val, (r1 - r0) * val_words * 8, hipMemcpyDeviceToHost, st),
			        "validity copy");
		}
		PGH_HIP(hipStreamSynchronize(st), "unpack sync");
	}
	return PGH_OK;
}

// ---------------------------------------------------------------------------
// dosage tracks
// ---------------------------------------------------------------------------

//! Local variant indices of [v_begin, v_begin + n) or of vidx[0..n) on the device; NULL when the run is contiguous.
//! `local` is the upload's (pageable) source: the caller keeps it alive behind a HostSourceFence on `st`.
static int UploadVariantList(const pgh_dataset *ds, uint32_t v_begin, uint32_t n, const uint32_t *vidx, DevBuf &d_list,
                             hipStream_t st, std::vector<uint32_t> &local, char *errbuf) {
	if (!vidx) {
		return CheckRange(ds, v_begin, v_begin + n, errbuf);
	}
	local.resize(n);
	for (uint32_t i = 0; i < n; i++) {
		if (vidx[i] < ds->v_begin || vidx[i] >= ds->v_end) {
			SetErr(errbuf, "variant index outside the resident range");
			return PGH_ERR_ARG;
		}
		local[i] = vidx[i] - ds->v_begin;
	}
	PGH_HIP(d_list.Alloc(sizeof(uint32_t) * n), "hipMalloc(variant list)");
	PGH_HIP(hipMemcpyAsync(d_list.p, local.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice, st), "variant list upload");
	return PGH_OK;
}

extern "C" int pgh_dosage_sums_dev(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin, uint32_t v_end,
                                   void *d_sums, void *stream, char *errbuf) {
	int rc = CheckRange(ds, v_begin, v_end, errbuf);
	if (rc == PGH_OK) {
		rc = CheckSubset(ds, subset, errbuf);
	}
	if (rc != PGH_OK) {
		return rc;
	}
	if (!d_sums && v_end > v_begin) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	PGH_ONE_DEVICE(ds);
	PGH_ENTER(ds);
	PGH_HIP(pgh::LaunchDosageSums(ds->View(), ds->Dosage(), v_begin - ds->v_begin, nullptr, v_end - v_begin,
	                              subset ? subset->d_include : nullptr, static_cast<uint64_t *>(d_sums),
	                              static_cast<hipStream_t>(stream)),
	        "dosage sums kernel");
	return PGH_OK;
}

extern "C" int pgh_dosage_sums(const pgh_dataset *ds, const pgh_subset *subset, uint32_t variant_begin,
                               uint32_t n_variants, const uint32_t *vidx, uint64_t (*sums)[3], char *errbuf) {
	if (!ds || (n_variants && !sums)) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	if (ds->IsGroup()) {
		return n_variants ? pgh_group::DosageSums(ds, subset, variant_begin, n_variants, vidx, sums, errbuf) : PGH_OK;
	}
	PGH_ENTER(ds);
	int rc = CheckSubset(ds, subset, errbuf);
	if (rc != PGH_OK || n_variants == 0) {
		return rc;
	}
	hipStream_t st = PghThreadStream();
	DevBuf d_list, d_sums;
	std::vector<uint32_t> local;
	HostSourceFence fence(st); // `local` feeds an asynchronous upload
	rc = UploadVariantList(ds, variant_begin, n_variants, vidx, d_list, st, local, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	PGH_HIP(d_sums.Alloc(24ull * n_variants), "hipMalloc(dosage sums)");
	PGH_HIP(pgh::LaunchDosageSums(ds->View(), ds->Dosage(), vidx ? 0 : variant_begin - ds->v_begin, d_list.As<uint32_t>(),
	                              n_variants, subset ? subset->d_include : nullptr, d_sums.As<uint64_t>(), st),
	        "dosage sums kernel");
	PGH_HIP(hipMemcpyAsync(sums, d_sums.p, 24ull * n_variants, hipMemcpyDeviceToHost, st), "dosage sums copy");
	PGH_HIP(hipStreamSynchronize(st), "dosage sums sync");
	return PGH_OK;
}

extern "C" int pgh_dosage_unpack_dev(const pgh_dataset *ds, const pgh_subset *subset, uint32_t v_begin, uint32_t v_end,
                                     void *d_out, size_t out_stride, void *stream, char *errbuf) {
	int rc = CheckRange(ds, v_begin, v_end, errbuf);
	if (rc == PGH_OK) {
		rc = CheckSubset(ds, subset, errbuf);
	}
	if (rc != PGH_OK) {
		return rc;
	}
	const uint32_t n_out = subset ? subset->n_out : ds->sample_ct;
	if (v_end > v_begin && (!d_out || out_stride < n_out)) {
		SetErr(errbuf, "out_stride must cover the row");
		return PGH_ERR_ARG;
	}
	PGH_ONE_DEVICE(ds);
	PGH_ENTER(ds);
	PGH_HIP(pgh::LaunchDosageUnpack(ds->View(), ds->Dosage(), v_begin - ds->v_begin, nullptr, v_end - v_begin,
	                                subset ? subset->d_sel : nullptr, n_out, static_cast<double *>(d_out), out_stride,
	                                static_cast<hipStream_t>(stream)),
	        "dosage unpack kernel");
	return PGH_OK;
}

extern "C" int pgh_dosage_unpack(const pgh_dataset *ds, const pgh_subset *subset, uint32_t variant_begin,
                                 uint32_t n_variants, const uint32_t *vidx, double *out, char *errbuf) {
	if (!ds || (n_variants && !out)) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	if (ds->IsGroup()) {
		return n_variants ? pgh_group::DosageUnpack(ds, subset, variant_begin, n_variants, vidx, out, errbuf) : PGH_OK;
	}
	PGH_ENTER(ds);
	int rc = CheckSubset(ds, subset, errbuf);
	const uint32_t n_out = subset ? subset->n_out : ds->sample_ct;
	if (rc != PGH_OK || n_variants == 0 || n_out == 0) {
		return rc;
	}
	hipStream_t st = PghThreadStream();
	DevBuf d_list, d_out;
	std::vector<uint32_t> local;
	HostSourceFence fence(st); // `local` feeds an asynchronous upload
	rc = UploadVariantList(ds, variant_begin, n_variants, vidx, d_list, st, local, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	// 8 bytes per sample leave the device: chunk the run so the staging stays bounded
	const uint32_t chunk = static_cast<uint32_t>(std::min<uint64_t>(n_variants, std::max<uint64_t>(1, (512ull << 20) / (8ull * n_out))));
	PGH_HIP(d_out.Alloc(8ull * chunk * n_out), "hipMalloc(dosage unpack)");
	for (uint32_t r0 = 0; r0 < n_variants; r0 += chunk) {
		const uint32_t cnt = std::min(chunk, n_variants - r0);
		PGH_HIP(pgh::LaunchDosageUnpack(ds->View(), ds->Dosage(), vidx ? 0 : variant_begin - ds->v_begin + r0,
		                                vidx ? d_list.As<uint32_t>() + r0 : nullptr, cnt, subset ? subset->d_sel : nullptr,
		                                n_out, d_out.As<double>(), n_out, st),
		        "dosage unpack kernel");
		PGH_HIP(hipMemcpyAsync(out + static_cast<uint64_t>(r0) * n_out, d_out.p, 8ull * cnt * n_out, hipMemcpyDeviceToHost, st),
		        "dosage unpack copy");
		PGH_HIP(hipStreamSynchronize(st), "dosage unpack sync");
	}
	return PGH_OK;
}

// read_pfile orient := 'sample': the variants x samples matrix, sample-major.  The output leaves the device in
// runs of samples so the staging stays bounded whatever the number of variants.
template <class T, class Launch>
static int UnpackSamples(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_variants, const uint32_t *vidx,
                         T *out, char *errbuf, Launch launch) {
	if (!ds || (n_variants && (!vidx || !out))) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	PGH_ENTER(ds);
	int rc = CheckSubset(ds, subset, errbuf);
	const uint32_t n_out = subset ? subset->n_out : ds->sample_ct;
	if (rc != PGH_OK || n_variants == 0 || n_out == 0) {
		return rc;
	}
	hipStream_t st = PghThreadStream();
	DevBuf d_list, d_out;
	std::vector<uint32_t> local;
	HostSourceFence fence(st); // `local` feeds an asynchronous upload
	rc = UploadVariantList(ds, 0, n_variants, vidx, d_list, st, local, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	const uint64_t row_bytes = sizeof(T) * static_cast<uint64_t>(n_variants);
	uint32_t chunk = static_cast<uint32_t>(std::min<uint64_t>(n_out, std::max<uint64_t>(64, (512ull << 20) / row_bytes)));
	chunk = (chunk + 63) / 64 * 64;
	PGH_HIP(d_out.Alloc(row_bytes * chunk), "hipMalloc(sample-major unpack)");
	for (uint32_t k0 = 0; k0 < n_out; k0 += chunk) {
		const uint32_t cnt = std::min(chunk, n_out - k0);
		PGH_HIP(launch(d_list.As<uint32_t>(), k0, cnt, d_out.As<T>(), st), "sample-major unpack kernel");
		PGH_HIP(hipMemcpyAsync(out + static_cast<uint64_t>(k0) * n_variants, d_out.p, row_bytes * cnt, hipMemcpyDeviceToHost, st),
		        "sample-major unpack copy");
		PGH_HIP(hipStreamSynchronize(st), "sample-major unpack sync");
	}
	return PGH_OK;
}

extern "C" int pgh_unpack_samples(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_variants,
                                  const uint32_t *vidx, int8_t *out, int missing_code, char *errbuf) {
	if (ds && ds->IsGroup()) {
		return n_variants ? pgh_group::UnpackSamples(ds, subset, n_variants, vidx, out, missing_code, errbuf) : PGH_OK;
	}
	return UnpackSamples<int8_t>(ds, subset, n_variants, vidx, out, errbuf,
	                             [&](const uint32_t *d_list, uint32_t k0, uint32_t cnt, int8_t *d_out, hipStream_t st) {
		                             return pgh::LaunchUnpackTransposed(ds->View(), d_list, n_variants,
		                                                                subset ? subset->d_sel : nullptr, k0, cnt, d_out,
		                                                                n_variants, static_cast<int8_t>(missing_code), st);
	                             });
}

extern "C" int pgh_dosage_unpack_samples(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_variants,
                                         const uint32_t *vidx, double *out, char *errbuf) {
	if (ds && ds->IsGroup()) {
		return n_variants ? pgh_group::DosageUnpackSamples(ds, subset, n_variants, vidx, out, errbuf) : PGH_OK;
	}
	return UnpackSamples<double>(ds, subset, n_variants, vidx, out, errbuf,
	                             [&](const uint32_t *d_list, uint32_t k0, uint32_t cnt, double *d_out, hipStream_t st) {
		                             return pgh::LaunchDosageUnpackTransposed(ds->View(), ds->Dosage(), d_list, n_variants,
		                                                                      subset ? subset->d_sel : nullptr, k0, cnt,
		                                                                      d_out, n_variants, st);
	                             });
}

// ---------------------------------------------------------------------------
// plink_score
// ---------------------------------------------------------------------------

// one prepared pass of the int8 contraction (score_i8.hpp): up to kI8MaxCols weight columns
struct ScoreI8Pass {
	uint32_t c0 = 0, n_cols = 0;
	pgh::ScoreI8Buffers buf;
	void *d_bmat = nullptr, *d_rowidx = nullptr, *d_cols = nullptr, *d_small = nullptr;
};

struct pgh_score_plan {
	const pgh_dataset *ds = nullptr;
	std::vector<ScoreI8Pass> i8; // the table-scored variants (hardcalls + sparse dosage tracks), cut into digits
	uint32_t n_table = 0;        // ... their number: the first n_table entries of d_vlist
	bool two_step = true;        // sparse dosage tracks ride the hardcall kernel + an explicit-entry kernel:
	bool records = false;        // ... k_score_dosage_records over the dataset's entry records, else k_score_dosage_fix
	uint32_t n_scored = 0, n_cols = 0;
	uint32_t n_hard = 0; // the first n_hard entries have hardcalls only; the rest carry dosage tracks:
	uint32_t n_gaps = 0; // ... then n_gaps whose tracks cover most samples (scored by the sample-owning kernel)
	uint32_t n_full = 0; // ... and the LAST n_full with an explicit dosage for every sample
	int mode = 0;
	void *d_vlist = nullptr, *d_weights = nullptr, *d_flip = nullptr, *d_counts = nullptr, *d_ts = nullptr,
	     *d_td = nullptr, *d_ac = nullptr, *d_lin = nullptr;
	void *d_special = nullptr; // non-finite weights (pgh::ScoreSpecial), scored apart in plain double arithmetic
	uint32_t n_special = 0;
	void *d_tiles = nullptr;   // tile-major copy of the table-scored rows (kept plans with many columns)
	// A weight column whose every coefficient is exactly 1.0 (-1: none).  w * scored is then `scored` itself, so in the
	// reference SCORE_SUM and NAMED_ALLELE_DOSAGE_SUM of such a column are the same doubles added in the same order
	// (src/plink_score.cpp:621-651; streaming_threading.test:226-233 asserts the equality).  Here the two are separate
	// columns of the contraction, equal as real numbers but cut into digits and added up apart: the run copies the
	// column into the dosage sum instead of leaving the last bits to the order of the atomic additions.
	int unit_col = -1;
};

// The entry records of every sparse dosage track of the dataset (dosage.hpp), built once -- by the first plan that
// scores such a track -- and resident until the dataset is closed: 4 bytes per explicit dosage, in the order and
// form k_score_dosage_records streams them.  "Sparse" is the plan's own cut (fewer than 40 % of the samples
// explicit).  Not fitting in HBM is not an error: the plans then keep to the bit-walking k_score_dosage_fix.
static int EnsureDosageRecords(const pgh_dataset *ds, hipStream_t st, char *errbuf) {
	std::lock_guard<std::mutex> lock(ds->dos_rec_mutex);
	if (ds->dos_rec_state != 0) {
		return PGH_OK;
	}
	const uint32_t rows = ds->dos_rows;
	std::vector<uint64_t> off(rows + 1, 0);
	for (uint32_t r = 0; r < rows; r++) {
		const uint64_t have = ds->dos_row_count[r];
		const bool sparse = have != ds->sample_ct && have * 5 < static_cast<uint64_t>(ds->sample_ct) * 2;
		off[r + 1] = off[r] + (sparse ? have : 0);
	}
	ds->dos_rec_state = -1;
	if (off[rows] == 0) {
		return PGH_OK;
	}
	std::vector<uint32_t> row_variant(rows, 0);
	for (size_t v = 0; v < ds->dos_row_of.size(); v++) {
		if (ds->dos_row_of[v] >= 0) {
			row_variant[static_cast<uint32_t>(ds->dos_row_of[v])] = static_cast<uint32_t>(v);
		}
	}
	uint32_t *d_rec = nullptr, *d_row_variant = nullptr;
	uint64_t *d_off = nullptr;
	if (PghMalloc(reinterpret_cast<void **>(&d_rec), 4ull * off[rows] + 64) != hipSuccess) {
		(void)hipGetLastError(); // no room: not an error
		return PGH_OK;
	}
	hipError_t e = PghMalloc(reinterpret_cast<void **>(&d_off), 8ull * (rows + 1));
	if (e == hipSuccess) {
		e = PghMalloc(reinterpret_cast<void **>(&d_row_variant), 4ull * rows);
	}
	if (e == hipSuccess) {
		e = hipMemcpyAsync(d_off, off.data(), 8ull * (rows + 1), hipMemcpyHostToDevice, st);
	}
	if (e == hipSuccess) {
		e = hipMemcpyAsync(d_row_variant, row_variant.data(), 4ull * rows, hipMemcpyHostToDevice, st);
	}
	if (e == hipSuccess) {
		e = pgh::LaunchDosageRecords(ds->View(), ds->Dosage(), rows, d_row_variant, d_off, d_rec, st);
	}
	if (e == hipSuccess) {
		e = hipStreamSynchronize(st); // off / row_variant die with this frame
	}
	(void)hipFree(d_row_variant);
	if (e != hipSuccess) {
		(void)hipFree(d_rec);
		(void)hipFree(d_off);
		return DeviceFail(errbuf, "dosage entry records", e);
	}
	ds->d_dos_rec = d_rec;
	ds->d_dos_rec_off = d_off;
	ds->dos_rec_ct = off[rows];
	ds->dos_rec_state = 1;
	return PGH_OK;
}

extern "C" void pgh_score_plan_destroy(pgh_score_plan *plan) {
	if (!plan) {
		return;
	}
	for (void *p : {plan->d_vlist, plan->d_weights, plan->d_flip, plan->d_counts, plan->d_ts, plan->d_td, plan->d_ac,
	                plan->d_lin, plan->d_special, plan->d_tiles}) {
		if (p) {
			(void)hipFree(p);
		}
	}
	for (ScoreI8Pass &pass : plan->i8) {
		for (void *p : {pass.d_bmat, pass.d_rowidx, pass.d_cols, pass.d_small}) {
			if (p) {
				(void)hipFree(p);
			}
		}
	}
	delete plan;
}

//! pgh_score_plan_create; `counts` (optional, host): the scored variants' class tallies over the included samples,
//! counts[i] for vidx[i] -- what a tally pass already holds -- in which case the plan does not read the rows for them.
//! keep_tiles: the plan will be run more than once (pgh_score_plan_create): with many weight columns it keeps a
//! tile-major copy of the scored rows (score_i8.hpp) when HBM has room -- a pass over the rows to build, contiguous
//! genotype DMA in every run.  A one-shot call (pgh_score) would pay more for the copy than it saves.
static int ScorePlanCreate(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_scored, const uint32_t *vidx,
                           const double *weights, const uint8_t *flip, uint32_t n_cols, int mode,
                           const uint32_t (*counts)[4], bool keep_tiles, pgh_score_plan **out, char *errbuf) {
	if (!ds || !out || (n_scored && (!vidx || !weights))) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	*out = nullptr;
	PGH_ONE_DEVICE(ds);
	PGH_ENTER(ds);
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
	// Variants that carry dosage tracks are scored by their own kernel (dosage.hip); they go to the back
	// of the list so each kernel sees one contiguous run.  Per-sample sums do not depend on the order.
	std::vector<uint32_t> order(n_scored);
	uint32_t n_hard = n_scored, n_gaps = 0, n_full = 0;
	if (ds->dos_rows) {
		// [hardcalls only][sparse tracks][tracks covering most samples][tracks covering every sample].  The
		// explicit-entry kernel costs ~2.2 ps per entry, the sample-owning one ~0.5 us per variant at 500 k
		// samples: they meet where ~40 % of the samples are explicit.
		n_hard = 0;
		auto kind = [&](uint32_t i) { // 0 hardcalls, 1 sparse, 2 mostly explicit, 3 fully explicit
			const int32_t r = ds->dos_row_of[local[i]];
			if (r < 0) {
				return 0;
			}
			const uint64_t have = ds->dos_row_count[static_cast<uint32_t>(r)];
			return have == ds->sample_ct ? 3 : (have * 5 >= static_cast<uint64_t>(ds->sample_ct) * 2 ? 2 : 1);
		};
		uint32_t at = 0;
		for (int want = 0; want < 4; want++) {
			for (uint32_t i = 0; i < n_scored; i++) {
				if (kind(i) == want) {
					order[at++] = i;
					n_hard += want == 0;
					n_gaps += want == 2;
					n_full += want == 3;
				}
			}
		}
	} else {
		for (uint32_t i = 0; i < n_scored; i++) {
			order[i] = i;
		}
	}
	// the plan's order of the list; a file without dosage tracks keeps the caller's (no copies: a million-variant
	// list is several milliseconds of host loops per array)
	const bool reorder = ds->dos_rows != 0;
	std::vector<uint32_t> p_local;
	std::vector<double> p_weights;
	std::vector<uint8_t> p_flip;
	std::vector<uint32_t> p_counts;
	if (reorder) {
		p_local.resize(n_scored);
		p_weights.resize(static_cast<size_t>(n_scored) * n_cols);
		p_flip.resize(flip ? n_scored : 0);
		p_counts.resize(counts ? 4ull * n_hard : 0);
		for (uint32_t k = 0; counts && k < n_hard; k++) {
			std::memcpy(&p_counts[4ull * k], counts[order[k]], 16);
		}
		for (uint32_t k = 0; k < n_scored; k++) {
			p_local[k] = local[order[k]];
			std::memcpy(&p_weights[static_cast<size_t>(k) * n_cols], weights + static_cast<size_t>(order[k]) * n_cols,
			            sizeof(double) * n_cols);
			if (flip) {
				p_flip[k] = flip[order[k]];
			}
		}
	}
	const uint32_t *up_local = reorder ? p_local.data() : local.data();
	const double *up_weights = reorder ? p_weights.data() : weights;
	// Non-finite weights: the fixed-point digits of the matrix-core contraction cannot hold them (llrint of a NaN or
	// an infinity is undefined: round 2 returned finite garbage for such a column).  They are zeroed for the
	// contraction and scored apart, term by term in double arithmetic as the reference does
	// (src/plink_score.cpp:621-651): NaN / +-Inf come out where the reference's sums have them.
	std::vector<pgh::ScoreSpecial> special;
	{
		double probe = 0.0;
		const size_t total = static_cast<size_t>(n_scored) * n_cols;
		for (size_t j = 0; j < total; j++) {
			probe += up_weights[j] * 0.0; // NaN as soon as one weight is not finite
		}
		if (!(probe == 0.0)) {
			if (!reorder) {
				p_weights.assign(weights, weights + total);
			}
			for (uint32_t k = 0; k < n_scored; k++) {
				for (uint32_t c = 0; c < n_cols; c++) {
					double &w = p_weights[static_cast<size_t>(k) * n_cols + c];
					if (!std::isfinite(w)) {
						if (k >= n_hard) {
							SetErr(errbuf, "non-finite weight on a variant that carries a dosage track");
							return PGH_ERR_ARG;
						}
						special.push_back(pgh::ScoreSpecial {k, c, w});
						w = 0.0;
					}
				}
			}
			up_weights = p_weights.data();
		}
	}
	const uint8_t *up_flip = reorder ? p_flip.data() : flip;
	const uint32_t *up_counts = counts ? (reorder ? p_counts.data() : &counts[0][0]) : nullptr;
	std::unique_ptr<pgh_score_plan, void (*)(pgh_score_plan *)> plan(new pgh_score_plan(), pgh_score_plan_destroy);
	plan->ds = ds;
	plan->n_scored = n_scored;
	plan->n_hard = n_hard;
	plan->n_full = n_full;
	plan->n_gaps = n_gaps;
	plan->n_cols = n_cols;
	plan->mode = mode;
	for (uint32_t c = 0; c < n_cols && plan->unit_col < 0 && n_scored; c++) {
		bool unit = true;
		for (uint32_t k = 0; k < n_scored && unit; k++) {
			unit = up_weights[static_cast<size_t>(k) * n_cols + c] == 1.0;
		}
		if (unit) {
			plan->unit_col = static_cast<int>(c);
		}
	}
	if (n_scored) {
		const uint32_t N = ds->sample_ct;
		const uint32_t n_dos = n_scored - n_hard;
		hipStream_t st = PghThreadStream();
		HostSourceFence fence(st); // p_local / p_weights / p_flip feed asynchronous uploads
		PGH_HIP(PghMalloc(&plan->d_vlist, sizeof(uint32_t) * n_scored), "hipMalloc(score)");
		PGH_HIP(PghMalloc(&plan->d_weights, sizeof(double) * n_scored * n_cols), "hipMalloc(score)");
		PGH_HIP(PghMalloc(&plan->d_counts, 24ull * n_scored), "hipMalloc(score)"); // counts[4] u32, or dosage sums[3] u64
		PGH_HIP(PghMalloc(&plan->d_ts, 32ull * n_scored), "hipMalloc(score)");
		PGH_HIP(PghMalloc(&plan->d_td, 32ull * n_scored), "hipMalloc(score)");
		PGH_HIP(PghMalloc(&plan->d_ac, 4ull * n_scored), "hipMalloc(score)");
		PGH_HIP(hipMemcpyAsync(plan->d_vlist, up_local, sizeof(uint32_t) * n_scored, hipMemcpyHostToDevice, st),
		        "score upload");
		PGH_HIP(hipMemcpyAsync(plan->d_weights, up_weights, sizeof(double) * n_scored * n_cols, hipMemcpyHostToDevice, st),
		        "score upload");
		if (flip) {
			PGH_HIP(PghMalloc(&plan->d_flip, n_scored), "hipMalloc(score)");
			PGH_HIP(hipMemcpyAsync(plan->d_flip, up_flip, n_scored, hipMemcpyHostToDevice, st), "score upload");
		}
		// per-variant statistics and contribution tables depend on the data only: once per plan
		uint32_t *vlist = static_cast<uint32_t *>(plan->d_vlist);
		uint8_t *d_flip = static_cast<uint8_t *>(plan->d_flip);
		if (n_hard) {
			if (counts) {
				PGH_HIP(hipMemcpyAsync(plan->d_counts, up_counts, 16ull * n_hard, hipMemcpyHostToDevice, st), "score upload");
			} else {
				PGH_HIP(pgh::LaunchCounts(ds->View(), 0, vlist, n_hard, subset ? subset->d_mask2 : nullptr,
				                          subset ? subset->n_out : N, static_cast<uint32_t *>(plan->d_counts), st),
				        "score counts kernel");
			}
			PGH_HIP(pgh::LaunchScoreTables(static_cast<uint32_t *>(plan->d_counts), d_flip, n_hard, mode,
			                               static_cast<double *>(plan->d_ts), static_cast<double *>(plan->d_td),
			                               static_cast<uint32_t *>(plan->d_ac), st),
			        "score table kernel");
		}
		if (n_dos) {
			PGH_HIP(PghMalloc(&plan->d_lin, 32ull * n_dos), "hipMalloc(score)");
			uint64_t *sums = reinterpret_cast<uint64_t *>(static_cast<uint8_t *>(plan->d_counts) + 16ull * n_hard);
			PGH_HIP(pgh::LaunchDosageSums(ds->View(), ds->Dosage(), 0, vlist + n_hard, n_dos,
			                              subset ? subset->d_include : nullptr, sums, st),
			        "dosage sums kernel");
			PGH_HIP(pgh::LaunchScoreTablesDosage(sums, d_flip ? d_flip + n_hard : nullptr, n_dos, mode,
			                                     static_cast<double *>(plan->d_ts) + 4ull * n_hard,
			                                     static_cast<double *>(plan->d_td) + 4ull * n_hard,
			                                     static_cast<double *>(plan->d_lin),
			                                     static_cast<uint32_t *>(plan->d_ac) + n_hard, st),
			        "score table kernel");
		}
		// The contraction itself runs on the int8 matrix cores: cut the coefficients of the table-scored
		// variants into digits once (score_i8.hpp).  PGH_SCORE_DOSAGE_LANES=1 keeps the sample-owning
		// k_score_dosage for every track with gaps (a cross-check): those variants then leave the tables.
		const char *lanes_env = std::getenv("PGH_SCORE_DOSAGE_LANES");
		plan->two_step = !(lanes_env && *lanes_env && *lanes_env != '0');
		const uint32_t n_sparse = n_scored - n_hard - n_gaps - n_full;
		plan->n_table = n_hard + (plan->two_step ? n_sparse : 0u);
		if (plan->two_step && n_sparse) {
			// PGH_SCORE_DOSAGE_RECORDS=0 keeps the bit-walking explicit-entry kernel (a cross-check, and what a
			// dataset whose records do not fit in HBM gets)
			const char *rec_env = std::getenv("PGH_SCORE_DOSAGE_RECORDS");
			if (!(rec_env && *rec_env == '0')) {
				rc = EnsureDosageRecords(ds, st, errbuf);
				if (rc != PGH_OK) {
					return rc;
				}
				plan->records = ds->dos_rec_state == 1;
			}
		}
		if (plan->n_table) {
			const char *tiles_env = std::getenv("PGH_SCORE_TILES"); // A/B switch
			if (keep_tiles && !(tiles_env && *tiles_env == '0') &&
			    pgh::ScoreI8UsesTiles(std::min(pgh::kI8MaxCols, n_cols), true)) {
				const size_t free_b = PghDeviceFreeBytes();
				const size_t need = pgh::ScoreI8TiledBytes(plan->n_table, N);
				if (free_b > need + (8ull << 30)) {
					PGH_HIP(PghMalloc(&plan->d_tiles, need), "hipMalloc(score tiles)");
					PGH_HIP(pgh::LaunchScoreI8TileMajor(ds->View(), vlist, plan->n_table, static_cast<uint8_t *>(plan->d_tiles), st),
					        "score tiles");
				}
			}
			for (uint32_t c0 = 0; c0 < n_cols; c0 += pgh::kI8MaxCols) {
				plan->i8.emplace_back();
				ScoreI8Pass &pass = plan->i8.back();
				pass.c0 = c0;
				pass.n_cols = std::min(pgh::kI8MaxCols, n_cols - c0);
				const pgh::ScoreI8Sizes z = pgh::ScoreI8Bytes(plan->n_table, pass.n_cols, true);
				PGH_HIP(PghMalloc(&pass.d_bmat, z.bmat), "hipMalloc(score digits)");
				PGH_HIP(PghMalloc(&pass.d_rowidx, z.rowidx), "hipMalloc(score digits)");
				PGH_HIP(PghMalloc(&pass.d_cols, z.cols), "hipMalloc(score digits)");
				PGH_HIP(PghMalloc(&pass.d_small, z.small), "hipMalloc(score digits)");
				pass.buf.bmat = static_cast<int8_t *>(pass.d_bmat);
				pass.buf.rowidx = static_cast<uint32_t *>(pass.d_rowidx);
				pass.buf.mult = static_cast<double *>(pass.d_cols);
				pass.buf.target = reinterpret_cast<uint32_t *>(pass.buf.mult + 16ull * z.n_tiles16);
				pass.buf.colmax = static_cast<unsigned long long *>(pass.d_small);
				pass.buf.k0 = reinterpret_cast<double *>(pass.buf.colmax + (pass.n_cols + 2));
				pass.buf.scale_exp = pass.buf.k0 + (pass.n_cols + 2);
				pass.buf.tiled = static_cast<const uint8_t *>(plan->d_tiles);
				PGH_HIP(pgh::LaunchScoreI8Prepare(vlist, plan->n_table, static_cast<double *>(plan->d_weights) + c0, n_cols,
				                                  pass.n_cols, static_cast<double *>(plan->d_ts),
				                                  static_cast<double *>(plan->d_td), static_cast<uint32_t *>(plan->d_ac),
				                                  mode != PGH_SCORE_MEAN_IMPUTE, true, pgh::kI8Tables, pass.buf, st),
				        "score digit kernels");
			}
		}
		if (!special.empty()) {
			plan->n_special = static_cast<uint32_t>(special.size());
			PGH_HIP(PghMalloc(&plan->d_special, sizeof(pgh::ScoreSpecial) * special.size()), "hipMalloc(score)");
			PGH_HIP(hipMemcpyAsync(plan->d_special, special.data(), sizeof(pgh::ScoreSpecial) * special.size(),
			                       hipMemcpyHostToDevice, st),
			        "score upload");
		}
		PGH_HIP(hipStreamSynchronize(st), "score plan sync"); // host staging vectors die with this frame
	}
	*out = plan.release();
	return PGH_OK;
}

extern "C" int pgh_score_plan_create(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_scored,
                                     const uint32_t *vidx, const double *weights, const uint8_t *flip, uint32_t n_cols,
                                     int mode, pgh_score_plan **out, char *errbuf) {
	return ScorePlanCreate(ds, subset, n_scored, vidx, weights, flip, n_cols, mode, nullptr, true, out, errbuf);
}

extern "C" int pgh_score_run_dev(const pgh_score_plan *plan, void *d_score_sum, void *d_dosage_sum, void *d_allele_ct,
                                 void *stream, char *errbuf) {
	if (!plan || !d_score_sum || !d_allele_ct) { // d_dosage_sum == NULL: the dosage sum is not wanted
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	const pgh_dataset *ds = plan->ds;
	PGH_ENTER(ds);
	const uint32_t N = ds->sample_ct;
	hipStream_t st = static_cast<hipStream_t>(stream);
	PGH_HIP(hipMemsetAsync(d_score_sum, 0, sizeof(double) * N * plan->n_cols, st), "score memset");
	if (d_dosage_sum) {
		PGH_HIP(hipMemsetAsync(d_dosage_sum, 0, sizeof(double) * N, st), "score memset");
	}
	if (plan->n_scored == 0) {
		PGH_HIP(hipMemsetAsync(d_allele_ct, 0, sizeof(uint32_t) * N, st), "score memset");
		return PGH_OK;
	}
	uint32_t *vlist = static_cast<uint32_t *>(plan->d_vlist);
	double *weights = static_cast<double *>(plan->d_weights);
	double *ts = static_cast<double *>(plan->d_ts);
	uint32_t *ac = static_cast<uint32_t *>(plan->d_ac);
	// dosage-bearing variants: n_dos sparse tracks, n_gaps tracks covering most samples, n_full covering all
	const uint32_t n_hard = plan->n_hard, n_full = plan->n_full;
	uint32_t n_gaps = plan->n_gaps, n_dos = plan->n_scored - plan->n_hard - plan->n_full - plan->n_gaps;
	// Dosage-bearing variants with sparse tracks ride the hardcall contraction with their dosage-mean tables
	// (every sample's term is ts[call], plus (affine(dosage) - ts[call]) where it has an explicit dosage), and
	// k_score_dosage_fix adds what the explicit entries change, one weight column per launch.
	if (!plan->two_step) {
		n_gaps += n_dos; // the cross-check: every track with gaps through the sample-owning kernel
		n_dos = 0;
	}
	const uint32_t n_table = n_hard + n_dos;
	const bool track = plan->mode != PGH_SCORE_CENTER && d_dosage_sum != nullptr;
	// ALLELE_CT is integer bookkeeping: 2 per scored, non-skipped variant, minus 2 per such
	// variant at which the sample is missing unless missing calls are mean-imputed
	// (src/plink_score.cpp:632-651).  Under a dosage track "missing" means no dosage and no call.  The
	// missing calls at the table-scored variants are one more digit column of the contraction.
	const bool count_missing = plan->mode != PGH_SCORE_MEAN_IMPUTE;
	void *miss = nullptr;
	hipError_t e = hipSuccess;
	if (count_missing) {
		const size_t miss_bytes = sizeof(uint32_t) * ((N + 63) / 64 * 64);
		PGH_HIP(PghThreadScratch(miss_bytes, st, &miss), "score scratch");
		e = hipMemsetAsync(miss, 0, miss_bytes, st);
	}
	// (The explicit-entry kernel is bound by LDS atomics and the contraction by HBM and the matrix cores, but side
	// by side on two streams they gained 1.5 %: each fills the CUs' LDS by itself, so the second only gets the CUs
	// the first drains.  They run one after the other.)
	if (e == hipSuccess && n_dos) {
		for (uint32_t c = 0; c < plan->n_cols && e == hipSuccess; c++) {
			e = (plan->records ? pgh::LaunchScoreDosageRecords : pgh::LaunchScoreDosageFix)(
			    ds->View(), ds->Dosage(), vlist + n_hard, n_dos,
			    weights + static_cast<uint64_t>(n_hard) * plan->n_cols + c, plan->n_cols, ts + 4ull * n_hard,
			    static_cast<double *>(plan->d_lin), ac + n_hard, static_cast<double *>(d_score_sum) + c, plan->n_cols,
			    (track && c == 0) ? static_cast<double *>(d_dosage_sum) : nullptr,
			    c == 0 ? static_cast<uint32_t *>(miss) : nullptr, st);
		}
	}
	for (const ScoreI8Pass &pass : plan->i8) {
		if (e != hipSuccess) {
			break;
		}
		const bool first = pass.c0 == 0;
		e = pgh::LaunchScoreI8(ds->View(), n_table, pass.n_cols, true, pgh::kI8Tables, pass.buf,
		                       static_cast<double *>(d_score_sum) + pass.c0,
		                       plan->n_cols, (first && track) ? static_cast<double *>(d_dosage_sum) : nullptr,
		                       first ? static_cast<uint32_t *>(miss) : nullptr, st);
	}
	if (e == hipSuccess && n_gaps) {
		const uint32_t at = n_hard + n_dos;
		e = pgh::LaunchScoreDosage(ds->View(), ds->Dosage(), vlist + at, n_gaps,
		                           weights + static_cast<uint64_t>(at) * plan->n_cols, plan->n_cols, plan->n_cols,
		                           ts + 4ull * at, static_cast<double *>(plan->d_lin) + 4ull * n_dos, ac + at, plan->mode,
		                           static_cast<double *>(d_score_sum), plan->n_cols,
		                           track ? static_cast<double *>(d_dosage_sum) : nullptr, static_cast<uint32_t *>(miss), st);
	}
	if (e == hipSuccess && n_full) {
		// every sample has a value at these variants: its affine map is the whole term, and nobody is missing
		const uint32_t at = n_hard + n_dos + n_gaps;
		e = pgh::LaunchScoreDosageFull(ds->View(), ds->Dosage(), vlist + at, n_full,
		                               weights + static_cast<uint64_t>(at) * plan->n_cols, plan->n_cols, plan->n_cols,
		                               static_cast<double *>(plan->d_lin) + 4ull * (n_dos + n_gaps), ac + at, plan->mode,
		                               static_cast<double *>(d_score_sum), plan->n_cols,
		                               track ? static_cast<double *>(d_dosage_sum) : nullptr, st);
	}
	if (e == hipSuccess && plan->n_special) {
		e = pgh::LaunchScoreNonFinite(ds->View(), vlist, ts, ac, static_cast<const pgh::ScoreSpecial *>(plan->d_special),
		                              plan->n_special, plan->n_cols, static_cast<double *>(d_score_sum), st);
	}
	if (e == hipSuccess) {
		e = pgh::LaunchAlleleCt(ac, plan->n_scored, static_cast<uint32_t *>(miss), N,
		                        static_cast<uint32_t *>(d_allele_ct), st);
	}
	if (e == hipSuccess && track && plan->unit_col >= 0) { // (pgh_score_plan::unit_col)
		e = pgh::LaunchCopyCols(static_cast<const double *>(d_score_sum) + plan->unit_col, plan->n_cols,
		                        static_cast<double *>(d_dosage_sum), 1, 1, N, st);
	}
	PGH_HIP(e, "score kernels");
	return PGH_OK;
}

int PghScoreDevCounts(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_scored, const uint32_t *vidx,
                      const double *weights, const uint8_t *flip, uint32_t n_cols, int mode, const uint32_t (*counts)[4],
                      void *d_score_sum, void *d_dosage_sum, void *d_allele_ct, void *stream, char *errbuf) {
	PGH_ONE_DEVICE(ds);
	PGH_ENTER(ds);
	pgh_score_plan *plan = nullptr;
	int rc = ScorePlanCreate(ds, subset, n_scored, vidx, weights, flip, n_cols, mode, counts, false, &plan, errbuf);
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

extern "C" int pgh_score_dev(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_scored, const uint32_t *vidx,
                             const double *weights, const uint8_t *flip, uint32_t n_cols, int mode, void *d_score_sum,
                             void *d_dosage_sum, void *d_allele_ct, void *stream, char *errbuf) {
	return PghScoreDevCounts(ds, subset, n_scored, vidx, weights, flip, n_cols, mode, nullptr, d_score_sum, d_dosage_sum,
	                         d_allele_ct, stream, errbuf);
}

extern "C" int pgh_score(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_scored, const uint32_t *vidx,
                         const double *weights, const uint8_t *flip, uint32_t n_cols, int mode, double *score_sum,
                         double *dosage_sum, uint32_t *allele_ct, char *errbuf) {
	return pgh_score_counts(ds, subset, n_scored, vidx, weights, flip, n_cols, mode, nullptr, score_sum, dosage_sum,
	                        allele_ct, errbuf);
}

extern "C" int pgh_score_counts(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_scored, const uint32_t *vidx,
                                const double *weights, const uint8_t *flip, uint32_t n_cols, int mode,
                                const uint32_t (*counts)[4], double *score_sum, double *dosage_sum, uint32_t *allele_ct,
                                char *errbuf) {
	if (!ds) {
		SetErr(errbuf, "null dataset");
		return PGH_ERR_ARG;
	}
	if (ds->IsGroup()) {
		return pgh_group::Score(ds, subset, n_scored, vidx, weights, flip, n_cols, mode, counts, score_sum, dosage_sum,
		                        allele_ct, errbuf);
	}
	PGH_ENTER(ds);
	// PGH_SCORE_TIMING=1: wall-clock of the call's phases on stderr
	static const bool timing = [] {
		const char *e = std::getenv("PGH_SCORE_TIMING");
		return e && *e && *e != '0';
	}();
	auto t_last = std::chrono::steady_clock::now();
	auto mark = [&](const char *what) {
		if (timing) {
			const auto now = std::chrono::steady_clock::now();
			std::fprintf(stderr, "pgh_score: %-22s %8.2f ms\n", what,
			             std::chrono::duration<double, std::milli>(now - t_last).count());
			t_last = now;
		}
	};
	const uint32_t N = ds->sample_ct;
	DevBuf d_score, d_dos, d_ac;
	PGH_HIP(d_score.Alloc(sizeof(double) * N * std::max<uint32_t>(1, n_cols)), "hipMalloc(score out)");
	if (dosage_sum) {
		PGH_HIP(d_dos.Alloc(sizeof(double) * N), "hipMalloc(score out)");
	}
	PGH_HIP(d_ac.Alloc(sizeof(uint32_t) * N), "hipMalloc(score out)");
	mark("output buffers");
	pgh_score_plan *plan = nullptr;
	int rc = ScorePlanCreate(ds, subset, n_scored, vidx, weights, flip, n_cols, mode, counts, false, &plan, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	mark("plan (tables, digits)");
	rc = pgh_score_run_dev(plan, d_score.p, d_dos.p, d_ac.p, PghThreadStream(), errbuf);
	if (rc == PGH_OK) {
		hipError_t e = hipStreamSynchronize(PghThreadStream()); // the plan's buffers are freed next
		if (e != hipSuccess) {
			rc = DeviceFail(errbuf, "score sync", e);
		}
	}
	mark("contraction");
	pgh_score_plan_destroy(plan);
	if (rc != PGH_OK) {
		return rc;
	}
	mark("plan release");
	std::vector<double> h_score(static_cast<size_t>(N) * n_cols), h_dos(N);
	std::vector<uint32_t> h_ac(N);
	PGH_HIP(hipMemcpy(h_score.data(), d_score.p, sizeof(double) * N * n_cols, hipMemcpyDeviceToHost), "score copy");
	if (dosage_sum) {
		PGH_HIP(hipMemcpy(h_dos.data(), d_dos.p, sizeof(double) * N, hipMemcpyDeviceToHost), "score copy");
	}
	PGH_HIP(hipMemcpy(h_ac.data(), d_ac.p, sizeof(uint32_t) * N, hipMemcpyDeviceToHost), "score copy");
	Compact<double>(subset, h_score.data(), n_cols, score_sum, N);
	if (dosage_sum) {
		Compact<double>(subset, h_dos.data(), 1, dosage_sum, N);
	}
	Compact<uint32_t>(subset, h_ac.data(), 1, allele_ct, N);
	mark("results to the host");
	return PGH_OK;
}

// ---------------------------------------------------------------------------
// plink_pca
// ---------------------------------------------------------------------------

extern "C" int pgh_pca(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_var, const uint32_t *vidx,
                       const double *center, const double *inv_stdev, uint32_t n_pcs, const double *g1_init,
                       double *eigenvalues, double *eigenvectors, char *errbuf) {
	if (ds && ds->IsGroup()) {
		return pgh_group::Pca(ds, subset, n_var, vidx, center, inv_stdev, n_pcs, g1_init, eigenvalues, eigenvectors,
		                      errbuf);
	}
	return pgh_pca_sharded(ds, subset, n_var, vidx, center, inv_stdev, n_var, n_pcs, g1_init, nullptr, nullptr,
	                       eigenvalues, eigenvectors, errbuf);
}

namespace {

//! A block of the rows of X that pgh_pca works on while it is resident.
struct PcaBlock {
	RowView view {nullptr, 0, 0, 0};
	const uint32_t *local = nullptr; // host: the block's effective variants as rows of `view`
	uint32_t row0 = 0, rows = 0;     // which of the call's effective variants
};

//! Where the rows come from: a resident dataset (one block, kept for the whole call) or a file beyond the HBM budget
//! (one window after the other; Acquire opens it, Release closes it).
struct PcaSource {
	uint32_t sample_ct = 0;
	virtual ~PcaSource() = default;
	virtual uint32_t Count() const = 0;
	virtual uint32_t MaxRows() const = 0;
	virtual bool Keep() const = 0;
	virtual int Acquire(uint32_t i, PcaBlock &out, char *errbuf) = 0;
	virtual void Release(uint32_t i) = 0;
};

struct ResidentPcaSource : PcaSource {
	const pgh_dataset *ds;
	std::vector<uint32_t> local;
	ResidentPcaSource(const pgh_dataset *d, std::vector<uint32_t> rows) : ds(d), local(std::move(rows)) {
		sample_ct = d->sample_ct;
	}
	uint32_t Count() const override {
		return 1;
	}
	uint32_t MaxRows() const override {
		return static_cast<uint32_t>(local.size());
	}
	bool Keep() const override {
		return true;
	}
	int Acquire(uint32_t, PcaBlock &out, char *) override {
		out.view = ds->View();
		out.local = local.data();
		out.row0 = 0;
		out.rows = static_cast<uint32_t>(local.size());
		return PGH_OK;
	}
	void Release(uint32_t) override {
	}
};

//! Windows of a .pgen: consecutive runs of the (ascending) effective variants whose file span fits `window` variants.
struct WindowPcaSource : PcaSource {
	std::string path, pgi;
	int device = -1;
	const uint32_t *vidx;
	struct Win {
		uint32_t first, count; // into vidx
	};
	std::vector<Win> wins;
	uint32_t max_rows = 0;
	pgh_dataset *open = nullptr;
	std::vector<uint32_t> local;
	uint32_t Count() const override {
		return static_cast<uint32_t>(wins.size());
	}
	uint32_t MaxRows() const override {
		return max_rows;
	}
	bool Keep() const override {
		return false;
	}
	int Acquire(uint32_t i, PcaBlock &out, char *errbuf) override {
		const Win &w = wins[i];
		const uint32_t v0 = vidx[w.first], v1 = vidx[w.first + w.count - 1] + 1;
		int rc = pgh_open(path.c_str(), pgi.empty() ? nullptr : pgi.c_str(), v0, v1, &open, errbuf);
		if (rc != PGH_OK) {
			open = nullptr;
			return rc;
		}
		local.resize(w.count);
		for (uint32_t k = 0; k < w.count; k++) {
			local[k] = vidx[w.first + k] - v0;
		}
		out.view = open->View();
		out.local = local.data();
		out.row0 = w.first;
		out.rows = w.count;
		return PGH_OK;
	}
	void Release(uint32_t) override {
		if (open) {
			pgh_close(open);
			open = nullptr;
		}
	}
	~WindowPcaSource() override {
		Release(0);
	}
};

} // namespace

static int PcaRun(PcaSource &src, const pgh_subset *subset, uint32_t n_var, const double *center, const double *inv_stdev,
                  uint64_t n_var_total, uint32_t n_pcs, const double *g1_init, pgh_allreduce_fn allreduce,
                  void *allreduce_ctx, double *eigenvalues, double *eigenvectors, char *errbuf) {
	// PGH_PCA_TIMING=1: wall-clock of the call's phases on stderr (each mark drains the stream first)
	static const bool timing = [] {
		const char *e = std::getenv("PGH_PCA_TIMING");
		return e && *e && *e != '0';
	}();
	auto t_last = std::chrono::steady_clock::now();
	hipStream_t t_stream = nullptr;
	auto mark = [&](const char *what) {
		if (!timing) {
			return;
		}
		if (t_stream) {
			(void)hipStreamSynchronize(t_stream);
		}
		const auto now = std::chrono::steady_clock::now();
		std::fprintf(stderr, "pca %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
		t_last = now;
	};
	const uint32_t N = src.sample_ct;
	const uint32_t n_out = subset ? subset->n_out : N;
	const uint32_t M = n_var; // this shard's rows of X; every 1/M and the eigenvalue divisor use the total
	const double m_total = static_cast<double>(n_var_total);
	const uint32_t k2 = 2 * n_pcs;
	const uint32_t qq = (n_pcs + 1) * k2;
	if (n_var_total < qq || n_out < qq) {
		SetErr(errbuf, "too few variants or samples for the requested number of PCs");
		return PGH_ERR_ARG;
	}
	// A stream of the call's own (not the per-thread default): the all-reduce callback is handed a real stream
	// handle that a host framework can wrap and order against its own streams with events.
	struct OwnStream {
		hipStream_t s = nullptr;
		~OwnStream() {
			if (s) {
				(void)hipStreamSynchronize(s);
				(void)hipStreamDestroy(s);
			}
		}
	} own;
	PGH_HIP(hipStreamCreateWithFlags(&own.s, hipStreamNonBlocking), "hipStreamCreate(pca)");
	hipStream_t st = own.s;
	t_stream = st;
	mark("checks + stream");
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
	DevBuf d_center, d_inv, d_ts, d_g1, d_g2, d_qq, d_bb;
	PGH_HIP(d_center.Alloc(sizeof(double) * m_alloc), "hipMalloc(pca)");
	PGH_HIP(d_inv.Alloc(sizeof(double) * m_alloc), "hipMalloc(pca)");
	PGH_HIP(d_ts.Alloc(32ull * m_alloc), "hipMalloc(pca)");
	PGH_HIP(d_g1.Alloc(sizeof(double) * N * k2), "hipMalloc(pca)");
	PGH_HIP(d_g2.Alloc(sizeof(double) * N * k2), "hipMalloc(pca)");
	PGH_HIP(d_qq.Alloc(sizeof(double) * m_alloc * qq), "hipMalloc(pca)");
	if (M) {
		PGH_HIP(hipMemcpy(d_center.p, center, sizeof(double) * M, hipMemcpyHostToDevice), "pca upload");
		PGH_HIP(hipMemcpy(d_inv.p, inv_stdev, sizeof(double) * M, hipMemcpyHostToDevice), "pca upload");
	}
	if (!subset) {
		// every sample is in: the caller's start matrix already is in raw-sample rows
		PGH_HIP(hipMemcpy(d_g1.p, g1_init, sizeof(double) * static_cast<size_t>(N) * k2, hipMemcpyHostToDevice), "pca upload");
	} else {
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
	mark("allocations + uploads");
	// Both contractions run on the int8 matrix cores (score_i8.hip): the dense factor of each pass is cut into
	// exact fixed-point digits, <= 18 columns per pass.  X^T (...) walks the resident rows; X G1 walks the
	// transposed packed matrix, built here once (pca_i8.hip).
	DevBuf d_iota, d_a, d_mm, d_colsum, i8_bmat, i8_rowidx, i8_cols, i8_small;
	pgh::ScoreI8Buffers i8;
	const uint32_t block_rows = std::max<uint32_t>(src.MaxRows(), 1);
	{
		const uint32_t longest = std::max(block_rows, N);
		const pgh::ScoreI8Sizes z = pgh::ScoreI8Bytes(longest, pgh::kI8MaxColsBare, false);
		PGH_HIP(i8_bmat.Alloc(z.bmat), "hipMalloc(pca digits)");
		PGH_HIP(i8_rowidx.Alloc(z.rowidx), "hipMalloc(pca digits)");
		PGH_HIP(i8_cols.Alloc(z.cols), "hipMalloc(pca digits)");
		PGH_HIP(i8_small.Alloc(z.small), "hipMalloc(pca digits)");
		i8.bmat = i8_bmat.As<int8_t>();
		i8.rowidx = i8_rowidx.As<uint32_t>();
		i8.mult = i8_cols.As<double>();
		i8.target = reinterpret_cast<uint32_t *>(i8.mult + 16ull * z.n_tiles16);
		i8.colmax = i8_small.As<unsigned long long>();
		i8.k0 = reinterpret_cast<double *>(i8.colmax + (pgh::kI8MaxColsBare + 2));
		i8.scale_exp = i8.k0 + (pgh::kI8MaxColsBare + 2);
		PGH_HIP(d_iota.Alloc(sizeof(uint32_t) * N), "hipMalloc(pca)");
		PGH_HIP(pgh::LaunchIota(d_iota.As<uint32_t>(), N, st), "pca iota");
		PGH_HIP(d_a.Alloc(sizeof(double) * block_rows * k2), "hipMalloc(pca)");
		PGH_HIP(d_mm.Alloc(sizeof(double) * block_rows * k2), "hipMalloc(pca)");
		PGH_HIP(d_colsum.Alloc(sizeof(double) * k2), "hipMalloc(pca)");
	}
	// The rows of X come in blocks (PcaSource): ONE for a resident dataset -- its transposed copy and tile images are
	// built once, here, and serve every pass -- or the windows of a file beyond the HBM budget, each opened, transposed,
	// used for a pass's Step A and Step B (or for phase 3) and closed again.  Rows row0 .. row0 + rows of the call's
	// per-variant arrays (centre, 1 / sd, tables, QQ) belong to the block.
	struct BlockState {
		PcaBlock blk;
		DevBuf d_vlist, d_xt, d_tiles_x, d_tiles_xt;
		RowView view_t {nullptr, 0, 0, 0};
		bool ready = false;
	};
	std::vector<BlockState> states(src.Count());
	struct BlockGuard { // (an early return between acquire and release must not leave a window open)
		PcaSource &s;
		std::vector<BlockState> &st;
		~BlockGuard() {
			for (uint32_t i = 0; i < st.size(); i++) {
				if (st[i].ready) {
					s.Release(i);
				}
			}
		}
	} block_guard {src, states};
	auto prepare_block = [&](uint32_t i, int &rc_out) -> hipError_t {
		BlockState &b = states[i];
		rc_out = PGH_OK;
		if (b.ready) {
			return hipSuccess;
		}
		b = BlockState();
		rc_out = src.Acquire(i, b.blk, errbuf);
		if (rc_out != PGH_OK) {
			return hipSuccess;
		}
		b.ready = true;
		const uint32_t rows = b.blk.rows;
		if (rows == 0) {
			return hipSuccess;
		}
		hipError_t e = b.d_vlist.Alloc(sizeof(uint32_t) * rows);
		if (e == hipSuccess) {
			e = hipMemcpyAsync(b.d_vlist.p, b.blk.local, sizeof(uint32_t) * rows, hipMemcpyHostToDevice, st);
		}
		if (e == hipSuccess) {
			e = hipStreamSynchronize(st); // (blk.local is the source's; a streamed one reuses it for the next window)
		}
		b.view_t = RowView {nullptr, pgh::TransposedPitch(rows), rows, (rows + 3) / 4};
		if (e == hipSuccess) {
			e = b.d_xt.Alloc(b.view_t.pitch * N);
		}
		if (e == hipSuccess) {
			e = pgh::LaunchTranspose2bit(b.blk.view, b.d_vlist.As<uint32_t>(), rows, b.d_xt.As<uint8_t>(), st);
		}
		b.view_t.rows = b.d_xt.As<uint8_t>();
		// Tile-major copies of both matrices for the many-column passes (score_i8.hpp): every pass over X or X^T
		// then streams its genotype bytes as contiguous 8 KB tile images.  One pass over each matrix to build
		// (~1 ms per GB), n_pcs + 1 resp. n_pcs + 10-ish contractions to use; skipped when the shapes of this
		// call do not use tiles (few PCs), when HBM is short, or when the block is a window that serves one pass
		// only (the contractions then read the rows as before).  PGH_PCA_TILES=0 turns it off (an A/B switch).
		const char *tiles_env = std::getenv("PGH_PCA_TILES");
		const bool want_tiles = src.Keep() && !(tiles_env && *tiles_env == '0');
		const bool step_tiles = pgh::ScoreI8UsesTiles(std::min(pgh::kI8MaxColsBare, k2), false);
		const bool final_tiles = pgh::ScoreI8UsesTiles(std::min<uint32_t>(pgh::kI8MaxColsBare, qq), false);
		const size_t free_b = want_tiles ? PghDeviceFreeBytes() : 0;
		const size_t need_x = pgh::ScoreI8TiledBytes(rows, N), need_xt = pgh::ScoreI8TiledBytes(N, rows);
		if (e == hipSuccess && want_tiles && (step_tiles || final_tiles) && free_b > need_x + need_xt + (2ull << 30)) {
			e = b.d_tiles_x.Alloc(need_x);
			if (e == hipSuccess) {
				e = pgh::LaunchScoreI8TileMajor(b.blk.view, b.d_vlist.As<uint32_t>(), rows, b.d_tiles_x.As<uint8_t>(), st);
			}
			if (e == hipSuccess && step_tiles) {
				e = b.d_tiles_xt.Alloc(need_xt);
				if (e == hipSuccess) {
					e = pgh::LaunchScoreI8TileMajor(b.view_t, d_iota.As<uint32_t>(), N, b.d_tiles_xt.As<uint8_t>(), st);
				}
			}
		}
		return e;
	};
	// a window is done with: its kernels drain, its device copies and the window itself go
	auto drop_block = [&](uint32_t i) -> hipError_t {
		if (src.Keep() || !states[i].ready) {
			return hipSuccess;
		}
		hipError_t e = hipStreamSynchronize(st);
		states[i] = BlockState();
		src.Release(i);
		return e;
	};
#define PGH_BLOCK(i)                                                                                                   \
	do {                                                                                                               \
		int rc_blk_ = PGH_OK;                                                                                          \
		hipError_t e_blk_ = prepare_block((i), rc_blk_);                                                               \
		if (rc_blk_ != PGH_OK) {                                                                                       \
			return rc_blk_;                                                                                            \
		}                                                                                                              \
		PGH_HIP(e_blk_, "pca block (transpose / tiles)");                                                              \
	} while (0)
	if (src.Keep()) {
		for (uint32_t i = 0; i < src.Count(); i++) {
			PGH_BLOCK(i);
		}
	}
	mark("i8 buffers + transpose");
	// out[s][c] += sum_v t_v[g(v,s)] W[v][c] over this shard's variants (Step B, phase 3); out zeroed by the caller
	// (w: the block's rows of the factor, i.e. already offset by blk.row0 * w_stride)
	auto contract_variants = [&](BlockState &b, const double *w, uint32_t w_stride, uint32_t n_cols, double *out,
	                             uint32_t out_stride) -> hipError_t {
		hipError_t e = hipSuccess;
		const uint32_t rows = b.blk.rows;
		for (uint32_t c0 = 0; c0 < n_cols && e == hipSuccess && rows; c0 += pgh::kI8MaxColsBare) {
			const uint32_t nc = std::min(pgh::kI8MaxColsBare, n_cols - c0);
			i8.tiled = b.d_tiles_x.As<uint8_t>();
			e = pgh::LaunchScoreI8Prepare(b.d_vlist.As<uint32_t>(), rows, w + c0, w_stride, nc,
			                              d_ts.As<double>() + 4ull * b.blk.row0, nullptr, nullptr, false, false, pgh::kI8Tables,
			                              i8, st);
			if (e == hipSuccess) {
				e = pgh::LaunchScoreI8(b.blk.view, rows, nc, false, pgh::kI8Tables, i8, out + c0, out_stride, nullptr, nullptr,
				                       st);
			}
		}
		return e;
	};
	// y[v][c] = sum_s x(v,s) G[s][c] for this shard's variants (Step A): the two integer planes over the transposed
	// matrix, then the per-variant normalisation
	// (y_out: the block's rows of QQ, i.e. already offset by blk.row0 * qq)
	auto contract_samples = [&](BlockState &b, const double *g, double *y_out) -> hipError_t {
		const uint32_t rows = b.blk.rows;
		if (rows == 0) {
			return hipSuccess;
		}
		hipError_t e = hipMemsetAsync(d_a.p, 0, sizeof(double) * rows * k2, st);
		if (e == hipSuccess) {
			e = hipMemsetAsync(d_mm.p, 0, sizeof(double) * rows * k2, st);
		}
		for (int plane = pgh::kI8CodePlane; plane <= pgh::kI8MissingPlane && e == hipSuccess; plane++) {
			double *dst = plane == pgh::kI8CodePlane ? d_a.As<double>() : d_mm.As<double>();
			for (uint32_t c0 = 0; c0 < k2 && e == hipSuccess; c0 += pgh::kI8MaxColsBare) {
				const uint32_t nc = std::min(pgh::kI8MaxColsBare, k2 - c0);
				i8.tiled = b.d_tiles_xt.As<uint8_t>();
				e = pgh::LaunchScoreI8Prepare(d_iota.As<uint32_t>(), N, g + c0, k2, nc, nullptr, nullptr, nullptr, false, false,
				                              plane, i8, st);
				if (e == hipSuccess) {
					e = pgh::LaunchScoreI8(b.view_t, N, nc, false, plane, i8, dst + c0, k2, nullptr, nullptr, st);
				}
			}
		}
		if (e == hipSuccess) {
			e = pgh::LaunchColumnSums(g, N, k2, k2, d_colsum.As<double>(), st);
		}
		if (e == hipSuccess) {
			e = pgh::LaunchPcaCombine(d_a.As<double>(), d_mm.As<double>(), d_colsum.As<double>(),
			                          d_center.As<double>() + b.blk.row0, d_inv.As<double>() + b.blk.row0, rows, k2, y_out, qq, st);
		}
		return e;
	};
	double *g1 = d_g1.As<double>();
	double *g2 = d_g2.As<double>();
	const uint8_t *mask2 = subset ? subset->d_mask2 : nullptr;
	for (uint32_t pass = 0; pass <= n_pcs; pass++) {
		double *y = d_qq.As<double>() + static_cast<size_t>(pass) * k2;
		if (pass < n_pcs) {
			PGH_HIP(hipMemsetAsync(g2, 0, sizeof(double) * N * k2, st), "pca memset");
		}
		for (uint32_t bi = 0; bi < src.Count(); bi++) {
			PGH_BLOCK(bi);
			BlockState &b = states[bi];
			double *y_b = y + static_cast<size_t>(b.blk.row0) * qq;
			// Step A: QQ[:, pass*2k : (pass+1)*2k] = X * G1   (rows of this block only)
			PGH_HIP(contract_samples(b, g1, y_b), "pca step A");
			if (pass < n_pcs) {
				// Step B: G2 += X^T Y over the block's rows
				PGH_HIP(contract_variants(b, y_b, qq, k2, g2, k2), "pca step B");
			}
			PGH_HIP(drop_block(bi), "pca block release");
		}
		if (pass < n_pcs) {
			// merge: G1 = X^T Y / M, summed over shards
			PGH_SUM(g2, static_cast<uint64_t>(N) * k2);
			PGH_HIP(pgh::LaunchMaskRows(g2, N, k2, k2, mask2, st), "pca mask");
			PGH_HIP(pgh::LaunchScale(g2, static_cast<uint64_t>(N) * k2, 1.0 / m_total, st), "pca scale");
			std::swap(g1, g2);
		}
	}
	mark("power iterations");
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
		HostSourceFence fence(st); // `t` feeds an asynchronous upload
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
	mark("orthonormal basis");
	// Phase 3: BB = X^T U   (src/plink_pca.cpp:664-676)
	PGH_HIP(d_bb.Alloc(sizeof(double) * static_cast<size_t>(N) * qq), "hipMalloc(pca)");
	PGH_HIP(hipMemsetAsync(d_bb.p, 0, sizeof(double) * static_cast<size_t>(N) * qq, st), "pca memset");
	for (uint32_t bi = 0; bi < src.Count(); bi++) {
		PGH_BLOCK(bi);
		BlockState &b = states[bi];
		PGH_HIP(contract_variants(b, d_qq.As<double>() + static_cast<size_t>(b.blk.row0) * qq, qq, qq, d_bb.As<double>(), qq),
		        "pca phase 3");
		PGH_HIP(drop_block(bi), "pca block release");
	}
	PGH_SUM(d_bb.As<double>(), static_cast<uint64_t>(N) * qq);
	PGH_HIP(pgh::LaunchMaskRows(d_bb.As<double>(), N, qq, qq, mask2, st), "pca mask");
	mark("phase 3");
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
		mark("gram of BB");
		pgh::SymmetricEigen(g, qq, lam, vec);
		mark("eigen (host)");
		std::vector<double> vk(static_cast<size_t>(qq) * n_pcs);
		HostSourceFence fence(st); // `vk` feeds an asynchronous upload
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
		if (!subset) {
			PGH_HIP(hipMemcpyAsync(eigenvectors, d_uk.p, sizeof(double) * static_cast<size_t>(N) * n_pcs,
			                       hipMemcpyDeviceToHost, st),
			        "pca download");
			PGH_HIP(hipStreamSynchronize(st), "pca sync");
		} else {
			std::vector<double> uk_raw(static_cast<size_t>(N) * n_pcs);
			PGH_HIP(hipMemcpyAsync(uk_raw.data(), d_uk.p, sizeof(double) * uk_raw.size(), hipMemcpyDeviceToHost, st),
			        "pca download");
			PGH_HIP(hipStreamSynchronize(st), "pca sync");
			Compact<double>(subset, uk_raw.data(), n_pcs, eigenvectors, N);
		}
		mark("eigenvectors + download");
	}
	return PGH_OK;
#undef PGH_SUM
#undef PGH_BLOCK
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
	PGH_ONE_DEVICE(ds);
	PGH_ENTER(ds);
	int rc = CheckSubset(ds, subset, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	std::vector<uint32_t> local(n_var);
	for (uint32_t i = 0; i < n_var; i++) {
		if (vidx[i] < ds->v_begin || vidx[i] >= ds->v_end) {
			SetErr(errbuf, "effective variant index outside the resident range");
			return PGH_ERR_ARG;
		}
		local[i] = vidx[i] - ds->v_begin;
	}
	ResidentPcaSource src(ds, std::move(local));
	return PcaRun(src, subset, n_var, center, inv_stdev, n_var_total, n_pcs, g1_init, allreduce, allreduce_ctx, eigenvalues,
	              eigenvectors, errbuf);
}

extern "C" int pgh_pca_streamed(const char *pgen_path, const char *pgi_path, const uint64_t *sample_include,
                                uint32_t n_var, const uint32_t *vidx, const double *center, const double *inv_stdev,
                                uint32_t n_pcs, const double *g1_init, uint64_t window_variants, double *eigenvalues,
                                double *eigenvectors, char *errbuf) {
	if (!pgen_path || !g1_init || !eigenvalues || !eigenvectors || n_pcs == 0 || n_var == 0 || !vidx || !center ||
	    !inv_stdev || window_variants == 0) {
		SetErr(errbuf, "null or empty argument");
		return PGH_ERR_ARG;
	}
	for (uint32_t i = 1; i < n_var; i++) {
		if (vidx[i] <= vidx[i - 1]) {
			SetErr(errbuf, "the effective variants of a streamed pgh_pca must be in ascending file order");
			return PGH_ERR_ARG;
		}
	}
	WindowPcaSource src;
	src.path = pgen_path;
	src.pgi = pgi_path ? pgi_path : "";
	src.vidx = vidx;
	for (uint32_t i = 0; i < n_var;) {
		uint32_t j = i + 1;
		while (j < n_var && static_cast<uint64_t>(vidx[j]) + 1 - vidx[i] <= window_variants) {
			j++;
		}
		src.wins.push_back({i, j - i});
		src.max_rows = std::max(src.max_rows, j - i);
		i = j;
	}
	// the sample count, and the staged form of the subset (it outlives the window it is made on): the first window
	pgh_dataset *first = nullptr;
	pgh_subset *ss = nullptr;
	int rc = pgh_open(pgen_path, pgi_path, vidx[0], vidx[src.wins[0].count - 1] + 1, &first, errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	src.sample_ct = first->sample_ct;
	if (sample_include) {
		rc = pgh_subset_create(first, sample_include, &ss, errbuf);
	}
	pgh_close(first);
	if (rc == PGH_OK) {
		rc = PcaRun(src, ss, n_var, center, inv_stdev, n_var, n_pcs, g1_init, nullptr, nullptr, eigenvalues, eigenvectors,
		            errbuf);
	}
	pgh_subset_destroy(ss);
	return rc;
}

// ---------------------------------------------------------------------------
// plink_ld
// ---------------------------------------------------------------------------

namespace {
// pgh_ld_pairs_dev's task list: one device buffer + one pinned host buffer per calling thread, plus the
// kernel's "malformed task" word and its pinned mirror
struct LdTaskBuffers {
	void *d = nullptr, *h = nullptr;
	uint32_t *d_bad = nullptr, *h_bad = nullptr;
	size_t cap = 0;
	int device = -1;           // the buffers belong to this device
	hipEvent_t done = nullptr; // recorded behind the kernel that reads d and the copy that fills h_bad
	bool used = false;
	void Release() {
		if (d) {
			(void)hipFree(d);
		}
		if (h) {
			(void)hipHostFree(h);
		}
		if (d_bad) {
			(void)hipFree(d_bad);
		}
		if (h_bad) {
			(void)hipHostFree(h_bad);
		}
		if (done) {
			(void)hipEventDestroy(done);
		}
		d = h = nullptr;
		d_bad = h_bad = nullptr;
		done = nullptr;
		cap = 0;
		used = false;
	}
	~LdTaskBuffers() {
		Release(); // thread exit: the runtime may already be gone at process exit, errors are of no interest here
	}
};
thread_local LdTaskBuffers t_ld_tasks;

//! Waits for this thread's last LD launch and fails if its kernel met a task the host cannot have built.
int LdCheckLast(char *errbuf) {
	LdTaskBuffers &buf = t_ld_tasks;
	if (!buf.used) {
		return PGH_OK;
	}
	PGH_HIP(hipEventSynchronize(buf.done), "ld pair kernel");
	buf.used = false;
	if (*buf.h_bad != UINT32_MAX) {
		char msg[200];
		std::snprintf(msg, sizeof msg,
		              "plink_ld: task %u of the launch was not well-formed on the device (the task list did not "
		              "arrive intact); its sums were not computed",
		              *buf.h_bad - 1);
		SetErr(errbuf, msg);
		return PGH_ERR_DEVICE;
	}
	return PGH_OK;
}
} // namespace

extern "C" int pgh_ld_pairs_status(char *errbuf) {
	return LdCheckLast(errbuf);
}

extern "C" int pgh_ld_pairs_dev(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_pairs,
                                const uint32_t *vidx_a, const uint32_t *vidx_b, void *d_sums, void *stream,
                                char *errbuf) {
	if (!ds || (n_pairs && (!vidx_a || !vidx_b || !d_sums))) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	PGH_ONE_DEVICE(ds);
	PGH_ENTER(ds);
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
	hipStream_t st = static_cast<hipStream_t>(stream);
	// The task list goes up through this thread's own pair of buffers (device + PINNED host), reused from call
	// to call.  Round 1's first form -- hipMallocAsync, hipMemcpyAsync straight from the frame-local pageable
	// `tasks` vector, launch, hipFreeAsync, return -- once delivered all-zero tasks.  What the box shows
	// (tools/pageable_async_probe.hip, profiles/r02_pageable_probe.txt): the runtime takes a pageable source's
	// bytes before hipMemcpyAsync returns, so the dying vector was not the cause; but a pool block that an
	// earlier owner zeroed with a still-queued hipMemsetAsync, freed stream-ordered and got handed out again was
	// seen to keep the EARLIER zeros after the LATER pageable copy (once in three probe runs) -- exactly this
	// call's shape after plink_score's `miss` block.  Round 2 then lost KERNEL-written data in a block of the same
	// pool (pgh_missing_per_sample's scratch, profiles/r02_async_pool_ab.txt): nothing here allocates
	// stream-ordered any more (api_internal.hpp:PghThreadScratch); uploads go through pinned memory into plain
	// allocations (here), or from pageable memory into plain allocations behind a HostSourceFence.
	// The previous launch of this thread is checked first: it must have finished with the buffers anyway.
	rc = LdCheckLast(errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	LdTaskBuffers &buf = t_ld_tasks;
	const size_t need = sizeof(pgh::LdTask) * tasks.size();
	if (buf.cap < need || buf.device != ds->device) {
		buf.Release();
		buf.device = ds->device;
		const size_t cap = std::max<size_t>(2 * need, 64u << 10);
		PGH_HIP(PghMalloc(&buf.d, cap), "hipMalloc(ld tasks)");
		PGH_HIP(hipHostMalloc(&buf.h, cap, hipHostMallocDefault), "hipHostMalloc(ld tasks)");
		PGH_HIP(PghMalloc(reinterpret_cast<void **>(&buf.d_bad), sizeof(uint32_t)), "hipMalloc(ld tasks)");
		PGH_HIP(hipHostMalloc(reinterpret_cast<void **>(&buf.h_bad), sizeof(uint32_t), hipHostMallocDefault),
		        "hipHostMalloc(ld tasks)");
		PGH_HIP(hipEventCreateWithFlags(&buf.done, hipEventDisableTiming), "ld task event");
		buf.cap = cap;
	}
	std::memcpy(buf.h, tasks.data(), need);
	if (const char *t = std::getenv("PGH_TEST_ZERO_LD_TASKS"); t && *t == '1') {
		std::memset(buf.h, 0, need); // test hook: what round 1's lost upload looked like on the device
	}
	*buf.h_bad = UINT32_MAX;
	// a pair whose task were refused must not hand back whatever the caller's buffer held
	PGH_HIP(hipMemsetAsync(d_sums, 0, 24ull * n_pairs, st), "ld memset");
	PGH_HIP(hipMemsetAsync(buf.d_bad, 0xff, sizeof(uint32_t), st), "ld memset");
	PGH_HIP(hipMemcpyAsync(buf.d, buf.h, need, hipMemcpyHostToDevice, st), "ld task upload");
	PGH_HIP(pgh::LaunchLdPairs(ds->View(), ds->v_end - ds->v_begin, static_cast<const pgh::LdTask *>(buf.d),
	                           static_cast<uint32_t>(tasks.size()), subset ? subset->d_mask2 : nullptr,
	                           static_cast<uint32_t(*)[6]>(d_sums), buf.d_bad, st),
	        "ld pair kernel");
	PGH_HIP(hipMemcpyAsync(buf.h_bad, buf.d_bad, sizeof(uint32_t), hipMemcpyDeviceToHost, st), "ld status copy");
	PGH_HIP(hipEventRecord(buf.done, st), "ld task event");
	buf.used = true;
	return PGH_OK;
}

extern "C" int pgh_ld_pairs(const pgh_dataset *ds, const pgh_subset *subset, uint32_t n_pairs, const uint32_t *vidx_a,
                            const uint32_t *vidx_b, uint32_t (*sums)[6], char *errbuf) {
	if (n_pairs && !sums) {
		SetErr(errbuf, "null argument");
		return PGH_ERR_ARG;
	}
	if (n_pairs == 0) {
		return PGH_OK;
	}
	if (ds && ds->IsGroup()) {
		return pgh_group::LdPairs(ds, subset, n_pairs, vidx_a, vidx_b, sums, errbuf);
	}
	PGH_ENTER(ds);
	DevBuf d_out;
	PGH_HIP(d_out.Alloc(24ull * n_pairs), "hipMalloc(ld)");
	int rc = pgh_ld_pairs_dev(ds, subset, n_pairs, vidx_a, vidx_b, d_out.p, PghThreadStream(), errbuf);
	if (rc != PGH_OK) {
		return rc;
	}
	PGH_HIP(hipMemcpyAsync(sums, d_out.p, 24ull * n_pairs, hipMemcpyDeviceToHost, PghThreadStream()), "ld copy");
	PGH_HIP(hipStreamSynchronize(PghThreadStream()), "ld sync");
	return LdCheckLast(errbuf);
}

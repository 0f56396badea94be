"""ctypes binding of libpgenhip.so -- one Python method per C entry point of
``include/pgenhip.h``.  Thin on purpose: argument marshalling and status ->
exception translation only (PGH_ERR_ARG -> ValueError ~ InvalidInputException,
everything else -> IOError ~ IOException, the reference's convention,
src/plink_freq.cpp:152,181,485)."""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PGENHIP_LIB: another build of the library (tools/i8_experiment.sh's knock-out builds live under /tmp and never
# replace the in-tree file)
LIB_PATH = os.environ.get("PGENHIP_LIB") or os.path.join(_HERE, "libpgenhip.so")

ERRBUF_LEN = 256
PGH_OK, PGH_ERR_OPEN, PGH_ERR_FORMAT, PGH_ERR_ARG, PGH_ERR_DEVICE, PGH_ERR_NOMEM, PGH_ERR_UNSUPPORTED = range(7)
SCORE_MEAN_IMPUTE, SCORE_NO_MEAN_IMPUTATION, SCORE_CENTER = 0, 1, 2

# every symbol include/pgenhip.h declares (tests/test_abi.py checks the .so exports each)
EXPORTED_SYMBOLS = [
    "pgh_version", "pgh_device_count", "pgh_set_device", "pgh_open", "pgh_probe", "pgh_normalize_range_host",
    "pgh_from_host_rows", "pgh_open_sharded", "pgh_group_create", "pgh_group_uses_rccl", "pgh_shard_count", "pgh_shard",
    "pgh_synth_create", "pgh_synth_record_host", "pgh_synth_write_files", "pgh_copy_rows_to_host", "pgh_get_info", "pgh_device_rows",
    "pgh_close", "pgh_subset_create", "pgh_subset_size", "pgh_subset_destroy", "pgh_counts_range",
    "pgh_counts_range_dev", "pgh_freq_from_counts_dev", "pgh_fused_tally_dev", "pgh_missing_per_sample", "pgh_missing_per_sample_dev", "pgh_unpack_range",
    "pgh_unpack_range_dev", "pgh_probe_unpack_shape_dev", "pgh_score", "pgh_score_counts", "pgh_score_dev", "pgh_score_plan_create", "pgh_score_run_dev",
    "pgh_score_plan_destroy", "pgh_pca", "pgh_pca_sharded", "pgh_pca_streamed", "pgh_ld_pairs", "pgh_ld_pairs_dev", "pgh_ld_pairs_status", "pgh_sample_counts", "pgh_sample_counts_dev",
    "pgh_synth_add_dosage", "pgh_synth_write_dosage_files", "pgh_dosage_sums", "pgh_dosage_sums_dev", "pgh_dosage_unpack", "pgh_dosage_unpack_dev", "pgh_unpack_samples", "pgh_dosage_unpack_samples", "pgh_reader_create", "pgh_reader_destroy",
    "pgh_reader_unpack_start", "pgh_reader_unpack_wait", "pgh_get_2bit", "pgh_get_counts", "pgh_get_missingness", "pgh_get_int8", "pgh_get_dosage_f64", "pgh_get_phased",
    "pgh_tally_start", "pgh_tally_request", "pgh_tally_wait", "pgh_tally_counts", "pgh_tally_hwe_lnp",
    "pgh_tally_sample_missing", "pgh_tally_destroy", "pgh_tally_passes_started", "pgh_host_alloc", "pgh_host_free", "pgh_trim_device_cache",
    "pgh_reader_error", "pgh_hwe_lnp", "pgh_hwe_xchr_lnp", "pgh_hwe_lnp_batch", "pgh_hwe_lnp_batch_dev", "pgh_hwe_xchr_lnp_batch",
]


class PghInfo(C.Structure):
    _fields_ = [
        ("raw_variant_ct", C.c_uint32), ("raw_sample_ct", C.c_uint32), ("variant_begin", C.c_uint32),
        ("variant_end", C.c_uint32), ("has_dosage", C.c_uint32), ("has_phase", C.c_uint32),
        ("max_record_bytes", C.c_uint32), ("record_bytes", C.c_uint32), ("pitch_bytes", C.c_uint64),
        ("vrtype_hist", C.c_uint32 * 8), ("device", C.c_int32),
        ("dosage_variant_ct", C.c_uint32), ("dosage_value_ct", C.c_uint64),
    ]


# pgh_allreduce_fn(ctx, d_buf, count, stream) -> 0 on success
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p)


class PghError(IOError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class PghArgError(ValueError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with plinking_duck_amd/csrc/build.sh "
            "(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
    # PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so).  A process that
    # ends up with two HIP runtimes -- this library bound to /opt/rocm first, torch's loaded later --
    # leaves the second one without a device ("No HIP GPUs are available").  Loading torch first makes
    # both bind the same runtime, so do that whenever torch is installed; the library itself needs
    # neither torch nor Python.
    if os.environ.get("PGH_NO_TORCH_PRELOAD", "") in ("", "0"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    lib = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, cp = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32, C.c_char_p
    sigs = {
        "pgh_version": (cp, []),
        "pgh_device_count": (C.c_int, []),
        "pgh_set_device": (C.c_int, [C.c_int, cp]),
        "pgh_open": (C.c_int, [cp, cp, u32, u32, C.POINTER(vp), cp]),
        "pgh_probe": (C.c_int, [cp, cp, C.POINTER(PghInfo), cp]),
        "pgh_normalize_range_host": (C.c_int, [cp, cp, u32, u32, vp, C.c_size_t, cp]),
        "pgh_from_host_rows": (C.c_int, [vp, C.c_size_t, u32, u32, C.POINTER(vp), cp]),
        "pgh_synth_create": (C.c_int, [u32, u32, u32, u64, C.c_double, C.POINTER(vp), cp]),
        "pgh_open_sharded": (C.c_int, [cp, cp, u32, u32, vp, u32, C.POINTER(vp), cp]),
        "pgh_group_create": (C.c_int, [vp, u32, C.POINTER(vp), cp]),
        "pgh_shard_count": (u32, [vp]),
        "pgh_group_uses_rccl": (C.c_int, [vp]),
        "pgh_shard": (vp, [vp, u32]),
        "pgh_synth_record_host": (C.c_int, [u32, u32, u64, C.c_double, vp]),
        "pgh_synth_write_files": (C.c_int, [cp, u32, u32, u64, C.c_double, cp]),
        "pgh_copy_rows_to_host": (C.c_int, [vp, u32, u32, vp, C.c_size_t, cp]),
        "pgh_freq_from_counts_dev": (C.c_int, [vp, u32, vp, vp, vp, cp]),
        "pgh_hwe_lnp_batch_dev": (C.c_int, [vp, u32, u32, vp, vp, cp]),
        "pgh_get_info": (C.c_int, [vp, C.POINTER(PghInfo)]),
        "pgh_device_rows": (vp, [vp]),
        "pgh_close": (None, [vp]),
        "pgh_subset_create": (C.c_int, [vp, vp, C.POINTER(vp), cp]),
        "pgh_subset_size": (u32, [vp]),
        "pgh_subset_destroy": (None, [vp]),
        "pgh_counts_range": (C.c_int, [vp, vp, u32, u32, vp, cp]),
        "pgh_counts_range_dev": (C.c_int, [vp, vp, u32, u32, vp, vp, cp]),
        "pgh_fused_tally_dev": (C.c_int, [vp, u32, u32, vp, vp, vp, cp]),
        "pgh_missing_per_sample": (C.c_int, [vp, vp, u32, u32, vp, cp]),
        "pgh_missing_per_sample_dev": (C.c_int, [vp, u32, u32, vp, vp, cp]),
        "pgh_unpack_range": (C.c_int, [vp, vp, u32, u32, vp, vp, C.c_int, cp]),
        "pgh_unpack_range_dev": (C.c_int, [vp, vp, u32, u32, vp, C.c_size_t, vp, C.c_int, vp, cp]),
        "pgh_probe_unpack_shape_dev": (C.c_int, [vp, C.c_size_t, vp, vp, vp, cp]),
        "pgh_score": (C.c_int, [vp, vp, u32, vp, vp, vp, u32, C.c_int, vp, vp, vp, cp]),
        "pgh_score_dev": (C.c_int, [vp, vp, u32, vp, vp, vp, u32, C.c_int, vp, vp, vp, vp, cp]),
        "pgh_score_counts": (C.c_int, [vp, vp, u32, vp, vp, vp, u32, C.c_int, vp, vp, vp, vp, cp]),
        "pgh_score_plan_create": (C.c_int, [vp, vp, u32, vp, vp, vp, u32, C.c_int, C.POINTER(vp), cp]),
        "pgh_score_run_dev": (C.c_int, [vp, vp, vp, vp, vp, cp]),
        "pgh_score_plan_destroy": (None, [vp]),
        "pgh_pca": (C.c_int, [vp, vp, u32, vp, vp, vp, u32, vp, vp, vp, cp]),
        "pgh_ld_pairs": (C.c_int, [vp, vp, u32, vp, vp, vp, cp]),
        "pgh_hwe_xchr_lnp_batch": (C.c_int, [vp, u32, u32, vp, cp]),
        "pgh_sample_counts": (C.c_int, [vp, vp, u32, u32, vp, vp, cp]),
        "pgh_sample_counts_dev": (C.c_int, [vp, u32, u32, vp, vp, cp]),
        "pgh_synth_add_dosage": (C.c_int, [vp, C.c_double, C.c_uint64, cp]),
        "pgh_synth_write_dosage_files": (C.c_int, [cp, u32, u32, C.c_uint64, C.c_double, C.c_double, cp]),
        "pgh_dosage_sums": (C.c_int, [vp, vp, u32, u32, vp, vp, cp]),
        "pgh_dosage_sums_dev": (C.c_int, [vp, vp, u32, u32, vp, vp, cp]),
        "pgh_dosage_unpack": (C.c_int, [vp, vp, u32, u32, vp, vp, cp]),
        "pgh_unpack_samples": (C.c_int, [vp, vp, u32, vp, vp, C.c_int, cp]),
        "pgh_dosage_unpack_samples": (C.c_int, [vp, vp, u32, vp, vp, cp]),
        "pgh_dosage_unpack_dev": (C.c_int, [vp, vp, u32, u32, vp, C.c_size_t, vp, cp]),
        "pgh_ld_pairs_dev": (C.c_int, [vp, vp, u32, vp, vp, vp, vp, cp]),
        "pgh_ld_pairs_status": (C.c_int, [cp]),
        "pgh_pca_sharded": (C.c_int, [vp, vp, u32, vp, vp, vp, C.c_uint64, u32, vp, ALLREDUCE_FN, vp, vp, vp, cp]),
        "pgh_pca_streamed": (C.c_int, [C.c_char_p, C.c_char_p, vp, u32, vp, vp, vp, u32, vp, C.c_uint64, vp, vp, cp]),
        "pgh_reader_create": (C.c_int, [vp, vp, C.POINTER(vp), cp]),
        "pgh_reader_destroy": (None, [vp]),
        "pgh_reader_unpack_start": (C.c_int, [vp, C.c_int, u32, u32, vp, vp, C.c_int]),
        "pgh_reader_unpack_wait": (C.c_int, [vp, C.c_int]),
        "pgh_get_2bit": (C.c_int, [vp, u32, vp]),
        "pgh_get_counts": (C.c_int, [vp, u32, vp]),
        "pgh_get_missingness": (C.c_int, [vp, u32, vp]),
        "pgh_get_int8": (C.c_int, [vp, u32, vp]),
        "pgh_get_dosage_f64": (C.c_int, [vp, u32, vp]),
        "pgh_get_phased": (C.c_int, [vp, u32, vp, vp, vp]),
        "pgh_reader_error": (cp, [vp]),
        "pgh_hwe_lnp": (C.c_double, [i32, i32, i32, u32]),
        "pgh_hwe_xchr_lnp": (C.c_double, [i32, i32, i32, i32, i32, u32]),
        "pgh_hwe_lnp_batch": (C.c_int, [vp, u32, u32, vp, cp]),
        "pgh_tally_start": (C.c_int, [vp, vp, u32, u32, u32, C.POINTER(vp), cp]),
        "pgh_tally_request": (C.c_int, [vp, u32, cp]),
        "pgh_tally_wait": (C.c_int, [vp, u32, u32, u32, cp]),
        "pgh_tally_counts": (vp, [vp]),
        "pgh_tally_hwe_lnp": (vp, [vp, u32]),
        "pgh_tally_sample_missing": (C.c_int, [vp, vp, cp]),
        "pgh_tally_destroy": (None, [vp]),
        "pgh_tally_passes_started": (u64, []),
        "pgh_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(vp), cp]),
        "pgh_host_free": (None, [vp]),
        "pgh_trim_device_cache": (None, []),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


_lib = _load()


def raw():
    """The ctypes CDLL (for tests that probe the ABI directly)."""
    return _lib


def _check(rc, errbuf):
    if rc == PGH_OK:
        return
    msg = errbuf.value.decode("utf-8", "replace") if errbuf is not None else f"pgenhip error {rc}"
    if rc == PGH_ERR_ARG:
        raise PghArgError(rc, msg)
    raise PghError(rc, msg)


def _errbuf():
    return C.create_string_buffer(ERRBUF_LEN)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def version() -> str:
    return _lib.pgh_version().decode()


def device_count() -> int:
    return _lib.pgh_device_count()


def set_device(dev: int):
    eb = _errbuf()
    _check(_lib.pgh_set_device(dev, eb), eb)


def probe(path: str, pgi_path: str | None = None) -> PghInfo:
    info = PghInfo()
    eb = _errbuf()
    _check(_lib.pgh_probe(path.encode(), pgi_path.encode() if pgi_path else None, C.byref(info), eb), eb)
    return info


def normalize_range_host(path: str, v_begin: int = 0, v_end: int | None = None, pgi_path: str | None = None):
    """Host normaliser only: plain 2-bit rows uint8[v_end-v_begin][ceil(N/4)]."""
    info = probe(path, pgi_path)
    v_end = info.raw_variant_ct if v_end is None else v_end
    rows = np.zeros((max(0, v_end - v_begin), info.record_bytes), dtype=np.uint8)
    eb = _errbuf()
    _check(_lib.pgh_normalize_range_host(path.encode(), pgi_path.encode() if pgi_path else None, v_begin, v_end,
                                         _ptr(rows), info.record_bytes, eb), eb)
    return rows


def hwe_lnp(hets: int, hom1: int, hom2: int, midp: bool = False) -> float:
    return _lib.pgh_hwe_lnp(hets, hom1, hom2, 1 if midp else 0)


def hwe_xchr_lnp_batch(strata, midp: bool = False) -> np.ndarray:
    """strata: int32[n][5] = {female_hets, female_hom1, female_hom2, male1, male2} -> ln p per variant (device)."""
    st = np.ascontiguousarray(strata, dtype=np.int32)
    assert st.ndim == 2 and st.shape[1] == 5
    out = np.zeros(len(st), dtype=np.float64)
    eb = _errbuf()
    _check(_lib.pgh_hwe_xchr_lnp_batch(_ptr(st), len(st), 1 if midp else 0, _ptr(out), eb), eb)
    return out


def hwe_xchr_lnp(fhets: int, fhom1: int, fhom2: int, male1: int, male2: int, midp: bool = False) -> float:
    return _lib.pgh_hwe_xchr_lnp(fhets, fhom1, fhom2, male1, male2, 1 if midp else 0)


def hwe_lnp_batch(counts: np.ndarray, midp: bool = False) -> np.ndarray:
    counts = np.ascontiguousarray(counts, dtype=np.uint32).reshape(-1, 4)
    out = np.empty(len(counts), dtype=np.float64)
    eb = _errbuf()
    _check(_lib.pgh_hwe_lnp_batch(_ptr(counts), len(counts), 1 if midp else 0, _ptr(out), eb), eb)
    return out


def freq_from_counts_dev(d_counts: int, n: int, d_alt_freq: int, d_obs_ct: int, stream: int = 0):
    eb = _errbuf()
    _check(_lib.pgh_freq_from_counts_dev(d_counts, n, d_alt_freq, d_obs_ct, stream, eb), eb)


def hwe_lnp_batch_dev(d_counts: int, n: int, d_ln_p: int, midp: bool = False, stream: int = 0):
    eb = _errbuf()
    _check(_lib.pgh_hwe_lnp_batch_dev(d_counts, n, 1 if midp else 0, d_ln_p, stream, eb), eb)


def probe_unpack_shape_dev(d_src: int, n_vec: int, d_dst: int, d_val: int, stream: int = 0):
    eb = _errbuf()
    _check(_lib.pgh_probe_unpack_shape_dev(d_src, n_vec, d_dst, d_val, stream, eb), eb)


def synth_record_host(v: int, n: int, seed: int, missing_rate: float) -> np.ndarray:
    out = np.zeros((n + 3) // 4, dtype=np.uint8)
    rc = _lib.pgh_synth_record_host(v, n, seed, missing_rate, _ptr(out))
    if rc != PGH_OK:
        raise PghArgError(rc, "pgh_synth_record_host: bad argument")
    return out


def synth_write_files(prefix: str, m: int, n: int, seed: int, missing_rate: float):
    eb = _errbuf()
    _check(_lib.pgh_synth_write_files(prefix.encode(), m, n, seed, missing_rate, eb), eb)


def synth_write_dosage_files(prefix: str, m: int, n: int, seed: int, missing_rate: float, dosage_rate: float):
    eb = _errbuf()
    _check(_lib.pgh_synth_write_dosage_files(prefix.encode(), m, n, seed, missing_rate, dosage_rate, eb), eb)


TALLY_COUNTS, TALLY_SAMPLE_MISSING, TALLY_HWE, TALLY_HWE_MIDP = 1, 2, 4, 8


def pca_streamed(path: str, vidx, center, inv_stdev, n_pcs: int, g1_init, window_variants: int, sample_include=None,
                 pgi_path: str | None = None):
    """pgh_pca over a .pgen that is not made resident: windows of at most `window_variants` variants, one at a time.
    sample_include: uint64 words of the subset's bit mask over the raw samples (None: everybody)."""
    vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
    center = np.ascontiguousarray(center, dtype=np.float64)
    inv_stdev = np.ascontiguousarray(inv_stdev, dtype=np.float64)
    g1_init = np.ascontiguousarray(g1_init, dtype=np.float64)
    n_out = g1_init.shape[0]
    mask = None if sample_include is None else np.ascontiguousarray(sample_include, dtype=np.uint64)
    ev = np.zeros(n_pcs, dtype=np.float64)
    vec = np.zeros((n_out, n_pcs), dtype=np.float64)
    eb = _errbuf()
    rc = raw().pgh_pca_streamed(path.encode(), pgi_path.encode() if pgi_path else None, None if mask is None else _ptr(mask),
                                len(vidx), _ptr(vidx), _ptr(center), _ptr(inv_stdev), n_pcs, _ptr(g1_init),
                                int(window_variants), _ptr(ev), _ptr(vec), eb)
    _check(rc, eb)
    return ev, vec


def trim_device_cache():
    """Hand the call-scoped work blocks the library keeps between calls back to the driver."""
    raw().pgh_trim_device_cache()


def tally_passes_started() -> int:
    return int(_lib.pgh_tally_passes_started())


class TallyPass:
    """pgh_tally: one asynchronous walk of [v_begin, v_end) whose products land in pinned host memory."""

    def __init__(self, ds: "Dataset", v_begin: int = 0, v_end: int | None = None, products: int = TALLY_COUNTS,
                 subset: "Subset | None" = None):
        self.ds, self.subset = ds, subset
        self.v_begin = v_begin
        self.v_end = ds.info.variant_end if v_end is None else v_end
        self.n_out = subset.size if subset is not None else ds.info.raw_sample_ct
        self._h = C.c_void_p()
        eb = _errbuf()
        _check(_lib.pgh_tally_start(ds._h, subset._h if subset is not None else None, self.v_begin, self.v_end,
                                    products, C.byref(self._h), eb), eb)

    def request(self, products: int):
        eb = _errbuf()
        _check(_lib.pgh_tally_request(self._h, products, eb), eb)

    def wait(self, products: int = TALLY_COUNTS, v_begin: int | None = None, v_end: int | None = None):
        eb = _errbuf()
        _check(_lib.pgh_tally_wait(self._h, products, self.v_begin if v_begin is None else v_begin,
                                   self.v_end if v_end is None else v_end, eb), eb)

    def _view(self, addr, dtype, shape):
        n = int(np.prod(shape))
        if n == 0:
            return np.zeros(shape, dtype=dtype)
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(addr)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def counts(self) -> np.ndarray:
        """uint32[v_end - v_begin][4], a copy of the pass's pinned array (everything waited for)."""
        self.wait(TALLY_COUNTS)
        return self._view(_lib.pgh_tally_counts(self._h), np.uint32, (self.v_end - self.v_begin, 4)).copy()

    def hwe_lnp(self, midp: bool = False) -> np.ndarray:
        bit = TALLY_HWE_MIDP if midp else TALLY_HWE
        self.request(bit)
        self.wait(bit)
        return self._view(_lib.pgh_tally_hwe_lnp(self._h, 1 if midp else 0), np.float64,
                          (self.v_end - self.v_begin,)).copy()

    def sample_missing(self) -> np.ndarray:
        out = np.zeros(self.n_out, dtype=np.uint32)
        eb = _errbuf()
        _check(_lib.pgh_tally_sample_missing(self._h, _ptr(out), eb), eb)
        return out

    def close(self):
        if self._h:
            _lib.pgh_tally_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        self.close()


class Subset:
    def __init__(self, ds: "Dataset", include_mask: np.ndarray):
        """include_mask: bool[N]."""
        include_mask = np.asarray(include_mask, dtype=bool)
        n = ds.info.raw_sample_ct
        assert include_mask.shape == (n,)
        words = np.zeros((n + 63) // 64, dtype=np.uint64)
        bits = np.packbits(include_mask, bitorder="little")
        words.view(np.uint8)[: len(bits)] = bits
        self._h = C.c_void_p()
        self.ds = ds
        eb = _errbuf()
        _check(_lib.pgh_subset_create(ds._h, _ptr(words), C.byref(self._h), eb), eb)
        self.size = _lib.pgh_subset_size(self._h)

    def close(self):
        if self._h:
            _lib.pgh_subset_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        self.close()


class Dataset:
    """A genotype matrix resident in HBM (pgh_dataset)."""

    def __init__(self, handle):
        self._h = handle
        self.info = PghInfo()
        _lib.pgh_get_info(self._h, C.byref(self.info))

    @classmethod
    def open(cls, path: str, pgi_path: str | None = None, variant_begin: int = 0, variant_end: int | None = None):
        h = C.c_void_p()
        eb = _errbuf()
        _check(_lib.pgh_open(path.encode(), pgi_path.encode() if pgi_path else None, variant_begin,
                             0xFFFFFFFF if variant_end is None else variant_end, C.byref(h), eb), eb)
        return cls(h)

    @classmethod
    def open_sharded(cls, path: str, devices, pgi_path: str | None = None, variant_begin: int = 0,
                     variant_end: int | None = None):
        """One handle over len(devices) contiguous variant ranges of the file, shard k resident on devices[k]."""
        h, eb = C.c_void_p(), _errbuf()
        dev = np.ascontiguousarray(devices, dtype=np.int32)
        _check(_lib.pgh_open_sharded(path.encode(), pgi_path.encode() if pgi_path else None, variant_begin,
                                     0xFFFFFFFF if variant_end is None else variant_end, _ptr(dev), len(dev),
                                     C.byref(h), eb), eb)
        return cls(h)

    @classmethod
    def group(cls, shards):
        """A shard group over datasets that hold contiguous, ascending variant ranges; takes them over."""
        h, eb = C.c_void_p(), _errbuf()
        arr = (C.c_void_p * len(shards))(*[s._h for s in shards])
        _check(_lib.pgh_group_create(arr, len(shards), C.byref(h), eb), eb)
        for s in shards:
            s._h = None  # owned by the group now
        return cls(h)

    @property
    def uses_rccl(self) -> bool:
        """The group's per-sample merges are RCCL collectives (shards on distinct devices)."""
        return bool(_lib.pgh_group_uses_rccl(self._h))

    @property
    def shard_count(self) -> int:
        return int(_lib.pgh_shard_count(self._h))

    @classmethod
    def from_host_rows(cls, rows: np.ndarray, n_samples: int):
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        assert rows.ndim == 2
        h = C.c_void_p()
        eb = _errbuf()
        _check(_lib.pgh_from_host_rows(_ptr(rows), rows.shape[1], rows.shape[0], n_samples, C.byref(h), eb), eb)
        return cls(h)

    @classmethod
    def synth(cls, variant_begin: int, variant_end: int, n_samples: int, seed: int, missing_rate: float):
        h = C.c_void_p()
        eb = _errbuf()
        _check(_lib.pgh_synth_create(variant_begin, variant_end, n_samples, seed, missing_rate, C.byref(h), eb), eb)
        return cls(h)

    # -- properties -------------------------------------------------------
    @property
    def n_samples(self):
        return self.info.raw_sample_ct

    @property
    def v_begin(self):
        return self.info.variant_begin

    @property
    def v_end(self):
        return self.info.variant_end

    @property
    def device_rows(self) -> int:
        return _lib.pgh_device_rows(self._h)

    def copy_rows_to_host(self, v_begin: int, v_end: int) -> np.ndarray:
        rows = np.zeros((max(0, v_end - v_begin), self.info.record_bytes), dtype=np.uint8)
        eb = _errbuf()
        _check(_lib.pgh_copy_rows_to_host(self._h, v_begin, v_end, _ptr(rows), self.info.record_bytes, eb), eb)
        return rows

    def subset(self, include_mask) -> Subset:
        return Subset(self, include_mask)

    def close(self):
        if self._h:
            _lib.pgh_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        self.close()

    # -- batched calls ------------------------------------------------------
    def counts_range(self, v_begin=None, v_end=None, subset: Subset | None = None) -> np.ndarray:
        v_begin = self.v_begin if v_begin is None else v_begin
        v_end = self.v_end if v_end is None else v_end
        out = np.zeros((max(0, v_end - v_begin), 4), dtype=np.uint32)
        eb = _errbuf()
        _check(_lib.pgh_counts_range(self._h, subset._h if subset else None, v_begin, v_end, _ptr(out), eb), eb)
        return out

    def counts_range_dev(self, v_begin, v_end, d_out: int, stream: int = 0, subset: Subset | None = None):
        eb = _errbuf()
        _check(_lib.pgh_counts_range_dev(self._h, subset._h if subset else None, v_begin, v_end, d_out, stream, eb), eb)

    def missing_per_sample(self, v_begin=None, v_end=None, subset: Subset | None = None) -> np.ndarray:
        v_begin = self.v_begin if v_begin is None else v_begin
        v_end = self.v_end if v_end is None else v_end
        n_out = subset.size if subset else self.n_samples
        out = np.zeros(n_out, dtype=np.uint32)
        eb = _errbuf()
        _check(_lib.pgh_missing_per_sample(self._h, subset._h if subset else None, v_begin, v_end, _ptr(out), eb), eb)
        return out

    def fused_tally_dev(self, v_begin, v_end, d_counts: int, d_missing: int, stream: int = 0):
        eb = _errbuf()
        _check(_lib.pgh_fused_tally_dev(self._h, v_begin, v_end, d_counts, d_missing, stream, eb), eb)

    def missing_per_sample_dev(self, v_begin, v_end, d_out: int, stream: int = 0):
        eb = _errbuf()
        _check(_lib.pgh_missing_per_sample_dev(self._h, v_begin, v_end, d_out, stream, eb), eb)

    def unpack_range(self, v_begin=None, v_end=None, subset: Subset | None = None, missing_code: int = -9,
                     want_validity: bool = True):
        v_begin = self.v_begin if v_begin is None else v_begin
        v_end = self.v_end if v_end is None else v_end
        n_out = subset.size if subset else self.n_samples
        rows = max(0, v_end - v_begin)
        out = np.zeros((rows, n_out), dtype=np.int8)
        val = np.zeros((rows, (n_out + 63) // 64), dtype=np.uint64) if want_validity else None
        eb = _errbuf()
        _check(_lib.pgh_unpack_range(self._h, subset._h if subset else None, v_begin, v_end, _ptr(out), _ptr(val),
                                     missing_code, eb), eb)
        return out, val

    def unpack_range_dev(self, v_begin, v_end, d_out: int, out_pitch: int, d_validity: int, missing_code: int = 0,
                         stream: int = 0, subset: Subset | None = None):
        eb = _errbuf()
        _check(_lib.pgh_unpack_range_dev(self._h, subset._h if subset else None, v_begin, v_end, d_out, out_pitch,
                                         d_validity, missing_code, stream, eb), eb)

    def score(self, vidx, weights, flip=None, mode: int = SCORE_MEAN_IMPUTE, subset: Subset | None = None,
              want_dosage_sum: bool = True, counts=None):
        """counts (optional): uint32[n_scored][4], the scored variants' class tallies (pgh_score_counts)."""
        vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
        weights = np.ascontiguousarray(weights, dtype=np.float64)
        if weights.ndim == 1:
            weights = weights.reshape(-1, 1)
        n_scored, n_cols = weights.shape
        assert len(vidx) == n_scored
        flip_a = None if flip is None else np.ascontiguousarray(flip, dtype=np.uint8)
        n_out = subset.size if subset else self.n_samples
        score = np.zeros((n_out, n_cols), dtype=np.float64)
        dos = np.zeros(n_out, dtype=np.float64) if want_dosage_sum else None
        ac = np.zeros(n_out, dtype=np.uint32)
        eb = _errbuf()
        if counts is not None:
            counts = np.ascontiguousarray(counts, dtype=np.uint32).reshape(n_scored, 4)
            _check(_lib.pgh_score_counts(self._h, subset._h if subset else None, n_scored, _ptr(vidx), _ptr(weights),
                                         _ptr(flip_a), n_cols, mode, _ptr(counts), _ptr(score), _ptr(dos), _ptr(ac), eb), eb)
            return score, dos, ac
        _check(_lib.pgh_score(self._h, subset._h if subset else None, n_scored, _ptr(vidx), _ptr(weights), _ptr(flip_a),
                              n_cols, mode, _ptr(score), _ptr(dos), _ptr(ac), eb), eb)
        return score, dos, ac

    def score_dev(self, vidx, weights, d_score: int, d_dosage: int, d_allele: int, flip=None,
                  mode: int = SCORE_MEAN_IMPUTE, stream: int = 0, subset: Subset | None = None):
        vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
        weights = np.ascontiguousarray(weights, dtype=np.float64)
        if weights.ndim == 1:
            weights = weights.reshape(-1, 1)
        flip_a = None if flip is None else np.ascontiguousarray(flip, dtype=np.uint8)
        eb = _errbuf()
        _check(_lib.pgh_score_dev(self._h, subset._h if subset else None, weights.shape[0], _ptr(vidx), _ptr(weights),
                                  _ptr(flip_a), weights.shape[1], mode, d_score, d_dosage, d_allele, stream, eb), eb)

    def score_plan(self, vidx, weights, flip=None, mode: int = SCORE_MEAN_IMPUTE, subset: Subset | None = None):
        return ScorePlan(self, vidx, weights, flip, mode, subset)

    def pca(self, vidx, center, inv_stdev, n_pcs: int, g1_init, subset: Subset | None = None):
        vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
        center = np.ascontiguousarray(center, dtype=np.float64)
        inv_stdev = np.ascontiguousarray(inv_stdev, dtype=np.float64)
        g1 = np.ascontiguousarray(g1_init, dtype=np.float64)
        n_out = subset.size if subset else self.n_samples
        assert g1.shape == (n_out, 2 * n_pcs)
        ev = np.zeros(n_pcs, dtype=np.float64)
        vecs = np.zeros((n_out, n_pcs), dtype=np.float64)
        eb = _errbuf()
        _check(_lib.pgh_pca(self._h, subset._h if subset else None, len(vidx), _ptr(vidx), _ptr(center),
                            _ptr(inv_stdev), n_pcs, _ptr(g1), _ptr(ev), _ptr(vecs), eb), eb)
        return ev, vecs

    def sample_counts(self, v_begin: int | None = None, v_end: int | None = None, vidx=None,
                      subset: Subset | None = None) -> np.ndarray:
        """uint32[n_out][4] = {hom_ref, het, hom_alt, missing} per sample over a variant range or list."""
        n_out = subset.size if subset else self.n_samples
        out = np.zeros((n_out, 4), dtype=np.uint32)
        eb = _errbuf()
        if vidx is not None:
            v = np.ascontiguousarray(vidx, dtype=np.uint32)
            _check(_lib.pgh_sample_counts(self._h, subset._h if subset else None, 0, len(v), _ptr(v), _ptr(out), eb), eb)
        else:
            v0 = self.v_begin if v_begin is None else v_begin
            v1 = self.v_end if v_end is None else v_end
            _check(_lib.pgh_sample_counts(self._h, subset._h if subset else None, v0, v1 - v0, None, _ptr(out), eb), eb)
        return out

    def synth_add_dosage(self, rate: float, seed: int):
        """Seeded synthetic dosage tracks on every resident variant (benchmark input)."""
        eb = _errbuf()
        _check(_lib.pgh_synth_add_dosage(self._h, rate, seed, eb), eb)
        _lib.pgh_get_info(self._h, C.byref(self.info))

    def _range_or_list(self, v_begin, v_end, vidx):
        if vidx is not None:
            v = np.ascontiguousarray(vidx, dtype=np.uint32)
            return 0, len(v), v
        v0 = self.v_begin if v_begin is None else v_begin
        v1 = self.v_end if v_end is None else v_end
        return v0, v1 - v0, None

    def dosage_sums(self, v_begin: int | None = None, v_end: int | None = None, vidx=None,
                    subset: Subset | None = None) -> np.ndarray:
        """uint64[n][3] = {sum, sum of squares, observed samples} of the dosages (16384 per ALT copy)."""
        v0, n, v = self._range_or_list(v_begin, v_end, vidx)
        out = np.zeros((n, 3), dtype=np.uint64)
        eb = _errbuf()
        _check(_lib.pgh_dosage_sums(self._h, subset._h if subset else None, v0, n, _ptr(v) if v is not None else None,
                                    _ptr(out), eb), eb)
        return out

    def dosage_unpack(self, v_begin: int | None = None, v_end: int | None = None, vidx=None,
                      subset: Subset | None = None) -> np.ndarray:
        """float64[n][n_out]: dosages, -9 where a sample has neither a dosage nor a call."""
        v0, n, v = self._range_or_list(v_begin, v_end, vidx)
        out = np.zeros((n, subset.size if subset else self.n_samples), dtype=np.float64)
        eb = _errbuf()
        _check(_lib.pgh_dosage_unpack(self._h, subset._h if subset else None, v0, n, _ptr(v) if v is not None else None,
                                      _ptr(out), eb), eb)
        return out

    def unpack_samples(self, vidx, subset: Subset | None = None, missing_code: int = -9) -> np.ndarray:
        """int8[n_out][len(vidx)]: the calls sample-major (read_pfile orient := 'sample')."""
        v = np.ascontiguousarray(vidx, dtype=np.uint32)
        out = np.zeros((subset.size if subset else self.n_samples, len(v)), dtype=np.int8)
        eb = _errbuf()
        _check(_lib.pgh_unpack_samples(self._h, subset._h if subset else None, len(v), _ptr(v), _ptr(out), missing_code, eb), eb)
        return out

    def dosage_unpack_samples(self, vidx, subset: Subset | None = None) -> np.ndarray:
        """float64[n_out][len(vidx)]: the dosages sample-major, -9 = missing."""
        v = np.ascontiguousarray(vidx, dtype=np.uint32)
        out = np.zeros((subset.size if subset else self.n_samples, len(v)), dtype=np.float64)
        eb = _errbuf()
        _check(_lib.pgh_dosage_unpack_samples(self._h, subset._h if subset else None, len(v), _ptr(v), _ptr(out), eb), eb)
        return out

    def dosage_sums_dev(self, v_begin, v_end, d_sums: int, stream: int = 0, subset: Subset | None = None):
        eb = _errbuf()
        _check(_lib.pgh_dosage_sums_dev(self._h, subset._h if subset else None, v_begin, v_end, d_sums, stream, eb), eb)

    def dosage_unpack_dev(self, v_begin, v_end, d_out: int, out_stride: int, stream: int = 0,
                          subset: Subset | None = None):
        eb = _errbuf()
        _check(_lib.pgh_dosage_unpack_dev(self._h, subset._h if subset else None, v_begin, v_end, d_out, out_stride,
                                          stream, eb), eb)

    def sample_counts_dev(self, v_begin, v_end, d_classes: int, stream: int = 0):
        eb = _errbuf()
        _check(_lib.pgh_sample_counts_dev(self._h, v_begin, v_end, d_classes, stream, eb), eb)

    def ld_pairs_dev(self, vidx_a: np.ndarray, vidx_b: np.ndarray, d_sums: int, stream: int = 0,
                     subset: Subset | None = None):
        """vidx_a / vidx_b: contiguous uint32 host arrays (kept by the caller); sums land in d_sums."""
        eb = _errbuf()
        _check(_lib.pgh_ld_pairs_dev(self._h, subset._h if subset else None, len(vidx_a), _ptr(vidx_a), _ptr(vidx_b),
                                     d_sums, stream, eb), eb)

    def ld_pairs(self, vidx_a, vidx_b, subset: Subset | None = None) -> np.ndarray:
        """uint32[n_pairs][6] = {n, sum_a, sum_b, sum_ab, sum_a2, sum_b2} per pair."""
        a = np.ascontiguousarray(vidx_a, dtype=np.uint32)
        b = np.ascontiguousarray(vidx_b, dtype=np.uint32)
        assert a.shape == b.shape and a.ndim == 1
        out = np.zeros((len(a), 6), dtype=np.uint32)
        eb = _errbuf()
        _check(_lib.pgh_ld_pairs(self._h, subset._h if subset else None, len(a), _ptr(a), _ptr(b), _ptr(out), eb), eb)
        return out

    def pca_sharded(self, vidx, center, inv_stdev, n_var_total: int, n_pcs: int, g1_init, allreduce,
                    subset: Subset | None = None):
        """pgh_pca_sharded: this rank's effective variants + `allreduce(d_ptr, count, stream)`, a
        callable that sums `count` doubles at device pointer `d_ptr` in place over all ranks
        (see sharding.device_allreduce for the torch.distributed one)."""
        vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
        center = np.ascontiguousarray(center, dtype=np.float64)
        inv_stdev = np.ascontiguousarray(inv_stdev, dtype=np.float64)
        g1 = np.ascontiguousarray(g1_init, dtype=np.float64)
        n_out = subset.size if subset else self.n_samples
        assert g1.shape == (n_out, 2 * n_pcs)
        ev = np.zeros(n_pcs, dtype=np.float64)
        vecs = np.zeros((n_out, n_pcs), dtype=np.float64)
        failure = []

        def trampoline(_ctx, d_buf, count, stream):
            try:
                allreduce(int(d_buf), int(count), int(stream or 0))
                return 0
            except BaseException as exc:  # an exception must not unwind through the C frames
                failure.append(exc)
                return 1

        cb = ALLREDUCE_FN(trampoline)
        eb = _errbuf()
        rc = _lib.pgh_pca_sharded(self._h, subset._h if subset else None, len(vidx), _ptr(vidx), _ptr(center),
                                  _ptr(inv_stdev), n_var_total, n_pcs, _ptr(g1), cb, None, _ptr(ev), _ptr(vecs), eb)
        if failure:
            raise failure[0]
        _check(rc, eb)
        return ev, vecs

    def reader(self, subset: Subset | None = None) -> "Reader":
        return Reader(self, subset)


class ScorePlan:
    """Uploaded weights + per-variant tables of one plink_score call (pgh_score_plan)."""

    def __init__(self, ds: Dataset, vidx, weights, flip, mode, subset):
        vidx = np.ascontiguousarray(vidx, dtype=np.uint32)
        weights = np.ascontiguousarray(weights, dtype=np.float64)
        if weights.ndim == 1:
            weights = weights.reshape(-1, 1)
        flip_a = None if flip is None else np.ascontiguousarray(flip, dtype=np.uint8)
        self.ds = ds
        self.n_cols = weights.shape[1]
        self._h = C.c_void_p()
        eb = _errbuf()
        _check(_lib.pgh_score_plan_create(ds._h, subset._h if subset else None, weights.shape[0], _ptr(vidx),
                                          _ptr(weights), _ptr(flip_a), weights.shape[1], mode, C.byref(self._h), eb), eb)

    def run_dev(self, d_score: int, d_dosage: int, d_allele: int, stream: int = 0):
        eb = _errbuf()
        _check(_lib.pgh_score_run_dev(self._h, d_score, d_dosage, d_allele, stream, eb), eb)

    def close(self):
        if self._h:
            _lib.pgh_score_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        self.close()


class Reader:
    """Per-scan-thread view with pgenlib-shaped per-variant calls (pgh_reader)."""

    def __init__(self, ds: Dataset, subset: Subset | None = None):
        self.ds = ds
        self.subset = subset
        self.n_out = subset.size if subset else ds.n_samples
        self._h = C.c_void_p()
        eb = _errbuf()
        _check(_lib.pgh_reader_create(ds._h, subset._h if subset else None, C.byref(self._h), eb), eb)

    def _chk(self, rc):
        if rc != PGH_OK:
            msg = _lib.pgh_reader_error(self._h).decode("utf-8", "replace")
            raise (PghArgError if rc == PGH_ERR_ARG else PghError)(rc, msg)

    def get_counts(self, vidx: int) -> np.ndarray:
        out = np.zeros(4, dtype=np.uint32)
        self._chk(_lib.pgh_get_counts(self._h, vidx, _ptr(out)))
        return out

    def get_2bit(self, vidx: int) -> np.ndarray:
        out = np.zeros((self.n_out + 31) // 32, dtype=np.uint64)
        self._chk(_lib.pgh_get_2bit(self._h, vidx, _ptr(out)))
        return out

    def get_missingness(self, vidx: int) -> np.ndarray:
        out = np.zeros((self.n_out + 63) // 64, dtype=np.uint64)
        self._chk(_lib.pgh_get_missingness(self._h, vidx, _ptr(out)))
        return out

    def get_int8(self, vidx: int) -> np.ndarray:
        out = np.zeros(self.n_out, dtype=np.int8)
        self._chk(_lib.pgh_get_int8(self._h, vidx, _ptr(out)))
        return out

    def get_phased(self, vidx: int):
        g = np.zeros((self.n_out + 31) // 32, dtype=np.uint64)
        pp = np.zeros((self.n_out + 63) // 64, dtype=np.uint64)
        pi = np.zeros((self.n_out + 63) // 64, dtype=np.uint64)
        self._chk(_lib.pgh_get_phased(self._h, vidx, _ptr(g), _ptr(pp), _ptr(pi)))
        return g, pp, pi

    def get_dosage_f64(self, vidx: int) -> np.ndarray:
        out = np.zeros(self.n_out, dtype=np.float64)
        self._chk(_lib.pgh_get_dosage_f64(self._h, vidx, _ptr(out)))
        return out

    def close(self):
        if self._h:
            _lib.pgh_reader_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        self.close()

"""CPU-only checks of the product's host side: the C-ABI library loads and
exports every symbol include/pgenhip.h declares, the header probe and the host
record normaliser agree with the oracle on every reference fixture, and the HWE
routines agree with the oracle.  No device work happens here."""

import ctypes as C
import math
import os
import re

import numpy as np
import pytest

from conftest import ROOT, data_path

FIXTURES = ["pgen_example", "all_missing", "large_example", "streaming_example", "sexchr_example", "rare_small",
            "pca_example", "phased_example", "dosage_example", "pgen_split"]


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "pgenhip.h")).read()
    declared = set(re.findall(r"\b(pgh_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    cdll = lib.raw()
    for name in sorted(declared):
        assert hasattr(cdll, name), f"{name} declared in pgenhip.h but not exported"
    assert declared == set(lib.EXPORTED_SYMBOLS)
    assert lib.version().startswith("pgenhip")


def test_probe_matches_oracle(lib, oracle):
    for name in FIXTURES:
        info = lib.probe(data_path(name + ".pgen"))
        pg = oracle.Pgen(data_path(name + ".pgen"))
        assert (info.raw_variant_ct, info.raw_sample_ct) == (pg.M, pg.N)
        assert bool(info.has_dosage) == pg.has_dosage and bool(info.has_phase) == pg.has_phase
        hist = np.bincount([pg.vrtype(v) & 7 for v in range(pg.M)], minlength=8)
        assert list(info.vrtype_hist) == hist.tolist()
        assert info.record_bytes == (pg.N + 3) // 4 and info.pitch_bytes % 16 == 0


def test_probe_errors(lib, tmp_path):
    with pytest.raises(IOError):
        lib.probe(str(tmp_path / "nope.pgen"))
    bad = tmp_path / "bad.pgen"
    bad.write_bytes(b"\x00\x01\x02\x03" * 8)
    with pytest.raises(IOError) as e:
        lib.probe(str(bad))
    assert "magic" in str(e.value)
    trunc = tmp_path / "trunc.pgen"
    trunc.write_bytes(open(data_path("pgen_example.pgen"), "rb").read()[:20])
    with pytest.raises(IOError):
        lib.probe(str(trunc))


def pack_rows(codes):
    n = codes.shape[-1]
    pad = np.zeros(codes.shape[:-1] + (((n + 3) // 4) * 4,), dtype=np.uint8)
    pad[..., :n] = codes
    return (pad[..., 0::4] | (pad[..., 1::4] << 2) | (pad[..., 2::4] << 4) | (pad[..., 3::4] << 6)).astype(np.uint8)


@pytest.mark.parametrize("name", FIXTURES)
def test_host_normaliser_matches_oracle(lib, oracle, name):
    path = data_path(name + ".pgen")
    rows = lib.normalize_range_host(path)
    pg = oracle.Pgen(path)
    step = max(1, pg.M // 1500)
    for v in range(0, pg.M, step):
        assert np.array_equal(rows[v], pack_rows(pg.raw(v))), (name, v)


def test_host_normaliser_mid_chain_start(lib):
    """Starting inside an LD chain must resolve the base record first."""
    path = data_path("pca_example.pgen")
    full = lib.normalize_range_host(path)
    for vb in (1, 7, 133, 250, 499):
        part = lib.normalize_range_host(path, vb, min(500, vb + 9))
        assert np.array_equal(part, full[vb:vb + 9])


def test_hwe_matches_oracle(lib, oracle):
    rng = np.random.default_rng(7)
    cases = [(1, 1, 1), (2, 1, 1), (1, 2, 1), (0, 1, 1), (2, 2, 2), (0, 0, 5), (5, 0, 0), (0, 3, 0)]
    for _ in range(200):
        n = int(rng.integers(1, 3000))
        p = rng.uniform(0.01, 0.99)
        f = rng.uniform(-0.2, 0.4)  # inbreeding: pushes tables away from HWE
        probs = [(1 - p) ** 2 + f * p * (1 - p), 2 * p * (1 - p) * (1 - f), p * p + f * p * (1 - p)]
        probs = np.clip(probs, 0, None)
        hom1, het, hom2 = rng.multinomial(n, probs / probs.sum())
        cases.append((int(het), int(hom1), int(hom2)))
    for het, hom1, hom2 in cases:
        for midp in (False, True):
            a = lib.hwe_lnp(het, hom1, hom2, midp)
            b = oracle.hwe_lnp(het, hom1, hom2, midp)
            if math.isinf(b):
                assert a < -700
            else:
                assert a == pytest.approx(b, rel=1e-9, abs=1e-9), (het, hom1, hom2, midp)
    # large, biobank-sized counts
    for het, hom1, hom2 in [(49000, 26000, 25000), (120000, 300000, 80000), (249000, 125500, 125500)]:
        a, b = lib.hwe_lnp(het, hom1, hom2), oracle.hwe_lnp(het, hom1, hom2)
        assert a == pytest.approx(b, rel=1e-8, abs=1e-8)
    assert lib.hwe_lnp(0, 0, 0) == 0.0


def test_hwe_xchr_matches_oracle(lib, oracle):
    rng = np.random.default_rng(11)
    cases = [(1, 1, 1, 2, 1), (0, 0, 0, 3, 2), (2, 2, 2, 0, 0), (1, 0, 0, 0, 1)]
    for _ in range(120):
        nf, nm = int(rng.integers(0, 60)), int(rng.integers(0, 60))
        p = rng.uniform(0.05, 0.95)
        hom1, het, hom2 = rng.multinomial(nf, [(1 - p) ** 2, 2 * p * (1 - p), p * p])
        m2 = int(rng.binomial(nm, p))
        cases.append((int(het), int(hom1), int(hom2), nm - m2, m2))
    for c in cases:
        for midp in (False, True):
            a, b = lib.hwe_xchr_lnp(*c, midp), oracle.hwe_xchr_lnp(*c, midp)
            assert a == pytest.approx(b, rel=1e-8, abs=1e-8), (c, midp)


def test_synth_files_roundtrip_through_oracle(lib, oracle, tmp_path):
    """The product's .pgen writer and host generator against the oracle's reader."""
    prefix = str(tmp_path / "syn")
    M, N, seed = 300, 1003, 20260807
    lib.synth_write_files(prefix, M, N, seed, 0.02)
    pg = oracle.Pgen(prefix + ".pgen")
    assert (pg.M, pg.N) == (M, N)
    info = lib.probe(prefix + ".pgen")
    assert list(info.vrtype_hist)[0] == M
    tot = np.zeros(4, dtype=np.int64)
    for v in range(0, M, 7):
        rec = lib.synth_record_host(v, N, seed, 0.02)
        assert np.array_equal(rec, pack_rows(pg.raw(v)))
        tot += pg.counts(v)
    # generator sanity: ~2 % missing, all three genotype classes present
    assert 0.01 < tot[3] / tot.sum() < 0.03 and (tot[:3] > 0).all()
    pvar = oracle.load_pvar(prefix + ".pvar")
    psam = oracle.load_psam(prefix + ".psam")
    assert len(pvar["id"]) == M and len(psam["iid"]) == N
    rows = lib.normalize_range_host(prefix + ".pgen", 10, 20)
    assert np.array_equal(rows[3], lib.synth_record_host(13, N, seed, 0.02))


def test_synth_dosage_files_roundtrip_through_oracle(lib, oracle, tmp_path):
    """pgh_synth_write_dosage_files: records = the same 2-bit bytes + a 0x60 dosage track, variable lengths,
    tables written after the body; two 65,536-variant blocks.  Read back with the oracle."""
    prefix = str(tmp_path / "dz")
    M, N, seed = 66000, 37, 11
    lib.synth_write_dosage_files(prefix, M, N, seed, 0.05, 0.3)
    pg = oracle.Pgen(prefix + ".pgen")
    assert (pg.M, pg.N, pg.has_dosage) == (M, N, True)
    info = lib.probe(prefix + ".pgen")
    assert info.has_dosage == 1 and info.raw_variant_ct == M
    explicit = 0
    for v in (0, 1, 777, 65535, 65536, 65999):
        assert pg.vrtype(v) == 0x60
        assert np.array_equal(lib.synth_record_host(v, N, seed, 0.05), pack_rows(pg.raw(v)))  # hardcalls untouched
        d = pg.dosage(v)
        g = pg.geno(v).astype(np.float64)
        differs = d != g
        explicit += int(differs.sum())
        assert ((d == -9.0) | ((d >= 0.0) & (d <= 2.0))).all()
        assert np.array_equal(lib.normalize_range_host(prefix + ".pgen", v, v + 1)[0], pack_rows(pg.raw(v)))
    assert 0.1 < explicit / (6 * N) < 0.5  # ~30 % of the samples carry a value
    assert len(oracle.load_pvar(prefix + ".pvar")["id"]) == M and len(oracle.load_psam(prefix + ".psam")["iid"]) == N


def test_multi_block_header_roundtrip(lib, oracle, tmp_path):
    """> 65536 variants: per-block tables, parsed identically by both decoders."""
    prefix = str(tmp_path / "blocks")
    M, N = 70000, 9
    lib.synth_write_files(prefix, M, N, 5, 0.1)
    info = lib.probe(prefix + ".pgen")
    assert info.raw_variant_ct == M
    pg = oracle.Pgen(prefix + ".pgen")
    for v in (0, 65535, 65536, 69999):
        assert np.array_equal(lib.normalize_range_host(prefix + ".pgen", v, v + 1)[0], pack_rows(pg.raw(v)))
        assert np.array_equal(lib.synth_record_host(v, N, 5, 0.1), pack_rows(pg.raw(v)))


def test_no_gpu_means_loud_failure(lib):
    if lib.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(IOError):
        lib.Dataset.open(data_path("pgen_example.pgen"))
    with pytest.raises(IOError):
        lib.Dataset.synth(0, 4, 16, 1, 0.0)


def test_header_is_plain_c_and_links(tmp_path):
    """include/pgenhip.h is the boundary: it must compile as C99 (what cgo / JNI / ctypes generators
    consume) and as C++, and a C program must link against libpgenhip.so with nothing but the header."""
    import subprocess
    src = tmp_path / "host.c"
    src.write_text('#include "pgenhip.h"\n#include <stdio.h>\n'
                   'int main(void) { pgh_info info; char err[PGH_ERRBUF_LEN];\n'
                   '  int rc = pgh_probe("/nonexistent.pgen", 0, &info, err);\n'
                   '  printf("%s rc=%d %s\\n", pgh_version(), rc, err); return rc == PGH_OK; }\n')
    inc = os.path.join(ROOT, "include")
    libdir = os.path.join(ROOT, "plinking_duck_amd")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-c", str(src),
                    "-o", str(tmp_path / "host_c.o")], check=True, capture_output=True)
    subprocess.run(["g++", "-std=c++11", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", inc, "-x", "c++", "-c",
                    str(src), "-o", str(tmp_path / "host_cpp.o")], check=True, capture_output=True)
    exe = tmp_path / "host"
    subprocess.run(["gcc", str(tmp_path / "host_c.o"), "-o", str(exe), "-L", libdir, "-lpgenhip",
                    "-Wl,-rpath," + libdir], check=True, capture_output=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "pgenhip" in r.stdout and "cannot open" in r.stdout


def test_rccl_host_example_compiles(tmp_path):
    """tests/abi/rccl_host.cpp -- the RCCL all-reduce callback a C++ host hands to pgh_pca_sharded --
    compiles against pgenhip.h + rccl.h (hipcc cross-compiles without a GPU)."""
    import subprocess
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    subprocess.run([hipcc, "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), "-c",
                    os.path.join(ROOT, "tests", "abi", "rccl_host.cpp"), "-o", str(tmp_path / "rccl_host.o")],
                   check=True, capture_output=True)

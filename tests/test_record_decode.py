"""Every hardcall record type through the three decoders: the oracle, the product's
host normaliser, and the device decoder that pgh_open uses (decode.hip).

Fixtures from the reference pin types 0/1/2/3/4/6/7 with single-group difflists
(N <= 256).  Multi-group difflists and 2/3-byte sample ids come from
tests/pgen_writer.py; for those the three decoders are checked against each other
and against the matrix the writer encoded ("parity unpinned" against pgenlib)."""

import os

import numpy as np
import pytest

from conftest import data_path
import pgen_writer as W


def _pack_rows(geno):
    m, n = geno.shape
    pad = (-n) % 4
    c = np.concatenate([geno.astype(np.uint8), np.zeros((m, pad), dtype=np.uint8)], axis=1).reshape(m, -1, 4)
    return (c[:, :, 0] | (c[:, :, 1] << 2) | (c[:, :, 2] << 4) | (c[:, :, 3] << 6)).astype(np.uint8)


CASES = [  # (variants, samples, seed): 1-, 2- and 3-byte sample ids, ragged tails
    (60, 7, 1), (300, 255, 2), (300, 256, 3), (200, 1000, 4), (120, 5003, 5), (40, 70001, 6),
]


@pytest.fixture(scope="module")
def written(tmp_path_factory):
    out = {}
    root = tmp_path_factory.mktemp("pgen_writer")
    for m, n, seed in CASES:
        rng = np.random.default_rng(seed)
        geno = W.rare_matrix(m, n, rng)
        kinds = W.choose_kinds(geno, rng)
        path = str(root / f"w_{m}_{n}.pgen")
        W.write_pgen(path, geno, kinds)
        out[(m, n)] = (path, geno, kinds)
    return out


@pytest.mark.parametrize("m,n,seed", CASES)
def test_writer_roundtrips_through_oracle_and_host_normaliser(written, oracle, lib, m, n, seed):
    path, geno, kinds = written[(m, n)]
    assert set(kinds) >= {0, 1, 2, 3, 4, 6, 7} or m < 100
    pg = oracle.Pgen(path)
    assert (pg.M, pg.N) == (m, n)
    assert [pg.vrtype(v) & 7 for v in range(m)] == kinds
    for v in range(m):
        got = pg.geno(v).astype(np.int16)
        got[got == -9] = 3
        assert np.array_equal(got, geno[v]), (v, kinds[v])
    assert np.array_equal(lib.normalize_range_host(path), _pack_rows(geno))
    # a range that starts inside an LD run resolves its base by walking back
    first_ld = next((v for v in range(1, m) if kinds[v] in (2, 3)), None)
    if first_ld is not None:
        assert np.array_equal(lib.normalize_range_host(path, first_ld, m), _pack_rows(geno[first_ld:]))


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,seed", CASES)
def test_device_decode_equals_host_normaliser(written, gpu_lib, m, n, seed):
    path, geno, kinds = written[(m, n)]
    want = _pack_rows(geno)
    ds = gpu_lib.Dataset.open(path)
    assert np.array_equal(ds.copy_rows_to_host(0, m), want)
    counts = ds.counts_range()
    assert np.array_equal(counts, np.stack([(geno == c).sum(axis=1) for c in range(4)], axis=1).astype(np.uint32))
    ds.close()
    # shards: one that starts on an LD record (host rows for the leading run), one that ends early
    first_ld = next((v for v in range(1, m) if kinds[v] in (2, 3)), None)
    for v0, v1 in [(first_ld or 1, m), (0, m // 2), (m // 3, 2 * m // 3)]:
        part = gpu_lib.Dataset.open(path, variant_begin=v0, variant_end=v1)
        assert np.array_equal(part.copy_rows_to_host(v0, v1), want[v0:v1]), (v0, v1)
        part.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["pca_example", "rare_small", "streaming_example", "all_missing", "phased_example",
                                  "dosage_example", "large_example"])
def test_device_decode_on_reference_fixtures(gpu_lib, oracle, name, monkeypatch):
    path = data_path(name + ".pgen")
    pg = oracle.Pgen(path)
    host = gpu_lib.normalize_range_host(path)
    ds = gpu_lib.Dataset.open(path)
    assert np.array_equal(ds.copy_rows_to_host(0, pg.M), host)
    ds.close()
    monkeypatch.setenv("PGH_HOST_NORMALIZE", "1")
    ds = gpu_lib.Dataset.open(path)
    assert np.array_equal(ds.copy_rows_to_host(0, pg.M), host)
    ds.close()
    step = max(1, pg.M // 97)
    for v in range(0, pg.M, step):
        g = pg.geno(v).astype(np.int16)
        g[g == -9] = 3
        row = host[v]
        got = (row[np.arange(pg.N) // 4] >> (2 * (np.arange(pg.N) % 4))) & 3
        assert np.array_equal(got, g)


@pytest.mark.gpu
def test_malformed_records_are_reported_not_followed(written, gpu_lib, tmp_path):
    path, geno, kinds = written[(200, 1000)]
    blob = bytearray(open(path, "rb").read())
    m = 200
    # find a type-4 record with a difflist and claim more entries than samples
    pg_off = 12 + 8 + m * 5
    lens = [int.from_bytes(blob[12 + 8 + m + 4 * v:12 + 8 + m + 4 * v + 4], "little") for v in range(m)]
    starts = np.concatenate([[pg_off], pg_off + np.cumsum(lens)])
    victim = next(v for v in range(m) if kinds[v] == 4 and lens[v] > 3)
    bad = bytearray(blob)
    bad[starts[victim]] = 0xFF      # varint continues ...
    bad[starts[victim] + 1] = 0x7F  # ... to a length far above N
    p = str(tmp_path / "bad_len.pgen")
    open(p, "wb").write(bad)
    with pytest.raises(gpu_lib.PghError) as e:
        gpu_lib.Dataset.open(p)
    assert f"malformed variant record {victim}" in str(e.value)
    # a sample id past N in a group's first-id slot
    victim = next(v for v in range(m) if kinds[v] == 6 and lens[v] > 6 and (geno[v] != 2).sum() >= 2)
    bad = bytearray(blob)
    at = starts[victim] + 1  # one-byte length, then the first group's 2-byte id
    bad[at:at + 2] = (60000).to_bytes(2, "little")
    p = str(tmp_path / "bad_id.pgen")
    open(p, "wb").write(bad)
    with pytest.raises(gpu_lib.PghError) as e:
        gpu_lib.Dataset.open(p)
    assert f"malformed variant record {victim}" in str(e.value)


@pytest.mark.gpu
def test_device_decoder_survives_corrupt_record_bytes(written, gpu_lib, tmp_path):
    """Record bytes mutated at random (header and tables intact): pgh_open either loads the file or
    reports a malformed record -- the decode kernels bounds-check every read, write only inside their
    own row, and every loop consumes input.  The pristine file still opens afterwards."""
    path, geno, kinds = written[(200, 1000)]
    blob = bytearray(open(path, "rb").read())
    m = 200
    body = 12 + 8 + m * 5
    rng = np.random.default_rng(99)
    p = str(tmp_path / "mut.pgen")
    opened = failed = 0
    for it in range(250):
        bad = bytearray(blob)
        for _ in range(int(rng.integers(1, 6))):
            at = int(rng.integers(body, len(bad)))
            bad[at] = int(rng.integers(0, 256)) if rng.random() < 0.7 else 0xFF
        with open(p, "wb") as f:
            f.write(bad)
        try:
            ds = gpu_lib.Dataset.open(p)
            counts = ds.counts_range()
            assert (counts.sum(axis=1) == 1000).all()  # whatever was decoded, every row is a full row of calls
            ds.close()
            opened += 1
        except gpu_lib.PghError as e:
            assert "malformed variant record" in str(e)
            failed += 1
    assert opened > 0 and failed > 0
    ds = gpu_lib.Dataset.open(path)
    assert np.array_equal(ds.copy_rows_to_host(0, m), _pack_rows(geno))


# ---- multiallelic records (vrtype bit 0x08) ----------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("n", [37, 300, 5003])
def test_multiallelic_records_are_read_with_their_alt_alleles_collapsed(gpu_lib, oracle, tmp_path, n, monkeypatch):
    """A file with multiallelic variants used to be refused whole (PGH_ERR_UNSUPPORTED).  PgrGet / PgrGetCounts /
    PgrGetD read such a variant with its ALT alleles collapsed -- the main track as stored (src/pgen_reader.cpp:727,
    src/plink_freq.cpp:482) -- so the multiallelic track only has to be stepped over to reach the phase and dosage
    tracks behind it.  No reference fixture holds one (parity unpinned): the writer, the oracle, the host parser and
    the device decoder are held to the encoded matrix and to each other, on every record type, with and without
    phase / dosage tracks behind the multiallelic one, for 3, 4, 6, 9 and 20 alleles."""
    L = gpu_lib
    rng = np.random.default_rng(n)
    m = 90
    geno = W.rare_matrix(m, n, rng)
    geno[::3] = rng.integers(0, 4, size=geno[::3].shape, dtype=np.uint8)  # rows with many 1s and 2s to patch
    kinds = W.choose_kinds(geno, rng)
    alleles = [int(rng.choice([2, 2, 3, 4, 6, 9, 20])) for _ in range(m)]
    dos = np.full((m, n), 0xFFFF, dtype=np.uint16)
    dkinds = [int(rng.choice([0, 0, 0x20, 0x40, 0x60])) for _ in range(m)]
    for v in range(m):
        if dkinds[v]:
            hit = rng.random(n) < 0.3
            dos[v, hit] = rng.integers(0, 32769, hit.sum())
            if dkinds[v] == 0x40:
                dos[v, (geno[v] != 3) & ~hit] = 0xFFFF
    path = str(tmp_path / "multi.pgen")
    W.write_pgen(path, geno, kinds, dosage=dos, dosage_kinds=dkinds, phase_rng=np.random.default_rng(5), allele_cts=alleles,
                 aux1_rng=np.random.default_rng(6))
    pg = oracle.Pgen(path)
    multi = [v for v in range(m) if alleles[v] > 2]
    assert multi and all(pg.vrtype(v) & 0x08 for v in multi) and not any(pg.vrtype(v) & 0x08 for v in range(m) if alleles[v] == 2)
    assert any(pg.vrtype(v) & 0x10 for v in multi) and any(pg.vrtype(v) & 0x60 for v in multi)
    want_calls = np.where(geno == 3, -9, geno).astype(np.int8)
    for host_only in ("0", "1"):
        monkeypatch.setenv("PGH_HOST_NORMALIZE", host_only)
        ds = L.Dataset.open(path)
        assert np.array_equal(ds.counts_range(), pg.counts_range())
        out, _ = ds.unpack_range(missing_code=-9)
        assert np.array_equal(out, want_calls)
        rd = ds.reader()
        d_all = ds.dosage_unpack()
        for v in range(m):
            assert np.array_equal(pg.geno(v), want_calls[v])
            # the tracks BEHIND the multiallelic one: found by both decoders at the same place
            g2, pp, pi = rd.get_phased(v)
            og, opp, opi = pg.phase(v)
            bits = lambda words: np.unpackbits(words.view(np.uint8), bitorder="little")[:n]
            assert np.array_equal(bits(pp), opp) and np.array_equal(bits(pi) & bits(pp), opi & opp), v
            assert np.array_equal(d_all[v], pg.dosage(v)), v
            have = dos[v] != 0xFFFF
            assert np.array_equal(d_all[v][have], dos[v][have] / 16384.0)
    monkeypatch.delenv("PGH_HOST_NORMALIZE")


def test_multiallelic_records_on_the_host(lib, oracle, tmp_path):
    """The CPU half of the test above: the oracle and the product's host normaliser read a file with multiallelic
    records (every allele count class, both selector forms) and find the dosage track behind the multiallelic one."""
    rng = np.random.default_rng(11)
    m, n = 60, 700
    geno = rng.integers(0, 4, size=(m, n), dtype=np.uint8)
    kinds = W.choose_kinds(geno, rng)
    alleles = [int(rng.choice([2, 3, 4, 5, 7, 18, 19, 40])) for _ in range(m)]
    dos = np.full((m, n), 0xFFFF, dtype=np.uint16)
    dkinds = [int(rng.choice([0, 0x20, 0x60])) for _ in range(m)]
    for v in range(m):
        if dkinds[v]:
            hit = rng.random(n) < 0.2
            dos[v, hit] = rng.integers(0, 32769, hit.sum())
    path = str(tmp_path / "multi_host.pgen")
    W.write_pgen(path, geno, kinds, dosage=dos, dosage_kinds=dkinds, allele_cts=alleles, aux1_rng=np.random.default_rng(3))
    pg = oracle.Pgen(path)
    assert lib.probe(path).raw_variant_ct == m
    assert np.array_equal(lib.normalize_range_host(path), _pack_rows(geno))
    for v in range(m):
        assert np.array_equal(pg.geno(v), np.where(geno[v] == 3, -9, geno[v].astype(np.int8)))
        d = pg.dosage(v)
        have = dos[v] != 0xFFFF
        assert np.array_equal(d[have], dos[v][have] / 16384.0)
        assert np.array_equal(d[~have], np.where(geno[v][~have] == 3, -9.0, geno[v][~have].astype(np.float64)))

"""The tally pass (pgh_tally_*): one asynchronous walk of the matrix that serves plink_freq, plink_hardy and
plink_missing -- through the C ABI against the oracle, and through the shells (one pass for three functions,
the `synth:` source against the same fileset on disk, config 1 at its stated shape)."""

import numpy as np
import pytest

from conftest import data_path

pytestmark = pytest.mark.gpu

F = pytest.importorskip("plinking_duck_amd.functions")

SEED = 20260807
FIXTURES = ["pgen_example", "all_missing", "large_example", "rare_small", "pca_example", "dosage_example", "pgen_split"]


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(gpu_lib):
    return gpu_lib


def host_pgen(oracle, rows, n):
    m = len(rows)
    head = bytes([0x6c, 0x1b, 0x02]) + m.to_bytes(4, "little") + n.to_bytes(4, "little") + bytes([0x40])
    return oracle.Pgen(mem=np.frombuffer(head + rows.tobytes(), dtype=np.uint8))


# ---- C ABI -----------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("name", FIXTURES)
def test_pass_over_the_reference_fixtures(gpu_lib, oracle, name):
    L = gpu_lib
    path = data_path(name + ".pgen")
    ds, pg = L.Dataset.open(path), oracle.Pgen(path)
    t = L.TallyPass(ds, products=L.TALLY_COUNTS | L.TALLY_SAMPLE_MISSING | L.TALLY_HWE)
    counts = t.counts()
    assert np.array_equal(counts, pg.counts_range())
    assert np.array_equal(t.sample_missing(), pg.missing_per_sample())
    for midp in (False, True):  # midp asked for after the start
        lnp = t.hwe_lnp(midp)
        for v in range(min(pg.M, 200)):
            c = counts[v]
            assert lnp[v] == pytest.approx(oracle.hwe_lnp(int(c[1]), int(c[0]), int(c[2]), midp), abs=1e-6)
    t.close()


@pytest.mark.parametrize("m,n", [(1, 1), (37, 257), (5000, 1003), (70_000, 131)])
def test_pass_on_synthetic_matrices_with_subsets_and_ranges(gpu_lib, oracle, m, n):
    L = gpu_lib
    ds = L.Dataset.synth(0, m, n, SEED, 0.05)
    pg = host_pgen(oracle, np.stack([L.synth_record_host(v, n, SEED, 0.05) for v in range(min(m, 6000))]), n)
    mo = pg.M  # the oracle sees the first rows; the pass is checked there and against the library beyond
    rng = np.random.default_rng(m * 1000 + n)
    # whole range, no subset: counts and the per-sample tally come out of one kernel pass
    t = L.TallyPass(ds, products=L.TALLY_SAMPLE_MISSING)
    assert np.array_equal(t.counts()[:mo], pg.counts_range())
    assert np.array_equal(t.counts(), ds.counts_range())
    assert np.array_equal(t.sample_missing(), ds.missing_per_sample())
    if mo == m:
        assert np.array_equal(t.sample_missing(), pg.missing_per_sample())
    # the per-sample product asked for after the start (a sweep of its own), and a sub-range
    a, b = (0, m) if m < 3 else (m // 5, m - m // 7)
    t2 = L.TallyPass(ds, a, b)
    t2.wait(L.TALLY_COUNTS, a, min(b, a + 1))
    with pytest.raises(ValueError):
        t2.wait(L.TALLY_SAMPLE_MISSING)
    assert np.array_equal(t2.sample_missing(), ds.missing_per_sample(a, b))
    assert np.array_equal(t2.counts(), ds.counts_range(a, b))
    # subsets
    if n > 1:
        mask = rng.random(n) < 0.4
        mask[0] = True
        ss = ds.subset(mask)
        t3 = L.TallyPass(ds, a, b, products=L.TALLY_SAMPLE_MISSING | L.TALLY_HWE_MIDP, subset=ss)
        assert np.array_equal(t3.counts(), ds.counts_range(a, b, subset=ss))
        if b <= mo:
            assert np.array_equal(t3.counts(), pg.counts_range(a, b, include=mask))
            assert np.array_equal(t3.sample_missing(), pg.missing_per_sample(a, b, include=mask))
        assert np.array_equal(t3.sample_missing(), ds.missing_per_sample(a, b, subset=ss))
        assert np.allclose(t3.hwe_lnp(True), L.hwe_lnp_batch(t3.counts(), True), rtol=0, atol=1e-12)
    with pytest.raises(ValueError):
        L.TallyPass(ds, 0, m + 1)


def test_pass_over_a_shard_group(gpu_lib, monkeypatch):
    L = gpu_lib
    monkeypatch.setenv("PGH_TALLY_BATCH", "4096")
    m, n = 9000, 1501
    one = L.Dataset.synth(0, m, n, SEED, 0.03)
    cuts = [0, 2000, 2001, 7000, m]
    grp = L.Dataset.group([L.Dataset.synth(a, b, n, SEED, 0.03) for a, b in zip(cuts, cuts[1:])])
    t = L.TallyPass(grp, 500, 8000, products=L.TALLY_SAMPLE_MISSING | L.TALLY_HWE)
    assert np.array_equal(t.counts(), one.counts_range(500, 8000))
    assert np.array_equal(t.sample_missing(), one.missing_per_sample(500, 8000))
    assert np.array_equal(t.hwe_lnp(False), L.hwe_lnp_batch(one.counts_range(500, 8000), False))
    mask = np.arange(n) % 3 != 0
    t = L.TallyPass(grp, products=L.TALLY_SAMPLE_MISSING, subset=grp.subset(mask))
    assert np.array_equal(t.counts(), one.counts_range(subset=one.subset(mask)))
    assert np.array_equal(t.sample_missing(), one.missing_per_sample(subset=one.subset(mask)))


def test_many_threads_read_one_pass(gpu_lib, monkeypatch):
    """Scan threads wait for different batches of one pass while it is still running."""
    import threading
    L = gpu_lib
    monkeypatch.setenv("PGH_TALLY_BATCH", "20480")  # 15 batches here (a batch is ~16 GB of rows otherwise)
    m, n = 300_000, 4001
    ds = L.Dataset.synth(0, m, n, SEED, 0.02)
    expect = ds.counts_range()
    t = L.TallyPass(ds, products=L.TALLY_SAMPLE_MISSING | L.TALLY_HWE)
    base = t._view(L.raw().pgh_tally_counts(t._h), np.uint32, (m, 4))
    bad = []

    def worker(k):
        for b in range(k * 16384, m, 8 * 16384):
            e = min(m, b + 16384)
            t.wait(L.TALLY_COUNTS | L.TALLY_HWE, b, e)
            if not np.array_equal(base[b:e], expect[b:e]):
                bad.append(b)

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(8)]
    [x.start() for x in threads]
    [x.join() for x in threads]
    assert not bad
    assert np.array_equal(t.sample_missing(), ds.missing_per_sample())


# ---- shells ------------------------------------------------------------------------------------------------------

def test_one_pass_serves_freq_hardy_and_missing(gpu_lib, oracle, tmp_path):
    """BASELINE config 3 from SQL: three functions (four scans in the reference) for one walk of the matrix."""
    L = gpu_lib
    prefix = str(tmp_path / "c3")
    m, n = 50_000, 2003
    L.synth_write_files(prefix, m, n, SEED, 0.04)
    path = prefix + ".pgen"
    pg = oracle.Pgen(path)
    counts = pg.counts_range()
    key = lambda c, p: (int(c) - 1) * ((m + 21) // 22) + p // 100 - 1
    before = L.tally_passes_started()
    r = F.query("plink_freq", path, counts=True, threads=5, columns=["CHROM", "POS", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "MISSING_CT"])
    assert L.tally_passes_started() == before + 1
    got = np.zeros((m, 4), dtype=np.uint32)
    for c, p, *cc in r.rows:
        got[key(c, p)] = cc
    assert len(r) == m and np.array_equal(got, counts)
    # ... and nothing of the matrix is read for these three
    r = F.query("plink_hardy", path, threads=5, columns=["CHROM", "POS", "HET_CT", "P_HWE"])
    for c, p, het, pv in r.rows[::97]:
        v = key(c, p)
        assert het == counts[v][1] and pv == pytest.approx(oracle.hardy_from_counts(counts[v])[2], rel=1e-6)
    r = F.query("plink_missing", path, threads=5, columns=["CHROM", "POS", "MISSING_CT"])
    assert all(counts[key(c, p)][3] == mc for c, p, mc in r.rows) and len(r) == m
    r = F.query("plink_missing", path, mode="sample", threads=5, columns=["IID", "MISSING_CT", "OBS_CT"])
    miss = pg.missing_per_sample()
    assert len(r) == n and all(miss[int(i[1:])] == mc and oc == m - mc for i, mc, oc in r.rows)
    r = F.query("plink_hardy", path, midp=True, region="3:1-100000000", threads=3, columns=["POS", "P_HWE"])
    assert len(r) == (m + 21) // 22
    assert L.tally_passes_started() == before + 1
    # the option off: every call walks for itself, same answers
    r2 = F.query("plink_missing", path, mode="sample", threads=2, settings={"plinking_tally_cache": False},
                 columns=["IID", "MISSING_CT"])
    assert L.tally_passes_started() == before + 2
    assert dict(r2.rows) == {f"S{s}": int(miss[s]) for s in range(n)}
    # a subset gets a pass of its own; a region inside a cached range does not
    F.query("plink_freq", path, samples=[0, 5, 6], columns=["ID", "ALT_FREQ"])
    assert L.tally_passes_started() == before + 3
    r = F.query("plink_freq", path, region="2:1-100000000", counts=True, columns=["POS", "HET_CT"])
    assert L.tally_passes_started() == before + 3
    per = (m + 21) // 22
    assert all(counts[per + p // 100 - 1][1] == h for p, h in r.rows) and len(r) == per


def test_synth_source_equals_the_fileset_on_disk(gpu_lib, tmp_path):
    L = gpu_lib
    m, n = 20_000, 1003
    prefix = str(tmp_path / "s")
    L.synth_write_files(prefix, m, n, SEED, 0.02)
    spec = f"synth:{m}x{n}:{SEED}:0.02"
    w = [0.001 * ((i * 7919) % 2001 - 1000) for i in range(m)]
    calls = [("plink_freq", dict(counts=True)), ("plink_hardy", dict(midp=True)), ("plink_missing", {}),
             ("plink_missing", dict(mode="sample")), ("plink_score", dict(weights=w)),
             ("read_pgen", dict(genotypes="counts", af_range={"min": 0.2})),
             ("plink_freq", dict(region="5:1-50000", samples=[1, 2, 3, 500]))]
    for fn, kw in calls:
        a = F.query(fn, prefix + ".pgen", threads=4, **kw)
        b = F.query(fn, spec, threads=4, **kw)
        assert a.names == b.names and len(a) == len(b) > 0
        key = lambda r: tuple(str(x) for x in r[:3])
        ra, rb = sorted(a.rows, key=key), sorted(b.rows, key=key)
        if fn == "plink_score":  # the contraction's last additions are FP64 atomics: equal to the order of the sums
            for x, y in zip(ra, rb):
                assert x[:4] == y[:4] and np.allclose(x[4:], y[4:], rtol=1e-12, atol=1e-9), fn
        else:
            assert ra == rb, fn
        d = F.query(fn, spec, threads=4, drain=True, **kw)
        assert len(d) == len(a) and d.rows == [] and d.checksum is not None
    with pytest.raises(F.InvalidInputException):
        F.query("plink_freq", "synth:12")


def test_config_1_at_its_stated_shape(gpu_lib, oracle, tmp_path):
    """BASELINE config 1: plink_freq on a synthetic 10,000-variant x 1,000-sample .pgen on disk -- the oracle's
    multi-threaded scan of the file (the reference's scan structure) against plink_freq through the shell and
    against pgh_counts_range, bit-exact / ALT_FREQ exact."""
    L = gpu_lib
    m, n = 10_000, 1_000
    prefix = str(tmp_path / "cfg1")
    L.synth_write_files(prefix, m, n, SEED, 0.02)
    pg = oracle.Pgen(prefix + ".pgen")
    want = pg.scan_counts_mt(0, m, 4)
    assert np.array_equal(want, pg.counts_range())
    assert np.array_equal(L.Dataset.open(prefix + ".pgen").counts_range(), want)
    r = F.query("plink_freq", prefix + ".pgen", counts=True, threads=8,
                columns=["ID", "ALT_FREQ", "OBS_CT", "HOM_REF_CT", "HET_CT", "HOM_ALT_CT", "MISSING_CT"])
    assert len(r) == m
    for vid, af, obs, *cc in r.rows:
        v = int(vid[2:])
        exp_af, exp_obs = oracle.freq_from_counts(want[v])
        assert cc == [int(x) for x in want[v]] and obs == exp_obs and af == exp_af


def test_score_from_a_pass_s_counts(gpu_lib, oracle):
    """pgh_score_counts: the tallies of a pass in the place of the plan's own read of the rows."""
    L = gpu_lib
    m, n = 3000, 2003
    ds = L.Dataset.synth(0, m, n, SEED, 0.05)
    pg = host_pgen(oracle, np.stack([L.synth_record_host(v, n, SEED, 0.05) for v in range(m)]), n)
    rng = np.random.default_rng(11)
    vidx = np.sort(rng.choice(m, 1200, replace=False)).astype(np.uint32)
    w = rng.normal(size=(len(vidx), 3))
    flip = (rng.random(len(vidx)) < 0.3).astype(np.uint8)
    mask = rng.random(n) < 0.6
    for subset, include in ((None, None), (ds.subset(mask), mask)):
        t = L.TallyPass(ds, subset=subset)
        counts = t.counts()[vidx]
        for mode, name in ((L.SCORE_MEAN_IMPUTE, "default"), (L.SCORE_NO_MEAN_IMPUTATION, "no_mean_imputation"),
                           (L.SCORE_CENTER, "center")):
            a = ds.score(vidx, w, flip, mode, subset=subset)
            b = ds.score(vidx, w, flip, mode, subset=subset, counts=counts)
            assert np.allclose(a[0], b[0], rtol=1e-12, atol=1e-9) and np.allclose(a[1], b[1], rtol=1e-12, atol=1e-9)
            assert np.array_equal(a[2], b[2])
            exp = oracle.score(pg, vidx, w[:, 0], flip, name, include=include)
            assert np.allclose(b[0][:, 0], exp[0][:, 0], rtol=1e-9, atol=1e-9) and np.array_equal(b[2], exp[2])


@pytest.mark.parametrize("threads", [3, 11])
def test_a_file_beyond_the_hbm_budget_streams_its_tallies(gpu_lib, oracle, tmp_path, monkeypatch, threads):
    """A file whose rows exceed the HBM budget is not made resident: plink_freq / plink_hardy / plink_missing (both
    modes) / read_pgen's counts get their tallies from a pass that walks the file window by window through HBM,
    read_pfile's per-sample counts add over the windows, read_pgen's / read_pfile's hardcall output unpacks one window
    at a time -- the reference's own functions stream the file too -- and what needs the whole matrix at once (scores,
    PCA, LD, the sample-orient matrix, dosage and phase tracks) says that it does not fit instead of failing in an
    allocation."""
    L = gpu_lib
    m, n = 6000, 2003
    small = str(tmp_path / "fits")
    big = str(tmp_path / "too_big")
    L.synth_write_files(small, m, n, SEED, 0.04)
    L.synth_write_files(big, m, n, SEED, 0.04)
    calls = [("plink_freq", dict(counts=True)), ("plink_hardy", dict(midp=True)), ("plink_hardy", {}), ("plink_missing", {}),
             ("plink_missing", dict(mode="sample")), ("plink_freq", dict(samples=[0, 3, 700, 2002], region="4:1-1000000")),
             ("read_pgen", dict(genotypes="counts", af_range={"max": 0.2})),
             # read_pfile's per-sample tallies (the reference streams this mode: no matrix, src/pfile_reader.cpp:3287):
             # they add over the windows, with subsets, regions, count filters, variant lists and carrier row-skips
             ("read_pfile", dict(orient="sample", genotypes="counts")),
             ("read_pfile", dict(orient="sample", genotypes="stats", samples=[5, 0, 1999], region="3:1-900000")),
             ("read_pfile", dict(orient="sample", genotypes="counts", af_range={"min": 0.1, "max": 0.4})),
             ("read_pfile", dict(orient="sample", genotypes="counts", variants={"start": 100, "stop": 4100},
                                 include_genotypes=["hom_alt"])),
             # genotype output: the hardcalls of a window at a time (LeaseRows), every layout, subsets, regions, filters
             ("read_pgen", dict(genotypes="list")),
             ("read_pfile", dict(genotypes="array", samples=[1, 5, 9, 2002], region="2:1-800000")),
             ("read_pgen", dict(genotypes="columns", samples=[0, 2])),
             ("read_pgen", dict(genotypes="struct", samples=[7, 3], af_range={"max": 0.3})),
             ("read_pfile", dict(genotypes="list", include_genotypes=["hom_alt", "missing"], region="5:1-2000000")),
             ("read_pgen", dict(genotypes="list", variants=[5, 4000, 17, 5999])),
             # the dosage and phase forms of the output (no tracks in this file: hardcall dosages, REF|ALT hets)
             ("read_pgen", dict(dosages=True, samples=[3, 2, 1000])), ("read_pgen", dict(phased=True, samples=[0, 1, 2])),
             ("read_pgen", dict(dosages=True, variants=[5000, 12, 4999])), ("read_pfile", dict(phased=True, region="6:1-300000")),
             # read_pfile's other orients: the sample-major matrix is filled a window's columns at a time, a genotype-orient
             # batch leases the window(s) that hold it; count filters settle the effective variants window by window at bind
             ("read_pfile", dict(orient="sample", samples=[9, 1, 1200], region="2:1-20000")),
             ("read_pfile", dict(orient="sample", genotypes="columns", samples=[0, 5], af_range={"min": 0.3, "max": 0.5},
                                 region="1:1-27000")),
             ("read_pfile", dict(orient="sample", genotypes="list", samples=[3, 4, 5, 6], variants=[10, 3000, 5990],
                                 include_genotypes=["het"])),
             ("read_pfile", dict(orient="genotype", samples=[11, 12], region="4:1-9000", include_genotypes=["hom_alt", "missing"])),
             ("read_pfile", dict(orient="genotype", samples=[2, 40], ac_range={"max": 1})),
             # plink_ld: a claim's anchors and their partners inside a small LD window lie within one window of the file
             ("plink_ld", dict(window_kb=2, r2_threshold=0.0, region="7:1-20000")),
             ("plink_ld", dict(variant1="sv100", variant2="sv190")),
             ("plink_ld", dict(variant1="sv17", variant2="sv5801", samples=[5, 6, 7, 8, 900, 901, 1500, 2000])),  # windows apart
             # plink_pca: every pass walks the windows (eigenvalues compared with a tolerance below)
             ("plink_pca", dict(n_pcs=3)), ("plink_pca", dict(n_pcs=2, samples=list(range(0, 2003, 3)), region="2:1-27000")),
             # plink_score: a sum over variants, so the windows' partial sums add (compared with a tolerance below)
             ("plink_score", dict(weights=[((7 * i) % 13 - 6) / 5.0 for i in range(m)])),
             ("plink_score", dict(weights=[((3 * i) % 7) / 3.0 for i in range(m)], samples=[4, 9, 1500], center=True)),
             ("plink_score", dict(weights=[{"id": f"sv{i}", "allele": "G" if i % 3 else "A", "weight": 0.25 * (i % 5 + 1)} for i in range(100, 5000, 37)],
                                  no_mean_imputation=True))]
    want = [F.query(fn, small + (".pgen" if fn != "read_pfile" else ""), threads=threads, **kw) for fn, kw in calls]
    monkeypatch.setenv("PLINKING_HBM_CACHE_GB", "0.0005")  # 500 KB: windows of ~240 variants
    passes = L.tally_passes_started()
    for (fn, kw), w in zip(calls, want):
        got = F.query(fn, big + (".pgen" if fn != "read_pfile" else ""), threads=threads, **kw)
        key = lambda r: tuple(str(x) for x in r[:3]) + ((str(r[-2]),) if kw.get("orient") == "genotype" else ())
        if fn == "plink_pca":  # (sums of doubles in another order; PC signs are the SVD's)
            assert got.names == w.names and len(got) == len(w)
            for a, b in zip(sorted(got.rows, key=lambda r: str(r[:2])), sorted(w.rows, key=lambda r: str(r[:2]))):
                for x, y in zip(a, b):
                    assert x == y if not isinstance(x, float) else min(abs(x - y), abs(x + y)) <= 1e-7 * max(1.0, abs(y)), (a, b)
            continue
        if fn == "plink_score":  # (sums of doubles in another order)
            key = lambda r: str(r[1])
            assert got.names == w.names and len(got) == len(w)
            for a, b in zip(sorted(got.rows, key=key), sorted(w.rows, key=key)):
                assert all(x == y if not isinstance(x, float) else abs(x - y) <= 1e-9 * max(1.0, abs(y)) for x, y in zip(a, b)), fn
            continue
        assert got.names == w.names and sorted(got.rows, key=key) == sorted(w.rows, key=key), fn
    assert L.tally_passes_started() > passes + 8  # one resident pass per window of the file
    for fn, kw in (("plink_ld", {}),):  # (pairs a megabase apart reach further than a 240-variant window)
        with pytest.raises(F.IOException, match="does not fit the HBM budget"):
            F.query(fn, big + (".pgen" if fn != "read_pfile" else ""), threads=2, **kw)


def test_dosage_tracks_stream_with_the_rows(gpu_lib, tmp_path, monkeypatch):
    """A file WITH dosage tracks beyond the HBM budget: the windows are opened with their share of the tracks, so
    read_pgen(dosages := true), plink_freq(dosage := true) and plink_score (which scores a dosage-bearing variant from
    its dosages) give what the resident route gives."""
    L = gpu_lib
    m, n = 3000, 1501
    small, big = str(tmp_path / "fits"), str(tmp_path / "too_big")
    for prefix in (small, big):
        L.synth_write_dosage_files(prefix, m, n, SEED + 5, 0.03, 0.3)
    calls = [("read_pgen", dict(dosages=True)), ("read_pgen", dict(dosages=True, samples=[1500, 7, 8], variants=[2999, 0, 1400])),
             ("read_pgen", dict(genotypes="list")), ("plink_score", dict(weights=[((5 * i) % 11 - 5) / 4.0 for i in range(m)])),
             ("plink_freq", dict(dosage=True)), ("plink_freq", dict(dosage=True, samples=[2, 900, 901], region="3:1-50000"))]
    want = [F.query(fn, small + ".pgen", threads=3, **kw) for fn, kw in calls]
    monkeypatch.setenv("PLINKING_HBM_CACHE_GB", "0.0005")
    for (fn, kw), w in zip(calls, want):
        got = F.query(fn, big + ".pgen", threads=3, **kw)
        assert got.names == w.names and len(got) == len(w)
        key = (lambda r: str(r[1])) if fn == "plink_score" else (lambda r: (str(r[0]), r[1]))
        for a, b in zip(sorted(got.rows, key=key), sorted(w.rows, key=key)):
            if fn == "plink_score":
                assert all(x == y if not isinstance(x, float) else abs(x - y) <= 1e-9 * max(1.0, abs(y)) for x, y in zip(a, b))
            else:
                assert a == b, fn


def test_phase_tracks_stream_with_the_rows(gpu_lib, tmp_path, monkeypatch):
    """Phased output over a file beyond the HBM budget: every variant's phase track travels with its window, and a scan
    thread makes its PgrGetP reader on the window it holds (compressed record types included)."""
    import pgen_writer as W

    rng = np.random.default_rng(12)
    m, n = 900, 700
    geno = rng.choice(np.array([0, 1, 2, 3], dtype=np.uint8), size=(m, n), p=[0.55, 0.3, 0.1, 0.05])
    kinds = W.choose_kinds(geno, rng)
    for tag in ("fits", "too_big"):
        prefix = str(tmp_path / tag)
        W.write_pgen(prefix + ".pgen", geno, kinds, phase_rng=np.random.default_rng(8))
        with open(prefix + ".pvar", "w") as f:
            f.write("#CHROM\tPOS\tID\tREF\tALT\n" + "".join(f"{1 + v // 300}\t{100 * (v % 300 + 1)}\tp{v}\tA\tC\n" for v in range(m)))
        with open(prefix + ".psam", "w") as f:
            f.write("#IID\tSEX\n" + "".join(f"I{s}\t{1 + s % 2}\n" for s in range(n)))
    small, big = str(tmp_path / "fits"), str(tmp_path / "too_big")
    calls = [("read_pgen", dict(phased=True)), ("read_pgen", dict(phased=True, samples=[699, 0, 350], variants=[899, 3, 450])),
             ("read_pfile", dict(phased=True, genotypes="list", region="2:1-20000")),
             ("read_pfile", dict(phased=True, orient="sample", samples=[5, 6, 7], region="1:1-12000")),
             ("read_pfile", dict(phased=True, orient="genotype", samples=[100, 101], region="3:1-30000"))]
    cols = lambda kw: ["IID", "genotypes"] if kw.get("orient") == "sample" else ["ID", "IID", "genotype"] if kw.get("orient") else ["ID", "genotypes"]
    want = [F.query(fn, small + (".pgen" if fn == "read_pgen" else ""), threads=3, columns=cols(kw), **kw) for fn, kw in calls]
    assert any([1, 0] in g for _, g in want[0].rows)  # the file does carry ALT|REF hets
    monkeypatch.setenv("PLINKING_HBM_CACHE_GB", "0.00002")  # 20 KB: windows of ~40 variants
    for (fn, kw), w in zip(calls, want):
        got = F.query(fn, big + (".pgen" if fn == "read_pgen" else ""), threads=3, columns=cols(kw), **kw)
        assert sorted(got.rows, key=repr) == sorted(w.rows, key=repr), fn


@pytest.mark.parametrize("masked", [False, True])
def test_pca_over_windows_of_a_file_equals_pca_over_the_resident_rows(gpu_lib, tmp_path, masked):
    """pgh_pca_streamed: the effective variants cut into windows, every pass opening them one after the other, against
    pgh_pca on the same file made resident -- eigenvalues to 1e-10, eigenvectors up to their sign."""
    L = gpu_lib
    m, n, k = 5000, 1403, 5
    prefix = str(tmp_path / "x")
    L.synth_write_files(prefix, m, n, SEED + 21, 0.03)
    ds = L.Dataset.open(prefix + ".pgen")
    rng = np.random.default_rng(3)
    keep_mask = rng.random(n) < 0.8 if masked else np.ones(n, dtype=bool)
    ss = ds.subset(keep_mask) if masked else None
    c = ds.counts_range(subset=ss).astype(np.float64) if masked else ds.counts_range().astype(np.float64)
    obs = c[:, :3].sum(axis=1)
    af = (c[:, 1] + 2 * c[:, 2]) / (2 * np.maximum(obs, 1))
    keep = np.flatnonzero((obs > 0) & (af > 0) & (af < 1)).astype(np.uint32)
    keep = keep[rng.random(len(keep)) < 0.7]  # gaps in the list: windows of uneven row counts
    center, inv = 2 * af[keep], 1.0 / np.sqrt(2 * af[keep] * (1 - af[keep]))
    n_out = int(keep_mask.sum())
    g1 = np.random.default_rng(4).standard_normal((n_out, 2 * k))
    ev1, vec1 = ds.pca(keep, center, inv, k, g1, subset=ss)
    words = np.zeros((n + 63) // 64, dtype=np.uint64)
    for s in np.flatnonzero(keep_mask):
        words[s >> 6] |= np.uint64(1) << np.uint64(s & 63)
    for window in (700, 100000):  # eight windows; one
        ev2, vec2 = L.pca_streamed(prefix + ".pgen", keep, center, inv, k, g1, window, sample_include=words if masked else None)
        assert np.allclose(ev1, ev2, rtol=1e-10)
        for j in range(k):
            assert np.allclose(vec1[:, j], np.sign(np.dot(vec1[:, j], vec2[:, j])) * vec2[:, j], atol=1e-7)

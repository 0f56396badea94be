"""Dosage tracks (vrtype bits 0x20 / 0x40 / 0x60) through the resident device form.

The reference's own fixture (dosage_example, 4 x 4) pins the decode and the arithmetic
(`read_pgen_dosage.test:108-128`, `plink_freq_dosage.test:93-116`; see test_gpu_parity and the
table-function tests).  Everything larger comes from tests/pgen_writer.py: the three track
shapes, with and without a phase track in front, over every hardcall record type, so the
device path is checked against the oracle on multi-word presence arrays and multi-group id
lists; those shapes are "parity unpinned" against pgenlib."""

import numpy as np
import pytest

from conftest import data_path
import pgen_writer as W

CASES = [(40, 7, 11), (150, 257, 12), (120, 1000, 13), (60, 5003, 14)]  # (variants, samples, seed)


def make_dosage_file(path, m, n, seed, with_phase):
    rng = np.random.default_rng(seed)
    geno = W.rare_matrix(m, n, rng)
    for v in range(0, m, 5):  # some dense rows so that hets (and a phase track) are common
        geno[v] = rng.choice(4, size=n, p=[0.45, 0.3, 0.2, 0.05])
    kinds = W.choose_kinds(geno, rng)
    dos = np.full((m, n), 0xFFFF, dtype=np.uint16)
    dkinds = []
    for v in range(m):
        k = int(rng.choice([0, 0x20, 0x40, 0x60]))
        dkinds.append(k)
        if k == 0:
            continue
        rate = float(rng.choice([0.0, 0.002, 0.05, 0.6, 1.0]))
        hit = rng.random(n) < rate
        dos[v, hit] = rng.integers(0, 32769, int(hit.sum()), dtype=np.uint16)
    W.write_pgen(path, geno, kinds, dosage=dos, dosage_kinds=dkinds,
                 phase_rng=np.random.default_rng(seed + 100) if with_phase else None)
    want = np.where(dos != 0xFFFF, dos.astype(np.float64) / 16384.0,
                    np.where(geno == 3, -9.0, geno.astype(np.float64)))
    return geno, dos, dkinds, want


@pytest.fixture(scope="module")
def files(tmp_path_factory):
    root = tmp_path_factory.mktemp("dosage_tracks")
    out = {}
    for m, n, seed in CASES:
        for with_phase in (False, True):
            path = str(root / f"d_{m}_{n}_{int(with_phase)}.pgen")
            out[(m, n, with_phase)] = (path,) + make_dosage_file(path, m, n, seed, with_phase)
    return out


@pytest.mark.parametrize("m,n,seed", CASES)
@pytest.mark.parametrize("with_phase", [False, True])
def test_oracle_reads_the_written_tracks(files, oracle, m, n, seed, with_phase):
    path, geno, dos, dkinds, want = files[(m, n, with_phase)]
    pg = oracle.Pgen(path)
    assert pg.has_dosage and {k for k in dkinds} == {0, 0x20, 0x40, 0x60}
    for v in range(m):
        assert pg.vrtype(v) & 0x60 == dkinds[v]
        assert np.array_equal(pg.dosage(v), want[v]), (v, dkinds[v])


def _moments(want_rows):
    out = np.zeros((len(want_rows), 3), dtype=np.uint64)
    for i, d in enumerate(want_rows):
        u = np.rint(d[d != -9.0] * 16384.0).astype(np.uint64)
        out[i] = (u.sum(), (u * u).sum(), len(u))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,seed", CASES)
@pytest.mark.parametrize("with_phase", [False, True])
def test_device_dosages_equal_the_oracle(files, gpu_lib, oracle, m, n, seed, with_phase, monkeypatch):
    path, geno, dos, dkinds, want = files[(m, n, with_phase)]
    ds = gpu_lib.Dataset.open(path)  # tracks located and extracted on the device (k_dosage_locate / k_dosage_values)
    assert np.array_equal(ds.dosage_unpack(), want)
    assert ds.info.dosage_variant_ct == sum(1 for k in dkinds if k) and ds.info.dosage_value_ct == int((dos != 0xFFFF).sum())
    monkeypatch.setenv("PGH_HOST_NORMALIZE", "1")  # the host parser of the same tracks
    hosted = gpu_lib.Dataset.open(path)
    monkeypatch.delenv("PGH_HOST_NORMALIZE")
    assert np.array_equal(hosted.dosage_unpack(), want) and np.array_equal(hosted.dosage_sums(), ds.dosage_sums())
    assert hosted.info.dosage_value_ct == ds.info.dosage_value_ct
    assert np.array_equal(ds.dosage_sums(), _moments(want))
    rng = np.random.default_rng(seed)
    mask = rng.random(n) < 0.4
    mask[0] = True
    ss = ds.subset(mask)
    vidx = np.sort(rng.choice(m, size=min(m, 17), replace=False))
    assert np.array_equal(ds.dosage_unpack(vidx=vidx, subset=ss), want[vidx][:, mask])
    assert np.array_equal(ds.dosage_sums(vidx=vidx, subset=ss), _moments(want[vidx][:, mask]))
    assert np.array_equal(ds.dosage_sums(3, 9), _moments(want[3:9]))
    rd = ds.reader(ss)
    for v in vidx[:5]:
        assert np.array_equal(rd.get_dosage_f64(int(v)), want[v][mask])
    # a shard opened mid-file indexes its tracks from its own first variant
    part = gpu_lib.Dataset.open(path, variant_begin=m // 3, variant_end=m - 2)
    assert np.array_equal(part.dosage_unpack(), want[m // 3:m - 2])
    # the hardcall tallies are untouched by the aux tracks behind them
    pg = oracle.Pgen(path)
    assert np.array_equal(ds.counts_range(), np.stack([pg.counts(v) for v in range(m)]))


REL = 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["default", "no_mean_imputation", "center"])
@pytest.mark.parametrize("ncols,path", [(1, "records"), (3, "records"), (6, "records"), (1, "bitwalk"), (3, "bitwalk"),
                                        (1, "lanes"), (5, "lanes")])
def test_score_over_dosage_tracks(files, gpu_lib, oracle, mode, ncols, path, monkeypatch):
    """The three ways a sparse track is scored: the hardcall contraction + k_score_dosage_records over the dataset's
    entry records (the default), + the bit-walking k_score_dosage_fix (what a dataset whose records do not fit gets),
    and the one-lane-per-sample k_score_dosage."""
    if path == "lanes":
        monkeypatch.setenv("PGH_SCORE_DOSAGE_LANES", "1")
    elif path == "bitwalk":
        monkeypatch.setenv("PGH_SCORE_DOSAGE_RECORDS", "0")
    m, n = 120, 1000
    path, geno, dos, dkinds, want = files[(m, n, True)]
    ds = gpu_lib.Dataset.open(path)
    pg = oracle.Pgen(path)
    rng = np.random.default_rng(ncols)
    vidx = np.sort(rng.choice(m, size=97, replace=False))
    w = rng.standard_normal((len(vidx), ncols))
    flip = (rng.random(len(vidx)) < 0.3).astype(np.uint8)
    code = {"default": gpu_lib.SCORE_MEAN_IMPUTE, "no_mean_imputation": gpu_lib.SCORE_NO_MEAN_IMPUTATION,
            "center": gpu_lib.SCORE_CENTER}[mode]
    for mask in (None, rng.random(n) < 0.5):
        ss = None if mask is None else ds.subset(mask)
        inc = None if mask is None else mask.astype(np.uint8)
        s, d, ac = ds.score(vidx, w, flip=flip, mode=code, subset=ss)
        es, ed, eac = oracle.score(pg, vidx, w, flip=flip, mode=mode, include=inc)
        assert np.array_equal(ac, eac)
        scale = np.abs(w).sum(axis=0) * 2.0
        assert np.all(np.abs(s - es) <= REL * np.maximum(np.abs(es), 1e-3 * scale))
        assert np.allclose(d, ed, rtol=REL, atol=1e-9)
    # only dosage-bearing variants, and only hardcall ones, through the same entry point
    for pick in ([v for v in range(m) if dkinds[v]][:30], [v for v in range(m) if not dkinds[v]][:30]):
        wv = rng.standard_normal((len(pick), ncols))
        s, d, ac = ds.score(pick, wv, mode=code)
        es, ed, eac = oracle.score(pg, pick, wv, mode=mode)
        assert np.array_equal(ac, eac) and np.allclose(s, es, rtol=1e-9, atol=1e-9) and np.allclose(d, ed, rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
def test_reference_dosage_fixture_through_the_device_form(gpu_lib, oracle):
    ds = gpu_lib.Dataset.open(data_path("dosage_example.pgen"))
    pg = oracle.Pgen(data_path("dosage_example.pgen"))
    got = ds.dosage_unpack()
    for v in range(pg.M):
        assert np.array_equal(got[v], pg.dosage(v))
        counts, dosages, r2 = pg.dcounts(v)
        sums = ds.dosage_sums(v, v + 1)[0]
        assert int(sums[0]) == int(dosages[1]) and int(sums[2]) * 32768 - int(sums[0]) == int(dosages[0])


def _companions(prefix, m, n):
    with open(prefix + ".pvar", "w") as f:
        f.write("#CHROM\tPOS\tID\tREF\tALT\n")
        for v in range(m):
            f.write(f"{1 + v * 3 // m}\t{1000 + 10 * v}\tv{v}\tA\tG\n")
    with open(prefix + ".psam", "w") as f:
        f.write("#IID\tSEX\n")
        for s in range(n):
            f.write(f"S{s}\t{1 + s % 2}\n")


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [1, 5])
def test_table_functions_over_dosage_tracks(tmp_path, gpu_lib, oracle, threads):
    """plink_freq(dosage := true), plink_score and read_pgen(dosages := true) on a file whose variants mix the
    three track shapes: the shells against the oracle's PgrGetDCounts / PgrGetD restatements."""
    import plinking_duck_amd.functions as F
    m, n = 2600, 301  # several 2048-row chunks
    prefix = str(tmp_path / "dz")
    geno, dos, dkinds, want = make_dosage_file(prefix + ".pgen", m, n, 21, True)
    _companions(prefix, m, n)
    pg = oracle.Pgen(prefix + ".pgen")
    mask = np.zeros(n, dtype=bool)
    mask[[5, 17, 200, 300] + list(range(40, 140))] = True
    for samples, inc in ((None, None), ([int(i) for i in np.flatnonzero(mask)], mask.astype(np.uint8))):
        kw = {} if samples is None else {"samples": samples}
        r = F.query("plink_freq", prefix + ".pgen", dosage=True, threads=threads,
                    columns=["ID", "ALT_FREQ", "OBS_CT", "IMP_R2"], **kw)
        assert len(r) == m
        for vid, af, obs, r2 in r.rows:
            v = int(vid[1:])
            counts, dosages, er2 = pg.dcounts(v, inc)
            eaf, eobs = oracle.freq_from_dcounts(dosages)
            assert (af, obs) == (eaf, eobs), v
            if eobs and not np.isnan(er2):
                assert r2 == pytest.approx(er2, rel=1e-12, abs=1e-300), v
        rng = np.random.default_rng(3)
        w = rng.standard_normal(m)
        w[rng.random(m) < 0.2] = 0.0
        for mode, kwm in (("default", {}), ("no_mean_imputation", {"no_mean_imputation": True}), ("center", {"center": True})):
            r = F.query("plink_score", prefix + ".pgen", weights=[float(x) for x in w], threads=threads,
                        columns=["IID", "ALLELE_CT", "NAMED_ALLELE_DOSAGE_SUM", "SCORE_SUM"], **kw, **kwm)
            vidx = np.flatnonzero(w != 0.0)
            es, ed, eac = oracle.score(pg, vidx, w[vidx], mode=mode, include=inc)
            names = [f"S{s}" for s in (range(n) if samples is None else samples)]
            got = {iid: (ac, d, s) for iid, ac, d, s in r.rows}
            assert sorted(got) == sorted(names)
            for k, iid in enumerate(names):
                ac, d, s = got[iid]
                assert ac == int(eac[k])
                assert s == pytest.approx(es[k, 0], rel=1e-9, abs=1e-9) and d == pytest.approx(ed[k], rel=1e-9, abs=1e-9)
    r = F.query("read_pgen", prefix + ".pgen", dosages=True, threads=threads, columns=["ID", "genotypes"])
    assert len(r) == m
    for vid, g in r.rows:
        v = int(vid[1:])
        assert [(-9.0 if x is None else x) for x in g] == want[v].tolist(), v
    r = F.query("read_pgen", prefix + ".pgen", dosages=True, genotypes="columns", variants=[7, 2599, 64],
                samples=["S300", "S5"], columns=["ID", "S5", "S300"])
    assert [(vid, -9.0 if a is None else a, -9.0 if b is None else b) for vid, a, b in r.rows] == \
        [(f"v{v}", want[v][5], want[v][300]) for v in (7, 2599, 64)]


@pytest.mark.gpu
def test_malformed_dosage_tracks_are_reported(files, gpu_lib, tmp_path):
    """Truncated tracks, a value above 2.0 and a list that runs past the samples: pgh_open says which variant."""
    path, geno, dos, dkinds, want = files[(120, 1000, False)]
    raw = bytearray(open(path, "rb").read())
    pg_m = 120
    head = 12
    offs = int.from_bytes(raw[head:head + 8], "little")
    lens = [int.from_bytes(raw[head + 8 + pg_m + 4 * v:head + 12 + pg_m + 4 * v], "little") for v in range(pg_m)]
    starts = np.concatenate([[offs], offs + np.cumsum(lens)])
    for kind in (0x20, 0x40, 0x60):
        v = next(v for v in range(pg_m) if dkinds[v] == kind and (dos[v] != 0xFFFF).sum() > 3)
        bad = bytearray(raw)
        end = int(starts[v + 1])
        bad[end - 2:end] = (40000).to_bytes(2, "little")  # the last value of the track: above 2.0
        p = str(tmp_path / f"bad_{kind:x}.pgen")
        open(p, "wb").write(bad)
        with pytest.raises(gpu_lib.PghError) as e:
            gpu_lib.Dataset.open(p)
        assert f"record {v}" in str(e.value) or f"variant {v}" in str(e.value), str(e.value)
    # a record cut short: its length field shrunk by 3 bytes (the next record's bytes are then misread too,
    # but the first malformed variant is the one reported or an earlier one never)
    v = next(v for v in range(pg_m) if dkinds[v] == 0x60 and (dos[v] != 0xFFFF).sum() > 3)
    bad = bytearray(raw)
    at = head + 8 + pg_m + 4 * v
    bad[at:at + 4] = (lens[v] - 3).to_bytes(4, "little")
    p = str(tmp_path / "short.pgen")
    open(p, "wb").write(bad)
    with pytest.raises(gpu_lib.PghError):
        gpu_lib.Dataset.open(p)


@pytest.mark.gpu
def test_device_track_reader_survives_corrupt_bytes(files, gpu_lib, tmp_path):
    """Random byte damage inside the records (aux tracks included): every open either fails with a
    PGH_ERR_* or yields tracks whose own bookkeeping is consistent; the kernels' reads stay inside the
    staged bytes, their writes inside the rows they own."""
    path, geno, dos, dkinds, want = files[(120, 1000, True)]
    raw = np.frombuffer(open(path, "rb").read(), dtype=np.uint8)
    body_at = 12 + 8 + 5 * 120
    rng = np.random.default_rng(99)
    opened = failed = 0
    for trial in range(60):
        bad = raw.copy()
        hits = rng.integers(body_at, len(bad), size=int(rng.integers(1, 6)))
        bad[hits] = rng.integers(0, 256, size=len(hits), dtype=np.uint8)
        p = str(tmp_path / f"c{trial}.pgen")
        bad.tofile(p)
        try:
            ds = gpu_lib.Dataset.open(p)
        except gpu_lib.PghError:
            failed += 1
            continue
        opened += 1
        sums = ds.dosage_sums()
        d = ds.dosage_unpack()
        assert int(ds.info.dosage_value_ct) <= int((dos != 0xFFFF).sum()) + 1000 * 120
        assert ((d == -9.0) | ((d >= 0.0) & (d <= 2.0))).all()
        assert (sums[:, 2] <= 1000).all()
        ds.close()
    assert opened and failed  # both outcomes occur: damage to a value is silent, damage to the structure is not
    ds = gpu_lib.Dataset.open(path)  # the device is still healthy
    assert np.array_equal(ds.dosage_unpack(), want)


def _bits(words, n):
    return np.unpackbits(words.view(np.uint8), bitorder="little")[:n]


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,seed", CASES)
def test_phase_tracks_through_the_device_form(files, gpu_lib, oracle, m, n, seed, monkeypatch):
    """pgh_get_phased (PgrGetP): the phase tracks of the written files -- both shapes, all hets phased or
    flagged ones -- expanded on the device at open, against the oracle and against the host parser."""
    path, geno, dos, dkinds, want = files[(m, n, True)]
    pg = oracle.Pgen(path)
    assert pg.has_phase
    ds = gpu_lib.Dataset.open(path)
    monkeypatch.setenv("PGH_HOST_NORMALIZE", "1")
    hosted = gpu_lib.Dataset.open(path)
    monkeypatch.delenv("PGH_HOST_NORMALIZE")
    rng = np.random.default_rng(seed)
    mask = rng.random(n) < 0.5
    mask[0] = True
    for dset in (ds, hosted):
        for subset_mask in (None, mask):
            ss = dset.subset(subset_mask) if subset_mask is not None else None
            inc = None if subset_mask is None else subset_mask.astype(np.uint8)
            n_out = n if subset_mask is None else int(subset_mask.sum())
            rd = dset.reader(ss)
            for v in range(m):
                g, pp, pi = rd.get_phased(v)
                eg, epp, epi = pg.phase(v, inc)
                codes = (np.unpackbits(g.view(np.uint8), bitorder="little")[: 2 * n_out].reshape(-1, 2) * [1, 2]).sum(axis=1)
                assert np.array_equal(np.where(codes == 3, -9, codes), eg), v
                assert np.array_equal(_bits(pp, n_out), epp != 0), v
                assert np.array_equal(_bits(pi, n_out) & _bits(pp, n_out), (epi != 0) & (epp != 0)), v


@pytest.mark.gpu
def test_dosage_ingest_across_staging_batches(tmp_path, gpu_lib, oracle):
    """A file whose records fill several 64 MB staging buffers: value offsets carry over from batch to batch."""
    prefix = str(tmp_path / "wide")
    m, n = 3000, 100_003
    gpu_lib.synth_write_dosage_files(prefix, m, n, 3, 0.02, 0.1)
    assert m * (n // 4 + n // 8 + n // 5) > 2 * (64 << 20)  # three staging buffers' worth
    pg = oracle.Pgen(prefix + ".pgen")
    ds = gpu_lib.Dataset.open(prefix + ".pgen")
    assert ds.info.dosage_variant_ct == m
    picks = [0, 1, 1170, 1171, 2340, 2341, 2999]
    got = ds.dosage_unpack(vidx=picks)
    sums = ds.dosage_sums()
    total = 0
    for i, v in enumerate(picks):
        assert np.array_equal(got[i], pg.dosage(v)), v
    for v in range(0, m, 97):
        counts, dosages, _ = pg.dcounts(v)
        assert int(sums[v][0]) == int(dosages[1]) and int(sums[v][2]) * 32768 - int(sums[v][0]) == int(dosages[0]), v
    part = gpu_lib.Dataset.open(prefix + ".pgen", variant_begin=1171, variant_end=2341)
    assert np.array_equal(part.dosage_sums(), sums[1171:2341])
    # the file-free generator makes the same draws: tracks written to disk and read back = tracks made in HBM
    twin = gpu_lib.Dataset.synth(0, m, n, 3, 0.02)
    twin.synth_add_dosage(0.1, 3)
    assert np.array_equal(twin.dosage_sums(), sums) and np.array_equal(twin.dosage_unpack(vidx=picks), got)
    assert int(twin.info.dosage_value_ct) == int(ds.info.dosage_value_ct)


@pytest.mark.gpu
def test_wide_dosage_matrix_properties(gpu_lib, monkeypatch):
    """500,000 samples wide (the BASELINE width), 4,000 variants, synthetic tracks: size-independent properties --
    shards add up, the unpacked doubles reproduce the integer moments, the two score formulations agree."""
    n, m, seed = 500_000, 4_000, 20260807
    whole = gpu_lib.Dataset.synth(0, m, n, seed, 0.02)
    whole.synth_add_dosage(0.1, seed + 7)
    sums = whole.dosage_sums()
    assert int(whole.info.dosage_variant_ct) == m and 0.09 < int(whole.info.dosage_value_ct) / (m * n) < 0.11
    lo = gpu_lib.Dataset.synth(0, 1500, n, seed, 0.02)
    hi = gpu_lib.Dataset.synth(1500, m, n, seed, 0.02)
    lo.synth_add_dosage(0.1, seed + 7)
    hi.synth_add_dosage(0.1, seed + 7)
    assert np.array_equal(np.concatenate([lo.dosage_sums(), hi.dosage_sums()]), sums)
    picks = [0, 1499, 1500, 3999]
    d = whole.dosage_unpack(vidx=picks)
    for i, v in enumerate(picks):
        row = d[i]
        u = np.rint(row[row != -9.0] * 16384.0).astype(np.uint64)
        assert (int(u.sum()), int((u * u).sum()), len(u)) == tuple(int(x) for x in sums[v])
    counts = whole.counts_range()
    assert np.all(sums[:, 2] >= (counts[:, :3].sum(axis=1)).astype(np.uint64))  # a dosage can stand in for a missing call
    rng = np.random.default_rng(1)
    vidx = np.sort(rng.choice(m, size=1000, replace=False))
    w = rng.standard_normal((len(vidx), 1))
    for mode in (gpu_lib.SCORE_MEAN_IMPUTE, gpu_lib.SCORE_NO_MEAN_IMPUTATION, gpu_lib.SCORE_CENTER):
        s2, d2, ac2 = whole.score(vidx, w, mode=mode)
        monkeypatch.setenv("PGH_SCORE_DOSAGE_LANES", "1")
        s1, d1, ac1 = whole.score(vidx, w, mode=mode)
        monkeypatch.delenv("PGH_SCORE_DOSAGE_LANES")
        assert np.array_equal(ac1, ac2)
        assert np.allclose(s1, s2, rtol=1e-9, atol=1e-9) and np.allclose(d1, d2, rtol=1e-9, atol=1e-9)
        # ... and the entry records (round-major runs per 4096-sample tile, 123 tiles here) against the bit walk
        monkeypatch.setenv("PGH_SCORE_DOSAGE_RECORDS", "0")
        s3, d3, ac3 = whole.score(vidx, w, mode=mode)
        monkeypatch.delenv("PGH_SCORE_DOSAGE_RECORDS")
        assert np.array_equal(ac3, ac2)
        assert np.allclose(s3, s2, rtol=1e-12, atol=1e-12) and np.allclose(d3, d2, rtol=1e-12, atol=1e-12)
    # shards' scores add up to the whole's
    a = lo.score(vidx[vidx < 1500], w[vidx < 1500])
    b = hi.score(vidx[vidx >= 1500], w[vidx >= 1500])
    s, dsum, ac = whole.score(vidx, w)
    assert np.array_equal(a[2] + b[2], ac) and np.allclose(a[0] + b[0], s, rtol=1e-9, atol=1e-9)


@pytest.mark.gpu
def test_phase_and_dosage_tracks_beyond_64kb_of_lds(tmp_path, gpu_lib, oracle):
    """600,000 samples: the phase kernel's two per-word prefix tables need 75 KB of LDS (an opt-in above 64 KB),
    sample ids in difflists and dosage lists are three bytes wide."""
    n, m = 600_000, 6
    rng = np.random.default_rng(5)
    geno = rng.choice(4, size=(m, n), p=[0.55, 0.3, 0.1, 0.05]).astype(np.uint8)
    geno[3] = 0
    geno[3, rng.choice(n, 500, replace=False)] = 1  # a sparse row: type 4 with a multi-group difflist
    kinds = [0, 1, 0, 4, 0, 6]
    dos = np.full((m, n), 0xFFFF, dtype=np.uint16)
    dkinds = [0x60, 0x20, 0x40, 0, 0x60, 0x20]
    for v, k in enumerate(dkinds):
        if k:
            hit = rng.random(n) < (0.01 if k == 0x20 else 0.3)
            dos[v, hit] = rng.integers(0, 32769, int(hit.sum()), dtype=np.uint16)
    path = str(tmp_path / "big.pgen")
    W.write_pgen(path, geno, kinds, dosage=dos, dosage_kinds=dkinds, phase_rng=np.random.default_rng(6))
    pg = oracle.Pgen(path)
    ds = gpu_lib.Dataset.open(path)
    want = np.where(dos != 0xFFFF, dos.astype(np.float64) / 16384.0, np.where(geno == 3, -9.0, geno.astype(np.float64)))
    assert np.array_equal(ds.dosage_unpack(), want)
    rd = ds.reader()
    for v in range(m):
        g, pp, pi = rd.get_phased(v)
        eg, epp, epi = pg.phase(v)
        assert np.array_equal(_bits(pp, n), epp != 0), v
        assert np.array_equal(_bits(pi, n) & _bits(pp, n), (epi != 0) & (epp != 0)), v
    assert np.array_equal(ds.counts_range(), np.stack([pg.counts(v) for v in range(m)]))
    # the column-owning kernels and the score at this width
    g = np.stack([pg.geno(v) for v in range(m)])
    sc = ds.sample_counts()
    assert np.array_equal(sc[:, 1], (g == 1).sum(axis=0)) and np.array_equal(sc[:, 2], (g == 2).sum(axis=0))
    assert np.array_equal(sc[:, 3], (g == -9).sum(axis=0)) and np.array_equal(ds.missing_per_sample(), (g == -9).sum(axis=0))
    w = rng.standard_normal((m, 1))
    s_, d_, ac_ = ds.score(np.arange(m), w)
    es, ed, eac = oracle.score(pg, np.arange(m), w)
    assert np.array_equal(ac_, eac) and np.allclose(s_, es, rtol=1e-11, atol=1e-11) and np.allclose(d_, ed, rtol=1e-11, atol=1e-11)

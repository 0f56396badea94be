"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Integer outputs are compared bit-exact; floating outputs within 1e-6 relative
(BASELINE.json north_star), with tighter bounds where the arithmetic allows.
"""

import json
import os

import numpy as np
import pytest

from conftest import data_path

pytestmark = pytest.mark.gpu

FIXTURES = ["pgen_example", "all_missing", "large_example", "streaming_example", "sexchr_example", "rare_small",
            "pca_example", "phased_example", "dosage_example", "pgen_split"]
SEED = 20260807
REL = 1e-6

with open(os.path.join(os.path.dirname(__file__), "golden", "known_answers.json")) as f:
    KA = json.load(f)


def subset_masks(n, rng):
    """A few include masks: single sample, alternating, random, all-but-one."""
    masks = []
    m = np.zeros(n, dtype=bool)
    m[n // 2] = True
    masks.append(m)
    if n > 1:
        masks.append(np.arange(n) % 2 == 0)
        masks.append(rng.random(n) < 0.4)
        m = np.ones(n, dtype=bool)
        m[0] = False
        masks.append(m)
    return [m for m in masks if m.any()]


def validity_bits(val_row, n):
    return np.unpackbits(val_row.view(np.uint8), bitorder="little")[:n].astype(bool)


# --------------------------------------------------------------------------
# reference fixtures (all record types)
# --------------------------------------------------------------------------

@pytest.mark.parametrize("name", FIXTURES)
def test_fixture_counts_unpack_missing(gpu_lib, oracle, name):
    path = data_path(name + ".pgen")
    ds = gpu_lib.Dataset.open(path)
    pg = oracle.Pgen(path)
    assert (ds.info.raw_variant_ct, ds.n_samples) == (pg.M, pg.N)
    rng = np.random.default_rng(3)
    # counts: whole file, bit-exact
    assert np.array_equal(ds.counts_range(), pg.counts_range())
    # unpack + validity
    out, val = ds.unpack_range(missing_code=-9)
    out0, _ = ds.unpack_range(missing_code=0, want_validity=False)
    step = max(1, pg.M // 400)
    for v in range(0, pg.M, step):
        g = pg.geno(v)
        assert np.array_equal(out[v], g), (name, v)
        assert np.array_equal(out0[v], np.where(g == -9, 0, g))
        assert np.array_equal(validity_bits(val[v], pg.N), g != -9)
    # per-sample missing tallies
    assert np.array_equal(ds.missing_per_sample(), pg.missing_per_sample())
    # sub-range
    if pg.M > 10:
        a, b = pg.M // 3, pg.M // 3 + 7
        assert np.array_equal(ds.counts_range(a, b), pg.counts_range(a, b))
        assert np.array_equal(ds.missing_per_sample(a, b), pg.missing_per_sample(a, b))
    # sample subsets (pgenlib semantics: ascending file order)
    for mask in subset_masks(pg.N, rng):
        ss = ds.subset(mask)
        inc = mask.astype(np.uint8)
        assert ss.size == int(mask.sum())
        assert np.array_equal(ds.counts_range(subset=ss), pg.counts_range(include=inc))
        assert np.array_equal(ds.missing_per_sample(subset=ss), pg.missing_per_sample(include=inc))
        o, vb = ds.unpack_range(subset=ss, missing_code=-9)
        for v in range(0, pg.M, max(1, pg.M // 50)):
            g = pg.geno(v, inc)
            assert np.array_equal(o[v], g)
            assert np.array_equal(validity_bits(vb[v], ss.size), g != -9)
    ds.close()


@pytest.mark.parametrize("name", ["pgen_example", "rare_small", "dosage_example", "pca_example"])
def test_fixture_reader_calls(gpu_lib, oracle, name):
    """The pgenlib-shaped per-variant calls (PgrGet / PgrGetCounts / ...)."""
    path = data_path(name + ".pgen")
    ds = gpu_lib.Dataset.open(path)
    pg = oracle.Pgen(path)
    rng = np.random.default_rng(5)
    for mask in [None] + subset_masks(pg.N, rng)[:2]:
        ss = ds.subset(mask) if mask is not None else None
        inc = None if mask is None else mask.astype(np.uint8)
        rd = ds.reader(ss)
        n_out = pg.N if mask is None else int(mask.sum())
        for v in list(range(0, pg.M, max(1, pg.M // 40))) + [pg.M - 1, 0]:
            g = pg.geno(v, inc)
            assert np.array_equal(rd.get_counts(v), pg.counts(v, inc))
            assert np.array_equal(rd.get_int8(v), g)
            two = rd.get_2bit(v)
            codes = (np.repeat(two, 32) >> np.tile(np.arange(0, 64, 2, dtype=np.uint64), len(two))) & np.uint64(3)
            assert np.array_equal(codes[:n_out].astype(np.int8), np.where(g == -9, 3, g))
            assert np.array_equal(validity_bits(rd.get_missingness(v), n_out), g == -9)
            assert np.array_equal(rd.get_dosage_f64(v), pg.dosage(v, inc))
        with pytest.raises(ValueError):
            rd.get_counts(pg.M)
        rd.close()
    ds.close()


def test_known_answers_through_the_gpu(gpu_lib, oracle):
    """Spot-check the reference's goldens straight off the HIP path."""
    ds = gpu_lib.Dataset.open(data_path("pgen_example.pgen"))
    ka = KA["pgen_example_freq"]
    counts = ds.counts_range()
    assert counts.tolist() == ka["counts"]
    for v in range(4):
        assert oracle.freq_from_counts(counts[v]) == (ka["alt_freq"][v], ka["obs_ct"][v])
    out, _ = ds.unpack_range(missing_code=-9)
    assert out.tolist() == KA["pgen_example_genotypes"]["matrix"]
    assert ds.missing_per_sample().tolist() == KA["missing_pgen_example"]["sample_missing_ct"]
    rs = gpu_lib.Dataset.open(data_path("rare_small.pgen"))
    t = rs.counts_range().sum(axis=0)
    k = KA["rare_small_totals"]
    assert t.tolist() == [k["hom_ref"], k["het"], k["hom_alt"], k["missing"]]
    # HWE on the device from the device counts
    lnp = gpu_lib.hwe_lnp_batch(counts)
    for v, row in enumerate(KA["hardy_pgen_example"]["rows"]):
        assert round(float(np.exp(lnp[v])), 6) == row["p"]
    lnp = gpu_lib.hwe_lnp_batch(counts, midp=True)
    for v, row in enumerate(KA["hardy_pgen_example"]["rows"]):
        assert round(float(np.exp(lnp[v])), 6) == row["p_midp"]


def test_score_known_answers_through_the_gpu(gpu_lib):
    ds = gpu_lib.Dataset.open(data_path("pgen_example.pgen"))
    ka = KA["score_pgen_example"]
    s, d, ac = ds.score(np.arange(4), ka["weights"])
    assert s[:, 0].tolist() == ka["default"]["score_sum"]
    assert d.tolist() == ka["default"]["dosage_sum"]
    assert ac.tolist() == ka["default"]["allele_ct"]
    s, d, ac = ds.score(np.arange(4), ka["weights"], mode=gpu_lib.SCORE_NO_MEAN_IMPUTATION)
    assert (int(ac[1]), s[1, 0], d[1]) == (6, 1.5, 2.0) and (int(ac[3]), s[3, 0], d[3]) == (6, 5.0, 4.0)
    s, d, ac = ds.score([1], [1.0], mode=gpu_lib.SCORE_CENTER)
    assert np.allclose(s[:, 0], ka["center_rs2_weight1"]["score_sum"], rtol=1e-15, atol=0)
    assert set(ac.tolist()) == {2} and not d.any()
    s, _, _ = ds.score([0], [1.0], flip=[1])
    assert (s[0, 0], s[2, 0]) == (2.0, 0.0)
    am = gpu_lib.Dataset.open(data_path("all_missing.pgen"))
    s, d, ac = am.score([0, 1], [1.0, 0.5])
    assert not s.any() and not ac.any() and not d.any()


# --------------------------------------------------------------------------
# seeded synthetic data, sizes the oracle finishes in seconds
# --------------------------------------------------------------------------

SHAPES = [(64, 1), (33, 5), (257, 63), (130, 64), (90, 65), (50, 1000), (40, 4099), (24, 16384), (12, 70001)]


@pytest.mark.parametrize("m,n", SHAPES)
def test_synth_matches_host_generator_and_oracle(gpu_lib, oracle, m, n):
    ds = gpu_lib.Dataset.synth(0, m, n, SEED, 0.02)
    host = np.stack([gpu_lib.synth_record_host(v, n, SEED, 0.02) for v in range(m)])
    # the on-device generator reproduces the host twin bit for bit
    rd = ds.reader()
    for v in (0, m // 2, m - 1):
        two = rd.get_2bit(v)
        assert np.array_equal(two.view(np.uint8)[: host.shape[1]], host[v])
    # oracle over the same records (vrtype-0 .pgen built in memory)
    pg = oracle.Pgen(mem=mem_pgen(host, n))
    assert np.array_equal(ds.counts_range(), pg.counts_range())
    assert np.array_equal(ds.missing_per_sample(), pg.missing_per_sample())
    out, val = ds.unpack_range(missing_code=0)
    for v in range(0, m, max(1, m // 16)):
        g = pg.geno(v)
        assert np.array_equal(out[v], np.where(g == -9, 0, g))
        assert np.array_equal(validity_bits(val[v], n), g != -9)
    rng = np.random.default_rng(n)
    for mask in subset_masks(n, rng)[1:3]:
        ss = ds.subset(mask)
        inc = mask.astype(np.uint8)
        assert np.array_equal(ds.counts_range(subset=ss), pg.counts_range(include=inc))
        o, vb = ds.unpack_range(2, min(m, 9), subset=ss, missing_code=-9)
        for i, v in enumerate(range(2, min(m, 9))):
            assert np.array_equal(o[i], pg.geno(v, inc))
    # a dataset built from host rows behaves the same as the generated one
    ds2 = gpu_lib.Dataset.from_host_rows(host, n)
    assert np.array_equal(ds2.counts_range(), ds.counts_range())


def mem_pgen(rows, n):
    """mode-0x02 (fixed-width) .pgen image around plain 2-bit rows."""
    m = rows.shape[0]
    head = bytes([0x6c, 0x1b, 0x02]) + int(m).to_bytes(4, "little") + int(n).to_bytes(4, "little") + bytes([0x40])
    return np.frombuffer(head + rows.tobytes(), dtype=np.uint8)


@pytest.mark.parametrize("m,n", [(1, 1), (7, 63), (12, 64), (25, 1000), (300, 4099), (1000, 70001), (64, 500_000)])
def test_fused_tally_equals_the_two_separate_passes(gpu_lib, m, n):
    """plink_freq/hardy/missing off one pass: same integers as the separate kernels."""
    import torch
    ds = gpu_lib.Dataset.synth(3, 3 + m, n, SEED, 0.04)
    d_counts = torch.full((m, 4), -1, dtype=torch.int32, device="cuda")
    d_miss = torch.full(((n + 63) // 64 * 64,), -1, dtype=torch.int32, device="cuda")
    ds.fused_tally_dev(3, 3 + m, d_counts.data_ptr(), d_miss.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_counts.cpu().numpy().astype(np.uint32), ds.counts_range())
    assert np.array_equal(d_miss[:n].cpu().numpy().astype(np.uint32), ds.missing_per_sample())
    if m > 20:  # sub-range
        ds.fused_tally_dev(10, 3 + m - 5, d_counts.data_ptr(), d_miss.data_ptr(),
                           torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        k = 3 + m - 5 - 10
        assert np.array_equal(d_counts[:k].cpu().numpy().astype(np.uint32), ds.counts_range(10, 3 + m - 5))
        assert np.array_equal(d_miss[:n].cpu().numpy().astype(np.uint32), ds.missing_per_sample(10, 3 + m - 5))


def test_variant_sharded_dataset_is_a_slice_of_the_global_one(gpu_lib):
    """Rank r generates rows [a,b) of the same global matrix (multi-GPU sharding)."""
    n = 3001
    whole = gpu_lib.Dataset.synth(0, 96, n, SEED, 0.02)
    part = gpu_lib.Dataset.synth(32, 64, n, SEED, 0.02)
    assert np.array_equal(part.counts_range(), whole.counts_range(32, 64))
    with pytest.raises(ValueError):
        part.counts_range(0, 8)


@pytest.mark.parametrize("mode", ["default", "no_mean_imputation", "center"])
@pytest.mark.parametrize("ncols", [1, 16])
def test_score_matches_oracle(gpu_lib, oracle, mode, ncols):
    m, n = 300, 2000
    ds = gpu_lib.Dataset.synth(0, m, n, SEED, 0.05)
    host = np.stack([gpu_lib.synth_record_host(v, n, SEED, 0.05) for v in range(m)])
    # force an all-missing and a monomorphic variant into the scored set
    host[5] = 0xFF
    host[9] = 0x00
    ds = gpu_lib.Dataset.from_host_rows(host, n)
    pg = oracle.Pgen(mem=mem_pgen(host, n))
    rng = np.random.default_rng(17)
    vidx = np.sort(rng.choice(m, size=180, replace=False))
    vidx = np.union1d(vidx, [5, 9])
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
        scale = np.abs(w).sum(axis=0) * 2.0  # magnitude of the terms being summed
        assert np.all(np.abs(s - es) <= REL * np.maximum(np.abs(es), 1e-9 * scale))
        assert np.allclose(d, ed, rtol=REL, atol=1e-9)
        # without the dosage sum (projection pushdown of NAMED_ALLELE_DOSAGE_SUM) the rest is unchanged
        s2, d2, ac2 = ds.score(vidx, w, flip=flip, mode=code, subset=ss, want_dosage_sum=False)
        assert d2 is None and np.array_equal(ac2, ac)
        assert np.all(np.abs(s2 - es) <= REL * np.maximum(np.abs(es), 1e-9 * scale))


@pytest.mark.parametrize("ncols", [2, 3, 5, 9, 11, 13, 15, 17, 18, 20, 21, 24, 25, 28, 29, 33, 52])
def test_score_any_number_of_columns(gpu_lib, oracle, ncols):
    """Every count of weight columns: 7 digit columns each + 6 for the dosage sum and the missing count fill
    1..10 sixteen-column tiles of the int8 contraction (k_score_i8<NT, TS>: 9, 11, 13, 16-17, 18, 20-22 columns are the
    software-pipelined shapes of 5 .. 10 tiles); more than 22 columns take a second pass."""
    m, n = 150, 1500
    host = np.stack([gpu_lib.synth_record_host(v, n, SEED + 3, 0.05) for v in range(m)])
    ds = gpu_lib.Dataset.from_host_rows(host, n)
    pg = oracle.Pgen(mem=mem_pgen(host, n))
    rng = np.random.default_rng(ncols)
    vidx = np.sort(rng.choice(m, size=131, replace=False))
    w = rng.standard_normal((len(vidx), ncols))
    s, d, ac = ds.score(vidx, w)
    es, ed, eac = oracle.score(pg, vidx, w)
    assert np.array_equal(ac, eac)
    scale = np.abs(w).sum(axis=0) * 2.0
    assert np.all(np.abs(s - es) <= REL * np.maximum(np.abs(es), 1e-9 * scale))
    assert np.allclose(d, ed, rtol=REL, atol=1e-9)


@pytest.mark.parametrize("ncols,mode", [(1, "default"), (1, "no_mean_imputation"), (1, "center"), (2, "default"),
                                        (3, "center"), (7, "default"), (16, "default"), (19, "no_mean_imputation")])
def test_score_over_many_tiles_and_slices(gpu_lib, oracle, ncols, mode):
    """The contraction walks 64-variant tiles through a four-slot LDS ring, several slices per sample group:
    4,500 scored variants picked out of 6,000 (a non-contiguous list: 71 tiles, the last one padded) over
    2,500 samples (ragged last word, three sample groups of the one-column shape) against the oracle."""
    m, n = 6000, 2500
    host = np.stack([gpu_lib.synth_record_host(v, n, SEED + 21, 0.04) for v in range(m)])
    host[17] = 0xFF
    host[4000] = 0x00
    ds = gpu_lib.Dataset.from_host_rows(host, n)
    pg = oracle.Pgen(mem=mem_pgen(host, n))
    rng = np.random.default_rng(100 + ncols)
    vidx = np.union1d(np.sort(rng.choice(m, size=4498, replace=False)), [17, 4000])
    w = rng.standard_normal((len(vidx), ncols)) * np.exp(rng.normal(0, 2, size=(len(vidx), 1)))
    flip = (rng.random(len(vidx)) < 0.4).astype(np.uint8)
    code = {"default": gpu_lib.SCORE_MEAN_IMPUTE, "no_mean_imputation": gpu_lib.SCORE_NO_MEAN_IMPUTATION,
            "center": gpu_lib.SCORE_CENTER}[mode]
    mask = rng.random(n) < 0.7
    for ss, inc in ((None, None), (ds.subset(mask), mask.astype(np.uint8))):
        s, d, ac = ds.score(vidx, w, flip=flip, mode=code, subset=ss)
        es, ed, eac = oracle.score(pg, vidx, w, flip=flip, mode=mode, include=inc)
        assert np.array_equal(ac, eac)
        # sums of ~4,500 terms: compare against the magnitude of what was summed (a double accumulation in
        # another order differs by as much)
        scale = np.abs(w).sum(axis=0) * 2.0
        assert np.all(np.abs(s - es) <= 1e-12 * scale + REL * np.abs(es) * 1e-3)
        assert np.allclose(d, ed, rtol=1e-9, atol=1e-9)


def test_score_weights_spanning_twelve_orders_of_magnitude(gpu_lib, oracle):
    """Fixed-point digits are cut below each column's LARGEST coefficient (54 bits): small weights next to
    large ones keep 54 - log2(ratio) bits.  With a 1e12 spread the small terms still carry ~14 bits, and the
    sum stays within 1e-6 of the oracle relative to its own size wherever it is not dominated by cancellation."""
    m, n = 512, 1200
    host = np.stack([gpu_lib.synth_record_host(v, n, SEED + 22, 0.03) for v in range(m)])
    ds = gpu_lib.Dataset.from_host_rows(host, n)
    pg = oracle.Pgen(mem=mem_pgen(host, n))
    rng = np.random.default_rng(5)
    vidx = np.arange(m)
    w = (10.0 ** rng.uniform(-6, 6, size=(m, 2))) * rng.choice([-1.0, 1.0], size=(m, 2))
    s, d, ac = ds.score(vidx, w)
    es, ed, eac = oracle.score(pg, vidx, w)
    assert np.array_equal(ac, eac)
    bound = 2.0 ** -50 * np.abs(w).max(axis=0) * 3 * m + 1e-12 * np.abs(es)
    assert np.all(np.abs(s - es) <= bound)
    assert np.allclose(d, ed, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("mode", ["default", "no_mean_imputation", "center"])
def test_score_at_baseline_width(gpu_lib, oracle, mode):
    """BASELINE config 4's shape on the sample axis: 500,000 samples x 16 weight columns (and the one-column SQL
    contract) over 70 variants -- 1,954 workgroup columns, sample indices up to 5e5 in the epilogue, the ragged
    last group -- against the oracle, with flips and a sample subset."""
    m, n = 70, 500_000
    host = np.stack([gpu_lib.synth_record_host(v, n, SEED + 23, 0.02) for v in range(m)])
    ds = gpu_lib.Dataset.from_host_rows(host, n)
    pg = oracle.Pgen(mem=mem_pgen(host, n))
    rng = np.random.default_rng(9)
    vidx = np.arange(m)
    flip = (rng.random(m) < 0.3).astype(np.uint8)
    code = {"default": gpu_lib.SCORE_MEAN_IMPUTE, "no_mean_imputation": gpu_lib.SCORE_NO_MEAN_IMPUTATION,
            "center": gpu_lib.SCORE_CENTER}[mode]
    mask = rng.random(n) < 0.6
    for ncols in (16, 1):
        w = rng.standard_normal((m, ncols))
        for ss, inc in ((None, None), (ds.subset(mask), mask.astype(np.uint8))):
            s, d, ac = ds.score(vidx, w, flip=flip, mode=code, subset=ss)
            es, ed, eac = oracle.score(pg, vidx, w, flip=flip, mode=mode, include=inc)
            assert np.array_equal(ac, eac)
            scale = np.abs(w).sum(axis=0) * 2.0
            assert np.all(np.abs(s - es) <= 1e-12 * scale + 1e-9 * np.abs(es))
            assert np.allclose(d, ed, rtol=1e-9, atol=1e-9)


def test_pca_at_baseline_width(gpu_lib):
    """BASELINE config 5's sample axis: pgh_pca at N = 500,000 (k = 4, 2,500 variants).  No second implementation
    fits the budget at this width, so the checks are size-independent identities: the eigenvectors are
    orthonormal, the eigenvalues equal those of the same matrix held as a three-shard group (another
    summation order, another set of fixed-point scales), and the Rayleigh quotients |X u_k|^2 / M recomputed on the
    host in float64 bracket the reported eigenvalues as the algebra says they must."""
    m, n, k = 2500, 500_000, 4
    whole = gpu_lib.Dataset.synth(0, m, n, SEED + 31, 0.02)
    c = whole.counts_range().astype(np.float64)
    obs = c[:, :3].sum(axis=1)
    af = (c[:, 1] + 2 * c[:, 2]) / (2 * np.maximum(obs, 1))
    keep = np.flatnonzero((obs > 0) & (af > 0) & (af < 1)).astype(np.uint32)
    center, inv = 2 * af[keep], 1.0 / np.sqrt(2 * af[keep] * (1 - af[keep]))
    g1 = np.random.default_rng(8).standard_normal((n, 2 * k))
    ev, vec = whole.pca(keep, center, inv, k, g1)
    assert np.all(np.diff(ev) <= 0) and np.all(ev > 0)
    assert np.allclose(vec.T @ vec, np.eye(k), atol=1e-9)
    shards = [gpu_lib.Dataset.synth(a, b, n, SEED + 31, 0.02) for a, b in ((0, 900), (900, 1700), (1700, m))]
    ev2, vec2 = gpu_lib.Dataset.group(shards).pca(keep, center, inv, k, g1)
    assert np.allclose(ev, ev2, rtol=1e-9)
    # Rayleigh quotients on the host, X row by row from the unpacked calls (missing -> 0 after normalisation).
    # u_k = X^T z_k / S_k with |z_k| = 1 (a unit vector of the Krylov basis), so |X u_k|^2 >= (z^T X X^T z)^2 / S_k^2
    # = S_k^2: the quotient can exceed the reported eigenvalue (where the subspace has not converged) but never
    # fall below it -- and it cannot exceed the whole spectrum's sum, trace(X^T X) / M, known from the tallies.
    t2 = (np.stack([(0 - center) * inv, (1 - center) * inv, (2 - center) * inv], axis=1)) ** 2
    trace_over_m = float(np.sum(c[keep, :3] * t2)) / len(keep)
    assert ev.sum() <= trace_over_m * (1 + 1e-12) and ev[0] >= trace_over_m / len(keep)
    for pc in range(2):
        acc = 0.0
        for lo in range(0, len(keep), 250):
            rows = keep[lo:lo + 250]
            g, _ = whole.unpack_range(int(rows[0]), int(rows[-1]) + 1, missing_code=-9, want_validity=False)
            g = g[rows - rows[0]].astype(np.float64)
            x = np.where(g == -9, 0.0, (g - center[lo:lo + 250, None]) * inv[lo:lo + 250, None])
            acc += float(np.sum((x @ vec[:, pc]) ** 2))
        rayleigh = acc / len(keep)
        assert ev[pc] * (1 - 1e-9) <= rayleigh <= trace_over_m


@pytest.mark.parametrize("n_pcs,m,n", [(2, 700, 2100), (5, 700, 2100), (6, 700, 2100), (9, 700, 2100), (10, 700, 2100), (11, 700, 2100),
                                       (13, 700, 2100), (2, 300, 20000), (10, 900, 20000)])
def test_pca_matches_oracle_on_wide_rows(gpu_lib, oracle, n_pcs, m, n):
    """pgh_pca against the numpy restatement with rows wide enough (>= 512 B) for the MFMA Step A;
    2k = 4, 18, 20, 22, 26 columns cover a bare tile and 1, 2 and 3 quarter tiles; N = 20,000 makes
    Step A split the sample axis over two workgroups per variant tile (atomic combine)."""
    host = np.stack([gpu_lib.synth_record_host(v, n, SEED + 11, 0.03) for v in range(m)])
    ds = gpu_lib.Dataset.from_host_rows(host, n)
    pg = oracle.Pgen(mem=mem_pgen(host, n))
    ev, vecs, m_eff = oracle.pca(pg, n_pcs)
    c = ds.counts_range().astype(np.float64)
    obs = c[:, 0] + c[:, 1] + c[:, 2]
    af = np.where(obs > 0, (c[:, 1] + 2 * c[:, 2]) / np.maximum(2 * obs, 1), 0.0)
    keep = np.flatnonzero((obs > 0) & (af > 0) & (af < 1))
    assert len(keep) == m_eff
    g1 = oracle.fill_g1(n, 2 * n_pcs)
    got_ev, got_vecs = ds.pca(keep, 2 * af[keep], 1.0 / np.sqrt(2 * af[keep] * (1 - af[keep])), n_pcs, g1)
    assert np.allclose(got_ev, ev, rtol=1e-6)
    # unstructured data: the leading eigenvalues are close together, so single eigenvectors are
    # ill-conditioned; the spanned subspace is not (compare the projectors)
    assert np.allclose(got_vecs.T @ got_vecs, np.eye(n_pcs), atol=1e-8)
    if n_pcs == 2:
        assert np.allclose(got_vecs @ got_vecs.T @ vecs, vecs, atol=1e-5)


def test_hwe_batch_matches_oracle(gpu_lib, oracle):
    rng = np.random.default_rng(23)
    rows = []
    for _ in range(400):
        n = int(rng.integers(1, 200000))
        p = rng.uniform(0.001, 0.999)
        f = rng.uniform(-0.05, 0.1)
        pr = np.clip([(1 - p) ** 2 + f * p * (1 - p), 2 * p * (1 - p) * (1 - f), p * p + f * p * (1 - p)], 0, None)
        hom1, het, hom2 = rng.multinomial(n, pr / pr.sum())
        rows.append([hom1, het, hom2, int(rng.integers(0, 50))])
    rows += [[0, 0, 0, 7], [1, 1, 1, 0], [0, 10, 0, 0]]
    counts = np.array(rows, dtype=np.uint32)
    for midp in (False, True):
        got = gpu_lib.hwe_lnp_batch(counts, midp)
        for i in range(0, len(counts), 7):
            hom1, het, hom2 = (int(x) for x in counts[i, :3])
            exp = oracle.hwe_lnp(het, hom1, hom2, midp)
            if np.isinf(exp):
                assert got[i] < -700
            else:
                # 1e-6 relative on the p-value == 1e-6 absolute on ln p
                assert abs(got[i] - exp) < 1e-6, (counts[i], midp)
        # the device result equals the library's host routine
        host = np.array([gpu_lib.hwe_lnp(int(c[1]), int(c[0]), int(c[2]), midp) for c in counts[:50]])
        assert np.allclose(got[:50], host, rtol=1e-12, atol=1e-12)


# --------------------------------------------------------------------------
# BASELINE-sized rows (N = 500k) through size-independent properties
# --------------------------------------------------------------------------

def test_full_width_rows_properties(gpu_lib, oracle):
    """N = 500,000 (BASELINE configs 2/3 row width), a few thousand variants."""
    n, m = 500_000, 2048
    ds = gpu_lib.Dataset.synth(0, m, n, SEED, 0.02)
    counts = ds.counts_range()
    # every row tallies to N
    assert (counts.sum(axis=1, dtype=np.int64) == n).all()
    # checksum of checksums: per-sample missing tallies and per-variant missing
    # counts are two reductions of the same indicator matrix
    miss = ds.missing_per_sample()
    assert int(miss.sum(dtype=np.int64)) == int(counts[:, 3].sum(dtype=np.int64))
    # linearity over variant ranges
    a = ds.missing_per_sample(0, 700)
    b = ds.missing_per_sample(700, m)
    assert np.array_equal(a + b, miss)
    # complementary subsets partition the counts
    rng = np.random.default_rng(1)
    mask = rng.random(n) < 0.37
    c1 = ds.counts_range(0, 256, subset=ds.subset(mask))
    c2 = ds.counts_range(0, 256, subset=ds.subset(~mask))
    assert np.array_equal(c1 + c2, counts[:256])
    # unpack is consistent with the tallies, and three rows match the oracle exactly
    out, val = ds.unpack_range(0, 64, missing_code=0)
    vb = np.unpackbits(val.view(np.uint8), axis=1, bitorder="little")[:, :n]
    assert np.array_equal((vb == 0).sum(axis=1), counts[:64, 3])
    assert np.array_equal(((out == 1) & (vb == 1)).sum(axis=1), counts[:64, 1])
    assert np.array_equal((out == 2).sum(axis=1), counts[:64, 2])
    host = np.stack([gpu_lib.synth_record_host(v, n, SEED, 0.02) for v in (0, 31, 63)])
    pg = oracle.Pgen(mem=mem_pgen(host, n))
    for i, v in enumerate((0, 31, 63)):
        g = pg.geno(i)
        assert np.array_equal(out[v], np.where(g == -9, 0, g))
        assert np.array_equal(counts[v], pg.counts(i))
    # generator statistics: ~2 % missing, allele frequencies inside U(0.01, 0.5)
    assert 0.019 < counts[:, 3].mean() / n < 0.021
    af = (counts[:, 1] + 2.0 * counts[:, 2]) / (2.0 * counts[:, :3].sum(axis=1))
    assert 0.005 < af.min() and af.max() < 0.505
    # score: all-ones weights reproduce the dosage sum; shards add up
    vidx = np.arange(0, 512)
    s, d, ac = ds.score(vidx, np.ones(len(vidx)))
    assert np.allclose(s[:, 0], d, rtol=1e-12)
    assert set(ac.tolist()) == {2 * len(vidx)}
    s1, _, _ = ds.score(vidx[:200], np.ones(200))
    s2, _, _ = ds.score(vidx[200:], np.ones(312))
    assert np.allclose(s1 + s2, s, rtol=1e-9)


@pytest.mark.parametrize("n", [1, 63, 1000, 4097, 16385, 70001])
def test_ld_pair_sums_match_oracle(gpu_lib, oracle, n):
    """pgh_ld_pairs: exact integer sums, wave form (short rows) and workgroup form (rows >= 4 KiB),
    ragged tails, subsets, anchors with 1..9 consecutive partners, repeated and reversed pairs."""
    m = 40
    host = np.stack([gpu_lib.synth_record_host(v, n, SEED + 5, 0.08) for v in range(m)])
    host[3] = 0xFF  # all missing
    host[4] = 0x00  # monomorphic
    ds = gpu_lib.Dataset.from_host_rows(host, n)
    pg = oracle.Pgen(mem=mem_pgen(host, n))
    rng = np.random.default_rng(n)
    a, b = [], []
    for anchor in range(0, 30, 3):
        span = int(rng.integers(1, 10))
        for j in range(anchor + 1, min(m, anchor + 1 + span)):
            a.append(anchor)
            b.append(j)
    a += [7, 9, 9, 39, 3]
    b += [7, 2, 9, 0, 4]
    for mask in (None, rng.random(n) < 0.5):
        ss = None if mask is None else ds.subset(mask)
        inc = None if mask is None else mask.astype(np.uint8)
        got = ds.ld_pairs(a, b, subset=ss)
        want = np.stack([pg.ld_sums(x, y, include=inc) for x, y in zip(a, b)])
        assert np.array_equal(got.astype(np.uint64), want)


@pytest.mark.parametrize("n", [1000, 70001])
def test_ld_refuses_a_task_list_that_did_not_arrive(gpu_lib, monkeypatch, n):
    """Round 1 saw the task list reach the device as zeros (its pageable source had died before the
    asynchronous copy ran).  The kernel must neither read out of bounds nor hand back sums for such a
    task: the call fails with PGH_ERR_DEVICE, and the next, intact call works."""
    m = 12
    host = np.stack([gpu_lib.synth_record_host(v, n, SEED + 6, 0.05) for v in range(m)])
    ds = gpu_lib.Dataset.from_host_rows(host, n)
    a, b = [0, 0, 0, 5], [1, 2, 3, 9]
    good = ds.ld_pairs(a, b)
    monkeypatch.setenv("PGH_TEST_ZERO_LD_TASKS", "1")
    with pytest.raises(gpu_lib.PghError, match="did not arrive intact"):
        ds.ld_pairs(a, b)
    monkeypatch.setenv("PGH_TEST_ZERO_LD_TASKS", "0")
    assert np.array_equal(ds.ld_pairs(a, b), good)


@pytest.mark.parametrize("n", [5, 1000, 70001])
def test_sample_counts_match_oracle(gpu_lib, oracle, n):
    m = 300
    host = np.stack([gpu_lib.synth_record_host(v, n, SEED + 9, 0.06) for v in range(m)])
    ds = gpu_lib.Dataset.from_host_rows(host, n)
    pg = oracle.Pgen(mem=mem_pgen(host, n))
    rng = np.random.default_rng(n)
    assert np.array_equal(ds.sample_counts(), pg.sample_counts())
    assert np.array_equal(ds.sample_counts(40, 40), pg.sample_counts(vidx=[]))
    pick = np.sort(rng.choice(m, size=77, replace=False))
    mask = rng.random(n) < 0.4
    mask[0] = True
    assert np.array_equal(ds.sample_counts(vidx=pick, subset=ds.subset(mask)),
                          pg.sample_counts(vidx=[int(v) for v in pick], include=mask.astype(np.uint8)))
    assert np.array_equal(ds.sample_counts(10, 200), pg.sample_counts(vidx=range(10, 200)))


def test_full_size_matrix_properties(gpu_lib):
    """BASELINE configs 2/3 at full size -- 1,000,000 x 500,000 resident in HBM (125 GB) -- through
    properties that need no second implementation: every row tallies to N; the fused pass equals the
    two separate passes; per-sample and per-variant missing tallies are two reductions of one matrix;
    shards of the variant axis add up; unpack of a slice agrees with its tallies."""
    torch = pytest.importorskip("torch")
    free, _ = torch.cuda.mem_get_info()
    m, n = 1_000_000, 500_000
    if free < 135e9:
        pytest.skip("needs 135 GB of free HBM")
    ds = gpu_lib.Dataset.synth(0, m, n, SEED, 0.02)
    counts = ds.counts_range()
    assert counts.shape == (m, 4) and (counts.sum(axis=1, dtype=np.int64) == n).all()
    assert 0.019 < counts[:, 3].sum(dtype=np.int64) / (m * n) < 0.021
    st = torch.cuda.current_stream().cuda_stream
    d_counts = torch.zeros((m, 4), dtype=torch.int32, device="cuda")
    d_miss = torch.zeros((n + 63) // 64 * 64, dtype=torch.int32, device="cuda")
    ds.fused_tally_dev(0, m, d_counts.data_ptr(), d_miss.data_ptr(), st)
    torch.cuda.synchronize()
    assert np.array_equal(d_counts.cpu().numpy().astype(np.uint32), counts)
    miss = ds.missing_per_sample()
    assert np.array_equal(d_miss[:n].cpu().numpy().astype(np.uint32), miss)
    assert int(miss.sum(dtype=np.int64)) == int(counts[:, 3].sum(dtype=np.int64))
    cut = 611_111
    assert np.array_equal(ds.missing_per_sample(0, cut) + ds.missing_per_sample(cut, m), miss)
    sc = ds.sample_counts(cut, cut + 4096)
    assert (sc.sum(axis=1) == 4096).all() and int(sc[:, 3].sum()) == int(counts[cut:cut + 4096, 3].sum())
    geno, _ = ds.unpack_range(cut, cut + 8)
    for r in range(8):
        row = np.where(geno[r] == -9, 3, geno[r])
        assert [int((row == c).sum()) for c in range(4)] == [int(x) for x in counts[cut + r]]
    # ("rows add up to N" holds by construction -- hom-ref is N minus the rest -- so:) 300 random rows copied back from
    # HBM and tallied by the oracle's scan, bit-exact; 40 of them decoded by the oracle against the unpack kernel
    from oracle import oracle
    rng = np.random.default_rng(99)
    picks = np.sort(rng.choice(m, size=300, replace=False))
    rows = np.concatenate([ds.copy_rows_to_host(int(v), int(v) + 1) for v in picks])
    head = bytes([0x6c, 0x1b, 0x02]) + len(picks).to_bytes(4, "little") + n.to_bytes(4, "little") + bytes([0x40])
    pg = oracle.Pgen(mem=np.concatenate([np.frombuffer(head, dtype=np.uint8), rows.reshape(-1)]))
    assert np.array_equal(pg.scan_counts_mt(0, len(picks), 8), counts[picks])
    for i in range(0, 300, 8):
        g, _ = ds.unpack_range(int(picks[i]), int(picks[i]) + 1)
        assert np.array_equal(g[0], pg.geno(i))
    # the tally pass the SQL functions share, at full size (eight 131,072-variant batches): every product
    t = gpu_lib.TallyPass(ds, products=gpu_lib.TALLY_SAMPLE_MISSING | gpu_lib.TALLY_HWE)
    assert np.array_equal(t.counts(), counts)
    assert np.array_equal(t.sample_missing(), miss)
    lnp = t.hwe_lnp(False)
    assert np.array_equal(lnp[picks], gpu_lib.hwe_lnp_batch(counts[picks], False))
    for v in picks[:12]:
        c = counts[v]
        assert lnp[v] == pytest.approx(oracle.hwe_lnp(int(c[1]), int(c[0]), int(c[2]), False), abs=1e-6)
    t.close()


def test_hwe_xchr_batch_matches_host_and_oracle(gpu_lib, oracle):
    """The device chrX exact test (one workgroup per variant) against the library's host routine
    (same rule, running log-ratios instead of lgamma) and, for small tables, the oracle's full enumeration."""
    rng = np.random.default_rng(41)
    rows = []
    for _ in range(120):
        nf, nm = int(rng.integers(0, 4000)), int(rng.integers(0, 4000))
        p = rng.uniform(0.02, 0.98)
        f = rng.uniform(-0.1, 0.2)
        pr = np.clip([p * p + f * p * (1 - p), 2 * p * (1 - p) * (1 - f), (1 - p) ** 2 + f * p * (1 - p)], 0, None)
        hom1, het, hom2 = rng.multinomial(nf, pr / pr.sum())
        male1 = int(rng.binomial(nm, min(1.0, max(0.0, p + rng.uniform(-0.05, 0.05)))))
        rows.append([het, hom1, hom2, male1, nm - male1])
    rows += [[0, 0, 0, 0, 0], [0, 0, 0, 5, 7], [3, 2, 1, 0, 0], [1, 1, 1, 1, 1], [0, 10, 0, 10, 0], [200000, 150000, 70000, 130000, 90000]]
    strata = np.array(rows, dtype=np.int32)
    for midp in (False, True):
        got = gpu_lib.hwe_xchr_lnp_batch(strata, midp)
        for i, (fh, f1, f2, m1, m2) in enumerate(rows[:-1]):
            host = gpu_lib.hwe_xchr_lnp(fh, f1, f2, m1, m2, midp)
            if np.isinf(host):
                assert got[i] < -700
            else:
                assert abs(got[i] - host) < 1e-8, (rows[i], midp, got[i], host)
        for i in range(0, 40):
            fh, f1, f2, m1, m2 = rows[i]
            if fh + f1 + f2 + m1 + m2 <= 1500:
                exp = oracle.hwe_xchr_lnp(fh, f1, f2, m1, m2, midp)
                assert abs(got[i] - exp) < 1e-6 or (np.isinf(exp) and got[i] < -700)
    # biobank-sized strata finish (the host routine takes ~0.1 s for this one table)
    assert np.isfinite(got[-1]) or got[-1] < -700


@pytest.mark.gpu
@pytest.mark.parametrize("m,n", [(1, 1), (63, 65), (64, 64), (130, 1000), (1000, 4099), (17, 70001)])
def test_sample_major_unpack_is_the_transposed_unpack(gpu_lib, m, n):
    """pgh_unpack_samples (read_pfile orient := 'sample'): the variant-major unpack, transposed; ragged tiles,
    listed variants in any order, sample subsets."""
    ds = gpu_lib.Dataset.synth(0, m, n, SEED + 11, 0.07)
    rows, _ = ds.unpack_range(missing_code=-9, want_validity=False)
    rng = np.random.default_rng(m * 31 + n)
    order = rng.permutation(m)[: max(1, m - 3)]
    assert np.array_equal(ds.unpack_samples(np.arange(m)), rows.T)
    assert np.array_equal(ds.unpack_samples(order, missing_code=3), np.where(rows[order].T == -9, 3, rows[order].T))
    mask = rng.random(n) < 0.5
    mask[n - 1] = True
    ss = ds.subset(mask)
    assert np.array_equal(ds.unpack_samples(order, subset=ss), rows[order][:, mask].T)
    if m >= 63:
        ds.synth_add_dosage(0.2, 5)
        d = ds.dosage_unpack()
        assert np.array_equal(ds.dosage_unpack_samples(order), d[order].T)
        assert np.array_equal(ds.dosage_unpack_samples(order, subset=ss), d[order][:, mask].T)


@pytest.mark.gpu
def test_thread_scratch_grows_and_is_reused(gpu_lib):
    """Entry points that need device scratch keep one block per calling thread and stream (PghThreadScratch; no
    stream-ordered allocations -- DESIGN.md section 6): alternate a small and a larger dataset on one thread, through
    three entry points with different scratch needs, and on a second thread; every answer must equal the per-variant
    tallies' totals."""
    import threading

    small = gpu_lib.Dataset.synth(0, 700, 1000, 5, 0.05)
    large = gpu_lib.Dataset.synth(0, 6000, 70_001, 6, 0.02)
    want = {id(d): d.counts_range().astype(np.int64).sum(axis=0) for d in (small, large)}
    errors = []

    def work():
        try:
            for d in (small, large, small, large, large, small):
                w = want[id(d)]
                assert int(d.missing_per_sample().astype(np.int64).sum()) == int(w[3])
                cls = d.sample_counts().astype(np.int64).sum(axis=0)
                assert np.array_equal(cls, w)
                s, dsum, ac = d.score(np.arange(0, d.v_end, 7), np.ones((len(range(0, d.v_end, 7)), 1)),
                                      mode=gpu_lib.SCORE_NO_MEAN_IMPUTATION)
                assert ac.min() >= 0 and np.isfinite(s).all()
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append(repr(e))

    work()
    t = threading.Thread(target=work)
    t.start()
    t.join(timeout=120)
    assert not errors and not t.is_alive(), errors


@pytest.mark.gpu
def test_rccl_host_runs_on_one_rank(tmp_path):
    """tests/abi/rccl_host.cpp + rccl_host_main.cpp, built with hipcc against libpgenhip and librccl and RUN: a C++
    host whose all-reduce callback is ncclAllReduce on the stream pgh_pca_sharded hands it (one rank on the box's one
    GPU -- the RCCL launch, the stream handle and the library's ordering around the callback are what is exercised;
    the multi-rank arithmetic is covered by the gloo tests and the in-process shard groups)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libdir = os.path.join(root, "plinking_duck_amd")
    exe = tmp_path / "rccl_host"
    build = subprocess.run(["/opt/rocm/bin/hipcc", "-std=c++17", "-O1", "-I", os.path.join(root, "include"),
                            os.path.join(root, "tests", "abi", "rccl_host.cpp"),
                            os.path.join(root, "tests", "abi", "rccl_host_main.cpp"), "-o", str(exe), "-L", libdir,
                            "-lpgenhip", "-lrccl", "-Wl,-rpath," + libdir], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    run = subprocess.run([str(exe), str(tmp_path / "pcs")], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0, (run.stdout[-1500:], run.stderr[-1500:])
    assert "worst relative difference" in run.stdout


@pytest.mark.parametrize("mode,name", [(0, "default"), (1, "no_mean_imputation"), (2, "center")])
def test_score_with_non_finite_weights_matches_the_reference_arithmetic(gpu_lib, oracle, mode, name):
    """A NaN or infinite coefficient: the reference accumulates w * scored in doubles (src/plink_score.cpp:621-651), so
    the column comes out NaN / +-Inf sample by sample.  (Round 2's fixed-point digits turned such a weight into finite
    garbage.)  The other columns, the dosage sum and ALLELE_CT are untouched."""
    L = gpu_lib
    m, n = 300, 1003
    rows = np.stack([L.synth_record_host(v, n, SEED, 0.1) for v in range(m)])
    ds = L.Dataset.from_host_rows(rows, n)
    head = bytes([0x6c, 0x1b, 0x02]) + m.to_bytes(4, "little") + n.to_bytes(4, "little") + bytes([0x40])
    pg = oracle.Pgen(mem=np.frombuffer(head + rows.tobytes(), dtype=np.uint8))
    rng = np.random.default_rng(17)
    vidx = np.arange(m, dtype=np.uint32)
    w = rng.normal(size=(m, 3))
    w[7, 0] = np.nan
    w[120, 1] = np.inf
    w[121, 1] = np.inf
    w[200, 2] = -np.inf
    flip = (rng.random(m) < 0.3).astype(np.uint8)
    got = ds.score(vidx, w, flip, mode)
    with np.errstate(invalid="ignore"):
        exp = [oracle.score(pg, vidx, w[:, c], flip, name)[0][:, 0] for c in range(3)]
    for c in range(3):
        g, e = got[0][:, c], exp[c]
        assert np.array_equal(np.isnan(g), np.isnan(e)), (c, int(np.isnan(g).sum()), int(np.isnan(e).sum()))
        assert np.array_equal(np.isposinf(g), np.isposinf(e)) and np.array_equal(np.isneginf(g), np.isneginf(e))
        fin = np.isfinite(e)
        assert np.allclose(g[fin], e[fin], rtol=1e-9, atol=1e-9)
    clean = ds.score(vidx, np.nan_to_num(w, nan=0.0, posinf=0.0, neginf=0.0), flip, mode)
    assert np.array_equal(got[2], clean[2]) and np.allclose(got[1], clean[1])


@pytest.mark.parametrize("ncols", [10, 12, 14, 16, 18, 22, 30])
def test_kept_score_plans_stream_tile_major_copies(gpu_lib, oracle, ncols, monkeypatch):
    """A kept plan with many weight columns contracts a tile-major copy of its rows (contiguous 8 KB tile images per
    workgroup instead of 128 bytes of each of 64 rows); a one-shot pgh_score reads the rows themselves.  Same sums --
    the integer digit sums are exact, only the final FP64 additions may differ in order -- over a list with gaps, a
    ragged last tile, several slices, and a sample count that leaves a partial last stripe."""
    import torch
    L = gpu_lib
    m, n = 9000, 5003
    ds = L.Dataset.synth(0, m, n, SEED, 0.05)
    rng = np.random.default_rng(ncols)
    vidx = np.sort(rng.choice(m, 6001, replace=False)).astype(np.uint32)
    w = rng.normal(size=(len(vidx), ncols))
    flip = (rng.random(len(vidx)) < 0.3).astype(np.uint8)
    monkeypatch.setenv("PGH_I8_TPS_MAX", "16")
    for mode in (L.SCORE_MEAN_IMPUTE, L.SCORE_NO_MEAN_IMPUTATION, L.SCORE_CENTER):
        want = ds.score(vidx, w, flip, mode)
        plan = ds.score_plan(vidx, w, flip, mode)
        d_score = torch.zeros((n, ncols), dtype=torch.float64, device="cuda")
        d_dos = torch.zeros(n, dtype=torch.float64, device="cuda")
        d_ac = torch.zeros(n, dtype=torch.int32, device="cuda")
        plan.run_dev(d_score.data_ptr(), d_dos.data_ptr(), d_ac.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        scale = np.abs(w).sum(axis=0) * 2.0 + 1.0
        assert np.all(np.abs(d_score.cpu().numpy() - want[0]) <= 1e-12 * scale)
        assert np.allclose(d_dos.cpu().numpy(), want[1], rtol=1e-12, atol=1e-9)
        assert np.array_equal(d_ac.cpu().numpy().astype(np.uint32), want[2])
        plan.close()


def test_pca_with_and_without_tile_major_copies(gpu_lib, monkeypatch):
    L = gpu_lib
    m, n, k = 3000, 2100, 7  # 2k = 14 columns: the many-column shape
    ds = L.Dataset.synth(0, m, n, SEED + 3, 0.03)
    c = ds.counts_range().astype(np.float64)
    obs = c[:, :3].sum(axis=1)
    af = (c[:, 1] + 2 * c[:, 2]) / (2 * np.maximum(obs, 1))
    keep = np.flatnonzero((obs > 0) & (af > 0) & (af < 1)).astype(np.uint32)
    center, inv = 2 * af[keep], 1.0 / np.sqrt(2 * af[keep] * (1 - af[keep]))
    g1 = np.random.default_rng(4).standard_normal((n, 2 * k))
    ev1, vec1 = ds.pca(keep, center, inv, k, g1)
    monkeypatch.setenv("PGH_PCA_TILES", "0")
    ev2, vec2 = ds.pca(keep, center, inv, k, g1)
    assert np.allclose(ev1, ev2, rtol=1e-10)
    for j in range(k):
        assert np.allclose(vec1[:, j], np.sign(np.dot(vec1[:, j], vec2[:, j])) * vec2[:, j], atol=1e-7)


def test_work_blocks_are_kept_between_calls_and_given_back(gpu_lib):
    """plink_pca's big work matrices go to a per-device free list when a call ends (hipMalloc of tens of GB costs
    seconds in a long-lived process); the next call takes them from there and gives the same answer, and
    pgh_trim_device_cache() / pgh_close() hand them back to the driver."""
    import torch

    L = gpu_lib
    m, n, k = 24000, 24000, 7  # the transposed matrix and each tile-major copy: 144 MB = 137 MiB
    ds = L.Dataset.synth(0, m, n, SEED + 9, 0.02)
    c = ds.counts_range().astype(np.float64)
    obs = c[:, :3].sum(axis=1)
    af = (c[:, 1] + 2 * c[:, 2]) / (2 * np.maximum(obs, 1))
    keep = np.flatnonzero((obs > 0) & (af > 0) & (af < 1)).astype(np.uint32)
    center, inv = 2 * af[keep], 1.0 / np.sqrt(2 * af[keep] * (1 - af[keep]))
    g1 = np.random.default_rng(5).standard_normal((n, 2 * k))
    L.trim_device_cache()
    free0 = torch.cuda.mem_get_info(0)[0]
    ev1, vec1 = ds.pca(keep, center, inv, k, g1)
    held = free0 - torch.cuda.mem_get_info(0)[0]
    assert held >= 3 * 130 * 2**20  # X^T and the two tile-major copies (137 MiB each) at least
    ev2, vec2 = ds.pca(keep, center, inv, k, g1)
    assert free0 - torch.cuda.mem_get_info(0)[0] <= held + (64 << 20)  # the second call re-used them
    assert np.allclose(ev1, ev2, rtol=1e-10)
    for j in range(k):
        assert np.allclose(vec1[:, j], np.sign(np.dot(vec1[:, j], vec2[:, j])) * vec2[:, j], atol=1e-7)
    L.trim_device_cache()
    assert free0 - torch.cuda.mem_get_info(0)[0] <= (64 << 20)
    ev3, _ = ds.pca(keep, center, inv, k, g1)
    assert np.allclose(ev1, ev3, rtol=1e-10)
    ds.close()
    assert torch.cuda.mem_get_info(0)[0] >= free0  # pgh_close emptied the list too


def test_hwe_batches_tested_in_allele_fraction_order_give_the_same_doubles(gpu_lib):
    """Batches of 8,192 variants and more are tested in order of their minor-allele fraction (waves of equally long
    walks); every variant's double must be the one the plain launch gives, at the variant's own position."""
    rng = np.random.default_rng(77)
    m, n = 50_000, 120_000
    p = rng.uniform(0.0, 0.5, size=m)
    counts = np.zeros((m, 4), dtype=np.uint32)
    for i0 in range(0, m, 10_000):
        pp = p[i0:i0 + 10_000, None]
        probs = np.concatenate([(1 - pp) ** 2 * 0.98, 2 * pp * (1 - pp) * 0.98, pp ** 2 * 0.98, np.full_like(pp, 0.02)], axis=1)
        counts[i0:i0 + 10_000] = np.array([rng.multinomial(n, q) for q in probs], dtype=np.uint32)
    counts[17] = 0  # an empty variant
    counts[18] = (n, 0, 0, 0)  # monomorphic
    for midp in (False, True):
        ordered = gpu_lib.hwe_lnp_batch(counts, midp)
        plain = np.concatenate([gpu_lib.hwe_lnp_batch(counts[i:i + 4096], midp) for i in range(0, m, 4096)])
        assert np.array_equal(ordered, plain, equal_nan=True)
        # nothing observed: ln 1; one table possible: p = 1, and half of it under mid-p
        assert ordered[17] == 0.0 and ordered[18] == (np.log(0.5) if midp else 0.0)


def test_a_column_of_unit_weights_is_the_dosage_sum_bit_for_bit(gpu_lib):
    """w = 1.0 exactly: w * scored is `scored`, so the reference's SCORE_SUM and NAMED_ALLELE_DOSAGE_SUM are the same
    doubles added in the same order (streaming_threading.test:226-233 asserts it on 50,000 variants).  Here they are
    two columns of the contraction: the unit column must come out equal to the dosage sum to the last bit, whatever the
    order of the slices' atomic additions, next to other columns, with flips, in both imputation modes."""
    L = gpu_lib
    m, n = 30000, 4001
    ds = L.Dataset.synth(0, m, n, SEED + 31, 0.04)
    rng = np.random.default_rng(9)
    vidx = np.sort(rng.choice(m, 25000, replace=False)).astype(np.uint32)
    flip = (rng.random(len(vidx)) < 0.2).astype(np.uint8)
    for ncols, unit in ((1, 0), (3, 1), (16, 15)):
        w = rng.standard_normal((len(vidx), ncols))
        w[:, unit] = 1.0
        for mode in (L.SCORE_MEAN_IMPUTE, L.SCORE_NO_MEAN_IMPUTATION):
            s, d, _ = ds.score(vidx, w, flip, mode)
            assert np.array_equal(s[:, unit], d)

"""Shard groups: ONE process holding contiguous variant ranges of a file on several devices behind one handle
(pgh_open_sharded / pgh_group_create, api_sharded.cpp) -- the reference's in-process parallelism with devices in
the place of scan threads.  Every entry point must give a group exactly what it gives one resident dataset:
per-variant outputs bit for bit, per-sample sums to the last rounding of a different summation order.

The GPU box has one device, so the shards of these tests share device 0: the routing, the per-shard threads, the
device-to-device copy of the partials and the sum kernel are the code an 8-GPU node runs; only the copy's
transport (xGMI) differs."""

import numpy as np
import pytest

from conftest import data_path

pytestmark = pytest.mark.gpu

SEED = 20260807


def make_pair(L, m, n, cuts, missing=0.04, seed=SEED):
    """(one dataset, a group of len(cuts)+1 shards) over the same m x n synthetic matrix."""
    whole = L.Dataset.synth(0, m, n, seed, missing)
    edges = [0] + list(cuts) + [m]
    shards = [L.Dataset.synth(a, b, n, seed, missing) for a, b in zip(edges[:-1], edges[1:])]
    group = L.Dataset.group(shards)
    assert group.shard_count == len(edges) - 1 and whole.shard_count == 0
    assert (group.v_begin, group.v_end, group.n_samples) == (0, m, n)
    return whole, group


def test_per_variant_outputs_are_identical(gpu_lib):
    m, n = 1000, 5003
    whole, group = make_pair(gpu_lib, m, n, [1, 400, 401, 930])
    rng = np.random.default_rng(1)
    mask = rng.random(n) < 0.5
    for ds_mask in (None, mask):
        sw = None if ds_mask is None else whole.subset(ds_mask)
        sg = None if ds_mask is None else group.subset(ds_mask)
        assert np.array_equal(group.counts_range(subset=sg), whole.counts_range(subset=sw))
        for a, b in ((0, 1), (0, 2), (399, 402), (350, 990), (930, 1000), (500, 500)):
            assert np.array_equal(group.counts_range(a, b, subset=sg), whole.counts_range(a, b, subset=sw))
            go, gv = group.unpack_range(a, b, subset=sg)
            wo, wv = whole.unpack_range(a, b, subset=sw)
            assert np.array_equal(go, wo) and np.array_equal(gv, wv)
    assert np.array_equal(group.copy_rows_to_host(395, 935), whole.copy_rows_to_host(395, 935))
    pick = np.sort(rng.choice(m, size=300, replace=False)).astype(np.uint32)
    shuffled = rng.permutation(pick).astype(np.uint32)  # output order follows the list, not the shards
    assert np.array_equal(group.unpack_samples(shuffled), whole.unpack_samples(shuffled))
    rg, rw = group.reader(), whole.reader()
    for v in (0, 1, 400, 401, 929, 930, 999):
        assert np.array_equal(rg.get_counts(v), rw.get_counts(v))
        assert np.array_equal(rg.get_int8(v), rw.get_int8(v))
        assert np.array_equal(rg.get_2bit(v), rw.get_2bit(v))
        assert np.array_equal(rg.get_missingness(v), rw.get_missingness(v))
    with pytest.raises(ValueError):
        group.counts_range(10, 2000)


def test_per_sample_reductions_match(gpu_lib):
    m, n = 1500, 4099
    whole, group = make_pair(gpu_lib, m, n, [7, 800])
    rng = np.random.default_rng(2)
    mask = rng.random(n) < 0.6
    sw, sg = whole.subset(mask), group.subset(mask)
    assert np.array_equal(group.missing_per_sample(), whole.missing_per_sample())
    assert np.array_equal(group.missing_per_sample(5, 900, subset=sg), whole.missing_per_sample(5, 900, subset=sw))
    assert np.array_equal(group.sample_counts(), whole.sample_counts())
    pick = np.sort(rng.choice(m, size=700, replace=False))
    assert np.array_equal(group.sample_counts(vidx=pick, subset=sg), whole.sample_counts(vidx=pick, subset=sw))


@pytest.mark.parametrize("ncols", [1, 16])
@pytest.mark.parametrize("mode", ["MEAN_IMPUTE", "NO_MEAN_IMPUTATION", "CENTER"])
def test_score_over_shards_equals_one_dataset(gpu_lib, ncols, mode):
    """The merge of src/plink_score.cpp:657-664 as a device-to-device sum: integers identical, doubles equal up to
    the order of the final additions."""
    m, n = 2000, 3001
    whole, group = make_pair(gpu_lib, m, n, [640, 1300])
    rng = np.random.default_rng(3)
    vidx = np.sort(rng.choice(m, size=1500, replace=False))
    w = rng.standard_normal((len(vidx), ncols))
    flip = (rng.random(len(vidx)) < 0.3).astype(np.uint8)
    code = getattr(gpu_lib, "SCORE_" + mode)
    mask = rng.random(n) < 0.5
    for sw, sg in ((None, None), (whole.subset(mask), group.subset(mask))):
        s1, d1, a1 = whole.score(vidx, w, flip=flip, mode=code, subset=sw)
        s2, d2, a2 = group.score(vidx, w, flip=flip, mode=code, subset=sg)
        assert np.array_equal(a1, a2)
        scale = np.abs(w).sum(axis=0) * 2.0
        assert np.all(np.abs(s1 - s2) <= 1e-13 * scale)
        assert np.allclose(d1, d2, rtol=1e-12, atol=1e-9)
    # a shard without a single scored variant, and an empty list
    s1, _, a1 = whole.score(vidx[vidx >= 700], w[vidx >= 700])
    s2, _, a2 = group.score(vidx[vidx >= 700], w[vidx >= 700])
    assert np.array_equal(a1, a2) and np.allclose(s1, s2, rtol=1e-12, atol=1e-12)
    s0, _, a0 = group.score(vidx[:0], w[:0])
    assert not s0.any() and not a0.any()


@pytest.mark.parametrize("rate", [0.1, 0.7, 1.0])
def test_dosage_score_over_shards_equals_one_dataset(gpu_lib, rate, monkeypatch):
    """plink_score over dosage tracks through a group: each shard builds its own entry records (sparse tracks), or
    takes the sample-owning / fully explicit kernels (denser ones); the partial sums meet on the root device.  Sample
    count chosen to span two 4096-sample record tiles."""
    m, n, seed = 900, 5003, SEED
    cuts = [250, 251, 700]
    whole = gpu_lib.Dataset.synth(0, m, n, seed, 0.03)
    whole.synth_add_dosage(rate, seed + 7)
    edges = [0] + cuts + [m]
    shards = []
    for a, b in zip(edges[:-1], edges[1:]):
        sh = gpu_lib.Dataset.synth(a, b, n, seed, 0.03)
        sh.synth_add_dosage(rate, seed + 7)
        shards.append(sh)
    group = gpu_lib.Dataset.group(shards)
    assert np.array_equal(group.dosage_sums(), whole.dosage_sums())
    rng = np.random.default_rng(11)
    vidx = np.sort(rng.choice(m, size=600, replace=False))
    flip = (rng.random(len(vidx)) < 0.3).astype(np.uint8)
    for ncols, mode in ((1, "MEAN_IMPUTE"), (3, "NO_MEAN_IMPUTATION"), (1, "CENTER")):
        w = rng.standard_normal((len(vidx), ncols))
        code = getattr(gpu_lib, "SCORE_" + mode)
        s1, d1, a1 = whole.score(vidx, w, flip=flip, mode=code)
        s2, d2, a2 = group.score(vidx, w, flip=flip, mode=code)
        scale = np.abs(w).sum(axis=0) * 2.0
        assert np.array_equal(a1, a2) and np.all(np.abs(s1 - s2) <= 1e-12 * scale)
        assert np.allclose(d1, d2, rtol=1e-12, atol=1e-9)
        if rate < 0.4:  # the bit-walking kernel against the records, shard by shard
            monkeypatch.setenv("PGH_SCORE_DOSAGE_RECORDS", "0")
            s3, d3, a3 = group.score(vidx, w, flip=flip, mode=code)
            monkeypatch.delenv("PGH_SCORE_DOSAGE_RECORDS")
            assert np.array_equal(a3, a2) and np.all(np.abs(s3 - s2) <= 1e-12 * scale)


def test_pca_over_shards_equals_one_dataset(gpu_lib):
    m, n, k = 900, 2100, 4
    whole, group = make_pair(gpu_lib, m, n, [300, 650], missing=0.03, seed=SEED + 11)
    c = whole.counts_range().astype(np.float64)
    obs = c[:, :3].sum(axis=1)
    af = (c[:, 1] + 2 * c[:, 2]) / (2 * np.maximum(obs, 1))
    keep = np.flatnonzero((obs > 0) & (af > 0) & (af < 1)).astype(np.uint32)
    center, inv = 2 * af[keep], 1.0 / np.sqrt(2 * af[keep] * (1 - af[keep]))
    g1 = np.random.default_rng(4).standard_normal((n, 2 * k))
    ev1, vec1 = whole.pca(keep, center, inv, k, g1)
    ev2, vec2 = group.pca(keep, center, inv, k, g1)
    assert np.allclose(ev1, ev2, rtol=1e-9)
    for j in range(k):
        sgn = np.sign(np.dot(vec1[:, j], vec2[:, j]))
        assert np.allclose(vec1[:, j], sgn * vec2[:, j], atol=1e-7)


def _pca_inputs(whole, n, k):
    c = whole.counts_range().astype(np.float64)
    obs = c[:, :3].sum(axis=1)
    af = (c[:, 1] + 2 * c[:, 2]) / (2 * np.maximum(obs, 1))
    keep = np.flatnonzero((obs > 0) & (af > 0) & (af < 1)).astype(np.uint32)
    return keep, 2 * af[keep], 1.0 / np.sqrt(2 * af[keep] * (1 - af[keep])), np.random.default_rng(4).standard_normal((n, 2 * k))


def test_a_group_on_distinct_devices_merges_over_rccl(gpu_lib):
    """One shard on one device is the largest group of DISTINCT devices this box has: its merges run through the
    in-library RCCL path (ncclCommInitAll over the group's devices, ncclReduce for plink_score's partials,
    ncclAllReduce per plink_pca pass on the shard's own stream) -- the code an 8-GPU node runs with 8 ranks.
    Shards that share a device (every other test of this file) have no communicator and use the copies."""
    L = gpu_lib
    m, n, k = 900, 2100, 4
    whole = L.Dataset.synth(0, m, n, SEED + 11, 0.03)
    solo = L.Dataset.group([L.Dataset.synth(0, m, n, SEED + 11, 0.03)])
    shared = L.Dataset.group([L.Dataset.synth(0, 400, n, SEED + 11, 0.03), L.Dataset.synth(400, m, n, SEED + 11, 0.03)])
    assert solo.uses_rccl and not shared.uses_rccl and not whole.uses_rccl
    rng = np.random.default_rng(5)
    vidx = np.sort(rng.choice(m, 500, replace=False)).astype(np.uint32)
    w = rng.normal(size=(500, 3))
    for mode in (L.SCORE_MEAN_IMPUTE, L.SCORE_NO_MEAN_IMPUTATION, L.SCORE_CENTER):
        a, b = whole.score(vidx, w, mode=mode), solo.score(vidx, w, mode=mode)
        assert np.allclose(a[0], b[0], rtol=1e-12, atol=1e-9) and np.allclose(a[1], b[1], rtol=1e-12, atol=1e-9)
        assert np.array_equal(a[2], b[2])
    keep, center, inv, g1 = _pca_inputs(whole, n, k)
    ev1, vec1 = whole.pca(keep, center, inv, k, g1)
    ev2, vec2 = solo.pca(keep, center, inv, k, g1)
    assert np.allclose(ev1, ev2, rtol=1e-9)
    for j in range(k):
        assert np.allclose(vec1[:, j], np.sign(np.dot(vec1[:, j], vec2[:, j])) * vec2[:, j], atol=1e-7)


def test_a_failing_shard_does_not_leave_the_others_waiting(gpu_lib, monkeypatch):
    """pgh_pca over a group: a shard that gives up between two exchanges (a failed allocation, a launch error) used
    to leave the other shard threads in the rendezvous forever; now the meeting is called off and the call fails."""
    import threading
    L = gpu_lib
    m, n, k = 900, 2100, 4
    whole, group = make_pair(L, m, n, [300, 650], missing=0.03, seed=SEED + 11)
    keep, center, inv, g1 = _pca_inputs(whole, n, k)
    for victim in (0, 1, 2):
        monkeypatch.setenv("PGH_TEST_PCA_FAIL_SHARD", str(victim))
        result = {}

        def run():
            try:
                group.pca(keep, center, inv, k, g1)
                result["ok"] = True
            except IOError as e:
                result["err"] = str(e)

        t = threading.Thread(target=run, daemon=True)
        t.start()
        t.join(timeout=60)
        assert not t.is_alive(), "the surviving shard threads are still waiting"
        assert "err" in result
    monkeypatch.delenv("PGH_TEST_PCA_FAIL_SHARD")
    ev1, _ = whole.pca(keep, center, inv, k, g1)
    ev2, _ = group.pca(keep, center, inv, k, g1)  # and the group still works afterwards
    assert np.allclose(ev1, ev2, rtol=1e-9)


def test_ld_pairs_across_a_shard_boundary(gpu_lib):
    m, n = 300, 4500
    whole, group = make_pair(gpu_lib, m, n, [100, 200], missing=0.05)
    a, b = [], []
    for anchor in range(90, 215, 3):  # windows of 12 partners walk across both boundaries
        for j in range(anchor + 1, min(m, anchor + 13)):
            a.append(anchor)
            b.append(j)
    a += [250, 5, 100, 199]
    b += [3, 250, 99, 200]
    rng = np.random.default_rng(5)
    mask = rng.random(n) < 0.5
    assert np.array_equal(group.ld_pairs(a, b), whole.ld_pairs(a, b))
    assert np.array_equal(group.ld_pairs(a, b, subset=group.subset(mask)), whole.ld_pairs(a, b, subset=whole.subset(mask)))


@pytest.mark.parametrize("name", ["large_example", "rare_small", "dosage_example", "phased_example", "pgen_split"])
def test_open_sharded_files(gpu_lib, name):
    """pgh_open_sharded over the reference's fixtures (compressed records, LD runs that cross a shard's first
    variant, dosage and phase tracks): three shards on device 0 against one pgh_open."""
    path = data_path(name + ".pgen")
    whole = gpu_lib.Dataset.open(path)
    group = gpu_lib.Dataset.open_sharded(path, [0, 0, 0])
    assert group.shard_count == 3
    assert (group.info.raw_variant_ct, group.n_samples) == (whole.info.raw_variant_ct, whole.n_samples)
    assert (group.info.dosage_variant_ct, group.info.dosage_value_ct) == (whole.info.dosage_variant_ct,
                                                                         whole.info.dosage_value_ct)
    assert np.array_equal(group.counts_range(), whole.counts_range())
    go, gv = group.unpack_range()
    wo, wv = whole.unpack_range()
    assert np.array_equal(go, wo) and np.array_equal(gv, wv)
    assert np.array_equal(group.missing_per_sample(), whole.missing_per_sample())
    assert np.array_equal(group.sample_counts(), whole.sample_counts())
    m = whole.info.raw_variant_ct
    if whole.info.dosage_variant_ct:
        assert np.array_equal(group.dosage_sums(), whole.dosage_sums())
        assert np.array_equal(group.dosage_unpack(), whole.dosage_unpack())
        pick = np.arange(m - 1, -1, -2, dtype=np.uint32)
        assert np.array_equal(group.dosage_sums(vidx=pick), whole.dosage_sums(vidx=pick))
        assert np.array_equal(group.dosage_unpack_samples(pick), whole.dosage_unpack_samples(pick))
        w = np.linspace(-1, 1, m).reshape(-1, 1)
        s1, d1, a1 = whole.score(np.arange(m), w)
        s2, d2, a2 = group.score(np.arange(m), w)
        assert np.array_equal(a1, a2) and np.allclose(s1, s2, rtol=1e-12, atol=1e-12) and np.allclose(d1, d2, rtol=1e-12)
        rg, rw = group.reader(), whole.reader()
        for v in range(m):
            assert np.array_equal(rg.get_dosage_f64(v), rw.get_dosage_f64(v))
    if whole.info.has_phase:
        rg, rw = group.reader(), whole.reader()
        for v in range(m):
            for x, y in zip(rg.get_phased(v), rw.get_phased(v)):
                assert np.array_equal(x, y)


def test_group_handles_are_refused_where_device_pointers_cross(gpu_lib):
    whole, group = make_pair(gpu_lib, 64, 1000, [20])
    import torch

    buf = torch.empty((64, 4), dtype=torch.int32, device="cuda")
    with pytest.raises(ValueError, match="one device's dataset"):
        group.counts_range_dev(0, 64, buf.data_ptr())
    with pytest.raises(ValueError):
        gpu_lib.Dataset.open_sharded(data_path("pgen_example.pgen"), [0, 7])


def test_table_functions_over_device_shards():
    """Every SQL function with plinking_devices = '0,0,0' gives the rows it gives on one device (the option sits
    next to plinking_max_threads; the cache keeps both forms of the file apart)."""
    F = pytest.importorskip("plinking_duck_amd.functions")
    big = data_path("large_example.pgen")
    ex = data_path("pgen_example.pgen")
    one = {"plinking_devices": ""}
    three = {"plinking_devices": "0,0,0"}
    calls = [
        ("plink_freq", big, dict(counts=True)),
        ("plink_freq", big, dict(region="2:100-500", samples=[0, 2, 3])),
        ("plink_hardy", big, {}),
        ("plink_missing", big, {}),
        ("plink_missing", big, dict(mode="sample")),
        ("read_pgen", big, dict(columns=["ID", "genotypes"])),
        ("plink_ld", big, dict(window_kb=1, r2_threshold=0.0, region="1:100-3000")),
        ("plink_score", ex, dict(weights=[1.0, 0.5, -0.5, 2.0])),
        ("plink_score", ex, dict(weights=[1.0, 0.5, -0.5, 2.0], no_mean_imputation=True, samples=[0, 2])),
        ("read_pfile", data_path("large_example"), dict(orient="sample", genotypes="counts")),
    ]
    try:
        for fn, path, named in calls:
            named = dict(named)
            cols = named.pop("columns", None)
            a = F.query(fn, path, columns=cols, settings=one, threads=4, **named)
            b = F.query(fn, path, columns=cols, settings=three, threads=4, **named)
            key = lambda r: tuple(str(x) for x in r)
            assert a.names == b.names and a.types == b.types
            assert sorted(a.rows, key=key) == sorted(b.rows, key=key), (fn, named)
        pa = F.query("plink_pca", data_path("pca_example.pgen"), n_pcs=3, settings=one)
        pb = F.query("plink_pca", data_path("pca_example.pgen"), n_pcs=3, settings=three)
        assert len(pa) == len(pb)
        for ra, rb in zip(pa.rows, pb.rows):
            for xa, xb in zip(ra, rb):
                if isinstance(xa, float):
                    assert abs(abs(xa) - abs(xb)) <= 1e-6  # eigenvectors are defined up to sign
                else:
                    assert xa == xb
        with pytest.raises(Exception, match="plinking_devices"):
            F.query("plink_freq", ex, settings={"plinking_devices": "0,x"})
        with pytest.raises(Exception, match="does not exist"):
            F.query("plink_freq", ex, settings={"plinking_devices": "0,63"})
    finally:
        F.query("plink_freq", ex, settings=one)

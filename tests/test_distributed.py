"""The N > 1 path: variant shards per rank, concatenation of per-variant outputs,
sum-reduce of per-sample partials, max-over-ranks timing.

CPU tests run 2 ranks over gloo with the oracle standing in for the per-rank
compute (they check the sharding/collective plumbing, which is all that changes
between 1 and N GPUs).  The GPU test runs the real HIP path on 2 ranks sharing
the box's single MI355X, again over gloo (RCCL needs one GPU per rank)."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, data_path
from plinking_duck_amd import sharding


def test_shard_ranges_partition_the_variant_axis():
    for world in (1, 2, 3, 8):
        for m in (0, 1, 7, 1000, 1_000_000):
            got = [sharding.shard_range(r, world, m, "strong") for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == m
            assert all(a[1] == b[0] for a, b in zip(got, got[1:]))
            weak = [sharding.shard_range(r, world, m, "weak") for r in range(world)]
            assert weak[-1][1] == world * m == sharding.total_variants(world, m, "weak")
    with pytest.raises(ValueError):
        sharding.shard_range(2, 2, 10)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _cpu_worker(rank, world, port, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    from oracle import oracle
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pg = oracle.Pgen(data_path("large_example.pgen"))
    v0, v1 = sharding.shard_range(rank, world, pg.M, "strong")
    counts = torch.from_numpy(pg.counts_range(v0, v1).astype(np.int64))          # per-variant: concatenated
    miss = torch.from_numpy(pg.missing_per_sample(v0, v1).astype(np.int64))      # per-sample: summed
    w = np.linspace(-1, 1, pg.M)
    s, d, ac = oracle.score(pg, range(v0, v1), w[v0:v1])
    score = torch.from_numpy(s.copy())
    sharding.reduce_partials(dist, [miss, score], dst=0)
    gathered = [torch.zeros_like(counts) for _ in range(world)] if rank == 0 else None
    dist.gather(counts, gathered, dst=0)  # equal shard sizes in this test
    slowest = sharding.max_over_ranks(dist, 1.0 + rank)
    dist.barrier()
    if rank == 0:
        np.save(os.path.join(out_dir, "miss.npy"), miss.numpy())
        np.save(os.path.join(out_dir, "score.npy"), score.numpy())
        np.save(os.path.join(out_dir, "counts.npy"), torch.cat(gathered).numpy())
        np.save(os.path.join(out_dir, "slowest.npy"), np.array([slowest]))
    dist.destroy_process_group()


def test_two_rank_gloo_shards_equal_the_whole(tmp_path, oracle):
    world = 2
    mp.spawn(_cpu_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    pg = oracle.Pgen(data_path("large_example.pgen"))
    assert np.array_equal(np.load(tmp_path / "counts.npy"), pg.counts_range().astype(np.int64))
    assert np.array_equal(np.load(tmp_path / "miss.npy"), pg.missing_per_sample().astype(np.int64))
    s, _, _ = oracle.score(pg, range(pg.M), np.linspace(-1, 1, pg.M))
    assert np.allclose(np.load(tmp_path / "score.npy"), s, rtol=1e-12, atol=1e-12)
    assert np.load(tmp_path / "slowest.npy")[0] == 2.0


def _gpu_worker(rank, world, port, out_dir, m, n, seed):
    import sys
    sys.path.insert(0, ROOT)
    import plinking_duck_amd.lib as L
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    L.set_device(0)
    v0, v1 = sharding.shard_range(rank, world, m, "strong")
    ds = L.Dataset.synth(v0, v1, n, seed, 0.03)
    st = torch.cuda.current_stream().cuda_stream
    d_counts = torch.zeros((v1 - v0, 4), dtype=torch.int32, device="cuda")
    d_miss = torch.zeros((n + 63) // 64 * 64, dtype=torch.int32, device="cuda")
    ds.counts_range_dev(v0, v1, d_counts.data_ptr(), st)
    ds.missing_per_sample_dev(v0, v1, d_miss.data_ptr(), st)
    ncol = 4
    w = np.random.default_rng(9).standard_normal((m, ncol))
    d_score = torch.zeros((n, ncol), dtype=torch.float64, device="cuda")
    d_dos = torch.zeros(n, dtype=torch.float64, device="cuda")
    d_ac = torch.zeros(n, dtype=torch.int32, device="cuda")
    ds.score_dev(np.arange(v0, v1, dtype=np.uint32), w[v0:v1], d_score.data_ptr(), d_dos.data_ptr(),
                 d_ac.data_ptr(), None, L.SCORE_MEAN_IMPUTE, st)
    torch.cuda.synchronize()
    miss, score, ac = d_miss[:n].cpu(), d_score.cpu(), d_ac.cpu()
    sharding.reduce_partials(dist, [miss, score, ac], dst=0)  # gloo here; nccl == RCCL on a multi-GPU node
    counts = d_counts.cpu()
    gathered = [torch.zeros_like(counts) for _ in range(world)] if rank == 0 else None
    dist.gather(counts, gathered, dst=0)
    dist.barrier()
    if rank == 0:
        np.save(os.path.join(out_dir, "miss.npy"), miss.numpy())
        np.save(os.path.join(out_dir, "score.npy"), score.numpy())
        np.save(os.path.join(out_dir, "ac.npy"), ac.numpy())
        np.save(os.path.join(out_dir, "counts.npy"), torch.cat(gathered).numpy())
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_on_the_gpu_equal_one_rank(tmp_path, gpu_lib):
    m, n, seed, world = 512, 20_003, 77, 2
    mp.spawn(_gpu_worker, args=(world, _free_port(), str(tmp_path), m, n, seed), nprocs=world, join=True)
    whole = gpu_lib.Dataset.synth(0, m, n, seed, 0.03)
    assert np.array_equal(np.load(tmp_path / "counts.npy").astype(np.uint32), whole.counts_range())
    assert np.array_equal(np.load(tmp_path / "miss.npy").astype(np.uint32), whole.missing_per_sample())
    w = np.random.default_rng(9).standard_normal((m, 4))
    s, d, ac = whole.score(np.arange(m), w)
    assert np.array_equal(np.load(tmp_path / "ac.npy").astype(np.uint32), ac)
    assert np.allclose(np.load(tmp_path / "score.npy"), s, rtol=1e-9, atol=1e-9)


# ---- plink_pca over variant shards -------------------------------------------------

def _norms_from_counts(counts):
    """Effective variants + (center, inv_stdev), as the shell's bind does (src/plink_pca.cpp:392-416)."""
    c = counts.astype(np.float64)
    obs = c[:, 0] + c[:, 1] + c[:, 2]
    af = np.where(obs > 0, (c[:, 1] + 2 * c[:, 2]) / np.maximum(2 * obs, 1), 0.0)
    keep = np.flatnonzero((obs > 0) & (af > 0) & (af < 1))
    return keep, 2 * af[keep], 1.0 / np.sqrt(2 * af[keep] * (1 - af[keep]))


def _cpu_pca_worker(rank, world, port, out_dir):
    """numpy restatement of the sharded iteration: the same sums the HIP path all-reduces."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(3)
    m, n, k = 600, 80, 2
    x = rng.standard_normal((m, n))  # stands for the normalised genotype matrix
    g1 = rng.standard_normal((n, 2 * k))
    v0, v1 = sharding.shard_range(rank, world, m, "strong")
    xs = x[v0:v1]

    def all_sum(a):
        t = torch.from_numpy(np.ascontiguousarray(a))
        dist.all_reduce(t)
        return t.numpy()

    blocks = []
    for p in range(k + 1):
        y = xs @ g1
        blocks.append(y)
        if p < k:
            g1 = all_sum(xs.T @ y) / m
    q = np.concatenate(blocks, axis=1)
    k2 = 2 * k
    for p in range(k + 1):  # block Gram-Schmidt through all-reduced Gram matrices
        bp = q[:, p * k2:(p + 1) * k2]
        for _ in range(2):
            if p:
                prev = q[:, :p * k2]
                bp -= prev @ all_sum(prev.T @ bp)
        for _ in range(2):
            lam, vec = np.linalg.eigh(all_sum(bp.T @ bp))
            bp[:] = bp @ (vec / np.sqrt(lam))
    bb = all_sum(xs.T @ q)
    ev = np.sort(np.linalg.eigvalsh(bb.T @ bb))[::-1][:k] / m
    gram = all_sum(q.T @ q)  # every rank takes part in every all-reduce
    if rank == 0:
        np.save(os.path.join(out_dir, "ev.npy"), ev)
        np.save(os.path.join(out_dir, "gram.npy"), gram)
    dist.destroy_process_group()


def test_two_rank_gloo_pca_sums_equal_the_whole(tmp_path):
    mp.spawn(_cpu_pca_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    rng = np.random.default_rng(3)
    m, n, k = 600, 80, 2
    x = rng.standard_normal((m, n))
    g1 = rng.standard_normal((n, 2 * k))
    blocks = []
    for p in range(k + 1):
        y = x @ g1
        blocks.append(y)
        if p < k:
            g1 = x.T @ y / m
    u, _, _ = np.linalg.svd(np.concatenate(blocks, axis=1), full_matrices=False)
    s = np.linalg.svd(x.T @ u, compute_uv=False)
    assert np.allclose(np.load(tmp_path / "ev.npy"), s[:k] ** 2 / m, rtol=1e-9)
    assert np.allclose(np.load(tmp_path / "gram.npy"), np.eye(2 * k * (k + 1)), atol=1e-10)


def _gpu_pca_worker(rank, world, port, out_dir, m, n, seed, k):
    import sys
    sys.path.insert(0, ROOT)
    import plinking_duck_amd.lib as L
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    L.set_device(0)
    v0, v1 = sharding.shard_range(rank, world, m, "strong")
    ds = L.Dataset.synth(v0, v1, n, seed, 0.03)
    keep, center, inv = _norms_from_counts(ds.counts_range())
    total = torch.tensor([len(keep)], dtype=torch.int64)
    dist.all_reduce(total)
    g1 = np.random.default_rng(11).standard_normal((n, 2 * k))
    ev, vecs = ds.pca_sharded(keep + v0, center, inv, int(total.item()), k, g1, sharding.device_allreduce(dist))
    np.save(os.path.join(out_dir, f"ev{rank}.npy"), ev)
    np.save(os.path.join(out_dir, f"vecs{rank}.npy"), vecs)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_pca_equals_one_rank(tmp_path, gpu_lib):
    m, n, seed, k, world = 3000, 1501, 31, 3, 2
    mp.spawn(_gpu_pca_worker, args=(world, _free_port(), str(tmp_path), m, n, seed, k), nprocs=world, join=True)
    whole = gpu_lib.Dataset.synth(0, m, n, seed, 0.03)
    keep, center, inv = _norms_from_counts(whole.counts_range())
    g1 = np.random.default_rng(11).standard_normal((n, 2 * k))
    ev, vecs = whole.pca(keep, center, inv, k, g1)
    for rank in range(world):
        assert np.allclose(np.load(tmp_path / f"ev{rank}.npy"), ev, rtol=1e-9)
        got = np.load(tmp_path / f"vecs{rank}.npy")
        for pc in range(k):
            sign = np.sign(np.dot(got[:, pc], vecs[:, pc]))
            assert np.allclose(got[:, pc] * sign, vecs[:, pc], atol=1e-7)
    # world size 1 through the sharded entry point: the callback is called and changes nothing
    ev1, _ = whole.pca_sharded(keep, center, inv, len(keep), k, g1, lambda p, c, s: None)
    assert np.allclose(ev1, ev, rtol=1e-12)

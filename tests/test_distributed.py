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

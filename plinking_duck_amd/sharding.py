"""Variant sharding across ranks (one process per GPU).

plink_freq / plink_hardy / plink_missing(variant) / read_pgen shard by contiguous
variant ranges with no data-path collective; plink_missing(sample)
and plink_score additionally sum their per-sample partials with one reduce;
plink_pca all-reduces G2 / the Krylov Gram blocks / BB once per pass."""

from __future__ import annotations


def shard_range(rank: int, world: int, variants: int, scaling: str = "weak") -> tuple[int, int]:
    """[v_begin, v_end) of the global variant axis owned by `rank`.

    weak:   every rank owns `variants` rows of a world*variants-row matrix
    strong: the `variants` rows are split into `world` contiguous, near-equal ranges"""
    if not 0 <= rank < world:
        raise ValueError("rank outside world")
    if scaling == "weak":
        return rank * variants, (rank + 1) * variants
    if scaling == "strong":
        per = (variants + world - 1) // world
        return min(variants, rank * per), min(variants, (rank + 1) * per)
    raise ValueError("scaling must be 'weak' or 'strong'")


def total_variants(world: int, variants: int, scaling: str = "weak") -> int:
    return variants * world if scaling == "weak" else variants


def reduce_partials(dist, tensors, dst: int = 0):
    """Sum per-sample partials of the variant shards onto rank `dst` (RCCL over xGMI
    when the process group is nccl; gloo in the CPU tests)."""
    for t in tensors:
        dist.reduce(t, dst=dst)


def max_over_ranks(dist, seconds: float, device=None) -> float:
    """The bench contract: a step takes as long as the slowest rank."""
    import torch

    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class _DeviceDoubles:
    """A span of doubles at a raw device pointer, in the shape torch.as_tensor adopts without a copy."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f8", "data": (ptr, False), "version": 2}


def device_allreduce(dist, group=None):
    """The `allreduce(d_ptr, count, stream)` callable Dataset.pca_sharded wants, over a
    torch.distributed process group (nccl == RCCL over xGMI on a multi-GPU node; gloo stages
    through the host).  The library enqueued its kernels on ITS stream; torch reduces on torch's
    current stream.  The two are ordered with events, both ways -- no device-wide drain: the
    collective waits for the library's producer kernels, the library's consumer kernels wait for
    the collective, and the host thread goes straight on enqueueing."""
    import torch

    def allreduce(d_ptr: int, count: int, stream: int):
        if count == 0:
            return
        lib_stream = torch.cuda.ExternalStream(stream)
        cur = torch.cuda.current_stream()
        produced = torch.cuda.Event()
        produced.record(lib_stream)
        cur.wait_event(produced)
        t = torch.as_tensor(_DeviceDoubles(d_ptr, count), device="cuda")
        dist.all_reduce(t, group=group)
        reduced = torch.cuda.Event()
        reduced.record(cur)
        lib_stream.wait_event(reduced)

    return allreduce

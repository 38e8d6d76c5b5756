"""Multi-GPU protocol of this op: REPLICAS ONLY.

The FP4 GEMM is a single-GPU primitive with no exchange step: tensor-
parallel sharding is the caller's, which hands every rank its own (size_n, size_k).  What exists
at N > 1 is therefore only the measurement protocol of bench.py: every rank runs the same
workload on its own weights, there is NO data-path collective, and the job's throughput is

    value = world_size * units_per_rank / max_over_ranks(time)

This module holds that protocol (backend-agnostic, so it is testable with gloo on CPUs) and the
shard-shape rule a caller must respect.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def max_over_ranks(local_ms: float, device=None) -> float:
    """Whole-job time = the slowest rank's time (all_reduce MAX; identity at world size 1)."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(local_ms)
    t = torch.tensor([local_ms], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def job_throughput(units_per_rank: float, local_ms: float, device=None):
    """(aggregate units/s over all ranks, ms of the slowest rank)."""
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    ms = max_over_ranks(local_ms, device)
    return world * units_per_rank / (ms * 1e-3), ms


def check_shard_shape(size_n: int, size_k: int) -> None:
    """A per-rank (size_n, size_k) partition must be packable on its own: N-shards are multiples
    of 16 rows (32 for MXFP4 scale tensors), K-shards multiples of 256 (csrc/layout.h)."""
    if size_n % 16 or size_k % 256:
        raise ValueError(f"shard ({size_n}, {size_k}) is not packable: need size_n % 16 == 0 and size_k % 256 == 0")


def column_parallel_shards(size_n: int, world: int):
    """Split N (independent output columns, no collective) into `world` packable shards."""
    tiles = size_n // 16
    if size_n % 16:
        raise ValueError("size_n must be a multiple of 16")
    base, extra = divmod(tiles, world)
    out, start = [], 0
    for r in range(world):
        n = (base + (1 if r < extra else 0)) * 16
        out.append((start, n))
        start += n
    return out

"""N > 1 path on CPU: world_size-2 gloo processes exercising the replicas-only protocol that
bench.py uses on RCCL (petit_kernel/replicas.py).  No GPU, no compute call."""
import os
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank: int, world: int, port: int, q):
    sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from petit_kernel import replicas
    local_ms = 10.0 + 5.0 * rank            # rank 1 is the slow one
    dist.barrier()
    agg, ms = replicas.job_throughput(units_per_rank=1000.0, local_ms=local_ms)
    shards = replicas.column_parallel_shards(10240, world)
    q.put((rank, agg, ms, shards[rank]))
    dist.barrier()
    dist.destroy_process_group()


def test_replicas_protocol_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, agg, ms, shard in got:
        assert ms == 15.0                                   # max over ranks
        assert agg == pytest.approx(2 * 1000.0 / 15e-3)     # whole-job aggregate
    assert got[0][3] == (0, 5120) and got[1][3] == (5120, 5120)


def test_shard_rules_single_process():
    sys.path.insert(0, str(ROOT / "petit-kernel_amd"))
    from petit_kernel import replicas
    assert replicas.max_over_ranks(3.5) == 3.5              # no process group: identity
    assert replicas.column_parallel_shards(57344, 8) == [(i * 7168, 7168) for i in range(8)]
    parts = replicas.column_parallel_shards(8192 + 16, 3)
    assert sum(n for _, n in parts) == 8208 and all(n % 16 == 0 for _, n in parts)
    replicas.check_shard_shape(1280, 8192)                  # TP=8 qkv shard (tools/benchmarks/matmul.py:18-33)
    with pytest.raises(ValueError):
        replicas.check_shard_shape(1288, 8192)
    with pytest.raises(ValueError):
        replicas.check_shard_shape(1280, 1000)

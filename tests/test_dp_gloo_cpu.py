"""world_size-2 gloo test of the data-parallel layer (bucketed all-reduce + shard partition), CPU only."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from multi_task_breast_cancer_amd.trainer import allreduce_buckets, global_permutation, plan_buckets, shard_positions


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        numel = 10_007 // 4 * 4
        slots, off = [], 0
        sizes = [1000, 24, 3000, 8, 4000, 1972]
        for i, n in enumerate(sizes):
            slots.append((off, n, len(sizes) - i))
            off += n
        buckets = plan_buckets(slots, off, 3)
        # per-rank "gradient" = gradient of the mean loss over the rank's shard of a linear model
        perm = global_permutation(64, seed=5, epoch=0)
        G = 16
        data = torch.arange(64 * off, dtype=torch.float64).reshape(64, off).remainder(17.0) - 8.0
        mine = shard_positions(perm, rank, world, G, step=1)
        flat = data[mine].mean(dim=0).float().contiguous()
        for async_op in (False, True):
            g = flat.clone()
            works = allreduce_buckets(g, buckets, async_op=async_op)
            for w in works:
                w.wait()
            g *= 1.0 / world                     # grad_scale folded into Adam in the real step
            want = data[shard_positions(perm, 0, 1, G, step=1)].mean(dim=0).float()
            assert torch.allclose(g, want, atol=1e-5), (g - want).abs().max()
        q.put((rank, "ok"))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_equals_single_rank_gradient():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(0, "ok"), (1, "ok")]

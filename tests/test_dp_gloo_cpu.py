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


def test_bucket_order_is_a_function_of_the_layout_only():
    """Ranks may build different step programs (short last batch): their collectives still pair up because the bucket
    ranges AND their order depend on the parameter layout alone, never on the readiness indices."""
    sizes = [1000, 24, 3000, 8, 4000, 1972, 512, 64]
    offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist()
    total = int(sum(sizes))
    rng = np.random.default_rng(0)
    ref = None
    for _ in range(5):
        ready = rng.permutation(len(sizes)).tolist()
        b = plan_buckets([(o, n, r) for o, n, r in zip(offs, sizes, ready)], total, 3)
        rng_ = [(x.start, x.end) for x in b]
        assert rng_ == sorted(rng_, reverse=True) and rng_[-1][0] == 0 and rng_[0][1] == total
        assert sum(e - s_ for s_, e in rng_) == total
        for x in b:     # a bucket waits for the LAST op that writes any of its parameters
            assert x.ready_op == max(r for o, n, r in zip(offs, sizes, ready) if x.start <= o < x.end)
        ref = ref or rng_
        assert rng_ == ref

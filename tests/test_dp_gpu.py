"""Data-parallel step with REAL kernels on 2 ranks (both on the one visible GPU, gloo as transport: RCCL refuses
two ranks on one device): the sharded 2-rank step must reproduce the 1-rank step on the same global batch."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, G=4, cuts=None):
    import torch.distributed as dist
    from multi_task_breast_cancer_amd.miscellany import seed_everything
    from multi_task_breast_cancer_amd.nets import MTUNetPlusPlus
    from multi_task_breast_cancer_amd.optim import FusedAdam
    from multi_task_breast_cancer_amd.trainer import FusedTrainStep
    from oracle import torch_oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda:0")
        seed_everything(1993)
        m = MTUNetPlusPlus(in_channels=1, out_channels=1, n_classes=3, deep_supervision=True).to(dev)
        step = FusedTrainStep(m, FusedAdam(m, lr=1e-4, eps=1e-4), alpha=0.5, distributed=True, n_buckets=4)
        img, mask, label = O.synthetic_batch(G, 64, 64, seed=7)          # the GLOBAL batch
        if cuts is None:
            per = G // world
            sl, weight = slice(rank * per, (rank + 1) * per), None        # equal contiguous shards
        else:                                                             # the short last batch of drop_last=False: uneven shards
            sl = slice(cuts[rank], cuts[rank + 1])
            weight = (cuts[rank + 1] - cuts[rank]) / G                    # EpochIndex.weights
        losses = step(img[sl].to(dev), mask[sl].to(dev), label[sl].to(dev), weight=weight)
        torch.cuda.synchronize()
        assert len(step._st.buckets) >= 2 and [b.start for b in step._st.buckets] == sorted((b.start for b in step._st.buckets), reverse=True)
        if rank == 0:
            # numpy arrays travel by value; CPU tensors would travel as shared-memory handles that die with this process
            q.put((m.flat_p.cpu().numpy(), m.flat_g.cpu().numpy(), losses.cpu().numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("G,cuts", [(4, None), (5, (0, 3, 5))])
def test_two_rank_step_equals_single_rank_global_batch(G, cuts):
    import torch.multiprocessing as mp
    from multi_task_breast_cancer_amd.miscellany import seed_everything
    from multi_task_breast_cancer_amd.nets import MTUNetPlusPlus
    from multi_task_breast_cancer_amd.optim import FusedAdam
    from multi_task_breast_cancer_amd.trainer import FusedTrainStep
    from oracle import torch_oracle as O
    import queue as _queue
    world = 2
    ctx = mp.get_context("spawn")
    got = None
    for attempt in range(2):          # a 2-process gloo rendezvous on one box has been seen to stall once; a run takes ~6 s
        q, port = ctx.Queue(), _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, q, G, cuts)) for r in range(world)]
        for p in procs:
            p.start()
        try:
            got = q.get(timeout=120)
        except _queue.Empty:
            got = None
        for p in procs:
            p.join(60 if got is not None else 1)
            if p.is_alive():
                p.kill()              # exactly the processes this test started
                p.join(10)
        if got is not None:
            assert all(p.exitcode == 0 for p in procs)
            break
    assert got is not None, "the two-rank run produced no result in two attempts"
    p2, g2, l2 = (torch.from_numpy(a) for a in got)
    dev = torch.device("cuda:0")
    seed_everything(1993)
    m = MTUNetPlusPlus(in_channels=1, out_channels=1, n_classes=3, deep_supervision=True).to(dev)
    step = FusedTrainStep(m, FusedAdam(m, lr=1e-4, eps=1e-4), alpha=0.5)
    img, mask, label = O.synthetic_batch(G, 64, 64, seed=7)
    l1 = step(img.to(dev), mask.to(dev), label.to(dev)).cpu()
    p1, g1 = m.flat_p.cpu(), m.flat_g.cpu()
    # rank-0 loss is the mean over ITS shard; the gradient (sum over ranks, scaled 1/world in Adam) is the global one --
    # also for uneven shards, where each rank weighted its local mean-loss gradient by n_local / G on the device
    g2 = g2 / world
    rel = (g2 - g1).norm().item() / g1.norm().item()
    assert rel < 1e-5, rel
    assert (p2 - p1).abs().max().item() < 2e-6
    assert l2[3].item() == 0.0 and abs(l1[0].item()) > 0


def test_rccl_collectives_beside_the_cooperative_kernels_at_the_bench_plane_size(monkeypatch):
    """SURVEY 8(e) readiness on one GPU: the data-parallel step of the BENCH model -- MTUNetPlusPlus, bf16, 256 x 256 planes, i.e. the cooperative
    InstanceNorm backward in teams of 32 and the wide-block weight gradients -- with REAL RCCL all-reduce kernels on the communication stream
    (world 1: the collectives still launch) beside the compute stream's cooperative launches, planned the way an 8-rank trainer plans it
    (coop_reserve_cus = 64).  For 4 buckets (the default), ONE bucket (everything reduced after the last backward op) and 8 (more cuts than
    the default: every boundary between parameter groups is exercised) the step must reproduce the local step planned with the same
    reserve BIT FOR BIT, twice in a row, and the cooperative kernels' sticky error word must stay 0 (a member that was not resident would
    set it).  No reference counterpart: the reference is single-device (experiment_init.py:339-347)."""
    import torch.distributed as dist
    from multi_task_breast_cancer_amd.miscellany import seed_everything
    from multi_task_breast_cancer_amd.nets import MTUNetPlusPlus
    from multi_task_breast_cancer_amd.optim import FusedAdam
    from multi_task_breast_cancer_amd.trainer import FusedTrainStep
    from oracle import torch_oracle as O
    dev = torch.device("cuda:0")
    monkeypatch.setenv("MTBC_COOP_RESERVE_CUS", "64")         # world 1 plans like world 8 (trainer.py: the reserve applies when world > 1 or the variable is set)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        batches = [O.synthetic_batch(2, 256, 256, seed=60 + s) for s in range(2)]

        def run(distributed, n_buckets):
            seed_everything(1993)
            m = MTUNetPlusPlus(in_channels=1, out_channels=1, n_classes=3, deep_supervision=True).to(dev)
            m.set_compute("bf16")
            m.coop_reserve_cus = 64
            step = FusedTrainStep(m, FusedAdam(m, lr=1e-4, eps=1e-4), alpha=0.5, distributed=distributed, n_buckets=n_buckets)
            for img, mask, label in batches:
                l = step(img.to(dev), mask.to(dev), label.to(dev))
            torch.cuda.synchronize()
            step.check_nan()                                   # raises MtbcError on a non-zero cooperative error word
            assert m.coop_reserve_cus == 64
            assert m.coop_error_word() is not None and int(m.coop_error_word().item()) == 0
            kinds = {step._st.programs["bwd"].array[i].kind for i in range(step._st.programs["bwd"].n)}
            return m.flat_p.clone(), m.flat_g.clone(), l.clone(), step

        p0, g0, l0, _ = run(False, 4)
        for nb in (4, 1, 8):
            p1, g1, l1, step = run(True, nb)
            bk = step._st.buckets
            assert len(bk) == nb, (nb, len(bk))
            assert [b.start for b in bk] == sorted((b.start for b in bk), reverse=True)
            assert bk[-1].start == 0 and bk[0].end == step.model.flat_numel and all(a.start == b.end for a, b in zip(bk[:-1], bk[1:]))
            assert torch.equal(l0, l1), nb
            assert torch.equal(g0, g1), nb                      # world 1: the sum over ranks is the local gradient itself
            assert torch.equal(p0, p1), nb
    finally:
        dist.destroy_process_group()

"""Round 4: the training step (bench.py's configs[1] workload) replayed as ONE hipGraph -- trainer.FusedTrainStep(graph=True), the MTBC_GRAPH switch --
against the stream-ordered programs.  On one box, interleaved: host time to ISSUE a step (no sync), ms per step, and that both ways leave the same
parameters after the same steps.
usage: python tools/experiments/graph_replay.py [dtype=bf16] [batch=32] [size=256] [arch=MTUNetPlusPlus] [sync_each_step=0]
(sync_each_step=1: the loop reads the loss back after every step, as a training loop that logs `loss.item()` does: the host cannot run ahead)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multi_task_breast_cancer_amd.experiment_init import init_multitask_model, init_optimizer
from multi_task_breast_cancer_amd.miscellany import seed_everything
from multi_task_breast_cancer_amd.synthetic import synthetic_batch
from multi_task_breast_cancer_amd.trainer import FusedTrainStep

dtype = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
arch = sys.argv[4] if len(sys.argv) > 4 else "MTUNetPlusPlus"
SYNC = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
dev = "cuda:0"
STEPS, ROUNDS = 30, 3
batches = [synthetic_batch(B, S, S, seed=s, device=dev) for s in range(2)]
arms = {}
for graph in (False, True):
    seed_everything(1993)
    model = init_multitask_model(arch, sequences=1, regions=1, n_classes=3, deep_supervision=True).to(dev)
    model.set_compute(dtype)
    step = FusedTrainStep(model, init_optimizer(model, "Adam", 1e-4), alpha=0.5, inversely_weighted=True, graph=graph)
    for i in range(6):
        step(*batches[i % 2])
    torch.cuda.synchronize()
    arms[graph] = (model, step)
for rnd in range(ROUNDS):
    for graph in (False, True):
        model, step = arms[graph]
        t0 = time.perf_counter()
        for i in range(STEPS):
            l = step(*batches[i % 2])
            if SYNC:
                l[0].item()
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"round {rnd} [{'hipGraph replay' if graph else 'stream-ordered  '}]{' loss read back every step:' if SYNC else ''} host issue {1e3 * t_issue / STEPS:6.3f} ms per step, {1e3 * t_all / STEPS:7.3f} ms per step", flush=True)
pa, pb = arms[False][0].flat_p, arms[True][0].flat_p
print(f"parameters after {6 + ROUNDS * STEPS} steps: bit-identical = {bool(torch.equal(pa, pb))} (max |diff| {(pa - pb).abs().max().item():.3e}); "
      f"losses {[round(v, 6) for v in arms[False][1].losses.cpu().tolist()[:3]]} / {[round(v, 6) for v in arms[True][1].losses.cpu().tolist()[:3]]}")

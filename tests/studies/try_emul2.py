"""Scratch (round 3): HIP gradients of the 16-bit modes against the HIP fp32 mode, per layer, at 256x256 N=2 (is the fp16 mode further from
fp32 than bf16 is?  it keeps 3 more significant bits, so it should be closer)."""
import sys, os
sys.path.insert(0, os.getcwd())
import importlib.util
spec = importlib.util.spec_from_file_location("tm", "tests/test_model_gpu.py"); tm = importlib.util.module_from_spec(spec); spec.loader.exec_module(tm)
import torch
O, DEV = tm.O, tm.DEV
from multi_task_breast_cancer_amd.trainer import FusedTrainStep
from multi_task_breast_cancer_amd.optim import FusedAdam
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
size, N = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 2
g = {}
for dtype in ("f32", "bf16", "f16"):
    prod, ref = tm._oracle_and_product("MTUNetPlusPlus", 1993)
    prod.set_compute(dtype)
    img, mask, label = O.synthetic_batch(N, size, size, seed=21)
    step = FusedTrainStep(prod, FusedAdam(prod, lr=1e-4, eps=1e-4), alpha=0.5)
    st = step.load_batch(img.to(DEV), mask.to(DEV), label.to(DEV))
    step.run(st)
    g[dtype] = {n: (prod._grad_view(n) / prod.loss_scale).clone() for n in prod._order}
rows = [(rel(g["f16"][n], g["f32"][n]), rel(g["bf16"][n], g["f32"][n]), n) for n in g["f32"] if g["f32"][n].norm().item() > 0 and not n.endswith("conv.bias")]
rows.sort(reverse=True)
for a, b, n in rows[:12]:
    print(f"{n:44s} f16 vs f32 {a:.4f}   bf16 vs f32 {b:.4f}")
import statistics
print("median f16", statistics.median(r[0] for r in rows), "median bf16", statistics.median(r[1] for r in rows))

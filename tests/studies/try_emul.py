"""Scratch (round 3): the fp16 mode's whole-model gradient error against the fp64-accumulating emulation as a function of the static
loss scale and the plane size (tests/test_model_gpu.py::test_16bit_mfma_modes_match_their_emulation's body with the scale overridden).
usage: python tests/studies/try_emul.py"""
import sys, os, copy
sys.path.insert(0, os.getcwd())
import importlib.util
spec = importlib.util.spec_from_file_location("tm", "tests/test_model_gpu.py"); tm = importlib.util.module_from_spec(spec); spec.loader.exec_module(tm)
import torch
O, DEV = tm.O, tm.DEV
from multi_task_breast_cancer_amd.trainer import FusedTrainStep
from multi_task_breast_cancer_amd.optim import FusedAdam
rel = lambda a, b: ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()
for dtype, size, N, ls in [("f16", 256, 2, 4096.0), ("f16", 256, 2, 65536.0), ("f16", 256, 2, 1048576.0), ("f16", 64, 4, 4096.0), ("f16", 64, 4, 65536.0)]:
    prod, ref = tm._oracle_and_product("MTUNetPlusPlus", 1993)
    prod.set_compute(dtype)
    prod.loss_scale = ls
    ref64 = copy.deepcopy(ref).double()
    img, mask, label = O.synthetic_batch(N, size, size, seed=21)
    step = FusedTrainStep(prod, FusedAdam(prod, lr=1e-4, eps=1e-4), alpha=0.5)
    st = step.load_batch(img.to(DEV), mask.to(DEV), label.to(DEV))
    losses = step.run(st).cpu()
    with O.lowp_conv3x3(dtype, model=[ref, ref64]):
        t32 = O.train_step(ref, O.make_adam(ref, 1e-4), img, mask, label, 0.5, True, 3, loss_scale=ls)
        t64 = O.train_step(ref64, O.make_adam(ref64, 1e-4), img.double(), mask.double(), label, 0.5, True, 3, loss_scale=ls)
    g32, g64 = dict(ref.named_parameters()), dict(ref64.named_parameters())
    worst = []
    for name in prod._order:
        if name.endswith("conv.bias") or g64[name].grad.norm().item() == 0.0:
            continue
        e_hip, e_cpu = rel(prod._grad_view(name) / ls, g64[name].grad), rel(g32[name].grad, g64[name].grad)
        worst.append((e_hip / max(e_cpu, 1e-9), name, e_hip, e_cpu))
    worst.sort(reverse=True)
    print(dtype, size, "scale", ls, "nan flag", losses[3].item(), "worst ratios:", [(n, round(a, 3), round(b, 3)) for _, n, a, b in worst[:3]], flush=True)

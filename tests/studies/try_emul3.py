"""Scratch (round 4): the whole-model 16-bit-vs-emulation comparison of tests/test_model_gpu.py::test_16bit_mfma_modes_match_their_emulation at a state that
is NOT the initialisation: K optimisation steps of the (parity-tested) HIP fp32 mode first, then ONE step of the 16-bit mode against the emulation from
those weights.  Prints, per parameter tensor, e_hip / e_cpu / cosine -- are the level-0 encoder gradients still cancellation residues there?
usage: try_emul3.py SIZE N DTYPE K LR [ARM]      DTYPE f32 = the parity mode against the plain fp64 oracle; ARM in da16 | no_z16 | no_gather | z_bf16 (round 4's runs used the names of the time: the default plan had da16 on, 'no_da16' switched it off)"""
import sys, os, time, copy
sys.path.insert(0, os.getcwd())
import importlib.util
spec = importlib.util.spec_from_file_location("tm", "tests/test_model_gpu.py"); tm = importlib.util.module_from_spec(spec); spec.loader.exec_module(tm)
import torch
O, DEV = tm.O, tm.DEV
from multi_task_breast_cancer_amd.trainer import FusedTrainStep
from multi_task_breast_cancer_amd.optim import FusedAdam
size, N, dtype, K, lr = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), float(sys.argv[5])
arm = sys.argv[6] if len(sys.argv) > 6 else ""
import contextlib
from multi_task_breast_cancer_amd import engine
emu = {"": {"da16": False}, "da16": {"da16": True}, "no_gather": {"da16": False}, "no_z16": {"z16": False}, "z_bf16": {"z_fp16": False, "da16": False}}[arm]
if arm == "no_gather": engine._NO_GATHER = True
if arm == "no_z16": engine._NO_Z16 = True
if arm == "da16": engine._DA16 = True
if arm == "z_bf16": engine._Z_BF16 = True
rel = lambda a, b: ((a.double().cpu() - b.double()).norm() / b.double().norm()).item()
prod, ref = tm._oracle_and_product("MTUNetPlusPlus", 1993)
if K:
    pre = FusedTrainStep(prod, FusedAdam(prod, lr=lr, eps=1e-4), alpha=0.5)
    for s in range(K):
        img, mask, label = O.synthetic_batch(max(N, 4), size, size, seed=100 + s)
        l = pre(img.to(DEV), mask.to(DEV), label.to(DEV))
    print("pretrain losses", l.cpu().tolist(), flush=True)
    ref.load_state_dict({k: v.detach().cpu().clone() for k, v in prod.state_dict().items()})
prod.set_compute(dtype)
ref64 = copy.deepcopy(ref).double()
img, mask, label = O.synthetic_batch(N, size, size, seed=21)
step = FusedTrainStep(prod, FusedAdam(prod, lr=1e-4, eps=1e-4), alpha=0.5)
st = step.load_batch(img.to(DEV), mask.to(DEV), label.to(DEV))
losses = step.run(st).cpu()
ls = prod.loss_scale
t0 = time.time()
with (contextlib.nullcontext() if dtype == 'f32' else O.lowp_conv3x3(dtype, model=[ref, ref64], **emu)):
    t32 = O.train_step(ref, O.make_adam(ref, 1e-4), img, mask, label, 0.5, True, 3, loss_scale=ls)
    t64 = O.train_step(ref64, O.make_adam(ref64, 1e-4), img.double(), mask.double(), label, 0.5, True, 3, loss_scale=ls)
print(f"[{dtype} {arm or 'default'} K={K} lr={lr}] cpu emulation {time.time() - t0:.1f} s; loss hip {losses[0].item():.6f} cpu32 {t32[0].item():.6f} cpu64 {t64[0].item():.6f}", flush=True)
g32, g64 = dict(ref.named_parameters()), dict(ref64.named_parameters())
rows = []
for name in prod._order:
    if name.endswith("conv.bias") or g64[name].grad.norm().item() == 0.0:
        continue
    e_hip, e_cpu = rel(prod._grad_view(name) / ls, g64[name].grad), rel(g32[name].grad, g64[name].grad)
    gh, gr = (prod._grad_view(name) / ls).double().cpu().flatten(), g64[name].grad.double().flatten()
    rows.append((e_hip / max(e_cpu, 1e-12), e_hip, e_cpu, (gh @ gr / (gh.norm() * gr.norm())).item(), name))
rows.sort(reverse=True)
print(f"{'tensor':52s} e_hip   e_cpu   ratio  cos")
for r, eh, ec, cs, n in rows[:25]:
    print(f"{n:52s} {eh:.4f}  {ec:.4f}  {r:5.2f}  {cs:.4f}")
print("max e_cpu", max(r[2] for r in rows), "max e_hip", max(r[1] for r in rows), "tensors with e_cpu > 0.1:", sum(r[2] > 0.1 for r in rows),
      "failing 3x bar (floor 5e-2):", sum(r[1] >= max(3 * r[2], 5e-2) for r in rows))

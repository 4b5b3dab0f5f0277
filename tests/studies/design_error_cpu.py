"""Round 4, CPU only (no GPU, no product code): how far is each 16-bit DESIGN from the exact gradient?  The oracle's emulation of a mode (oracle.lowp_conv3x3,
fp64 accumulation) against the plain fp64 oracle, per parameter tensor, at weights that K fp32 oracle steps have moved away from the initialisation.
Asks whether the fp16 mode's arithmetic (fp16 operands AND fp16-stored conv outputs) carries a larger gradient error somewhere than the bf16 mode's
(profiles/r04_quality_hard.md: a slow mode of convergence that only fp16 runs with 16-bit conv outputs end in).
usage: python tests/studies/design_error_cpu.py [SIZE=128] [N=2] [K=30] [LR=1e-3] [hard=1]"""
import copy
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import torch_oracle as O

size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2
K = int(sys.argv[3]) if len(sys.argv) > 3 else 30
lr = float(sys.argv[4]) if len(sys.argv) > 4 else 1e-3
torch.set_num_threads(8)
O.seed_everything(1993)
ref = O.build_oracle_model("MTUNetPlusPlus", 1, 1, 3, True)
opt = O.make_adam(ref, lr)
t0 = time.time()
for s in range(K):
    img, mask, label = O.synthetic_batch(4, size, size, seed=100 + s)
    l = O.train_step(ref, opt, img, mask, label, 0.35, True, 3)
print(f"{K} fp32 oracle steps at {size} x {size}: loss {l[0].item():.4f} ({time.time() - t0:.0f} s)", flush=True)
img, mask, label = O.synthetic_batch(N, size, size, seed=21)


def grads(ctx, loss_scale=1.0):
    m = copy.deepcopy(ref).double()
    with ctx(m):
        O.train_step(m, O.make_adam(m, 1e-4), img.double(), mask.double(), label, 0.35, True, 3, loss_scale=loss_scale)
    return {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}


import contextlib
exact = grads(lambda m: contextlib.nullcontext())
designs = {
    "bf16, z fp16 (shipped bf16)": (lambda m: O.lowp_conv3x3("bf16", model=[m], da16=False), 1.0),
    "f16, z fp16 (shipped fp16)": (lambda m: O.lowp_conv3x3("f16", model=[m], da16=False), 4096.0),
    "f16, z fp32 (MTBC_NO_Z16)": (lambda m: O.lowp_conv3x3("f16", model=[m], z16=False), 4096.0),
    "bf16, z fp32 (MTBC_NO_Z16)": (lambda m: O.lowp_conv3x3("bf16", model=[m], z16=False), 1.0),
    "f16 + 16-bit gradients (MTBC_DA16)": (lambda m: O.lowp_conv3x3("f16", model=[m], da16=True), 4096.0),
    "f16, z fp16, loss scale 65536": (lambda m: O.lowp_conv3x3("f16", model=[m], da16=False), 65536.0),
    "f16, z fp16, loss scale 2^20": (lambda m: O.lowp_conv3x3("f16", model=[m], da16=False), 1048576.0),
    "f16, z fp16, loss scale 1": (lambda m: O.lowp_conv3x3("f16", model=[m], da16=False), 1.0),
    "bf16 + 16-bit gradients (MTBC_DA16)": (lambda m: O.lowp_conv3x3("bf16", model=[m], da16=True), 1.0),
    "f16, loss scale 65536 + 16-bit gradients": (lambda m: O.lowp_conv3x3("f16", model=[m], da16=True), 65536.0),
}
res = {}
for name, (ctx, ls) in designs.items():
    t0 = time.time()
    g = grads(ctx, ls)
    res[name] = {n: ((g[n] - exact[n]).norm() / exact[n].norm()).item() for n in exact if exact[n].norm().item() > 0 and not n.endswith("conv.bias")}
    print(f"{name}: {time.time() - t0:.0f} s", flush=True)
names = list(res)
keys = sorted(res[names[0]], key=lambda n: -res[names[int(os.environ.get('SORT_BY', '1'))]][n])
print(f"\nrelative gradient error of each design against the exact (fp64) gradient, the 30 tensors where the shipped fp16 design is worst; U-Net++ {size} x {size}, N = {N}, after {K} steps")
print(f"{'tensor':50s} " + " ".join(f"{i}" .rjust(8) for i in range(len(names))))
for n in keys[:30]:
    print(f"{n:50s} " + " ".join(f"{res[d][n]:8.4f}" for d in names))
for i, d in enumerate(names):
    v = torch.tensor(list(res[d].values()))
    print(f"[{i}] {d}: median {v.median().item():.4f}, mean {v.mean().item():.4f}, max {v.max().item():.4f}")

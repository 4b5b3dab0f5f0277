"""Margin of the cooperative InstanceNorm statistics (single pass against a pivot, team sums) against fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multi_task_breast_cancer_amd import ops
dev = "cuda:0"
for (N, C, H, W, shift) in [(32, 24, 256, 256, 0.5), (32, 24, 256, 256, 20.0), (8, 48, 128, 128, 0.5), (8, 48, 128, 128, -50.0)]:
    g = torch.Generator().manual_seed(N + C)
    z = (torch.randn(N, C, H, W, generator=g) * 2 + shift).to(dev)
    zd = z.double().view(N * C, -1)
    mean64 = zd.mean(1); rstd64 = 1.0 / torch.sqrt(zd.var(1, unbiased=False) + 1e-5)
    _, m1, r1 = ops.instnorm_lrelu_fwd(z)
    _, m2, r2, _ = ops.instnorm_lrelu_fwd_c8(z, compute=1)
    rel = lambda a, b: ((a.double() - b).abs() / b.abs().clamp_min(1e-12)).max().item()
    print(f"N={N} C={C} {H}x{W} shift {shift:+.1f}: one-plane kernel rstd err {rel(r1, rstd64):.2e} mean err {(m1.double()-mean64).abs().max().item():.2e} | "
          f"cooperative rstd err {rel(r2, rstd64):.2e} mean err {(m2.double()-mean64).abs().max().item():.2e}")

"""Planar-fp32-input 16-bit convs against the same convs on 16-bit channel-blocked operands (MTBC_LAYOUT_C8), through
the C-ABI, at the bench's layer shapes.  usage: python tools/c8_probe.py [compute=1]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_task_breast_cancer_amd import ops

compute = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = "cuda:0"
# (N, segs, Cout, S): U-Net++ B=32 256x256 nodes
SHAPES = [(32, [24], 24, 256), (32, [24, 24, 24], 24, 256), (32, [24] * 6, 24, 256), (32, [48, 48, 48], 48, 128),
          (32, [96, 96, 96], 96, 64), (32, [192, 192], 192, 32), (32, [384, 384], 384, 16)]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps


print(f"compute={compute}  (ms: planar -> c8)")
for N, segs, Cout, S in SHAPES:
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(N, c, S, S, generator=g).to(dev) for c in segs]
    Cin = sum(segs)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05).to(dev)
    b = torch.zeros(Cout, device=dev)
    dz = torch.randn(N, Cout, S, S, generator=g).to(dev)
    pf, pd = ops.conv3x3_pack_lp(w, compute)
    x8 = [ops.C8.pack(x, compute) for x in xs]
    dz8 = ops.C8.pack(dz, compute)
    dxs = [torch.zeros_like(x) for x in xs]
    acc = [1] * len(xs)
    t = {}
    t["fwd"] = (timeit(lambda: ops.conv3x3_fwd(xs, w, b, packed=pf, compute=compute)), timeit(lambda: ops.conv3x3_fwd_c8(x8, w, b, pf)))
    t["dgrad"] = (timeit(lambda: ops.conv3x3_dgrad(dz, w, dxs, acc, packed=pd, compute=compute)), timeit(lambda: ops.conv3x3_dgrad_c8(dz8, w, dxs, acc, pd)))
    dw = torch.empty_like(w)
    t["wgrad"] = (timeit(lambda: ops.conv3x3_wgrad(xs, dz, tuple(w.shape), want_bias=True, dw=dw, compute=compute)),
                  timeit(lambda: ops.conv3x3_wgrad_c8(x8, dz8, tuple(w.shape), want_bias=True, dw=dw)))
    fl = 2.0 * N * S * S * Cin * Cout * 9 / 1e9
    print(f"{Cin:4d}->{Cout:3d} @{S:3d} segs{len(segs)}: " + "  ".join(
        f"{k} {a:.3f}->{c:.3f} ({a / c:.2f}x, {fl / c:.0f} TF)" for k, (a, c) in t.items()), flush=True)

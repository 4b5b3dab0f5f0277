"""One channel-blocked 3x3 forward shape (16-bit channel-blocked output + epilogue statistics, the training step's form) through the C-ABI,
10 launches: for rocprofv3 --pmc runs.   usage: python tools/experiments/c8_fwd_one.py SEG[,SEG...] COUT SIZE [N=32]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from multi_task_breast_cancer_amd import ops
segs, cout, S = [int(v) for v in sys.argv[1].split(",")], int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
cin = sum(segs)
x8 = [ops.C8.pack(torch.randn(N, c, S, S, generator=g, device=dev), 1) for c in segs]
w = torch.randn(cout, cin, 3, 3, generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
b = torch.randn(cout, generator=g, device=dev)
pf, pd = ops.conv3x3_pack_lp(w, 1)
for _ in range(3):
    ops.conv3x3_fwd_c8(x8, w, b, pf, out_c8=True, stats=True, out_fp16=True)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    ops.conv3x3_fwd_c8(x8, w, b, pf, out_c8=True, stats=True, out_fp16=True)
e.record(); e.synchronize()
ms = s.elapsed_time(e) / 10
print(f"{cin}->{cout} @{S} N={N}: {ms * 1e3:.1f} us, {2.0 * N * S * S * cin * cout * 9 / ms / 1e9:.1f} TF")

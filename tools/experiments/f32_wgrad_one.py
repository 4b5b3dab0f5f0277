"""One fp32 (parity-mode) weight-gradient shape through the C-ABI, 10 launches: for rocprofv3 --pmc runs.
usage: python tools/experiments/f32_wgrad_one.py CIN COUT SIZE [N=32]"""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from multi_task_breast_cancer_amd import ops
cin, cout, S = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 32
dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn(N, cin, S, S, generator=g, device=dev)
dz = torch.randn(N, cout, S, S, generator=g, device=dev)
dw = torch.empty(cout, cin, 3, 3, device=dev)
for _ in range(3):
    ops.conv3x3_wgrad([x], dz, (cout, cin, 3, 3), want_bias=False, dw=dw)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(10):
    ops.conv3x3_wgrad([x], dz, (cout, cin, 3, 3), want_bias=False, dw=dw)
e.record(); e.synchronize()
ms = s.elapsed_time(e) / 10
print(f"{cin}->{cout} @{S} N={N}: {ms * 1e3:.1f} us, {2.0 * N * S * S * cin * cout * 9 / ms / 1e9:.1f} TF")

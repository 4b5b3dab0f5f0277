"""Times the ConvTranspose k=2 kernels per shape (U-Net++ B=32 256x256 up-convolutions)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_task_breast_cancer_amd import ops
DEV = torch.device("cuda:0")
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
for (N, ci, co, H) in [(32, 48, 48, 128), (32, 96, 48, 64), (32, 192, 96, 32), (32, 384, 192, 16)]:
    x = torch.randn(N, ci, H, H, device=DEV); w = torch.randn(ci, co, 2, 2, device=DEV) * 0.1
    dy = torch.randn(N, co, 2 * H, 2 * H, device=DEV); b = torch.zeros(co, device=DEV)
    gb = (x.numel() + dy.numel()) * 4 / 1e9
    line = f"convT {ci}->{co} @{H}: fwd {t(lambda: ops.convT_fwd(x, w, b, 2)):.3f} ms |"
    for mode in (0, 1):
        line += f" m{mode}: dgrad {t(lambda: ops.convT_dgrad(x, w, dy, 2, compute=mode)):.3f} wgrad {t(lambda: ops.convT_wgrad(x, w, dy, 2, want_bias=False, compute=mode)):.3f} |"
    print(line + f" floor@5TB/s {gb / 5 :.3f} ms", flush=True)

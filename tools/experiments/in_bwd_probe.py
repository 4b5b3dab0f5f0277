"""Times the channel-group InstanceNorm backward (z, dy 16-bit channel-blocked) stand-alone: tools/in_bwd_probe.py [N C H W extra]"""
import sys, torch
sys.path.insert(0, ".")
from multi_task_breast_cancer_amd import ops
N, C, H, W, extra = (int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (32, 24, 256, 256, 0)))
dev = torch.device("cuda")
z = ops.C8.pack(torch.randn(N, C, H, W, device=dev) * 2 + 0.5, 1)
dy = ops.C8.pack(torch.randn(N, C, H, W, device=dev), 1)
ex = torch.randn(N, C, H, W, device=dev) if extra else None
g, b = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
_, mean, rstd, _ = ops.instnorm_lrelu_fwd_c8(z, g, b, slope=0.1)
db = torch.zeros(C, device=dev)
for _ in range(3):
    ops.instnorm_lrelu_bwd_c8(z, dy, mean, rstd, g, b, slope=0.1, dbias_pre=db, dy_extra=ex)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.instnorm_lrelu_bwd_c8(z, dy, mean, rstd, g, b, slope=0.1, dbias_pre=db, dy_extra=ex)
    e.record(); e.synchronize()
    ts.append(s.elapsed_time(e))
print(f"in_bwd N{N} C{C} {H}x{W} extra={extra}: {sorted(ts)[len(ts)//2]*1e3:.1f} us")

"""Is a conv output that was just written served from the memory-side cache (MALL, 256 MB) when InstanceNorm reads it?
Times the cooperative InstanceNorm forward on z right after the conv that wrote it, against the same launch after the
cache was flushed with a 1 GB memset, for the full batch (201 MB z) and half batches (100 MB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multi_task_breast_cancer_amd import ops
dev = "cuda:0"
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)      # 1 GB
def run(N):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, 24, 256, 256, generator=g).to(dev)
    w = (torch.randn(24, 24, 3, 3, generator=g) * 0.1).to(dev)
    b = torch.zeros(24, device=dev)
    pf, _ = ops.conv3x3_pack_lp(w, 1)
    x8 = ops.C8.pack(x, 1)
    res = {}
    for mode in ("hot", "cold"):
        ts = []
        for _ in range(6):
            z = ops.conv3x3_fwd_c8([x8], w, b, pf)
            if mode == "cold":
                flush.zero_()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); ops.instnorm_lrelu_fwd_c8(z, None, None, slope=0.1, compute=1); e.record(); e.synchronize()
            ts.append(s.elapsed_time(e))
        res[mode] = sorted(ts)[2]
    print(f"N={N}: z = {N*24*65536*4/1e6:.0f} MB  IN forward right after the conv {res['hot']*1e3:.1f} us, after a cache flush {res['cold']*1e3:.1f} us", flush=True)
for N in (32, 16, 8):
    run(N)

"""Phase timestamps of the level-0 / level-1 igemm (conv3x3_igemm_c8_kernel) through the probes build: MTBC_LIB=.../libmtbc_hip_probes.so
MTBC_C8_TS=1 python tools/experiments/c8_ts_probe.py -- prints, per launch, when the blocks' first three tiles pass each phase (100 MHz clock)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multi_task_breast_cancer_amd import ops

dev = "cuda:0"
N = 32
for segs, Cout, S in [([24], 24, 256), ([24, 48], 24, 256), ([24, 24, 24, 24, 48], 24, 256), ([48], 48, 128), ([48, 48, 48], 48, 128), ([96], 96, 64)]:
    g = torch.Generator(device=dev).manual_seed(1)
    xs = [torch.randn(N, c, S, S, generator=g, device=dev) for c in segs]
    Cin = sum(segs)
    w = torch.randn(Cout, Cin, 3, 3, generator=g, device=dev) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g, device=dev)
    pf, pd = ops.conv3x3_pack_lp(w, 1)
    x8 = [ops.C8.pack(x, 1) for x in xs]
    del xs
    for rep in range(3):
        ops.conv3x3_fwd_c8(x8, w, b, pf, out_c8=True, stats=True, out_fp16=True)
    torch.cuda.synchronize()
    print(f"--- {Cin}->{Cout} @{S}", file=sys.stderr, flush=True)

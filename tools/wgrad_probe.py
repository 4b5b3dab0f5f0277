"""The 3x3 weight-gradient launches of one U-Net++ step (B=32, 256x256, 16-bit channel-blocked operands), one by one
through the C-ABI: time per launch (HIP events, kernel + split-K reduce), algorithmic bytes / time, and a correctness
check of every shape at N=2 against torch's fp32 conv2d_weight on the same rounded operands (GPU, fp32 accumulate:
tolerance 1e-4 of the largest element -- the strict 1e-5 fp64 check lives in tests/test_ops_gpu.py).
usage: python tools/wgrad_probe.py [compute=1] [N=32] [check=1] [bias_in_timed_launch=0]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_task_breast_cancer_amd import ops

compute = int(sys.argv[1]) if len(sys.argv) > 1 else 1
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 32
CHECK = int(sys.argv[3]) if len(sys.argv) > 3 else 1
BIAS = bool(int(sys.argv[4])) if len(sys.argv) > 4 else False      # the training step takes the conv-bias gradient from the InstanceNorm backward: timed without
dev = "cuda:0"
# (segments, Cout, S, launches per step): MTUNetPlusPlus, features (24, 48, 96, 192, 384), deep supervision + classifier branch
SHAPES = [([24], 24, 256, 5), ([24, 48], 24, 256, 1), ([24, 24, 48], 24, 256, 1), ([24, 24, 24, 48], 24, 256, 1), ([24] * 4 + [48], 24, 256, 1),
          ([24], 48, 128, 1), ([48], 48, 128, 4), ([48, 48], 48, 128, 1), ([48, 48, 48], 48, 128, 1), ([48] * 4, 48, 128, 1),
          ([48], 96, 64, 1), ([96], 96, 64, 3), ([96, 96], 96, 64, 1), ([96, 96, 96], 96, 64, 1),
          ([96], 192, 32, 1), ([192], 192, 32, 2), ([192, 192], 192, 32, 1),
          ([192], 384, 16, 3), ([384], 384, 16, 3), ([384, 384, 384], 512, 16, 1), ([512], 512, 16, 1)]


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps


def operands(N, segs, Cout, S, seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    xs = [torch.randn(N, c, S, S, generator=g, device=dev) for c in segs]
    dz = torch.randn(N, Cout, S, S, generator=g, device=dev)
    return xs, dz, [ops.C8.pack(x, compute) for x in xs], ops.C8.pack(dz, compute)


IKR = os.environ.get('PROBE_REDUCE', 'kernel') != 'launch'      # PROBE_REDUCE=launch: the split-K partials summed by a second launch (the pre-round-4 path)
ONLY = os.environ.get('PROBE_ONLY')          # e.g. PROBE_ONLY=144,24,256: one shape (Cin, Cout, size), for rocprofv3 --pmc runs
if ONLY:
    ci_, co_, s_ = (int(v) for v in ONLY.split(','))
    SHAPES = [sh for sh in SHAPES if (sum(sh[0]), sh[1], sh[2]) == (ci_, co_, s_)]
tot = tot_bytes = 0.0
worst = 0.0
for segs, Cout, S, cnt in SHAPES:
    Cin = sum(segs)
    shape = (Cout, Cin, 3, 3)
    err = float("nan")
    if CHECK:
        xs, dz, x8, dz8 = operands(2, segs, Cout, S, 7)
        dw, db = ops.conv3x3_wgrad_c8(x8, dz8, shape, want_bias=True)
        xr = torch.cat([t.unpack() for t in x8], 1)
        dzr = dz8.unpack()
        ref = torch.nn.grad.conv2d_weight(xr.double(), shape, dzr.double(), padding=1).float()
        err = ((dw - ref).abs().max() / ref.abs().max()).item()
        eb = ((db - dzr.sum((0, 2, 3))).abs().max() / dzr.sum((0, 2, 3)).abs().max()).item()
        dw2, _ = ops.conv3x3_wgrad_c8(x8, dz8, shape, want_bias=False)          # the no-bias kernel instance
        err = max(err, eb, ((dw2 - ref).abs().max() / ref.abs().max()).item())
        worst = max(worst, err)
        del xs, dz, x8, dz8, xr, dzr, ref
    xs, dz, x8, dz8 = operands(NB, segs, Cout, S, 1)
    del xs, dz
    dw = torch.empty(shape, device=dev)
    ms = timeit(lambda: ops.conv3x3_wgrad_c8(x8, dz8, shape, want_bias=BIAS, dw=dw, in_kernel_reduce=IKR))
    by = 2.0 * NB * S * S * (Cin + Cout)
    fl = 2.0 * NB * S * S * Cin * Cout * 9
    tot += ms * cnt
    tot_bytes += by * cnt
    print(f"{Cin:4d}->{Cout:3d} @{S:3d} segs{len(segs)} x{cnt}: {ms * 1e3:7.1f} us  {by / ms / 1e9:6.2f} TB/s alg  {fl / ms / 1e9:6.0f} TF  relerr {err:.1e}", flush=True)
    del x8, dz8
print(f"sum over the step's {sum(s[3] for s in SHAPES)} launches: {tot:.3f} ms, {tot_bytes / 1e9:.2f} GB algorithmic = {tot_bytes / tot / 1e9:.2f} TB/s; worst relerr {worst:.1e}")
assert not CHECK or worst < 1e-4, worst

"""C-ABI checks that need no GPU: the library loads, exports every symbol include/mtbc.h declares, and the
ctypes mirrors in _lib.py have exactly the C layout (sizes + a few offsets) of the header's structs."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mtbc.h")

from multi_task_breast_cancer_amd import _lib as L   # noqa: E402


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return L.load()


def test_library_exports_every_declared_symbol(lib):
    src = open(HEADER).read()
    declared = set(re.findall(r"\b(mtbc_[A-Za-z0-9_]+)\s*\(", src))
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.mtbc_version() == 202
    # header, library and binding agree on the layout version (the binding refuses any other library at load time)
    assert int(re.search(r"#define\s+MTBC_VERSION\s+(\d+)", src).group(1)) == lib.mtbc_version() == L.ABI_VERSION
    assert lib.mtbc_arch() == b"gfx950"
    assert lib.mtbc_strerror(0) == b"ok" and b"workspace" in lib.mtbc_strerror(-3)


def test_ctypes_layout_matches_header(tmp_path):
    structs = {"mtbc_seg": L.Seg, "mtbc_conv3x3_args": L.Conv3x3Args, "mtbc_instnorm_args": L.InstNormArgs,
               "mtbc_maxpool_args": L.MaxPoolArgs, "mtbc_convT_args": L.ConvTArgs, "mtbc_conv1x1_args": L.Conv1x1Args,
               "mtbc_gap_args": L.GapArgs, "mtbc_linear_args": L.LinearArgs, "mtbc_dice_args": L.DiceArgs,
               "mtbc_focal_args": L.FocalArgs, "mtbc_adam_args": L.AdamArgs, "mtbc_op": L.Op,
               "mtbc_pack_desc": L.PackDesc, "mtbc_head_fuse_args": L.HeadFuseArgs, "mtbc_wview_desc": L.WViewDesc}
    offs = [("mtbc_conv3x3_args", "workspace_bytes", L.Conv3x3Args.workspace_bytes.offset),
            ("mtbc_conv3x3_args", "w_packed", L.Conv3x3Args.w_packed.offset),
            ("mtbc_instnorm_args", "dgamma", L.InstNormArgs.dgamma.offset),
            ("mtbc_convT_args", "accumulate_dw", L.ConvTArgs.accumulate_dw.offset),
            ("mtbc_convT_args", "compute", L.ConvTArgs.compute.offset),
            ("mtbc_convT_args", "y_type", L.ConvTArgs.y_type.offset),
            ("mtbc_conv3x3_args", "compute", L.Conv3x3Args.compute.offset),
            ("mtbc_conv3x3_args", "operand_layout", L.Conv3x3Args.operand_layout.offset),
            ("mtbc_conv3x3_args", "out_accumulate", L.Conv3x3Args.out_accumulate.offset),
            ("mtbc_pack_desc", "kind", L.PackDesc.kind.offset),
            ("mtbc_dice_args", "gscale_dev", L.DiceArgs.gscale_dev.offset),
            ("mtbc_adam_args", "zero_grad", L.AdamArgs.zero_grad.offset), ("mtbc_adam_args", "dynamic", L.AdamArgs.dynamic.offset),
            ("mtbc_op", "u", L.Op.u.offset)]
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{HEADER}"', "int main(void){"]
    for name in structs:
        lines.append(f'printf("{name} %zu\\n", sizeof({name}));')
    for s, f, _ in offs:
        lines.append(f'printf("{s}.{f} %zu\\n", offsetof({s}, {f}));')
    lines.append("return 0;}")
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c11", "-o", str(exe), str(src)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)]).decode().splitlines())
    for name, typ in structs.items():
        assert int(got[name]) == C.sizeof(typ), (name, got[name], C.sizeof(typ))
    for s, f, off in offs:
        assert int(got[f"{s}.{f}"]) == off, (s, f)


def test_op_kind_enum_in_sync():
    src = open(HEADER).read()
    body = src[src.index("MTBC_OP_CONV3_FWD = 1"):]
    body = body[:body.index("};")]
    names = [n.strip().split("=")[0].strip() for n in body.replace("\n", " ").split(",") if n.strip()]
    want = ["CONV3_FWD", "CONV3_DGRAD", "CONV3_WGRAD", "CONV3_PACK_FWD", "CONV3_PACK_DGRAD", "IN_FWD", "IN_BWD",
            "POOL_FWD", "POOL_BWD", "CONVT_FWD", "CONVT_DGRAD", "CONVT_WGRAD", "CONV1_FWD", "CONV1_DGRAD", "CONV1_WGRAD",
            "GAP_FWD", "GAP_BWD", "LINEAR_FWD", "LINEAR_BWD", "DICE_FWD", "DICE_BWD", "FOCAL", "LOSS_MIX", "ADAM",
            "MEMSET", "DICE_COUNTS", "CONV3_PACK_LP", "HEAD_COMBINE", "HEAD_EXPAND", "C8_PACK", "C8_PACK16", "CONV3_WVIEW",
            "SET_STREAM", "EVENT_RECORD", "EVENT_WAIT", "IN_DPARAM"]
    assert names == ["MTBC_OP_" + w for w in want]
    for i, w in enumerate(want, start=1):
        assert getattr(L, "OP_" + w) == i


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from multi_task_breast_cancer_amd.nets import MTnnUNet
    from multi_task_breast_cancer_amd.criterions import DiceLoss, FocalLoss
    m = MTnnUNet(1, 1, 3)
    with pytest.raises(L.MtbcError):
        m(torch.rand(1, 1, 64, 64))
    with pytest.raises(L.MtbcError):
        DiceLoss()(torch.zeros(1, 1, 8, 8), torch.zeros(1, 1, 8, 8))
    with pytest.raises(L.MtbcError):
        FocalLoss()(torch.zeros(2, 3), torch.zeros(2, 3))


def test_library_is_loaded_after_torch():
    """_lib.load() must import torch before dlopen: libmtbc_hip.so has to bind to the HIP runtime torch ships, not bring the system one in
    beside it (two runtimes in one process: every launch fails) -- also when the package is imported first, as __graft_entry__.build() does."""
    import subprocess, sys
    code = ("import sys\n"
            "from multi_task_breast_cancer_amd import _lib\n"
            "assert 'torch' not in sys.modules, 'the package import itself stays light'\n"
            "_lib.load()\n"
            "assert 'torch' in sys.modules\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]


def test_adam_dynamic_scalars_are_torch_adams_scalar_path(lib):
    """mtbc_adam_dynamic is host arithmetic only (no GPU call): the three per-step scalars a replayed (hipGraph) Adam launch reads from device
    memory -- grad_scale, lr / (1 - b1^t), 1 / sqrt(1 - b2^t) -- with the bias corrections in double as torch.optim.Adam's scalar path computes them
    (torch/optim/adam.py: bias_correction1 = 1 - beta1 ** step; step_size = lr / bias_correction1; bias_correction2_sqrt = sqrt(1 - beta2 ** step)),
    the float32 betas / lr of the argument struct widened first.  experiment_init.py:186-187 (Adam, eps 1e-4), training_multitask.py:103."""
    import math
    import numpy as np
    for lr, b1, b2, t, gs in [(1e-4, 0.9, 0.999, 1, 1.0), (3e-4, 0.9, 0.999, 7, 1.0 / 4096.0), (5e-4, 0.8, 0.99, 12345, 0.125), (1e-6, 0.9, 0.999, 2_000_000, 1.0)]:
        a = L.AdamArgs()
        a.lr, a.beta1, a.beta2, a.eps, a.grad_scale, a.step = lr, b1, b2, 1e-4, gs, t
        out = (C.c_float * 3)()
        assert lib.mtbc_adam_dynamic(C.byref(a), C.byref(out)) == 0
        lr32, b132, b232 = (float(np.float32(v)) for v in (lr, b1, b2))
        want = (np.float32(gs), np.float32(lr32 / (1.0 - math.pow(b132, t))), np.float32(1.0 / math.sqrt(1.0 - math.pow(b232, t))))
        assert tuple(np.float32(v) for v in out) == want, (lr, b1, b2, t, list(out), want)
    bad = L.AdamArgs()
    bad.step = 0
    assert lib.mtbc_adam_dynamic(C.byref(bad), C.byref((C.c_float * 3)())) != 0        # t >= 1, as mtbc_adam_step

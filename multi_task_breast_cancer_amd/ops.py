"""Tensor-level wrappers over the per-op C-ABI (one call = one mtbc_* entry point).

Used by the parity tests and as the reference-side binding example of INTEGRATION.md; the training path itself
goes through step programs (engine.py).  Device fp32 tensors only; every wrapper raises on CPU input.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L


def _s() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if t.device.type != "cuda":
            L.require_gpu()
            raise L.MtbcError("device tensors required")
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise ValueError("fp32 contiguous tensors required")


def _p(t):
    return None if t is None else t.data_ptr()


def _fill_segs(arr, tensors: Sequence[torch.Tensor], accumulate: Sequence[int] = ()):
    for i, t in enumerate(tensors):
        arr[i].ptr = t.data_ptr()
        arr[i].batch_stride = t.shape[1] * t.shape[2] * t.shape[3]
        arr[i].channels = t.shape[1]
        arr[i].accumulate = accumulate[i] if i < len(accumulate) else 0


def _ws(nbytes: int, dev) -> torch.Tensor:
    return torch.empty(max(4, (nbytes + 3) // 4), dtype=torch.float32, device=dev)


# ------------------------------------------------------------------ conv3x3
def conv3x3_pack(w: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    _chk(w)
    lib = L.load()
    cout, cin = w.shape[0], w.shape[1]
    pf = torch.empty(lib.mtbc_conv3x3_packed_elems(cin, cout), dtype=torch.float32, device=w.device)
    pd = torch.empty(lib.mtbc_conv3x3_packed_dgrad_elems(cin, cout), dtype=torch.float32, device=w.device)
    L.check(lib.mtbc_conv3x3_pack_fwd(w.data_ptr(), pf.data_ptr(), cin, cout, _s()), "pack_fwd")
    L.check(lib.mtbc_conv3x3_pack_dgrad(w.data_ptr(), pd.data_ptr(), cin, cout, _s()), "pack_dgrad")
    return pf, pd


def conv3x3_pack_lp(w: torch.Tensor, compute: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """16-bit operand images (compute 1 = bf16, 2 = fp16) for fwd and dgrad, as int16 tensors."""
    _chk(w)
    lib = L.load()
    cout, cin = w.shape[0], w.shape[1]
    outs = []
    for dgrad in (0, 1):
        t = torch.empty(lib.mtbc_conv3x3_packed_lp_elems(cin, cout, dgrad), dtype=torch.int16, device=w.device)
        L.check(lib.mtbc_conv3x3_pack_lp(w.data_ptr(), t.data_ptr(), cin, cout, dgrad, compute, _s()), "pack_lp")
        outs.append(t)
    return outs[0], outs[1]


def conv3x3_pack_many(weights, compute: int = 0):
    """All weight images of a list of conv weights in one launch (mtbc_conv3x3_pack_many).  Returns, per weight, the
    (fwd, dgrad) images: fp32 tensors for compute 0, int16 tensors for compute 1 (bf16) / 2 (fp16)."""
    lib = L.load()
    descs = (L.PackDesc * (2 * len(weights)))()
    outs = []
    for i, w in enumerate(weights):
        _chk(w)
        cout, cin = w.shape[0], w.shape[1]
        pair = []
        for dg in (0, 1):
            if compute:
                t = torch.empty(lib.mtbc_conv3x3_packed_lp_elems(cin, cout, dg), dtype=torch.int16, device=w.device)
            else:
                n = lib.mtbc_conv3x3_packed_dgrad_elems(cin, cout) if dg else lib.mtbc_conv3x3_packed_elems(cin, cout)
                t = torch.empty(n, dtype=torch.float32, device=w.device)
            d = descs[2 * i + dg]
            d.w, d.packed, d.Cin, d.Cout = w.data_ptr(), t.data_ptr(), cin, cout
            d.kind, d.compute = (2 + dg if compute else dg), compute
            pair.append(t)
        outs.append(tuple(pair))
    L.check(lib.mtbc_conv3x3_pack_many(descs, len(descs), _s()), "pack_many")
    return outs


def _conv_args(xs, w, N, H, W):
    a = L.Conv3x3Args()
    cin = sum(x.shape[1] for x in xs)
    a.N, a.H, a.W, a.Cin, a.Cout, a.n_in = N, H, W, cin, w.shape[0], len(xs)
    a.w = w.data_ptr()
    return a


def conv3x3_fwd(xs: Sequence[torch.Tensor], w: torch.Tensor, bias: Optional[torch.Tensor] = None,
                packed: Optional[torch.Tensor] = None, force_direct: bool = False, compute: int = 0) -> torch.Tensor:
    _chk(*xs, w, bias)
    N, _, H, W = xs[0].shape
    a = _conv_args(xs, w, N, H, W)
    _fill_segs(a.in_, xs)
    out = torch.empty(N, w.shape[0], H, W, dtype=torch.float32, device=w.device)
    a.w_packed, a.bias, a.out, a.force_direct = _p(packed), _p(bias), out.data_ptr(), int(force_direct)
    a.compute = compute
    L.check(L.load().mtbc_conv3x3_fwd(C.byref(a), _s()), "conv3x3_fwd")
    return out


def conv3x3_stem_fwd_c8(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], compute: int, out_fp16: bool = False, stats: bool = False):
    """The 1-channel stem conv of the 16-bit modes: fp32 operands, the output channel-blocked 16-bit (out_layout = C8 with
    operand_layout = planar) + optional InstanceNorm statistics partials."""
    _chk(x, w, bias)
    N, _, H, W = x.shape
    a = _conv_args([x], w, N, H, W)
    _fill_segs(a.in_, [x])
    out = torch.empty(N, w.shape[0] // 8, H * W, 8, dtype=torch.int16, device=w.device)
    a.bias, a.out, a.compute, a.out_layout = _p(bias), out.data_ptr(), compute, L.LAYOUT_C8
    if out_fp16:
        a.out_type = 2
    part = None
    if stats:
        slots = L.load().mtbc_conv3x3_stats_slots(C.byref(a))
        if slots <= 0:
            raise L.MtbcError("conv3x3_fwd: no epilogue statistics for this launch")
        part = torch.full((N, slots, w.shape[0], 2), float("nan"), dtype=torch.float32, device=w.device)
        a.stats_partial = part.data_ptr()
    L.check(L.load().mtbc_conv3x3_fwd(C.byref(a), _s()), "conv3x3_fwd(stem c8)")
    z = C8(out, (N, w.shape[0], H, W), 2 if out_fp16 else compute)
    return (z, part) if stats else z


def conv3x3_dgrad(dz: torch.Tensor, w: torch.Tensor, dxs: Sequence[torch.Tensor], accumulate: Sequence[int] = (),
                  packed: Optional[torch.Tensor] = None, force_direct: bool = False, compute: int = 0) -> None:
    _chk(dz, w, *dxs)
    N, _, H, W = dz.shape
    a = _conv_args(dxs, w, N, H, W)
    _fill_segs(a.in_, dxs, accumulate)
    a.w_packed, a.dout, a.force_direct = _p(packed), dz.data_ptr(), int(force_direct)
    a.compute = compute
    L.check(L.load().mtbc_conv3x3_dgrad(C.byref(a), _s()), "conv3x3_dgrad")


def conv3x3_wgrad(xs: Sequence[torch.Tensor], dz: torch.Tensor, w_shape, want_bias: bool = False,
                  force_direct: bool = False, dw: Optional[torch.Tensor] = None, accumulate: bool = False,
                  compute: int = 0):
    _chk(*xs, dz)
    N, _, H, W = dz.shape
    dev = dz.device
    if dw is None:
        dw = torch.empty(*w_shape, dtype=torch.float32, device=dev)
    db = torch.empty(w_shape[0], dtype=torch.float32, device=dev) if want_bias else None
    a = L.Conv3x3Args()
    a.N, a.H, a.W, a.Cin, a.Cout, a.n_in = N, H, W, w_shape[1], w_shape[0], len(xs)
    _fill_segs(a.in_, xs)
    a.dout, a.dw, a.dbias, a.force_direct, a.accumulate_dw = dz.data_ptr(), dw.data_ptr(), _p(db), int(force_direct), int(accumulate)
    a.compute = compute
    nb = L.load().mtbc_conv3x3_wgrad_workspace(C.byref(a))
    ws = _ws(nb, dev)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    L.check(L.load().mtbc_conv3x3_wgrad(C.byref(a), _s()), "conv3x3_wgrad")
    return dw, db


# ------------------------------------------------------------------ conv3x3 on 16-bit channel-blocked operands
class C8:
    """A tensor in MTBC_LAYOUT_C8: [N][C/8][H*W][8] in bf16 (compute 1) or fp16 (compute 2), held as int16 storage.
    What the 3x3 convolutions' MFMAs read when `operand_layout = 1` (include/mtbc.h)."""

    def __init__(self, data: torch.Tensor, shape, compute: int):
        self.data, self.shape, self.compute = data, tuple(shape), compute

    @staticmethod
    def pack(x: torch.Tensor, compute: int) -> "C8":
        _chk(x)
        N, Cc, H, W = x.shape
        d = torch.empty(N, Cc // 8, H * W, 8, dtype=torch.int16, device=x.device)
        L.check(L.load().mtbc_c8_pack(x.data_ptr(), Cc * H * W, d.data_ptr(), N, Cc, H * W, compute, _s()), "c8_pack")
        return C8(d, x.shape, compute)

    @staticmethod
    def pack16(x16: torch.Tensor, compute: int) -> "C8":
        """From 16-bit planes (N,C,H,W as int16 storage), mtbc_c8_pack16."""
        N, Cc, H, W = x16.shape
        d = torch.empty(N, Cc // 8, H * W, 8, dtype=torch.int16, device=x16.device)
        L.check(L.load().mtbc_c8_pack16(x16.data_ptr(), Cc * H * W, d.data_ptr(), N, Cc, H * W, _s()), "c8_pack16")
        return C8(d, x16.shape, compute)

    def unpack(self) -> torch.Tensor:
        N, Cc, H, W = self.shape
        y = torch.empty(N, Cc, H, W, dtype=torch.float32, device=self.data.device)
        L.check(L.load().mtbc_c8_unpack(self.data.data_ptr(), y.data_ptr(), N, Cc, H * W, self.compute, _s()), "c8_unpack")
        return y


def _fill_segs_c8(arr, tensors: Sequence["C8"]):
    for i, t in enumerate(tensors):
        arr[i].ptr = t.data.data_ptr()
        arr[i].batch_stride = t.shape[1] * t.shape[2] * t.shape[3]      # 16-bit elements
        arr[i].channels = t.shape[1]
        arr[i].accumulate = 0


def conv3x3_fwd_c8(xs: Sequence["C8"], w: torch.Tensor, bias: Optional[torch.Tensor], packed: torch.Tensor, out_c8: bool = False,
                   stats: bool = False, norm=None, out_partial: Optional[torch.Tensor] = None, out_fp16: bool = False):
    """z = conv3x3(concat(xs)) with the inputs already in the MFMA's 16-bit channel-blocked layout; z comes back as fp32
    planes, or (out_c8, `out_layout = C8` of the C-ABI) as a channel-blocked 16-bit tensor of the inputs' type."""
    _chk(w, bias)
    N, _, H, W = xs[0].shape
    a = L.Conv3x3Args()
    a.N, a.H, a.W, a.Cin, a.Cout, a.n_in = N, H, W, sum(x.shape[1] for x in xs), w.shape[0], len(xs)
    _fill_segs_c8(a.in_, xs)
    if out_c8:
        out = torch.empty(N, w.shape[0] // 8, H * W, 8, dtype=torch.int16, device=w.device)
        a.out_layout = L.LAYOUT_C8
        if out_fp16:            # bf16 operands, the output stored as fp16 (out_type of the C-ABI)
            a.out_type = 2
    else:
        out = torch.empty(N, w.shape[0], H, W, dtype=torch.float32, device=w.device)
    a.w, a.w_packed, a.bias, a.out = w.data_ptr(), packed.data_ptr(), _p(bias), out.data_ptr()
    a.compute, a.operand_layout = xs[0].compute, L.LAYOUT_C8
    part = None
    if norm is not None:    # gathered dgrad + norm-backward reductions: norm = (z8, mean, rstd, gamma, beta, slope) of the output tensor
        z8n, mean, rstd, gamma, beta, slope = norm
        _chk(mean, rstd, gamma, beta, out_partial)
        a.norm_z, a.norm_mean, a.norm_rstd, a.norm_gamma, a.norm_beta, a.norm_slope = z8n.data.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(gamma), _p(beta), slope
        a.out_partial = _p(out_partial)
        stats = True
    if stats:       # InstanceNorm statistics from the epilogue: (partials [N][slots][Cout][2], slots)
        slots = L.load().mtbc_conv3x3_stats_slots(C.byref(a))
        if slots <= 0:
            raise L.MtbcError("conv3x3_fwd: no epilogue statistics for this launch")
        part = torch.full((N, slots, w.shape[0], 2), float("nan"), dtype=torch.float32, device=w.device)
        a.stats_partial = part.data_ptr()
    L.check(L.load().mtbc_conv3x3_fwd(C.byref(a), _s()), "conv3x3_fwd(c8)")
    z = C8(out, (N, w.shape[0], H, W), 2 if out_fp16 else xs[0].compute) if out_c8 else out
    return (z, part) if stats else z


def conv3x3_dgrad_c8(dz: "C8", w: torch.Tensor, dxs: Sequence[torch.Tensor], accumulate: Sequence[int], packed: torch.Tensor) -> None:
    """dx segments (fp32 planar, accumulate honoured; accumulate 2 = the segment is an int16 (N,C,H,W) tensor written as
    16-bit planar values of dz's type) from dz in the 16-bit channel-blocked layout."""
    if any(isinstance(d, C8) for d in dxs):       # accumulate = 3: channel-blocked 16-bit segments (all of them)
        N, _, H, W = dz.shape
        a = L.Conv3x3Args()
        a.N, a.H, a.W, a.Cin, a.Cout, a.n_in = N, H, W, sum(d.shape[1] for d in dxs), w.shape[0], len(dxs)
        a.w = w.data_ptr()
        _fill_segs_c8(a.in_, dxs)
        for i in range(len(dxs)):
            a.in_[i].accumulate = 3
        a.w_packed, a.dout = packed.data_ptr(), dz.data.data_ptr()
        a.compute, a.operand_layout = dz.compute, L.LAYOUT_C8
        L.check(L.load().mtbc_conv3x3_dgrad(C.byref(a), _s()), "conv3x3_dgrad(c8 -> c8)")
        return
    _chk(w, *[d for i, d in enumerate(dxs) if not (i < len(accumulate) and accumulate[i] == 2)])
    N, _, H, W = dz.shape
    a = _conv_args(dxs, w, N, H, W)
    _fill_segs(a.in_, dxs, accumulate)
    a.w_packed, a.dout = packed.data_ptr(), dz.data.data_ptr()
    a.compute, a.operand_layout = dz.compute, L.LAYOUT_C8
    L.check(L.load().mtbc_conv3x3_dgrad(C.byref(a), _s()), "conv3x3_dgrad(c8)")


_WGRAD_SYNC: dict = {}       # device -> zeroed int32 counters of the in-kernel split-K reduction (every launch leaves them at zero)


def wgrad_sync_buffer(dev, nbytes: int) -> torch.Tensor:
    buf = _WGRAD_SYNC.get(dev)
    if buf is None or buf.numel() * 4 < nbytes:
        buf = torch.zeros(max(4096, (nbytes + 3) // 4), dtype=torch.int32, device=dev)
        _WGRAD_SYNC[dev] = buf
    return buf


def conv3x3_wgrad_c8(xs: Sequence["C8"], dz: "C8", w_shape, want_bias: bool = False, dw: Optional[torch.Tensor] = None,
                     accumulate: bool = False, in_kernel_reduce: bool = True):
    """in_kernel_reduce: the split-K partials are summed inside the launch by the last-arriving block of each group
    (mtbc_conv3x3_args.wgrad_sync); False = the reduction launch behind the kernel (another, equally fixed, summation order)."""
    N, _, H, W = dz.shape
    dev = dz.data.device
    if dw is None:
        dw = torch.empty(*w_shape, dtype=torch.float32, device=dev)
    db = torch.empty(w_shape[0], dtype=torch.float32, device=dev) if want_bias else None
    a = L.Conv3x3Args()
    a.N, a.H, a.W, a.Cin, a.Cout, a.n_in = N, H, W, w_shape[1], w_shape[0], len(xs)
    if w_shape[1] == 1 and isinstance(xs[0], torch.Tensor):      # the stem: fp32 planar 1-channel input, channel-blocked dz
        _fill_segs(a.in_, xs)
    else:
        _fill_segs_c8(a.in_, xs)
    a.dout, a.dw, a.dbias, a.accumulate_dw = dz.data.data_ptr(), dw.data_ptr(), _p(db), int(accumulate)
    a.compute, a.operand_layout = dz.compute, L.LAYOUT_C8
    nb = L.load().mtbc_conv3x3_wgrad_workspace(C.byref(a))
    ws = _ws(nb, dev)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    nsync = int(L.load().mtbc_conv3x3_wgrad_sync_bytes(C.byref(a))) if in_kernel_reduce else 0
    if nsync:
        sync = wgrad_sync_buffer(dev, nsync)
        a.wgrad_sync, a.wgrad_sync_bytes = sync.data_ptr(), sync.numel() * 4
    L.check(L.load().mtbc_conv3x3_wgrad(C.byref(a), _s()), "conv3x3_wgrad(c8)")
    return dw, db


# ------------------------------------------------------------------ instance norm + leaky relu
def instnorm_lrelu_fwd(z, gamma=None, beta=None, eps=1e-5, slope=0.01, out16: int = 0):
    """out16 = 1 (bf16) / 2 (fp16): the activation comes back as 16-bit planes (int16 storage), y16 of the C-ABI."""
    _chk(z, gamma, beta)
    N, Cc, H, W = z.shape
    y = torch.empty_like(z, dtype=torch.int16 if out16 else torch.float32)
    mean = torch.empty(N * Cc, dtype=torch.float32, device=z.device)
    rstd = torch.empty_like(mean)
    a = L.InstNormArgs()
    a.N, a.C, a.H, a.W, a.eps, a.slope = N, Cc, H, W, eps, slope
    a.z, a.gamma, a.beta, a.y, a.y_batch_stride = z.data_ptr(), _p(gamma), _p(beta), y.data_ptr(), Cc * H * W
    a.mean, a.rstd = mean.data_ptr(), rstd.data_ptr()
    if out16:
        a.y, a.y16, a.out16_type = None, y.data_ptr(), out16
    nb = L.load().mtbc_instnorm_fwd_workspace(C.byref(a))
    if nb:
        ws = _ws(nb, z.device)
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    L.check(L.load().mtbc_instnorm_lrelu_fwd(C.byref(a), _s()), "instnorm_fwd")
    return y, mean, rstd


def instnorm_lrelu_bwd(z, dy, mean, rstd, gamma=None, beta=None, eps=1e-5, slope=0.01, inplace=False,
                       dbias_pre=None, dy_extra=(), out16: int = 0):
    _chk(z, dy, mean, rstd, gamma, beta, dbias_pre, *dy_extra)
    N, Cc, H, W = z.shape
    dz = torch.empty_like(z, dtype=torch.int16) if out16 else (dy if inplace else torch.empty_like(z))
    dg = torch.empty(Cc, dtype=torch.float32, device=z.device) if gamma is not None else None
    db = torch.empty(Cc, dtype=torch.float32, device=z.device) if gamma is not None else None
    ws = _ws(N * Cc * 12, z.device)
    a = L.InstNormArgs()
    a.N, a.C, a.H, a.W, a.eps, a.slope = N, Cc, H, W, eps, slope
    a.z, a.gamma, a.beta, a.mean, a.rstd = z.data_ptr(), _p(gamma), _p(beta), mean.data_ptr(), rstd.data_ptr()
    a.dy, a.dy_batch_stride, a.dz, a.dgamma, a.dbeta = dy.data_ptr(), Cc * H * W, dz.data_ptr(), _p(dg), _p(db)
    a.dbias_pre = _p(dbias_pre)
    if out16:
        a.dz, a.dz16, a.out16_type = None, dz.data_ptr(), out16
    a.n_dy_extra = len(dy_extra)
    for k_, t_ in enumerate(dy_extra):
        a.dy_extra[k_] = t_.data_ptr()
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    L.check(L.load().mtbc_instnorm_lrelu_bwd(C.byref(a), _s()), "instnorm_bwd")
    return dz, dg, db


_coop_states = {}


def coop_state(dev) -> torch.Tensor:
    """The persistent, zero-initialised mailbox block of the cooperative InstanceNorm kernels (one per device here;
    the C-ABI wants one per stream that launches them)."""
    key = str(dev)
    if key not in _coop_states:
        _coop_states[key] = torch.zeros(L.load().mtbc_instnorm_coop_state_bytes() // 4, dtype=torch.int32, device=dev)
    return _coop_states[key]


def instnorm_lrelu_fwd_c8(z, gamma=None, beta=None, eps=1e-5, slope=0.01, compute: Optional[int] = None, want_planar: bool = False,
                          stats: Optional[torch.Tensor] = None, want_pool: bool = False, planar16: bool = False):
    """InstanceNorm + LeakyReLU straight into the 16-bit channel-blocked layout (y8 of the C-ABI).  z: fp32 planes, or a
    C8 tensor (z_layout = C8: the conv output of the 16-bit modes).  planar16 (with stats): the planar copy is an int16
    (N,C,H,W) tensor of the output type (y16 beside y8)."""
    z8 = z if isinstance(z, C8) else None
    if compute is None:         # output type: given, else the type of a channel-blocked z, else bf16
        compute = z8.compute if z8 is not None else 1
    zt = z8.data if z8 is not None else z
    _chk(zt if z8 is None else None, gamma, beta)
    N, Cc, H, W = z.shape
    dev = zt.device
    y8 = torch.empty(N, Cc // 8, H * W, 8, dtype=torch.int16, device=dev)
    y = torch.empty(N, Cc, H, W, dtype=torch.int16 if planar16 else torch.float32, device=dev) if (want_planar or planar16) else None
    mean = torch.empty(N * Cc, dtype=torch.float32, device=dev)
    rstd = torch.empty_like(mean)
    a = L.InstNormArgs()
    a.N, a.C, a.H, a.W, a.eps, a.slope = N, Cc, H, W, eps, slope
    a.z, a.gamma, a.beta, a.y, a.y_batch_stride = zt.data_ptr(), _p(gamma), _p(beta), (None if planar16 else _p(y)), Cc * H * W
    if planar16:
        a.y16 = y.data_ptr()
    a.z_layout = L.LAYOUT_C8 if z8 is not None else L.LAYOUT_PLANAR
    if z8 is not None and z8.compute != compute:
        a.z_type = z8.compute           # fp16 z with bf16 outputs
    a.mean, a.rstd = mean.data_ptr(), rstd.data_ptr()
    a.y8, a.out16_type, a.coop_state = y8.data_ptr(), compute, coop_state(dev).data_ptr()
    if stats is not None:       # [N][slots][C][2] from conv3x3_fwd_c8(stats=True)
        a.stats_partial, a.stats_slots = stats.data_ptr(), stats.shape[1]
    yp8 = parg = None
    if want_pool:               # the streaming pass also writes the activation's 2x2 max-pool + argmax codes (pool_y8 / pool_arg)
        yp8 = torch.empty(N, Cc // 8, (H // 2) * (W // 2), 8, dtype=torch.int16, device=dev)
        parg = torch.empty(N, Cc // 8, (H // 2) * (W // 2), dtype=torch.int16, device=dev)
        a.pool_y8, a.pool_arg = yp8.data_ptr(), parg.data_ptr()
    if not L.load().mtbc_instnorm_c8_supported(C.byref(a), 0):
        raise L.MtbcError("instnorm_fwd: shape not supported with a channel-blocked output")
    L.check(L.load().mtbc_instnorm_lrelu_fwd(C.byref(a), _s()), "instnorm_fwd(c8)")
    if want_pool:
        return C8(y8, z.shape, compute), mean, rstd, y, C8(yp8, (N, Cc, H // 2, W // 2), compute), parg
    return C8(y8, z.shape, compute), mean, rstd, y


def instnorm_lrelu_bwd_c8(z, dy, mean, rstd, gamma=None, beta=None, eps=1e-5, slope=0.01, dbias_pre=None, compute: Optional[int] = None,
                          dy_extra: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None, rank1=None, rank1_grads: bool = False,
                          pool=None, defer_dparams: bool = False):
    """z / dy: fp32 planes or C8 tensors (z_layout / dy_layout = C8); dy_extra (with a C8 dy only): an fp32 planar partial
    gradient added while loading; rank1 = (dyhead (N,1,H,W), w (C)): the rank-1 gradient term of a one-output 1x1 head (dy may
    then be None)."""
    zt = z.data if isinstance(z, C8) else z
    dyt = dy.data if isinstance(dy, C8) else dy
    if compute is None:         # output (and channel-blocked dy) type: given, else dy's, else z's, else bf16
        compute = dy.compute if isinstance(dy, C8) else (z.compute if isinstance(z, C8) else 1)
    _chk(mean, rstd, gamma, beta, dbias_pre, dy_extra)
    N, Cc, H, W = z.shape
    dev = zt.device
    dz8 = torch.empty(N, Cc // 8, H * W, 8, dtype=torch.int16, device=dev)
    dg = torch.empty(Cc, dtype=torch.float32, device=dev) if gamma is not None else None
    db = torch.empty(Cc, dtype=torch.float32, device=dev) if gamma is not None else None
    ws = _ws(N * (Cc + 1) * 262 * 4, dev)
    a = L.InstNormArgs()
    a.N, a.C, a.H, a.W, a.eps, a.slope = N, Cc, H, W, eps, slope
    a.z, a.gamma, a.beta, a.mean, a.rstd = zt.data_ptr(), _p(gamma), _p(beta), mean.data_ptr(), rstd.data_ptr()
    a.dy, a.dy_batch_stride, a.dgamma, a.dbeta, a.dbias_pre = (dyt.data_ptr() if dyt is not None else None), Cc * H * W, _p(dg), _p(db), _p(dbias_pre)
    if rank1 is not None:
        _chk(*rank1)
        a.dy_rank1, a.dy_rank1_w = rank1[0].data_ptr(), rank1[1].data_ptr()
    if pool is not None:        # (gradient of the 2x2 max-pool of this activation (N,C,H/2,W/2) fp32, argmax codes from maxpool2_fwd_c8)
        _chk(pool[0])
        a.dy_pool, a.dy_pool_arg = pool[0].data_ptr(), pool[1].data_ptr()
    hdw = hdb = None
    if rank1_grads:             # the head's own weight / bias gradient from the same pass
        hdw, hdb = torch.empty(Cc, dtype=torch.float32, device=dev), torch.empty(1, dtype=torch.float32, device=dev)
        a.dy_rank1_dw, a.dy_rank1_db = hdw.data_ptr(), hdb.data_ptr()
    a.z_layout = L.LAYOUT_C8 if isinstance(z, C8) else L.LAYOUT_PLANAR
    if isinstance(z, C8) and z.compute != compute:
        a.z_type = z.compute
    a.dy_layout = L.LAYOUT_C8 if isinstance(dy, C8) else L.LAYOUT_PLANAR
    if dy_extra is not None:
        a.n_dy_extra = 1
        a.dy_extra[0] = dy_extra.data_ptr()
    a.dz8, a.out16_type, a.coop_state = dz8.data_ptr(), compute, coop_state(dev).data_ptr()
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    if stats is not None:       # {sum g, sum g * xhat} partials from conv3x3_fwd_c8(norm=...)
        a.stats_partial, a.stats_slots = stats.data_ptr(), stats.shape[1]
    if not L.load().mtbc_instnorm_c8_supported(C.byref(a), 1):
        raise L.MtbcError("instnorm_bwd: shape not supported with a channel-blocked output")
    if defer_dparams:           # leave the partials in the workspace, reduce them with the batched entry point afterwards
        a.defer_dparams = 1
    L.check(L.load().mtbc_instnorm_lrelu_bwd(C.byref(a), _s()), "instnorm_bwd(c8)")
    if defer_dparams:
        d = (L.DparamDesc * 1)()
        d[0].part, d[0].dgamma, d[0].dbeta, d[0].dbias_pre = ws.data_ptr(), _p(dg), _p(db), _p(dbias_pre)
        d[0].N, d[0].C, d[0].T, d[0].accumulate = N, Cc, L.load().mtbc_instnorm_bwd_team(C.byref(a)), 0
        L.check(L.load().mtbc_instnorm_dparam_many(d, 1, _s()), "instnorm_dparam_many")
    if rank1_grads:
        return C8(dz8, z.shape, compute), dg, db, hdw, hdb
    return C8(dz8, z.shape, compute), dg, db


def coop_error(dev) -> int:
    """1 if a cooperative kernel ever gave up polling its mailbox (protocol failure; results are then garbage)."""
    return int(coop_state(dev)[2].item())


# ------------------------------------------------------------------ maxpool
def maxpool2_fwd(x):
    _chk(x)
    N, Cc, H, W = x.shape
    y = torch.empty(N, Cc, H // 2, W // 2, dtype=torch.float32, device=x.device)
    a = L.MaxPoolArgs()
    a.N, a.C, a.H, a.W = N, Cc, H, W
    a.x, a.x_batch_stride, a.y, a.y_batch_stride = x.data_ptr(), Cc * H * W, y.data_ptr(), Cc * H * W // 4
    L.check(L.load().mtbc_maxpool2_fwd(C.byref(a), _s()), "maxpool_fwd")
    return y


def maxpool2_bwd(x, dy, dx=None, accumulate=False):
    _chk(x, dy, dx)
    N, Cc, H, W = x.shape
    if dx is None:
        dx = torch.empty_like(x)
    a = L.MaxPoolArgs()
    a.N, a.C, a.H, a.W = N, Cc, H, W
    a.x, a.x_batch_stride = x.data_ptr(), Cc * H * W
    a.dy, a.dy_batch_stride, a.dx, a.dx_batch_stride = dy.data_ptr(), Cc * H * W // 4, dx.data_ptr(), Cc * H * W
    a.accumulate_dx = int(accumulate)
    L.check(L.load().mtbc_maxpool2_bwd(C.byref(a), _s()), "maxpool_bwd")
    return dx


def maxpool2_fwd_c8(x8: "C8", want_argmax: bool = False):
    """2x2 max-pool of a channel-blocked 16-bit tensor into one (mtbc_maxpool_args.layout = MTBC_LAYOUT_C8); want_argmax: also the
    (N, C/8, H/2*W/2) int16 codes of where each window's maximum sits (mtbc_maxpool_args.argmax)."""
    N, Cc, H, W = x8.shape
    y = torch.empty(N, Cc // 8, (H // 2) * (W // 2), 8, dtype=torch.int16, device=x8.data.device)
    a = L.MaxPoolArgs()
    a.N, a.C, a.H, a.W, a.layout, a.type16 = N, Cc, H, W, L.LAYOUT_C8, x8.compute
    a.x, a.x_batch_stride, a.y, a.y_batch_stride = x8.data.data_ptr(), Cc * H * W, y.data_ptr(), Cc * H * W // 4
    arg = torch.empty(N, Cc // 8, (H // 2) * (W // 2), dtype=torch.int16, device=x8.data.device) if want_argmax else None
    if arg is not None:
        a.argmax = arg.data_ptr()
    L.check(L.load().mtbc_maxpool2_fwd(C.byref(a), _s()), "maxpool_fwd c8")
    out = C8(y, (N, Cc, H // 2, W // 2), x8.compute)
    return (out, arg) if want_argmax else out


def maxpool2_bwd_c8(x8: "C8", dy, dx=None, accumulate=False):
    _chk(dy, dx)
    N, Cc, H, W = x8.shape
    if dx is None:
        dx = torch.empty(N, Cc, H, W, dtype=torch.float32, device=dy.device)
    a = L.MaxPoolArgs()
    a.N, a.C, a.H, a.W, a.layout, a.type16 = N, Cc, H, W, L.LAYOUT_C8, x8.compute
    a.x, a.x_batch_stride = x8.data.data_ptr(), Cc * H * W
    a.dy, a.dy_batch_stride, a.dx, a.dx_batch_stride = dy.data_ptr(), Cc * H * W // 4, dx.data_ptr(), Cc * H * W
    a.accumulate_dx = int(accumulate)
    L.check(L.load().mtbc_maxpool2_bwd(C.byref(a), _s()), "maxpool_bwd c8")
    return dx


# ------------------------------------------------------------------ conv transpose (k == stride)
def _ct_args(x, w, k):
    a = L.ConvTArgs()
    N, Cin, H, W = x.shape
    a.N, a.H, a.W, a.Cin, a.Cout, a.k = N, H, W, Cin, w.shape[1], k
    a.x, a.x_batch_stride, a.w = x.data_ptr(), Cin * H * W, w.data_ptr()
    return a


def convT_fwd(x, w, bias, k):
    _chk(x, w, bias)
    N, Cin, H, W = x.shape
    y = torch.empty(N, w.shape[1], H * k, W * k, dtype=torch.float32, device=x.device)
    a = _ct_args(x, w, k)
    a.bias, a.y, a.y_batch_stride = _p(bias), y.data_ptr(), y[0].numel()
    L.check(L.load().mtbc_convT_fwd(C.byref(a), _s()), "convT_fwd")
    return y


def convT_fwd_c8(x, w, bias, k, compute: int) -> "C8":
    """ConvTranspose2d(k = s) forward written straight into the 16-bit channel-blocked layout (y_layout = C8)."""
    _chk(x, w, bias)
    a = _ct_args(x, w, k)
    N, _, H, W = x.shape
    cout = w.shape[1]
    y = torch.empty(N, cout // 8, H * k * W * k, 8, dtype=torch.int16, device=x.device)
    a.bias, a.y, a.y_batch_stride = _p(bias), y.data_ptr(), cout * H * k * W * k
    a.y_layout, a.y_type = L.LAYOUT_C8, compute
    if not L.load().mtbc_convT_fwd_c8_supported(C.byref(a)):
        raise L.MtbcError("convT_fwd: shape not supported with a channel-blocked output")
    L.check(L.load().mtbc_convT_fwd(C.byref(a), _s()), "convT_fwd(c8)")
    return C8(y, (N, cout, H * k, W * k), compute)


def convT_fwd_c8_lp(x8: "C8", w, bias, k) -> "C8":
    """ConvTranspose2d(k = s = 2) forward on the 16-bit MFMA, input and output channel-blocked (x_layout = y_layout = C8)."""
    _chk(w, bias)
    N, Cin, H, W = x8.shape
    cout = w.shape[1]
    a = L.ConvTArgs()
    a.N, a.H, a.W, a.Cin, a.Cout, a.k = N, H, W, Cin, cout, k
    a.x, a.x_batch_stride, a.w = x8.data.data_ptr(), Cin * H * W, w.data_ptr()
    y = torch.empty(N, cout // 8, H * k * W * k, 8, dtype=torch.int16, device=w.device)
    a.bias, a.y, a.y_batch_stride = _p(bias), y.data_ptr(), cout * H * k * W * k
    a.y_layout, a.y_type, a.x_layout = L.LAYOUT_C8, x8.compute, L.LAYOUT_C8
    if not L.load().mtbc_convT_fwd_c8_supported(C.byref(a)):
        raise L.MtbcError("convT_fwd: shape not supported with channel-blocked input and output")
    L.check(L.load().mtbc_convT_fwd(C.byref(a), _s()), "convT_fwd(c8 -> c8)")
    return C8(y, (N, cout, H * k, W * k), x8.compute)


def convT_dgrad(x, w, dy, k, dx=None, accumulate=False, compute=0, dy16=False):
    """dy16: dy is an int16 (N,Cout,kH,kW) tensor holding 16-bit planar values of the type of `compute`."""
    _chk(x, w, None if dy16 else dy, dx)
    if dx is None:
        dx = torch.empty_like(x)
    a = _ct_args(x, w, k)
    a.compute = compute
    a.dy_type16 = compute if dy16 else 0
    a.dy, a.dy_batch_stride, a.dx, a.dx_batch_stride, a.accumulate_dx = dy.data_ptr(), dy[0].numel(), dx.data_ptr(), x[0].numel(), int(accumulate)
    L.check(L.load().mtbc_convT_dgrad(C.byref(a), _s()), "convT_dgrad")
    return dx


def convT_wgrad(x, w, dy, k, want_bias=True, compute=0, dy16=False, x16=False):
    """x16 (with dy16): x is an int16 (N,Cin,H,W) tensor of 16-bit planar values of the type of `compute` too."""
    _chk(None if x16 else x, w, None if dy16 else dy)
    dw = torch.empty_like(w)
    db = torch.empty(w.shape[1], dtype=torch.float32, device=x.device) if want_bias else None
    a = _ct_args(x, w, k)
    a.compute = compute
    a.dy_type16 = compute if dy16 else 0
    a.x_type16 = compute if x16 else 0
    a.dy, a.dy_batch_stride, a.dw, a.dbias = dy.data_ptr(), dy[0].numel(), dw.data_ptr(), _p(db)
    ws = _ws(L.load().mtbc_convT_wgrad_workspace(C.byref(a)), x.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    L.check(L.load().mtbc_convT_wgrad(C.byref(a), _s()), "convT_wgrad")
    return dw, db


# ------------------------------------------------------------------ conv1x1
def _c1_args(x, w):
    a = L.Conv1x1Args()
    N, Cin, H, W = x.shape
    a.N, a.H, a.W, a.Cin, a.Cout = N, H, W, Cin, w.shape[0]
    a.x, a.x_batch_stride, a.w = x.data_ptr(), Cin * H * W, w.data_ptr()
    return a


def conv1x1_fwd(x, w, bias):
    _chk(x, w, bias)
    N, _, H, W = x.shape
    y = torch.empty(N, w.shape[0], H, W, dtype=torch.float32, device=x.device)
    a = _c1_args(x, w)
    a.bias, a.y = _p(bias), y.data_ptr()
    L.check(L.load().mtbc_conv1x1_fwd(C.byref(a), _s()), "conv1x1_fwd")
    return y


def conv1x1_bwd(x, w, dy):
    _chk(x, w, dy)
    dx, dw = torch.empty_like(x), torch.empty_like(w)
    db = torch.empty(w.shape[0], dtype=torch.float32, device=x.device)
    a = _c1_args(x, w)
    a.dy, a.dx, a.dx_batch_stride = dy.data_ptr(), dx.data_ptr(), x[0].numel()
    L.check(L.load().mtbc_conv1x1_dgrad(C.byref(a), _s()), "conv1x1_dgrad")
    a.dw, a.dbias = dw.data_ptr(), db.data_ptr()
    ws = _ws(L.load().mtbc_conv1x1_wgrad_workspace(C.byref(a)), x.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    L.check(L.load().mtbc_conv1x1_wgrad(C.byref(a), _s()), "conv1x1_wgrad")
    return dx, dw, db


def conv1x1_fwd_c8(x8: "C8", w, bias):
    """1x1 conv reading the channel-blocked 16-bit activation (mtbc_conv1x1_args.x_layout = MTBC_LAYOUT_C8)."""
    _chk(w, bias)
    N, Cin, H, W = x8.shape
    y = torch.empty(N, w.shape[0], H, W, dtype=torch.float32, device=w.device)
    a = L.Conv1x1Args()
    a.N, a.H, a.W, a.Cin, a.Cout = N, H, W, Cin, w.shape[0]
    a.x, a.x_batch_stride, a.w, a.x_layout, a.x_type = x8.data.data_ptr(), Cin * H * W, w.data_ptr(), L.LAYOUT_C8, x8.compute
    a.bias, a.y = _p(bias), y.data_ptr()
    L.check(L.load().mtbc_conv1x1_fwd(C.byref(a), _s()), "conv1x1_fwd c8")
    return y


def conv1x1_wgrad_c8(x8: "C8", w, dy, accumulate=False, dw=None, db=None):
    _chk(w, dy)
    N, Cin, H, W = x8.shape
    dw = torch.empty_like(w) if dw is None else dw
    db = torch.empty(w.shape[0], dtype=torch.float32, device=w.device) if db is None else db
    a = L.Conv1x1Args()
    a.N, a.H, a.W, a.Cin, a.Cout = N, H, W, Cin, w.shape[0]
    a.x, a.x_batch_stride, a.w, a.x_layout, a.x_type = x8.data.data_ptr(), Cin * H * W, w.data_ptr(), L.LAYOUT_C8, x8.compute
    a.dy, a.dw, a.dbias, a.accumulate_dw = dy.data_ptr(), dw.data_ptr(), db.data_ptr(), int(accumulate)
    ws = _ws(L.load().mtbc_conv1x1_wgrad_workspace(C.byref(a)), w.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    L.check(L.load().mtbc_conv1x1_wgrad(C.byref(a), _s()), "conv1x1_wgrad c8")
    return dw, db


# ------------------------------------------------------------------ head
def gap_fwd(x):
    _chk(x)
    N, Cc, H, W = x.shape
    y = torch.empty(N, Cc, dtype=torch.float32, device=x.device)
    a = L.GapArgs()
    a.N, a.C, a.H, a.W, a.x, a.y = N, Cc, H, W, x.data_ptr(), y.data_ptr()
    L.check(L.load().mtbc_gap_fwd(C.byref(a), _s()), "gap_fwd")
    return y


def gap_bwd(dy, H, W):
    _chk(dy)
    N, Cc = dy.shape
    dx = torch.empty(N, Cc, H, W, dtype=torch.float32, device=dy.device)
    a = L.GapArgs()
    a.N, a.C, a.H, a.W, a.dy, a.dx = N, Cc, H, W, dy.data_ptr(), dx.data_ptr()
    L.check(L.load().mtbc_gap_bwd(C.byref(a), _s()), "gap_bwd")
    return dx


def linear_fwd(x, w, b, relu=False):
    _chk(x, w, b)
    y = torch.empty(x.shape[0], w.shape[0], dtype=torch.float32, device=x.device)
    a = L.LinearArgs()
    a.N, a.In, a.Out, a.relu = x.shape[0], x.shape[1], w.shape[0], int(relu)
    a.x, a.w, a.bias, a.y = x.data_ptr(), w.data_ptr(), _p(b), y.data_ptr()
    L.check(L.load().mtbc_linear_fwd(C.byref(a), _s()), "linear_fwd")
    return y


def linear_bwd(x, w, y, dy, relu=False):
    _chk(x, w, y, dy)
    dx, dw = torch.empty_like(x), torch.empty_like(w)
    db = torch.empty(w.shape[0], dtype=torch.float32, device=x.device)
    ws = _ws(x.shape[0] * w.shape[0] * 4, x.device)
    a = L.LinearArgs()
    a.N, a.In, a.Out, a.relu = x.shape[0], x.shape[1], w.shape[0], int(relu)
    a.x, a.w, a.y, a.dy = x.data_ptr(), w.data_ptr(), y.data_ptr(), dy.data_ptr()
    a.dx, a.dw, a.dbias = dx.data_ptr(), dw.data_ptr(), db.data_ptr()
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    L.check(L.load().mtbc_linear_bwd(C.byref(a), _s()), "linear_bwd")
    return dx, dw, db


# ------------------------------------------------------------------ losses / optimiser
def dice_multihead(xs: Sequence[torch.Tensor], target, weights: Sequence[float], gscale: float = 1.0):
    """returns (loss[n_heads+1], [dx per head])"""
    _chk(*xs, target)
    nh = len(xs)
    N, Cc, H, W = xs[0].shape
    dev = target.device
    stats = torch.empty(nh * N * Cc * 3, dtype=torch.float32, device=dev)
    loss = torch.empty(nh + 1, dtype=torch.float32, device=dev)
    dxs = [torch.empty_like(x) for x in xs]
    a = L.DiceArgs()
    a.n_heads, a.N, a.C, a.H, a.W, a.smooth_nr, a.smooth_dr = nh, N, Cc, H, W, 1.0, 1.0
    for i in range(nh):
        a.x[i], a.dx[i], a.head_weight[i] = xs[i].data_ptr(), dxs[i].data_ptr(), weights[i]
    a.target, a.stats, a.loss, a.gscale = target.data_ptr(), stats.data_ptr(), loss.data_ptr(), gscale
    L.check(L.load().mtbc_dice_fwd(C.byref(a), _s()), "dice_fwd")
    L.check(L.load().mtbc_dice_bwd(C.byref(a), _s()), "dice_bwd")
    return loss, dxs


def focal(x, t, alpha=1.0, gamma=2.0, weight=None, gscale=1.0):
    _chk(x, t, weight)
    loss = torch.empty(1, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    a = L.FocalArgs()
    a.N, a.C, a.alpha, a.gamma = x.shape[0], x.shape[1], alpha, gamma
    a.x, a.target, a.weight, a.loss, a.dx, a.gscale = x.data_ptr(), t.data_ptr(), _p(weight), loss.data_ptr(), dx.data_ptr(), gscale
    L.check(L.load().mtbc_focal_fwd_bwd(C.byref(a), _s()), "focal")
    return loss, dx


def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-4, grad_scale=1.0, zero_grad=False):
    _chk(p, g, m, v)
    a = L.AdamArgs()
    a.n, a.p, a.g, a.m, a.v = p.numel(), p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
    a.lr, a.beta1, a.beta2, a.eps, a.grad_scale, a.step, a.zero_grad = lr, beta1, beta2, eps, grad_scale, step, int(zero_grad)
    L.check(L.load().mtbc_adam_step(C.byref(a), _s()), "adam")


# ------------------------------------------------------------------ fused ConvT + 1x1 head (MTnnUNet deep supervision)
def convT_head_fwd_bwd(x, wT, bT, w1, b1, k, dout):
    """Forward and backward of Conv2d_1x1(ConvTranspose2d_k(x)) through the combined-weight path
    (mtbc_convT_head_combine -> mtbc_convT_* with Cout = R -> mtbc_convT_head_expand).
    Returns (y, dx, dwT, dbT, dw1, db1)."""
    _chk(x, wT, bT, w1, b1, dout)
    lib = L.load()
    cin, cmid, R = wT.shape[0], wT.shape[1], w1.shape[0]
    dev = x.device
    Wc, bc = torch.empty(cin, R, k, k, device=dev), torch.empty(R, device=dev)
    a = L.HeadFuseArgs()
    a.Cin, a.Cmid, a.R, a.k = cin, cmid, R, k
    a.wT, a.bT, a.w1, a.b1 = wT.data_ptr(), bT.data_ptr(), w1.data_ptr(), b1.data_ptr()
    a.Wc, a.bc = Wc.data_ptr(), bc.data_ptr()
    L.check(lib.mtbc_convT_head_combine(C.byref(a), _s()), "head_combine")
    y = convT_fwd(x, Wc, bc, k)
    dx = convT_dgrad(x, Wc, dout, k)
    G, gb = convT_wgrad(x, Wc, dout, k)
    dwT, dbT, dw1, db1 = torch.empty_like(wT), torch.empty_like(bT), torch.empty_like(w1), torch.empty_like(b1)
    a.G, a.gb = G.data_ptr(), gb.data_ptr()
    a.dwT, a.dbT, a.dw1, a.db1 = dwT.data_ptr(), dbT.data_ptr(), dw1.data_ptr(), db1.data_ptr()
    L.check(lib.mtbc_convT_head_expand(C.byref(a), _s()), "head_expand")
    return y, dx, dwT, dbT, dw1, db1

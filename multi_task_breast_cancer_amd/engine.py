"""Static step programs: the forward / loss / backward / optimiser sequence of one training step as a
list of C-ABI op descriptors with every device pointer resolved at plan time.

The reference builds the same sequence dynamically each step through torch autograd
(src/training_multitask.py:87-103); here the graph is static, so the backward program is emitted once by
walking the forward tape in reverse, and `mtbc_program_run` issues it with no host work between kernels.
torch is used for device memory (arena tensors) and streams only.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib as L


from . import switches as _sw

_sw.refuse_removed()        # a removed switch in the environment is an error here, for every consumer of the engine (bench, trainers, tools)
# Plan switches (switches.py): each one selects between two step programs that a -m gpu test runs against the oracle / emulation
_NO_COOP = _sw.flag("MTBC_NO_COOP")
_NO_GATHER = _sw.flag("MTBC_NO_GATHER")
_NO_Z16 = _sw.flag("MTBC_NO_Z16")
_Z_BF16 = _sw.flag("MTBC_Z_BF16")
_DA16 = _sw.flag("MTBC_DA16")            # 16-bit gathered activation gradients: opt-in again since round 4 (switches.py)
# measured crossovers of the cooperative (split-plane) InstanceNorm kernels on fp32 conv outputs (the MTBC_NO_Z16 arm): they win on
# planes >= 128x128 forward / 256x256 backward; on small planes their barriers and 512-thread workgroups lose to one-plane kernels + pack
_COOP_MIN_FWD, _COOP_MIN_BWD = 16384, 65536
_IN_KERNEL_SPLITK = False   # see _conv_cell_backward: the in-kernel split-K reduction is slower than the reduction launches on this chip
_DPARAM_BATCH = 12          # cells per batched InstanceNorm parameter-gradient reduction (36 launches of 5 us -> 3)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


@dataclass
class Act:
    """An NCHW activation living in its own contiguous arena tensor (virtual concat = list of Acts)."""
    name: str
    data: torch.Tensor                 # (N, C, H, W)
    needs_grad: bool = True
    grad: Optional[torch.Tensor] = None
    grad_written: bool = False         # has any backward op produced (part of) this grad yet?
    c8: Optional[torch.Tensor] = None  # 16-bit channel-blocked copy [N][C/8][H*W][8] read by the 3x3 convs' MFMAs
    planar_valid: bool = True          # False: the producer wrote only `c8` (ConvT forward in the 16-bit modes)
    planar_used: bool = False          # some op reads `data` (pool, ConvT, 1x1 heads, GAP, a planar-staged conv)
    p16_users: list = field(default_factory=list)      # ConvT weight-gradient ops that take 16-bit planes if nobody needs the fp32 ones
    c8_used: bool = False              # some 3x3 conv reads `c8`
    conv_consumers: int = 0            # 3x3 conv cells (16-bit, channel-blocked backward) that read this activation
    no_gather: bool = False            # ... and one that cannot take part in a gathered dgrad
    pending: List[tuple] = field(default_factory=list)   # gathered dgrad: (dz8, weight, channel offset, consumer Cin, consumer Cout)
    readers: int = 0                   # ops that read this activation (any kind): 1 = its gradient has a single writer
    grad16_ok: bool = False            # ConvT output whose backward MFMAs round dy anyway: a lone 3x3-conv reader may hand
    grad16: Optional[torch.Tensor] = None   # them the gradient as 16-bit planes (mtbc_seg.accumulate = 2) instead of fp32
    dy8_ok: bool = False               # conv-cell output whose InstanceNorm backward can read a channel-blocked 16-bit dy
    grad8: Optional[torch.Tensor] = None    # ... the gradient from ALL its 3x3 consumers (one gathered launch), 16-bit channel-blocked
    z16: bool = False                  # conv-cell output whose InstanceNorm runs on the channel-group kernels (16-bit z)
    r1: Optional[tuple] = None         # (dy of a one-output 1x1 head, its weight): rank-1 gradient term formed inside the InstanceNorm backward
    pool: Optional[tuple] = None       # (gradient of this tensor's 2 x 2 max-pool, argmax codes): routed inside the InstanceNorm backward
    pooled: Optional["Act"] = None     # this tensor's 2 x 2 max-pool (planned once, however many modules pool it)
    parent: Optional["Act"] = None     # this Act is images [view_index * N, (view_index + 1) * N) of `parent` (StepPlan.batch_pair / split_batch)
    view_index: int = 0
    views: list = field(default_factory=list)
    c8_filled: bool = False            # (views) the channel-blocked half has a producer or a pack op already
    in_op: Optional[object] = None     # the InstanceNorm forward op that writes this activation (conv cells)
    pack_op: Optional[object] = None   # the op that fills `c8` from `data`

    @property
    def N(self): return self.data.shape[0]
    @property
    def C(self): return self.data.shape[1]
    @property
    def H(self): return self.data.shape[2]
    @property
    def W(self): return self.data.shape[3]
    @property
    def bstride(self): return self.data.shape[1] * self.data.shape[2] * self.data.shape[3]


@dataclass
class _Cell:
    """What the forward half of a conv cell hands to its backward half (StepPlan.conv_cell / _conv_cell_backward)."""
    inputs: List["Act"]
    y: "Act"
    N: int                                    # images of THIS cell (2 x the plan's batch for a module applied to two tensors at once)
    cin: int
    cout: int
    H: int
    W: int
    tag: int
    w: torch.Tensor
    wname: str
    bname: Optional[str]
    gname: Optional[str]
    betaname: Optional[str]
    slope: float
    mean: torch.Tensor
    rstd: torch.Tensor
    use_packed: bool
    wp_d: Optional[torch.Tensor] = None       # full dgrad weight image (dropped again when every input's gradient is gathered)
    wp_d_op: Optional[object] = None
    c8: bool = False                          # the MFMAs read channel-blocked 16-bit operands
    c8_bwd: bool = False
    stem16: bool = False
    z16: bool = False                         # conv output stored in 16 bits, channel-blocked
    zf16: bool = False                        # ... as fp16 (also in the bf16 mode)
    z: Optional[torch.Tensor] = None


@dataclass
class ParamSlot:
    name: str
    shape: Tuple[int, ...]
    offset: int = 0                    # element offset in the flat buffers
    grad_written: bool = False
    ready_at: int = -1                 # index of the last backward op that writes this grad

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n


class Program:
    """A contiguous ctypes array of ops + the stream-ordered runner (mtbc_program_run): the ops run on the caller's current stream.
    (The library can also run a program with stream-control ops on two streams, mtbc_program_run_ms: the weight gradient of a layer
    beside its input gradient was built and measured in round 2 -- +0.25 .. +1.0 ms per step -- and the plan no longer emits it.)"""

    def __init__(self, ops: List[L.Op], keep: list):
        self.n = len(ops)
        self.array = (L.Op * max(1, self.n))(*ops)
        self.keep = keep                 # tensors referenced by raw pointer
        self._failed = C.c_int32(-1)

    def run(self, first: int = 0, count: Optional[int] = None, stream: Optional[torch.cuda.Stream] = None) -> None:
        if count is None:
            count = self.n - first
        if count <= 0:
            return
        s = (stream or torch.cuda.current_stream()).cuda_stream
        rc = L.load().mtbc_program_run(self.array, first, count, C.c_void_p(s), C.byref(self._failed))
        if rc != 0:
            i = self._failed.value
            L.check(rc, f"program op #{i} (kind {self.array[i].kind}, tag {self.array[i].tag})")


def _mk(kind: int, tag: int = 0) -> L.Op:
    op = L.Op()
    op.kind = kind
    op.tag = tag
    return op


class StepPlan:
    """Builds the forward tape of a network for a fixed (N, H, W) and derives loss/backward/Adam programs.

    params: flat fp32 buffers (values p, grads g) with one ParamSlot per named tensor.
    """

    def __init__(self, device: torch.device, N: int, param_view: Callable[[str], torch.Tensor],
                 grad_view: Callable[[str], torch.Tensor], slots: Dict[str, ParamSlot], force_direct: bool = False,
                 compute: int = 0, coop_reserve_cus: int = 0, coop_state: Optional[Callable[[], torch.Tensor]] = None):
        self.dev = device
        self._shared_coop = coop_state   # the model's ONE mailbox / error-word block, shared by all of its step plans
        self.coop_reserve_cus = int(coop_reserve_cus)    # CUs the cooperative InstanceNorm grids leave to other streams (data parallel)
        self.compute = int(compute)      # MFMA operand type of the 3x3 convs: 0 fp32 (parity path), 1 bf16, 2 fp16
        self.N = N
        self.pv, self.gv, self.slots = param_view, grad_view, slots
        self.force_direct = 1 if force_direct else 0
        self.keep: list = []
        self.fwd_ops: List[L.Op] = []
        self.pack_ops: List[L.Op] = []
        self.wview_ops: List[L.Op] = []      # weight views of the gathered backward: run ahead of the pack ops
        self.bwd_emitters: List[Callable[[], None]] = []
        self.bwd_ops: List[L.Op] = []
        self.loss_ops: List[L.Op] = []
        self.ws_bytes = 0
        self.ws_users: List[Tuple[L.Op, str]] = []
        self._pending_dparams: List[Tuple[L.Op, tuple]] = []   # deferred InstanceNorm parameter-gradient reductions: (op, parameter names)
        self._sync_users: List[L.Op] = []                  # weight gradients with the in-kernel split-K reduction: one zeroed counter buffer
        self._sync_bytes = 0
        self._stat_users: List[Tuple[L.Op, str]] = []      # conv forward / InstanceNorm forward pairs sharing the epilogue-statistics scratch
        self._stat_bytes = 0
        self.arena_bytes = 0
        self._tag = 0
        self.acts: Dict[str, Act] = {}     # every activation by (reference module path) name, for parity probes
        self.lib = L.load()
        for s in slots.values():
            s.grad_written = False
            s.ready_at = -1

    # ------------------------------------------------------------------ memory
    def alloc(self, *shape, dtype=torch.float32) -> torch.Tensor:
        t = torch.empty(*shape, dtype=dtype, device=self.dev)
        self.arena_bytes += t.numel() * t.element_size()
        self.keep.append(t)
        return t

    def new_act(self, name: str, C_: int, H: int, W: int, needs_grad: bool = True, N: Optional[int] = None) -> Act:
        a = Act(name, self.alloc(self.N if N is None else N, C_, H, W), needs_grad)
        self.acts[name] = a
        return a

    # ---- one module applied to TWO tensors with shared weights (U-Net++: process_level_3 on pool(x_3_0) and pool(x_3_1),
    # MTUNetPlusPlus.py:79,128) runs ONCE over their batch concatenation: InstanceNorm is per sample, so the result is the two
    # applications' results exactly, in half the launches and with ONE weight gradient instead of two accumulated ones.  The two
    # halves of the 2N-image tensor are Acts of their own (views): producers write them, consumers read them, their gradients are
    # the halves of the parent's gradient.
    def _make_views(self, parent: Act) -> List[Act]:
        n = parent.N // 2
        parent.grad = self.grad_of(parent)
        for i in range(2):
            v = Act(f"{parent.name}[{i}]", parent.data[i * n:(i + 1) * n], parent.needs_grad)
            v.grad = parent.grad[i * n:(i + 1) * n]
            v.parent, v.view_index = parent, i
            if parent.c8 is not None:
                v.c8, v.c8_filled = parent.c8[i * n:(i + 1) * n], True
            v.planar_valid = parent.planar_valid
            parent.views.append(v)
            self.acts[v.name] = v
        return parent.views

    def batch_pair(self, name: str, C_: int, H: int, W: int) -> Tuple[Act, List[Act]]:
        """A 2N-image tensor whose halves are written by two producers (the returned views)."""
        parent = self.new_act(name, C_, H, W, N=2 * self.N)
        return parent, self._make_views(parent)

    def split_batch(self, parent: Act) -> List[Act]:
        """The two N-image halves of a 2N-image tensor as Acts for N-image consumers."""
        assert parent.N == 2 * self.N and not parent.views
        return self._make_views(parent)

    def _view_c8(self, v: Act) -> torch.Tensor:
        """channel-blocked storage of a view = its half of the parent's (allocated on first use)"""
        par = v.parent
        if par.c8 is None:
            par.c8 = self.alloc(par.N, par.C // 8, par.H * par.W, 8, dtype=torch.int16)
        for w in par.views:
            n = par.N // 2
            w.c8 = par.c8[w.view_index * n:(w.view_index + 1) * n]
        return v.c8

    def grad_of(self, a: Act) -> torch.Tensor:
        if a.grad is None:
            a.grad = self.alloc(*a.data.shape)
        return a.grad

    def grad_slot(self, a: Act) -> Tuple[torch.Tensor, int]:
        """Where the next backward contribution to `a` goes: (buffer, accumulate flag).  The first writer overwrites the buffer, later
        writers read-modify-write it (the static plan knows who is first: no memsets).  (Private fan-in buffers summed by the
        InstanceNorm backward were built and measured in round 1: dgrad -0.6 ms, norm backward +1.25 ms; removed.)"""
        if a.views:                 # a 2N-image tensor written as a whole: its halves are written with it
            if a.grad_written != all(v.grad_written for v in a.views) or (not a.grad_written and any(v.grad_written for v in a.views)):
                raise NotImplementedError(f"{a.name}: a batch-concatenated gradient written whole after one half was written alone")
            acc = 1 if a.grad_written else 0
            a.grad_written = True
            for v in a.views:
                v.grad_written = True
            return self.grad_of(a), acc
        if not a.grad_written:
            a.grad_written = True
            if a.parent is not None and all(v.grad_written for v in a.parent.views):
                a.parent.grad_written = True
            return self.grad_of(a), 0
        return self.grad_of(a), 1

    def _rd(self, a: Act) -> int:
        """Pointer to the fp32 planar values of an activation, for an op that reads them."""
        if not a.planar_valid:
            raise NotImplementedError(f"{a.name} exists only in the channel-blocked 16-bit layout")
        a.planar_used = True
        if a.parent is not None:
            a.parent.planar_used = True
        return a.data.data_ptr()

    def _c8_pack_op(self, src: torch.Tensor, dst: torch.Tensor, C_: int, HW: int, N: Optional[int] = None) -> L.Op:
        op = _mk(L.OP_C8_PACK)
        a = op.u.c8pack
        a.src, a.src_batch_stride, a.dst = src.data_ptr(), C_ * HW, dst.data_ptr()
        a.N, a.C, a.HW, a.compute = (self.N if N is None else N), C_, HW, self.compute
        return op

    def c8_of(self, a: Act) -> torch.Tensor:
        """The 16-bit channel-blocked copy of an activation (MTBC_LAYOUT_C8), converted ONCE -- by an op placed right
        behind the ops emitted so far, i.e. after its producer -- and then read by every 3x3 conv and weight gradient
        that consumes it (X0_0 of the U-Net++ feeds 4 convs and 4 wgrads): their staging becomes LDS-DMA."""
        a.c8_used = True

        def pack_half(v: Act) -> None:      # a view whose producer wrote fp32 planes: re-block ITS images (the other half may not exist yet)
            self._view_c8(v)
            v.pack_op = self._c8_pack_op(v.data, v.c8, v.C, v.H * v.W, v.N)
            self.fwd_ops.append(v.pack_op)
            v.c8_filled = v.planar_used = v.parent.planar_used = True

        if a.parent is not None:
            a.parent.c8_used = True
            if not a.c8_filled:
                pack_half(a)
            return a.c8
        if a.views:
            for v in a.views:
                if not v.c8_filled:
                    pack_half(v)
            return a.c8
        if a.c8 is None:
            a.c8 = self.alloc(a.N, a.C // 8, a.H * a.W, 8, dtype=torch.int16)
            a.pack_op = self._c8_pack_op(a.data, a.c8, a.C, a.H * a.W, a.N)
            self.fwd_ops.append(a.pack_op)
        return a.c8

    def _coop_state(self) -> int:
        """The zeroed mailbox block of the cooperative InstanceNorm kernels.  ONE per model, shared by every step plan the
        model compiles (training batch, short last batch, evaluation batch: they run one after the other on one stream),
        so that the sticky error word any of them sets is the word `check_nan()` / `result()` read, whichever plan ran
        last.  A plan built without a model (op tests) owns a block of its own."""
        if getattr(self, "_coop_buf", None) is None:
            if self._shared_coop is not None:
                self._coop_buf = self._shared_coop()
            else:
                self._coop_buf = torch.zeros(self.lib.mtbc_instnorm_coop_state_bytes() // 4, dtype=torch.int32, device=self.dev)
            self.keep.append(self._coop_buf)
        return self._coop_buf.data_ptr()

    def coop_error_word(self) -> Optional[torch.Tensor]:
        """Device view (1 x int32) of the sticky error word of this plan's cooperative kernels, or None when the plan has
        none.  Non-zero = a team member was not resident and a mailbox poll gave up: the step's results are garbage."""
        buf = getattr(self, "_coop_buf", None)
        if buf is None:
            return None
        i = self.lib.mtbc_instnorm_coop_error_offset() // 4
        return buf[i:i + 1]

    def _scratch16(self, attr: str, numel: int) -> torch.Tensor:
        """Shared 16-bit scratch of the backward pass (IN-backward -> pack -> wgrad -> dgrad run back to back on one
        stream): grows to the largest user, earlier ops keep the smaller buffer they were built with."""
        buf = getattr(self, attr, None)
        if buf is None or buf.numel() < numel:
            buf = self.alloc(numel, dtype=torch.int16)
            setattr(self, attr, buf)
        return buf

    def flush_dparams(self) -> None:
        """Emit the pending InstanceNorm parameter-gradient reductions (the runner issues a run of them as ONE launch); the gradients
        they finish are ready from the last of them on -- what the data-parallel bucket schedule keys on."""
        for dop, names in self._pending_dparams:
            self.bwd_ops.append(dop)
        for dop, names in self._pending_dparams:
            for name in names:
                self.slots[name].ready_at = len(self.bwd_ops) - 1
        self._pending_dparams = []

    def _need_ws(self, op: L.Op, fieldname: str, nbytes: int) -> None:
        self.ws_bytes = max(self.ws_bytes, int(nbytes))
        self.ws_users.append((op, fieldname))

    def _mark_param(self, name: str) -> int:
        """returns accumulate flag for a grad write to `name` and records readiness."""
        s = self.slots[name]
        acc = 1 if s.grad_written else 0
        s.grad_written = True
        s.ready_at = len(self.bwd_ops)      # index of the op about to be appended
        return acc

    def _segs(self, arr, acts: Sequence[Act], grads: bool = False, g16: bool = False) -> None:
        for i, a in enumerate(acts):
            if grads and g16 and a.grad16_ok and a.readers == 1 and not a.grad_written:
                # the only gradient this tensor will ever get, and its reader (the ConvT backward) rounds it to 16 bits
                # while loading: written as 16-bit planes by this dgrad launch (mtbc_seg.accumulate = 2)
                a.grad16 = self.alloc(*a.data.shape, dtype=torch.int16)
                a.grad_written = True
                arr[i].ptr = a.grad16.data_ptr()
                arr[i].accumulate = 2
            elif grads:
                g, acc = self.grad_slot(a)
                arr[i].ptr = g.data_ptr()
                arr[i].accumulate = acc
            else:
                arr[i].ptr = self._rd(a)
                arr[i].accumulate = 0
            arr[i].batch_stride = a.bstride
            arr[i].channels = a.C

    # ------------------------------------------------------------------ layers
    def conv_cell(self, inputs: Sequence[Act], cout: int, wname: str, bname: Optional[str],
                  gname: Optional[str], betaname: Optional[str], slope: float, out_name: str) -> Act:
        """conv3x3(pad 1) -> InstanceNorm(eps 1e-5, affine optional) -> LeakyReLU(slope): the forward ops now, the backward ops
        when `emit_backward` walks the tape (`_conv_cell_backward`)."""
        inputs = list(inputs)
        for a_ in inputs:
            a_.readers += 1
        N, H, W = inputs[0].N, inputs[0].H, inputs[0].W
        assert all(a.N == N for a in inputs), [(a.name, a.N) for a in inputs]
        cin = sum(a.C for a in inputs)
        w = self.pv(wname)
        assert tuple(w.shape) == (cout, cin, 3, 3), (wname, tuple(w.shape), cout, cin)
        use_packed = cin % 8 == 0 and all(a.C % 8 == 0 for a in inputs)
        wp_f = wp_d = wp_d_op = None
        if use_packed and self.compute:
            # 16-bit operand images, re-converted from the fp32 master weights every step
            wp_f, _ = self._pack_lp(w, cin, cout, 0)
            if any(a.needs_grad for a in inputs) and cout % 8 == 0:
                wp_d, wp_d_op = self._pack_lp(w, cin, cout, 1)
        elif use_packed:
            wp_f = self.alloc(self.lib.mtbc_conv3x3_packed_elems(cin, cout))
            op = _mk(L.OP_CONV3_PACK_FWD)
            op.u.pack.w, op.u.pack.packed, op.u.pack.Cin, op.u.pack.Cout = w.data_ptr(), wp_f.data_ptr(), cin, cout
            self.pack_ops.append(op)
            if any(a.needs_grad for a in inputs) and cout % 8 == 0:
                wp_d = self.alloc(self.lib.mtbc_conv3x3_packed_dgrad_elems(cin, cout))
                op = _mk(L.OP_CONV3_PACK_DGRAD)
                op.u.pack.w, op.u.pack.packed, op.u.pack.Cin, op.u.pack.Cout = w.data_ptr(), wp_d.data_ptr(), cin, cout
                self.pack_ops.append(op)
        y = self.new_act(out_name, cout, H, W, N=N)
        mean, rstd = self.alloc(N * cout), self.alloc(N * cout)
        self._tag += 1
        cell = _Cell(inputs=inputs, y=y, N=N, cin=cin, cout=cout, H=H, W=W, tag=self._tag, w=w, wname=wname, bname=bname, gname=gname,
                     betaname=betaname, slope=slope, mean=mean, rstd=rstd, use_packed=use_packed, wp_d=wp_d, wp_d_op=wp_d_op)

        # 16-bit modes: the MFMA operands are converted once per tensor into the channel-blocked 16-bit layout instead
        # of once per consumer inside the staging (same RNE, same MFMA order: bit-identical forward / dgrad)
        c8 = use_packed and self.compute != 0 and not self.force_direct and W % 4 == 0 and H >= 8 and W >= 8
        c8_bwd = c8 and cout % 8 == 0
        # 16-bit modes: the conv output z is stored once, in 16 bits, channel-blocked (what torch.autocast keeps between a
        # convolution and its normalisation): written by the igemm's epilogue (fp32 accumulate + bias, one RNE), read by the
        # InstanceNorm forward and backward as 16-byte pieces -- 2 + 2 + 2 instead of 4 + 4 + 4 bytes per element.  Needs the
        # channel-group InstanceNorm kernels in both directions (norm_coop.hip).  The stem (Cin = 1: fp32 operands, no MFMA) takes
        # part: its forward writes z in the same layout and leaves the same statistics, its weight gradient reads the channel-blocked dz
        stem16 = bool(self.compute) and cin == 1 and len(inputs) == 1 and cout % 8 == 0 and not self.force_direct and W % 4 == 0 \
            and H >= 8 and W >= 8 and inputs[0].planar_valid and not inputs[0].needs_grad
        z16 = False
        if (c8_bwd or stem16) and not _NO_Z16 and not _NO_COOP:      # (the one-plane InstanceNorm kernels read fp32 planes)
            q = L.InstNormArgs()
            q.N, q.C, q.H, q.W, q.out16_type, q.z_layout, q.coop_reserve_cus = N, cout, H, W, self.compute, L.LAYOUT_C8, self.coop_reserve_cus
            z16 = bool(self.lib.mtbc_instnorm_c8_supported(C.byref(q), 0)) and bool(self.lib.mtbc_instnorm_c8_supported(C.byref(q), 1))
        z = self.alloc(N, cout // 8, H * W, 8, dtype=torch.int16) if z16 else self.alloc(N, cout, H, W)
        # ... as fp16 also in the bf16 mode: a conv output in front of a norm is O(1), and fp16 keeps 11 significant bits of it in the
        # same 2 bytes (bf16: 8) -- measured on the held-out Dice of 3000-step runs (profiles/r02b_quality_sweep.md); the MFMA
        # operands (activations, dz, weights) stay bf16
        zf16 = z16 and self.compute == 1 and not _Z_BF16
        stem16 = stem16 and z16
        y.dy8_ok = z16 and _DA16 and not stem16
        y.z16 = z16
        cell.c8, cell.c8_bwd, cell.stem16, cell.z16, cell.zf16, cell.z = c8, c8_bwd, stem16, z16, zf16, z
        if not c8 and not all(a_.planar_valid for a_ in inputs):
            raise NotImplementedError(f"{out_name}: an input exists only in the channel-blocked 16-bit layout")
        for a_ in inputs:
            if c8_bwd and wp_d is not None:
                a_.conv_consumers += 1
            else:
                a_.no_gather = True

        op = self._cell_conv_op(cell, L.OP_CONV3_FWD)
        if c8:
            self._segs_c8(op.u.conv3.in_, inputs)
            op.u.conv3.operand_layout = L.LAYOUT_C8
        else:
            self._segs(op.u.conv3.in_, inputs)
        op.u.conv3.w_packed = _ptr(wp_f)
        op.u.conv3.bias = _ptr(self.pv(bname)) if bname else None
        op.u.conv3.out = z.data_ptr()
        stats_slots = 0
        if stem16:
            op.u.conv3.compute = self.compute
        if z16:
            op.u.conv3.out_layout = L.LAYOUT_C8
            if zf16:
                op.u.conv3.out_type = 2
            # InstanceNorm statistics from the conv epilogue ({sum, sum of squares} of the stored values per wave): the
            # normalisation that follows is then one streaming pass with no reduction / team exchange of its own
            stats_slots = int(self.lib.mtbc_conv3x3_stats_slots(C.byref(op.u.conv3)))
            if stats_slots > 0:
                self._stat_users.append((op, "conv3"))
                self._stat_bytes = max(self._stat_bytes, N * stats_slots * cout * 2 * 4)
        self.fwd_ops.append(op)

        op = self._cell_norm_op(cell, L.OP_IN_FWD)
        op.u.inorm.y, op.u.inorm.y_batch_stride = y.data.data_ptr(), y.bstride
        y.in_op = op
        if stats_slots > 0:
            op.u.inorm.stats_slots = stats_slots
            self._stat_users.append((op, "inorm"))
        if z16 or (self.compute and not _NO_COOP and not self.force_direct and cout % 8 == 0 and H * W >= _COOP_MIN_FWD
                   and self.lib.mtbc_instnorm_c8_supported(C.byref(op.u.inorm), 0)):
            # 16-bit modes: the channel-group / cooperative kernel writes the channel-blocked operand tensor itself (and fp32 planes
            # only if something reads them -- decided in finalize(), when all consumers are known)
            y.c8 = self.alloc(N, cout // 8, H * W, 8, dtype=torch.int16)
            op.u.inorm.y8, op.u.inorm.out16_type, op.u.inorm.coop_state = y.c8.data_ptr(), self.compute, self._coop_state()
        nb = self.lib.mtbc_instnorm_fwd_workspace(C.byref(op.u.inorm))      # > 0 only for planes larger than 64K elements
        if nb:
            self._need_ws(op, "inorm", nb)
        self.fwd_ops.append(op)
        self.bwd_emitters.append(lambda: self._conv_cell_backward(cell))
        return y

    # ---- helpers shared by the two halves of a conv cell
    def _pack_lp(self, wsrc: torch.Tensor, ci: int, co: int, dg: int) -> Tuple[torch.Tensor, L.Op]:
        """The 16-bit MFMA image of a weight tensor (forward image dg = 0, transposed / tap-flipped dgrad image dg = 1), rebuilt every step."""
        t = self.alloc(self.lib.mtbc_conv3x3_packed_lp_elems(ci, co, dg), dtype=torch.int16)
        op = _mk(L.OP_CONV3_PACK_LP)
        op.u.pack.w, op.u.pack.packed, op.u.pack.Cin, op.u.pack.Cout = wsrc.data_ptr(), t.data_ptr(), ci, co
        op.u.pack.dgrad, op.u.pack.compute = dg, self.compute
        self.pack_ops.append(op)
        return t, op

    def _wview(self, wsrc, dst, co, ci_total, off, cnt, mode, koff, K) -> None:
        op = _mk(L.OP_CONV3_WVIEW)
        v = op.u.wview
        v.w, v.dst, v.Cout, v.Cin, v.ci_off, v.ci_cnt, v.mode, v.k_off, v.K = wsrc.data_ptr(), dst.data_ptr(), co, ci_total, off, cnt, mode, koff, K
        self.wview_ops.append(op)

    def _segs_c8(self, arr, acts: Sequence[Act]) -> None:
        for i, a_ in enumerate(acts):
            arr[i].ptr = self.c8_of(a_).data_ptr()
            arr[i].batch_stride, arr[i].channels, arr[i].accumulate = a_.bstride, a_.C, 0

    def _cell_conv_op(self, cell: "_Cell", kind: int) -> L.Op:
        op = _mk(kind, cell.tag)
        a = op.u.conv3
        a.N, a.H, a.W, a.Cin, a.Cout, a.n_in = cell.N, cell.H, cell.W, cell.cin, cell.cout, len(cell.inputs)
        a.w = cell.w.data_ptr()
        a.force_direct = self.force_direct
        a.compute = self.compute if cell.use_packed else 0
        return op

    def _cell_norm_op(self, cell: "_Cell", kind: int) -> L.Op:
        op = _mk(kind, cell.tag)
        a = op.u.inorm
        a.N, a.C, a.H, a.W, a.eps, a.slope = cell.N, cell.cout, cell.H, cell.W, 1e-5, cell.slope
        a.z = cell.z.data_ptr()
        a.z_layout = L.LAYOUT_C8 if cell.z16 else L.LAYOUT_PLANAR
        a.z_type = 2 if cell.zf16 else 0
        a.gamma = _ptr(self.pv(cell.gname)) if cell.gname else None
        a.beta = _ptr(self.pv(cell.betaname)) if cell.betaname else None
        a.mean, a.rstd = cell.mean.data_ptr(), cell.rstd.data_ptr()
        a.coop_reserve_cus = self.coop_reserve_cus
        return op

    def _gathered_dgrad(self, cell: "_Cell") -> None:
        """The gradient of y with respect to ALL its 3x3 consumers in ONE forward-type launch over their channel-blocked dz (K = sum
        of their Cout), instead of one read-modify-write of y's fp32 gradient per consumer.  Weights: the consumers' slices for y's
        channels, transposed / tap-flipped, side by side (rebuilt each step: weight views, then the ordinary 16-bit image)."""
        y, cout, N, H, W = cell.y, cell.cout, cell.N, cell.H, cell.W
        K = sum(pj[4] for pj in y.pending)
        wg = self.alloc(cout, K, 3, 3)
        koff = 0
        for (dzj, wj, offj, cinj, coutj) in y.pending:
            self._wview(wj, wg, coutj, cinj, offj, cout, 1, koff, K)
            koff += coutj
        wpg, _ = self._pack_lp(wg, K, cout, 0)
        if y.dy8_ok:
            # (MTBC_DA16=1) ... written ONCE as a 16-bit channel-blocked tensor (fp32 sum over all consumers in the MFMA accumulators,
            # one RNE); what the tensor's other readers (pool / ConvT / 1x1 backward) wrote stays an fp32 planar partial that the
            # InstanceNorm backward adds while loading
            y.grad8 = self.alloc(N, cout // 8, H * W, 8, dtype=torch.int16)
            gbuf, acc = y.grad8, 0
        else:
            gbuf, acc = self.grad_slot(y)
        op = _mk(L.OP_CONV3_FWD, cell.tag)
        a = op.u.conv3
        a.N, a.H, a.W, a.Cin, a.Cout, a.n_in = N, H, W, K, cout, len(y.pending)
        for i_, (dzj, wj, offj, cinj, coutj) in enumerate(y.pending):
            a.in_[i_].ptr, a.in_[i_].batch_stride, a.in_[i_].channels, a.in_[i_].accumulate = dzj.data_ptr(), coutj * H * W, coutj, 0
        a.w, a.w_packed, a.out = wg.data_ptr(), wpg.data_ptr(), gbuf.data_ptr()
        a.compute, a.operand_layout, a.out_accumulate = self.compute, L.LAYOUT_C8, acc
        if y.dy8_ok:
            a.out_layout = L.LAYOUT_C8
        self.bwd_ops.append(op)

    def _conv_cell_backward(self, cell: "_Cell") -> None:
        """Backward of a conv cell: [gathered dgrad of y] -> InstanceNorm + LeakyReLU backward (dz channel-blocked, or in place over dy)
        -> weight gradient -> input gradient (whole, or the suffix of the inputs whose gradient is not gathered by THEIR cell)."""
        y, inputs, cout, N, H, W = cell.y, cell.inputs, cell.cout, cell.N, cell.H, cell.W
        gname, betaname, bname, z16, c8_bwd = cell.gname, cell.betaname, cell.bname, cell.z16, cell.c8_bwd
        if not y.grad_written and not y.pending and y.r1 is None and y.pool is None:
            return
        written_before = y.grad_written           # something else (ConvT / pool / head backward) has written y's fp32 gradient
        g8 = bool(y.pending) and y.dy8_ok
        if y.pending:
            self._gathered_dgrad(cell)
        only_folded = (y.r1 is not None or y.pool is not None) and not g8 and not y.grad_written       # no gradient TENSOR: only folded terms
        dy = y.grad8 if g8 else (None if only_folded else self.grad_of(y))
        op = self._cell_norm_op(cell, L.OP_IN_BWD)
        a = op.u.inorm
        a.dy, a.dy_batch_stride, a.dz = _ptr(dy), y.bstride, _ptr(dy)      # dz in place over dy unless a 16-bit output is asked for below
        if y.pool is not None:
            a.dy_pool, a.dy_pool_arg = y.pool[0].data_ptr(), y.pool[1].data_ptr()
        if y.r1 is not None:
            a.dy_rank1, a.dy_rank1_w = y.r1[0].data_ptr(), y.r1[1].data_ptr()
            a.dy_rank1_dw, a.dy_rank1_db, a.dy_rank1_accumulate = y.r1[2].data_ptr(), y.r1[3].data_ptr(), y.r1[4]
        if g8:
            a.dy_layout, a.dz = L.LAYOUT_C8, None
            if written_before:                    # the other readers' fp32 planar partial, added while loading
                a.n_dy_extra = 1
                a.dy_extra[0] = y.grad.data_ptr()
        # inputs whose gradient is gathered later (by THEIR backward) from all their 3x3 consumers: a prefix of the
        # segment list, so that this conv's own dgrad covers a contiguous suffix of its input channels
        defer: List[Act] = []
        if c8_bwd and cell.wp_d is not None and not _NO_GATHER:
            for a_ in inputs:
                if a_.in_op is not None and a_.needs_grad and a_.conv_consumers >= (1 if a_.dy8_ok else 2) and not a_.no_gather and a_.C % 8 == 0:
                    defer.append(a_)
                else:
                    break

        def dz8_buffer() -> torch.Tensor:      # the gathered launches read it later: it cannot be the shared scratch
            return self.alloc(N * cout * H * W, dtype=torch.int16) if defer else self._scratch16("_dz8_buf", N * cout * H * W)

        coop = z16 or (c8_bwd and not _NO_COOP and H * W >= _COOP_MIN_BWD and bool(self.lib.mtbc_instnorm_c8_supported(C.byref(a), 1)))
        dz8 = dz16 = None
        if coop:        # one pass straight into the channel-blocked dz the wgrad / dgrad MFMAs read
            dz8 = dz8_buffer()
            a.dz8, a.out16_type, a.coop_state = dz8.data_ptr(), self.compute, self._coop_state()
        p16 = c8_bwd and not coop and (H * W) % 4 == 0 and H * W <= 65536
        if p16:         # dz feeds MFMAs only: 16-bit planar here, channel-blocked by the pack below
            dz16 = self._scratch16("_dz16_buf", N * cout * H * W)
            a.dz16, a.out16_type = dz16.data_ptr(), self.compute
        if gname or bname:
            acc = None
            if gname:
                acc = self._mark_param(gname)
                self._mark_param(betaname)
                a.dgamma, a.dbeta = self.gv(gname).data_ptr(), self.gv(betaname).data_ptr()
            if bname:                       # conv bias gradient = sum of dz, produced by the same pass
                accb = self._mark_param(bname)
                acc = accb if acc is None else acc
                assert acc == accb
                a.dbias_pre = self.gv(bname).data_ptr()
            a.accumulate_dparams = acc
            self._need_ws(op, "inorm", N * cout * (131 if coop else 3) * 4)
        if y.r1 is not None:
            self._need_ws(op, "inorm", N * (cout + 1) * 262 * 4)
        dop = None
        if (gname or bname) and z16 and self.dev.type == "cuda":
            # the per-plane partial sums of dgamma / dbeta / dbias stay in a (small) buffer of this cell's own and are reduced later,
            # many cells per launch (a step has 36 of these 5 us reductions): after every _DPARAM_BATCH cells and at the end
            if a.accumulate_dparams:
                self.flush_dparams()            # an earlier contribution to the same parameters lands first
            buf = self.alloc(N * (cout + 1) * 262)
            self.ws_users = [u for u in self.ws_users if u[0] is not op]
            a.workspace, a.workspace_bytes, a.defer_dparams = buf.data_ptr(), buf.numel() * 4, 1
            dop = _mk(L.OP_IN_DPARAM, cell.tag)
            d = dop.u.dparam
            d.part, d.dgamma, d.dbeta, d.dbias_pre = buf.data_ptr(), a.dgamma, a.dbeta, a.dbias_pre
            d.N, d.C, d.T, d.accumulate = N, cout, int(self.lib.mtbc_instnorm_bwd_team(C.byref(a))), a.accumulate_dparams
        self.bwd_ops.append(op)
        if dop is not None:
            self._pending_dparams.append((dop, tuple(n_ for n_ in (gname, betaname, bname) if n_)))
            if len(self._pending_dparams) >= _DPARAM_BATCH:
                self.flush_dparams()
        if y.r1 is not None:            # the head's parameter gradients are written by THIS op
            for name in y.r1[5]:
                self.slots[name].ready_at = len(self.bwd_ops) - 1
        if c8_bwd and not coop:
            dz8 = dz8_buffer()
            pk = self._c8_pack_op(dz16 if p16 else dy, dz8, cout, H * W, N)
            if p16:
                pk.kind = L.OP_C8_PACK16
            self.bwd_ops.append(pk)
        # weight gradient
        op = self._cell_conv_op(cell, L.OP_CONV3_WGRAD)
        a = op.u.conv3
        if c8_bwd:
            self._segs_c8(a.in_, inputs)
            a.dout, a.operand_layout = dz8.data_ptr(), L.LAYOUT_C8
        elif cell.stem16:          # fp32 planar 1-channel input, channel-blocked dz
            self._segs(a.in_, inputs)
            a.dout, a.operand_layout, a.compute = dz8.data_ptr(), L.LAYOUT_C8, self.compute
        else:
            self._segs(a.in_, inputs)
            a.dout = dy.data_ptr()
        a.accumulate_dw = self._mark_param(cell.wname)
        a.dw = self.gv(cell.wname).data_ptr()
        # (batching the ~40 split-K reductions of a step into a few launches was built and measured: 15.20 vs 14.54 ms -- the
        #  partials then live in buffers of their own and travel to HBM and back instead of being reduced out of the cache)
        self._need_ws(op, "conv3", self.lib.mtbc_conv3x3_wgrad_workspace(C.byref(a)))
        # (round 4: the library can reduce the split-K partials of a channel-blocked weight gradient INSIDE the launch -- last arriver per
        #  group of rows, mtbc_conv3x3_args.wgrad_sync -- instead of by a reduction launch.  Built, tested, measured, NOT used: the 8 XCDs'
        #  L2s are not coherent with each other, so the partials have to travel as agent-scope (write-through) accesses and the last
        #  arriver of a tile pulls the whole tile's rows through ONE CU: every launch got 50 - 100 us slower, the step 11.2 -> 14.4 ms
        #  (profiles/r04_wgrad_in_kernel_reduction.txt).  _IN_KERNEL_SPLITK stays False.)
        nsync = int(self.lib.mtbc_conv3x3_wgrad_sync_bytes(C.byref(a))) if _IN_KERNEL_SPLITK else 0
        if nsync:
            self._sync_users.append(op)
            self._sync_bytes = max(self._sync_bytes, nsync)
        self.bwd_ops.append(op)
        # input gradient into every input that needs one
        need = [a_ for a_ in inputs if a_.needs_grad]
        if need and defer:
            off = 0
            for a_ in defer:
                a_.pending.append((dz8, cell.w, off, cell.cin, cout))
                off += a_.C
            self.pack_ops.remove(cell.wp_d_op)          # the full dgrad image is not needed
            rest = inputs[len(defer):]
            if rest:
                crest = cell.cin - off
                ws = self.alloc(cout, crest, 3, 3)
                self._wview(cell.w, ws, cout, cell.cin, off, crest, 0, 0, 0)
                wps, _ = self._pack_lp(ws, crest, cout, 1)
                op = self._cell_conv_op(cell, L.OP_CONV3_DGRAD)
                a = op.u.conv3
                a.Cin, a.n_in, a.w = crest, len(rest), ws.data_ptr()
                self._segs(a.in_, rest, grads=True, g16=True)
                a.dout, a.operand_layout, a.w_packed = dz8.data_ptr(), L.LAYOUT_C8, wps.data_ptr()
                self.bwd_ops.append(op)
        elif need:
            if len(need) != len(inputs):
                raise NotImplementedError("mixed grad / no-grad concat inputs")
            op = self._cell_conv_op(cell, L.OP_CONV3_DGRAD)
            a = op.u.conv3
            self._segs(a.in_, inputs, grads=True, g16=bool(c8_bwd and cell.wp_d is not None))
            a.dout = _ptr(dy)
            if c8_bwd and cell.wp_d is not None:
                a.dout, a.operand_layout = dz8.data_ptr(), L.LAYOUT_C8
            a.w_packed = _ptr(cell.wp_d)
            self.bwd_ops.append(op)

    def _c8_small_ok(self, x: Act) -> bool:
        """16-bit modes: may a small consumer (max-pool, 1x1 head) read the channel-blocked copy of x instead of fp32 planes?"""
        return bool(self.compute) and not self.force_direct and x.C % 8 == 0 and (x.H * x.W) % 4 == 0

    def maxpool(self, x: Act, out_name: str, out: Optional[Act] = None) -> Act:
        """out: the pooled tensor is written into this (view) Act -- one half of a batch-concatenated tensor (batch_pair)."""
        # A tensor pooled twice (U-Net++: x_3_0 feeds conv_4_0 AND the first application of process_level_3, MTUNetPlusPlus.py:75,128) is
        # pooled ONCE: both consumers read the same pooled tensor and its gradient is their fan-in.  (Round 4: two pools of one tensor had
        # each claimed the fold of their backward into the tensor's InstanceNorm backward -- ONE slot, the later emitter overwrote the
        # earlier one and the 16-bit modes lost the classification head's gradient into x_3_0; found by comparing gradients with the
        # emulation at a TRAINED state, tests/studies/try_emul3.py.  fp32 mode was not affected: no fold there.)
        if x.pooled is not None:
            assert out is None or out is x.pooled
            self.acts[out_name] = x.pooled
            return x.pooled
        x.readers += 1
        if out is not None:
            assert out.N == x.N and (out.C, out.H, out.W) == (x.C, x.H // 2, x.W // 2)
            y = out
            self.acts[out_name] = y
        else:
            y = self.new_act(out_name, x.C, x.H // 2, x.W // 2, N=x.N)
        x.pooled = y
        # 16-bit modes: pool the channel-blocked 16-bit tensor into a channel-blocked 16-bit tensor (max commutes with the
        # rounding: bit-identical to the fp32 pool + pack) -- the pooled tensor feeds 3x3 convs only, and x then needs no
        # fp32 planes on the pool's account.  Only when those convs can take the layout (else they need planes of y).
        c8 = self._c8_small_ok(x) and (x.W // 2) % 4 == 0 and x.H // 2 >= 8 and x.W // 2 >= 8 and x.H % 2 == 0 and x.W % 2 == 0
        fold = False
        if c8:
            y.c8 = self._view_c8(y) if y.parent is not None else self.alloc(x.N, x.C // 8, y.H * y.W, 8, dtype=torch.int16)
            y.c8_filled = True
            y.planar_valid = False
            if y.parent is not None:
                y.parent.planar_valid = False
            x8 = self.c8_of(x)
            # the pool's backward inside the InstanceNorm backward of x (mtbc_instnorm_args.dy_pool): the forward records where
            # each window's maximum sits (2 bits per channel), the pooled gradient is routed while the norm backward loads its
            # slab -- no 4x larger, three-quarters-zero fp32 tensor written, read-modify-written by the fan-in and read back
            fold = x.z16 and x.needs_grad
        arg = self.alloc(x.N, x.C // 8, y.H * y.W, dtype=torch.int16) if fold else None

        def base() -> L.Op:
            op = _mk(0)
            a = op.u.pool
            a.N, a.C, a.H, a.W = x.N, x.C, x.H, x.W
            if c8:
                a.layout, a.type16 = L.LAYOUT_C8, self.compute
                a.x, a.x_batch_stride = x8.data_ptr(), x.bstride
                a.y, a.y_batch_stride = y.c8.data_ptr(), y.bstride
            else:
                a.x, a.x_batch_stride = self._rd(x), x.bstride
                a.y, a.y_batch_stride = y.data.data_ptr(), y.bstride
            return op

        if c8 and x.in_op is not None and x.in_op.u.inorm.stats_slots > 0 and not x.in_op.u.inorm.pool_y8:
            # x's InstanceNorm forward is the streaming pass: it writes the pooled tensor (and the argmax codes) while it has the window
            # in registers -- no pool launch, no second read of x (mtbc_instnorm_args.pool_y8)
            x.in_op.u.inorm.pool_y8 = y.c8.data_ptr()
            x.in_op.u.inorm.pool_arg = arg.data_ptr() if fold else None
        else:
            op = base()
            op.kind = L.OP_POOL_FWD
            if fold:
                op.u.pool.argmax = arg.data_ptr()
            self.fwd_ops.append(op)

        def emit_bwd() -> None:
            if not y.grad_written or not x.needs_grad:
                return
            if fold:
                assert x.pool is None, f"{x.name}: a second max-pool backward folded into one InstanceNorm backward"
                x.pool = (self.grad_of(y), arg)
                return
            op = base()
            op.kind = L.OP_POOL_BWD
            a = op.u.pool
            a.dy, a.dy_batch_stride = self.grad_of(y).data_ptr(), y.bstride
            gx, acc = self.grad_slot(x)
            a.dx, a.dx_batch_stride, a.accumulate_dx = gx.data_ptr(), x.bstride, acc
            self.bwd_ops.append(op)

        self.bwd_emitters.append(emit_bwd)
        return y

    def convT(self, x: Act, cout: int, k: int, wname: str, bname: Optional[str], out_name: str) -> Act:
        assert x.N == self.N, f"{x.name}: only conv cells and max-pools are planned over a batch-concatenated (2N) tensor"
        w = self.pv(wname)
        assert tuple(w.shape) == (x.C, cout, k, k), (wname, tuple(w.shape))
        x.readers += 1
        y = self.new_act(out_name, cout, x.H * k, x.W * k)

        def base(reads_x: bool = True) -> L.Op:
            op = _mk(0)
            a = op.u.convT
            a.N, a.H, a.W, a.Cin, a.Cout, a.k = self.N, x.H, x.W, x.C, cout, k
            a.x, a.x_batch_stride = (self._rd(x) if reads_x else x.data.data_ptr()), x.bstride
            a.w = w.data_ptr()
            a.bias = _ptr(self.pv(bname)) if bname else None
            a.y, a.y_batch_stride = y.data.data_ptr(), y.bstride
            return op

        op = base(reads_x=False)
        op.kind = L.OP_CONVT_FWD
        if self.compute and cout % 8 == 0:
            # 16-bit modes: the up-sampled tensor feeds 3x3 convs only -- write it straight into their channel-blocked
            # 16-bit layout; y.data then stays unwritten.  First choice: the 16-bit MFMA forward that also READS the
            # channel-blocked copy of x (the same one the 3x3 convs read); else the fp32-MFMA forward on fp32 x (same
            # arithmetic as planar forward + pack); else planar output + pack.
            a = op.u.convT
            y.c8 = self.alloc(self.N, cout // 8, y.H * y.W, 8, dtype=torch.int16)
            a.y, a.y_batch_stride, a.y_layout, a.y_type = y.c8.data_ptr(), y.bstride, L.LAYOUT_C8, self.compute
            a.x_layout = L.LAYOUT_C8
            if k == 2 and x.C % 8 == 0 and self.lib.mtbc_convT_fwd_c8_supported(C.byref(a)):
                a.x = self.c8_of(x).data_ptr()
            else:
                a.x_layout = L.LAYOUT_PLANAR
                if not self.lib.mtbc_convT_fwd_c8_supported(C.byref(a)):
                    y.c8 = None
                    a.y, a.y_layout, a.y_type = y.data.data_ptr(), L.LAYOUT_PLANAR, 0
            y.planar_valid = y.c8 is None
        if op.u.convT.x_layout != L.LAYOUT_C8:
            op.u.convT.x = self._rd(x)
        self.fwd_ops.append(op)
        # 16-bit modes: the backward MFMAs of the k = 2 up-convolutions take rounded operands too (they are fp32-MFMA
        # bound otherwise); only where BOTH direct-to-fragment kernels of convt2.hip take the shape
        lp = self.compute if (k == 2 and (x.H * x.W) % 32 == 0 and x.W % 8 == 0 and cout % 2 == 0) else 0
        # ... and since they round dy while loading it, a dy with a single writer can arrive as 16-bit planes (half the bytes
        # written by the 3x3 conv's dgrad and read twice here); the writer decides (conv_cell.emit_bwd)
        y.grad16_ok = bool(lp) and (y.bstride % 8 == 0)

        def emit_bwd() -> None:
            if not y.grad_written:
                return
            g16 = y.grad16 is not None
            dy = y.grad16 if g16 else self.grad_of(y)
            # x as 16-bit planes too, where the streaming InstanceNorm pass can write them beside the channel-blocked copy and
            # nothing else wants fp32 planes of x (decided in _narrow_activations, once every reader is known)
            x16 = (g16 and x.in_op is not None and bool(x.in_op.u.inorm.y8) and x.in_op.u.inorm.stats_slots > 0
                   and (x.H * x.W) % 8 == 0 and x.bstride % 8 == 0)
            op = base(reads_x=not x16)
            if x16:
                x.p16_users.append(op)
            op.kind = L.OP_CONVT_WGRAD
            a = op.u.convT
            a.compute = lp
            a.dy, a.dy_batch_stride, a.dy_type16 = dy.data_ptr(), y.bstride, (lp if g16 else 0)
            a.accumulate_dw = self._mark_param(wname)
            a.dw = self.gv(wname).data_ptr()
            if bname:
                self._mark_param(bname)
                a.dbias = self.gv(bname).data_ptr()
            self._need_ws(op, "convT", self.lib.mtbc_convT_wgrad_workspace(C.byref(a)))
            self.bwd_ops.append(op)
            if x.needs_grad:
                op = base(reads_x=False)
                op.kind = L.OP_CONVT_DGRAD
                a = op.u.convT
                a.compute = lp
                a.dy, a.dy_batch_stride, a.dy_type16 = dy.data_ptr(), y.bstride, (lp if g16 else 0)
                gx, acc = self.grad_slot(x)
                a.dx, a.dx_batch_stride, a.accumulate_dx = gx.data_ptr(), x.bstride, acc
                self.bwd_ops.append(op)

        self.bwd_emitters.append(emit_bwd)
        return y

    def convT_head(self, x: Act, cmid: int, k: int, wTname: str, bTname: str, w1name: str, b1name: str, out_name: str) -> Act:
        """ConvTranspose2d(x.C -> cmid, k = s) followed by Conv2d(cmid -> R, 1x1) as ONE transposed conv with the
        combined weights (mtbc_convT_head_combine / _expand, include/mtbc.h): the cmid-channel full-resolution
        intermediate is never formed."""
        assert x.N == self.N
        wT, w1 = self.pv(wTname), self.pv(w1name)
        x.readers += 1
        R = w1.shape[0]
        assert tuple(wT.shape) == (x.C, cmid, k, k) and tuple(w1.shape) == (R, cmid, 1, 1), (wTname, w1name)
        Wc, bc = self.alloc(x.C, R, k, k), self.alloc(R)
        G, gb = self.alloc(x.C, R, k, k), self.alloc(R)
        y = self.new_act(out_name, R, x.H * k, x.W * k)

        def head() -> L.Op:
            op = _mk(0)
            a = op.u.head
            a.Cin, a.Cmid, a.R, a.k = x.C, cmid, R, k
            a.wT, a.bT, a.w1, a.b1 = wT.data_ptr(), _ptr(self.pv(bTname)), w1.data_ptr(), _ptr(self.pv(b1name))
            a.Wc, a.bc, a.G, a.gb = Wc.data_ptr(), bc.data_ptr(), G.data_ptr(), gb.data_ptr()
            return op

        def base() -> L.Op:
            op = _mk(0)
            a = op.u.convT
            a.N, a.H, a.W, a.Cin, a.Cout, a.k = self.N, x.H, x.W, x.C, R, k
            a.x, a.x_batch_stride = self._rd(x), x.bstride
            a.w, a.bias = Wc.data_ptr(), bc.data_ptr()
            a.y, a.y_batch_stride = y.data.data_ptr(), y.bstride
            return op

        op = head()
        op.kind = L.OP_HEAD_COMBINE
        self.fwd_ops.append(op)
        op = base()
        op.kind = L.OP_CONVT_FWD
        self.fwd_ops.append(op)

        def emit_bwd() -> None:
            if not y.grad_written:
                return
            dy = self.grad_of(y)
            op = base()
            op.kind = L.OP_CONVT_WGRAD
            a = op.u.convT
            a.dy, a.dy_batch_stride = dy.data_ptr(), y.bstride
            a.dw, a.dbias, a.accumulate_dw = G.data_ptr(), gb.data_ptr(), 0
            self._need_ws(op, "convT", self.lib.mtbc_convT_wgrad_workspace(C.byref(a)))
            self.bwd_ops.append(op)
            op = head()
            op.kind = L.OP_HEAD_EXPAND
            a = op.u.head
            a.acc_wT, a.acc_bT = self._mark_param(wTname), self._mark_param(bTname)
            a.acc_w1, a.acc_b1 = self._mark_param(w1name), self._mark_param(b1name)
            a.dwT, a.dbT = self.gv(wTname).data_ptr(), self.gv(bTname).data_ptr()
            a.dw1, a.db1 = self.gv(w1name).data_ptr(), self.gv(b1name).data_ptr()
            self.bwd_ops.append(op)
            if x.needs_grad:
                op = base()
                op.kind = L.OP_CONVT_DGRAD
                a = op.u.convT
                a.dy, a.dy_batch_stride = dy.data_ptr(), y.bstride
                gx, acc = self.grad_slot(x)
                a.dx, a.dx_batch_stride, a.accumulate_dx = gx.data_ptr(), x.bstride, acc
                self.bwd_ops.append(op)

        self.bwd_emitters.append(emit_bwd)
        return y

    def conv1x1(self, x: Act, cout: int, wname: str, bname: str, out_name: str) -> Act:
        assert x.N == self.N, f"{x.name}: only conv cells and max-pools are planned over a batch-concatenated (2N) tensor"
        w = self.pv(wname)
        assert tuple(w.shape) == (cout, x.C, 1, 1), (wname, tuple(w.shape))
        x.readers += 1
        y = self.new_act(out_name, cout, x.H, x.W)
        # 16-bit modes: the head reads the channel-blocked 16-bit activation the 3x3 convs read (fp32 weights, products
        # and sums): x then needs no fp32 planes on the head's account
        c8 = self._c8_small_ok(x) and cout <= 8
        x8 = self.c8_of(x) if c8 else None

        def base() -> L.Op:
            op = _mk(0)
            a = op.u.conv1
            a.N, a.H, a.W, a.Cin, a.Cout = self.N, x.H, x.W, x.C, cout
            if c8:
                a.x, a.x_batch_stride, a.x_layout, a.x_type = x8.data_ptr(), x.bstride, L.LAYOUT_C8, self.compute
            else:
                a.x, a.x_batch_stride = self._rd(x), x.bstride
            a.w, a.bias, a.y = w.data_ptr(), _ptr(self.pv(bname)), y.data.data_ptr()
            return op

        op = base()
        op.kind = L.OP_CONV1_FWD
        self.fwd_ops.append(op)

        def emit_bwd() -> None:
            if not y.grad_written:
                return
            dy = self.grad_of(y)
            r1 = x.needs_grad and c8 and cout == 1 and x.z16 and x.r1 is None
            if r1:
                # dx = w[c] * dy[n, pixel] is rank 1: the InstanceNorm backward of x forms it from dy (4 B per pixel) and w instead of
                # this head writing C fp32 planes that the fan-in and the norm read back (mtbc_instnorm_args.dy_rank1) -- and since it
                # re-forms the activation anyway, it also leaves the head's own weight / bias gradient (dy_rank1_dw / _db): no launch
                # of this head in the backward program at all
                acc = self._mark_param(wname)
                self._mark_param(bname)
                x.r1 = (dy, w, self.gv(wname), self.gv(bname), acc, (wname, bname))
                return
            op = base()
            op.kind = L.OP_CONV1_WGRAD
            a = op.u.conv1
            a.dy = dy.data_ptr()
            a.accumulate_dw = self._mark_param(wname)
            self._mark_param(bname)
            a.dw, a.dbias = self.gv(wname).data_ptr(), self.gv(bname).data_ptr()
            self._need_ws(op, "conv1", self.lib.mtbc_conv1x1_wgrad_workspace(C.byref(a)))
            self.bwd_ops.append(op)
            if x.needs_grad:
                op = base()
                op.kind = L.OP_CONV1_DGRAD
                a = op.u.conv1
                a.dy = dy.data_ptr()
                gx, acc = self.grad_slot(x)
                a.dx, a.dx_batch_stride, a.accumulate_dx = gx.data_ptr(), x.bstride, acc
                self.bwd_ops.append(op)

        self.bwd_emitters.append(emit_bwd)
        return y

    def gap(self, x: Act, out_name: str) -> Act:
        assert x.N == self.N, f"{x.name}: only conv cells and max-pools are planned over a batch-concatenated (2N) tensor"
        x.readers += 1
        y = Act(out_name, self.alloc(self.N, x.C, 1, 1))

        def base() -> L.Op:
            op = _mk(0)
            a = op.u.gap
            a.N, a.C, a.H, a.W = self.N, x.C, x.H, x.W
            a.x, a.y = self._rd(x), y.data.data_ptr()
            return op

        op = base()
        op.kind = L.OP_GAP_FWD
        self.fwd_ops.append(op)

        def emit_bwd() -> None:
            if not y.grad_written:
                return
            if x.grad_written:
                raise NotImplementedError("gap backward overwrites dx")
            op = base()
            op.kind = L.OP_GAP_BWD
            op.u.gap.dy, op.u.gap.dx = self.grad_of(y).data_ptr(), self.grad_of(x).data_ptr()
            x.grad_written = True
            self.bwd_ops.append(op)

        self.bwd_emitters.append(emit_bwd)
        return y

    def linear(self, x: Act, out_f: int, wname: str, bname: str, relu: bool, out_name: str) -> Act:
        assert x.N == self.N, f"{x.name}: only conv cells and max-pools are planned over a batch-concatenated (2N) tensor"
        in_f = x.C * x.H * x.W
        x.readers += 1
        w = self.pv(wname)
        assert tuple(w.shape) == (out_f, in_f), (wname, tuple(w.shape))
        y = Act(out_name, self.alloc(self.N, out_f, 1, 1))

        def base() -> L.Op:
            op = _mk(0)
            a = op.u.linear
            a.N, a.In, a.Out, a.relu = self.N, in_f, out_f, 1 if relu else 0
            a.x, a.w, a.bias, a.y = self._rd(x), w.data_ptr(), _ptr(self.pv(bname)), y.data.data_ptr()
            return op

        op = base()
        op.kind = L.OP_LINEAR_FWD
        self.fwd_ops.append(op)

        def emit_bwd() -> None:
            if not y.grad_written:
                return
            if x.grad_written:
                raise NotImplementedError("linear backward overwrites dx")
            op = base()
            op.kind = L.OP_LINEAR_BWD
            a = op.u.linear
            a.dy = self.grad_of(y).data_ptr()
            a.dx = self.grad_of(x).data_ptr() if x.needs_grad else None
            a.accumulate_dw = self._mark_param(wname)
            self._mark_param(bname)
            a.dw, a.dbias = self.gv(wname).data_ptr(), self.gv(bname).data_ptr()
            if relu:
                self._need_ws(op, "linear", self.N * out_f * 4)
            x.grad_written = True
            self.bwd_ops.append(op)

        self.bwd_emitters.append(emit_bwd)
        return y

    # ------------------------------------------------------------------ losses (fused-step path)
    def fused_losses(self, seg_heads: Sequence[Act], logits: Act, mask: torch.Tensor, onehot: torch.Tensor,
                     alpha: float, inversely_weighted: bool, focal_weight: Optional[torch.Tensor] = None,
                     loss_scale: float = 1.0, binary: bool = False, cls_gamma: float = 2.0):
        """criterions.py:52-76 + training_multitask.py:98 on device: Dice over the heads (weights 1/(j+1) from the
        LAST head backwards), Focal on the logits, alpha-mix, NaN flag.  Gradients land in the heads' grad buffers."""
        nh = len(seg_heads)
        assert 1 <= nh <= 4
        N, C_, H, W = seg_heads[0].data.shape
        self.dice_stats = self.alloc(nh * N * C_ * 3)
        self.dice_loss = self.alloc(nh + 1)
        self.focal_loss = self.alloc(1)
        self.loss_out = self.alloc(4)
        # device scalar multiplied into dL/d(logits) of both losses: 1 except on a rank whose shard of the global batch
        # is not 1/world of it (trainer.FusedTrainStep.load_batch(weight=...)); the reported losses are not scaled
        self.grad_weight = torch.ones(1, dtype=torch.float32, device=self.dev)
        self.keep.append(self.grad_weight)
        weights = [(1.0 / (nh - i)) if inversely_weighted else 1.0 for i in range(nh)]   # head i is reversed index nh-1-i

        def dice_base(kind: int) -> L.Op:
            op = _mk(kind)
            a = op.u.dice
            a.n_heads, a.N, a.C, a.H, a.W, a.smooth_nr, a.smooth_dr = nh, N, C_, H, W, 1.0, 1.0
            for i, h in enumerate(seg_heads):
                a.x[i] = h.data.data_ptr()
                a.head_weight[i] = weights[i]
            a.target, a.stats, a.loss = mask.data_ptr(), self.dice_stats.data_ptr(), self.dice_loss.data_ptr()
            return op

        self.loss_ops.append(dice_base(L.OP_DICE_FWD))
        op = dice_base(L.OP_DICE_BWD)
        for i, h in enumerate(seg_heads):
            op.u.dice.dx[i] = self.grad_of(h).data_ptr()
            h.grad_written = True
        op.u.dice.gscale_dev = self.grad_weight.data_ptr()
        op.u.dice.gscale = alpha * loss_scale      # loss_scale: fp16 mode keeps dz inside the fp16 range; Adam divides it out
        self.loss_ops.append(op)
        # classification criterion (experiment_init.py:232-262): "Focal" = FocalLoss(alpha 1, gamma 2); "CE" = CrossEntropyLoss = the same
        # formula with gamma 0; the ONE-logit head (n_classes == 2, `binary`) = BCEWithLogitsLoss on the {0, 1} label, which the focal
        # kernel evaluates when C == 1 (gamma 0): `onehot` then holds the (N, 1) float label itself
        assert (logits.C == 1) == bool(binary), "one logit <=> binary head"
        op = _mk(L.OP_FOCAL)
        a = op.u.focal
        a.N, a.C, a.alpha, a.gamma = N, logits.C, 1.0, (0.0 if binary else float(cls_gamma))
        a.x, a.target, a.weight = logits.data.data_ptr(), onehot.data_ptr(), _ptr(focal_weight)
        a.loss, a.dx, a.gscale = self.focal_loss.data_ptr(), self.grad_of(logits).data_ptr(), (1.0 - alpha) * loss_scale
        a.gscale_dev = self.grad_weight.data_ptr()
        logits.grad_written = True
        self.loss_ops.append(op)
        op = _mk(L.OP_LOSS_MIX)
        op.u.mix.seg = self.dice_loss.data_ptr() + 4 * nh
        op.u.mix.cls, op.u.mix.alpha, op.u.mix.out4 = self.focal_loss.data_ptr(), alpha, self.loss_out.data_ptr()
        self.loss_ops.append(op)
        self.keep += [mask, onehot] + ([focal_weight] if focal_weight is not None else [])

    # ------------------------------------------------------------------ finalisation
    def emit_backward(self) -> None:
        for em in reversed(self.bwd_emitters):
            em()
        self.flush_dparams()

    def _narrow_activations(self) -> None:
        """Conv-cell outputs that ONLY 3x3 convs read (the first conv of every double-conv block): InstanceNorm writes
        them as 16-bit planes instead of fp32 and the pack re-blocks 16-bit words -- 4 + 2 + 2 + 2 instead of
        4 + 4 + 4 + 2 bytes per element, same values (one RNE either way)."""
        if not self.compute:
            return
        for a in self.acts.values():
            hw = a.H * a.W
            if a.p16_users:           # ConvT weight gradients that can take 16-bit planes: only if nobody reads the fp32 ones
                if a.planar_used:
                    for op in a.p16_users:
                        op.u.convT.x = a.data.data_ptr()
                else:
                    y16 = self.alloc(a.N, a.C, a.H, a.W, dtype=torch.int16)
                    a.in_op.u.inorm.y16, a.in_op.u.inorm.y = y16.data_ptr(), None
                    for op in a.p16_users:
                        op.u.convT.x, op.u.convT.x_type16 = y16.data_ptr(), self.compute
                    a.planar_valid = False
                    continue
            if a.in_op is not None and a.in_op.u.inorm.y8:          # cooperative kernel: drop the output nobody reads
                if not a.c8_used and a.in_op.u.inorm.z_layout != L.LAYOUT_C8:      # (a channel-blocked z is read by the channel-group kernels only: they keep their output)
                    a.in_op.u.inorm.y8 = None
                elif not a.planar_used:
                    a.in_op.u.inorm.y = None
                    a.planar_valid = False
                continue
            if a.in_op is None or a.pack_op is None or a.planar_used or hw % 4 or hw > 65536:
                continue
            y16 = self.alloc(a.N, a.C, a.H, a.W, dtype=torch.int16)
            a.in_op.u.inorm.y16, a.in_op.u.inorm.out16_type = y16.data_ptr(), self.compute
            a.pack_op.kind = L.OP_C8_PACK16
            a.pack_op.u.c8pack.src = y16.data_ptr()
            a.planar_valid = False

    def finalize(self) -> Dict[str, Program]:
        self._narrow_activations()
        ws = self.alloc(max(16, (self.ws_bytes + 15) // 16 * 4)) if self.ws_bytes else None
        for op, fieldname in self.ws_users:
            a = getattr(op.u, fieldname)
            a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel() * 4
        if self._sync_bytes:
            sync = torch.zeros((self._sync_bytes + 3) // 4, dtype=torch.int32, device=self.dev)
            self.keep.append(sync)
            self.arena_bytes += sync.numel() * 4
            for op in self._sync_users:
                op.u.conv3.wgrad_sync, op.u.conv3.wgrad_sync_bytes = sync.data_ptr(), sync.numel() * 4
        if self._stat_bytes:       # one scratch for all cells: a conv's partials are consumed by the InstanceNorm op right behind it
            sb = self.alloc((self._stat_bytes + 15) // 16 * 4)
            for op, fieldname in self._stat_users:
                getattr(op.u, fieldname).stats_partial = sb.data_ptr()
        return {
            "pack": Program(self.wview_ops + self.pack_ops, self.keep),
            "fwd": Program(self.fwd_ops, self.keep),
            "loss": Program(self.loss_ops, self.keep),
            "bwd": Program(self.bwd_ops, self.keep),
        }

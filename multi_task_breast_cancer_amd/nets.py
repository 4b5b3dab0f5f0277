"""MTnnUNet and MTUNetPlusPlus as thin nn.Module shells over HIP step programs.

Drop-in surface: `model(x) -> (logits, segs)` (lists under deep supervision), `.parameters()`,
`.state_dict()` with the reference's key names (src/models/multitask/MTnnUNet.py:79-132,
MTUNetPlusPlus.py:47-87 on MONAI 1.3.0 blocks), `.train()`, `.to()`.  All arithmetic runs in
libmtbc_hip.so; there is no torch / CPU fallback -- forward on a CPU tensor raises.

Parameters live in ONE flat fp32 buffer (plus a flat gradient buffer) so Adam is one launch and the
data-parallel all-reduce works on contiguous buckets; each nn.Parameter is a view into it.
"""
from __future__ import annotations

import os

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import switches as _sw
from .engine import Act, ParamSlot, StepPlan

NNUNET_WIDTHS = (32, 64, 128, 256, 320)              # MTnnUNet.py:72
UNETPP_FEATURES = (24, 48, 96, 192, 384, 24)         # MTUNetPlusPlus.py:18


# ----------------------------------------------------------------------------------------------
# Parameter creation.  Values are drawn by instantiating the same torch.nn layer types in the
# reference's construction order, so that a seeded global RNG gives bit-identical initial weights.
# ----------------------------------------------------------------------------------------------
def _draw_conv(cin: int, cout: int, k: int, bias: bool):
    m = nn.Conv2d(cin, cout, kernel_size=k, padding=k // 2, bias=bias)
    return m.weight.detach(), (m.bias.detach() if bias else None)


def _draw_convT(cin: int, cout: int, k: int):
    m = nn.ConvTranspose2d(cin, cout, kernel_size=k, stride=k)
    return m.weight.detach(), m.bias.detach()


def _draw_linear(i: int, o: int):
    m = nn.Linear(i, o)
    return m.weight.detach(), m.bias.detach()


def _mtnnunet_params(sequences: int, regions: int, n_classes: int) -> List[Tuple[str, torch.Tensor]]:
    w = NNUNET_WIDTHS
    out: List[Tuple[str, torch.Tensor]] = []
    conv_names: List[str] = []            # every nn.Conv2d weight created before the head (re-initialised below)

    def level(name, cin, cmid, cout):
        for cell, (a, b) in (("ConvInNormLRelu1", (cin, cmid)), ("ConvInNormLRelu2", (cmid, cout))):
            wt, _ = _draw_conv(a, b, 3, False)
            out.append((f"{name}.{cell}.Conv.weight", wt))
            conv_names.append(f"{name}.{cell}.Conv.weight")

    level("encoder1", sequences, w[0], w[0])
    for i in range(1, 5):
        level(f"encoder{i + 1}", w[i - 1], w[i], w[i])
    level("bottleneck", w[4], w[4], w[4])
    level("decoder5", 2 * w[4], w[3], w[3])
    level("decoder4", 2 * w[3], w[2], w[2])
    level("decoder3", 2 * w[2], w[1], w[1])
    level("decoder2", 2 * w[1], w[0], w[0])
    level("decoder1", 2 * w[0], w[0], w[0] // 2)
    for i in (5, 4, 3, 2, 1):
        wt, b = _draw_convT(w[i - 1], w[i - 1], 2)
        out += [(f"upsample{i}.weight", wt), (f"upsample{i}.bias", b)]
    for name, c, k in (("output4", w[2], 8), ("output3", w[1], 4), ("output2", w[0], 2)):
        wt, b = _draw_convT(c, c, k)
        out += [(f"{name}.0.weight", wt), (f"{name}.0.bias", b)]
        wt, b = _draw_conv(c, regions, 1, True)
        out += [(f"{name}.1.weight", wt), (f"{name}.1.bias", b)]
        conv_names.append(f"{name}.1.weight")
    wt, b = _draw_conv(w[0] // 2, regions, 1, True)
    out += [("output1.weight", wt), ("output1.bias", b)]
    conv_names.append("output1.weight")
    # weights_initialization(), MTnnUNet.py:120,134-140: kaiming-normal on the Conv2d modules that exist so far
    d = dict(out)
    for nm in conv_names:
        nn.init.kaiming_normal_(d[nm], nonlinearity="leaky_relu")
        bn = nm[:-6] + "bias"
        if bn in d:
            nn.init.constant_(d[bn], 0)
    # classification head (default torch init), MTnnUNet.py:123-132
    out.append(("process_encoder_5.Conv.weight", _draw_conv(w[4], w[4], 3, False)[0]))
    out.append(("process_decoder_5.Conv.weight", _draw_conv(w[3], w[4], 3, False)[0]))
    out.append(("classifier.0.Conv.weight", _draw_conv(3 * w[4], 512, 3, False)[0]))
    wt, b = _draw_linear(512, 256)
    out += [("classifier.3.weight", wt), ("classifier.3.bias", b)]
    wt, b = _draw_linear(256, n_classes)
    out += [("classifier.5.weight", wt), ("classifier.5.bias", b)]
    return out


def _unetpp_params(in_ch: int, out_ch: int, n_classes: int) -> List[Tuple[str, torch.Tensor]]:
    f = UNETPP_FEATURES
    out: List[Tuple[str, torch.Tensor]] = []

    def convolution(name, cin, cout):       # MONAI Convolution: conv(k3, bias) + ADN(N: InstanceNorm affine)
        wt, b = _draw_conv(cin, cout, 3, True)
        out.extend([(f"{name}.conv.weight", wt), (f"{name}.conv.bias", b),
                    (f"{name}.adn.N.weight", torch.ones(cout)), (f"{name}.adn.N.bias", torch.zeros(cout))])

    def two_conv(name, cin, cout):
        convolution(f"{name}.conv_0", cin, cout)
        convolution(f"{name}.conv_1", cout, cout)

    def upcat(name, in_chns, cat_chns, out_chns, halves=True):
        up = in_chns // 2 if halves else in_chns
        wt, b = _draw_convT(in_chns, up, 2)
        out.extend([(f"{name}.upsample.deconv.weight", wt), (f"{name}.upsample.deconv.bias", b)])
        two_conv(f"{name}.convs", cat_chns + up, out_chns)

    two_conv("conv_0_0", in_ch, f[0])
    for i in range(1, 5):
        two_conv(f"conv_{i}_0.convs", f[i - 1], f[i])
    upcat("upcat_0_1", f[1], f[0], f[0], halves=False)
    upcat("upcat_1_1", f[2], f[1], f[1])
    upcat("upcat_2_1", f[3], f[2], f[2])
    upcat("upcat_3_1", f[4], f[3], f[3])
    upcat("upcat_0_2", f[1], f[0] * 2, f[0], halves=False)
    upcat("upcat_1_2", f[2], f[1] * 2, f[1])
    upcat("upcat_2_2", f[3], f[2] * 2, f[2])
    upcat("upcat_0_3", f[1], f[0] * 3, f[0], halves=False)
    upcat("upcat_1_3", f[2], f[1] * 3, f[1])
    upcat("upcat_0_4", f[1], f[0] * 4, f[5], halves=False)
    for j, c in ((1, f[0]), (2, f[0]), (3, f[0]), (4, f[5])):
        wt, b = _draw_conv(c, out_ch, 1, True)
        out += [(f"final_conv_0_{j}.weight", wt), (f"final_conv_0_{j}.bias", b)]
    two_conv("process_level_3.convs", f[3], f[4])
    two_conv("classifier.0", f[4] * 3, 512)
    wt, b = _draw_linear(512, 256)
    out += [("classifier.3.weight", wt), ("classifier.3.bias", b)]
    wt, b = _draw_linear(256, n_classes)
    out += [("classifier.5.weight", wt), ("classifier.5.bias", b)]
    return out


# ----------------------------------------------------------------------------------------------
# Forward graphs (emit ops into a StepPlan)
# ----------------------------------------------------------------------------------------------
def _graph_mtnnunet(plan: StepPlan, x: Act):
    w = NNUNET_WIDTHS

    def cell(inputs, cout, name):
        return plan.conv_cell(inputs, cout, f"{name}.Conv.weight", None, None, None, 0.01, name)

    def level(inputs, cmid, cout, name):
        return cell([cell(inputs, cmid, f"{name}.ConvInNormLRelu1")], cout, f"{name}.ConvInNormLRelu2")

    enc, t = [], x
    for i in range(5):
        e = level([t], w[i], w[i], f"encoder{i + 1}")
        enc.append(e)
        t = plan.maxpool(e, f"pool{i + 1}")
    bott = level([t], w[4], w[4], "bottleneck")
    up5 = plan.convT(bott, w[4], 2, "upsample5.weight", "upsample5.bias", "up5")   # used twice (F10), computed once
    dec_out = {5: w[3], 4: w[2], 3: w[1], 2: w[0], 1: w[0] // 2}
    dec_mid = {5: w[3], 4: w[2], 3: w[1], 2: w[0], 1: w[0]}
    dec, d, up = {}, None, up5
    for i in (5, 4, 3, 2, 1):
        if i != 5:
            up = plan.convT(d, w[i - 1], 2, f"upsample{i}.weight", f"upsample{i}.bias", f"up{i}")
        d = level([enc[i - 1], up], dec_mid[i], dec_out[i], f"decoder{i}")
        dec[i] = d
    pe5 = cell([enc[4]], w[4], "process_encoder_5")
    pd5 = cell([dec[5]], w[4], "process_decoder_5")
    feat = cell([pe5, up5, pd5], 512, "classifier.0")
    g = plan.gap(feat, "gap")
    h = plan.linear(g, 256, "classifier.3.weight", "classifier.3.bias", True, "fc1")
    n_cls = plan.pv("classifier.5.weight").shape[0]
    logits = plan.linear(h, n_cls, "classifier.5.weight", "classifier.5.bias", False, "logits")
    regions = plan.pv("output1.weight").shape[0]
    # deep-supervision heads: ConvT(k = 8 / 4 / 2) + 1x1 conv with nothing in between -> one transposed conv with the
    # combined weights (StepPlan.convT_head); MTBC_NOFUSE_HEADS=1 keeps the reference's two layers (A/B, parity check)
    if _sw.flag("MTBC_NOFUSE_HEADS"):
        o4 = plan.conv1x1(plan.convT(dec[4], w[2], 8, "output4.0.weight", "output4.0.bias", "o4up"), regions,
                          "output4.1.weight", "output4.1.bias", "output4")
        o3 = plan.conv1x1(plan.convT(dec[3], w[1], 4, "output3.0.weight", "output3.0.bias", "o3up"), regions,
                          "output3.1.weight", "output3.1.bias", "output3")
        o2 = plan.conv1x1(plan.convT(dec[2], w[0], 2, "output2.0.weight", "output2.0.bias", "o2up"), regions,
                          "output2.1.weight", "output2.1.bias", "output2")
    else:
        o4 = plan.convT_head(dec[4], w[2], 8, "output4.0.weight", "output4.0.bias", "output4.1.weight", "output4.1.bias", "output4")
        o3 = plan.convT_head(dec[3], w[1], 4, "output3.0.weight", "output3.0.bias", "output3.1.weight", "output3.1.bias", "output3")
        o2 = plan.convT_head(dec[2], w[0], 2, "output2.0.weight", "output2.0.bias", "output2.1.weight", "output2.1.bias", "output2")
    o1 = plan.conv1x1(dec[1], regions, "output1.weight", "output1.bias", "output1")
    return logits, [o4, o3, o2, o1]


def _graph_unetpp(plan: StepPlan, x: Act):
    f = UNETPP_FEATURES

    def convolution(inputs, cout, name):
        return plan.conv_cell(inputs, cout, f"{name}.conv.weight", f"{name}.conv.bias", f"{name}.adn.N.weight",
                              f"{name}.adn.N.bias", 0.1, name)

    def two_conv(inputs, cout, name):
        return convolution([convolution(inputs, cout, f"{name}.conv_0")], cout, f"{name}.conv_1")

    def down(t, cout, name, tag=""):
        return two_conv([plan.maxpool(t, f"{name}.pool{tag}")], cout, f"{name}.convs")

    def upcat(t, skips, cout, name, halves=True):
        up_c = t.C // 2 if halves else t.C
        up = plan.convT(t, up_c, 2, f"{name}.upsample.deconv.weight", f"{name}.upsample.deconv.bias", f"{name}.up")
        return two_conv(list(skips) + [up], cout, f"{name}.convs")

    x00 = two_conv([x], f[0], "conv_0_0")
    x10 = down(x00, f[1], "conv_1_0")
    x01 = upcat(x10, [x00], f[0], "upcat_0_1", halves=False)
    x20 = down(x10, f[2], "conv_2_0")
    x11 = upcat(x20, [x10], f[1], "upcat_1_1")
    x02 = upcat(x11, [x00, x01], f[0], "upcat_0_2", halves=False)
    x30 = down(x20, f[3], "conv_3_0")
    x21 = upcat(x30, [x20], f[2], "upcat_2_1")
    x12 = upcat(x21, [x10, x11], f[1], "upcat_1_2")
    x03 = upcat(x12, [x00, x01, x02], f[0], "upcat_0_3", halves=False)
    # process_level_3 (shared weights) is applied to pool(x_3_0) AND pool(x_3_1) (MTUNetPlusPlus.py:79,128): planned ONCE over their
    # batch concatenation (InstanceNorm is per sample: exact).  The two pooled tensors are the halves of one 2N-image tensor; conv_4_0
    # reads the first half (x_3_0 is pooled once, for both of its consumers)
    pl3_in, (p30, p31) = plan.batch_pair("process_level_3.in", f[3], x30.H // 2, x30.W // 2)
    x40 = two_conv([plan.maxpool(x30, "conv_4_0.pool", out=p30)], f[4], "conv_4_0.convs")
    x31 = upcat(x40, [x30], f[3], "upcat_3_1")
    x22 = upcat(x31, [x20, x21], f[2], "upcat_2_2")
    x13 = upcat(x22, [x10, x11, x12], f[1], "upcat_1_3")
    x04 = upcat(x13, [x00, x01, x02, x03], f[5], "upcat_0_4", halves=False)
    regions = plan.pv("final_conv_0_1.weight").shape[0]
    outs = [plan.conv1x1(t, regions, f"final_conv_0_{j}.weight", f"final_conv_0_{j}.bias", f"final_conv_0_{j}")
            for j, t in ((1, x01), (2, x02), (3, x03), (4, x04))]
    plan.maxpool(x31, "process_level_3.poolb", out=p31)
    pa, pb = plan.split_batch(two_conv([pl3_in], f[4], "process_level_3.convs"))      # shared weights, both applications in one pass (F10)
    feat = two_conv([pa, x40, pb], 512, "classifier.0")
    g = plan.gap(feat, "gap")
    h = plan.linear(g, 256, "classifier.3.weight", "classifier.3.bias", True, "fc1")
    n_cls = plan.pv("classifier.5.weight").shape[0]
    logits = plan.linear(h, n_cls, "classifier.5.weight", "classifier.5.bias", False, "logits")
    return logits, outs


# ----------------------------------------------------------------------------------------------
# nn.Module shell
# ----------------------------------------------------------------------------------------------
class _Node(nn.Module):
    """Bare container used to give parameters the reference's dotted names."""


class CompiledStep:
    """Everything built for one (N, H, W): arena, programs, I/O buffers."""

    def __init__(self, net: "HipMultiTaskNet", N: int, H: int, W: int, fused_loss: Optional[dict] = None):
        dev = net.flat_p.device
        self.N, self.H, self.W = N, H, W
        plan = StepPlan(dev, N, net._param_view, net._grad_view, net.slots, force_direct=net.force_direct,
                        compute=net.compute, coop_reserve_cus=net.coop_reserve_cus, coop_state=net._coop_state)
        self.x = Act("input", plan.alloc(N, net.in_channels, H, W), needs_grad=False)
        self.logits, self.segs = net._graph(plan, self.x)
        self.mask = self.onehot = None
        if fused_loss is not None:
            self.mask = plan.alloc(N, self.segs[0].C, H, W)
            self.onehot = plan.alloc(N, self.logits.C)
            heads = self.segs if net.deep_supervision_outputs else self.segs[-1:]
            self.loss_scale = float(net.loss_scale)
            plan.fused_losses(heads, self.logits, self.mask, self.onehot, fused_loss["alpha"],
                              fused_loss["inversely_weighted"], fused_loss.get("focal_weight"), loss_scale=self.loss_scale,
                              binary=bool(fused_loss.get("binary")), cls_gamma=float(fused_loss.get("cls_gamma", 2.0)))
            self.grad_weight = plan.grad_weight
        else:
            for h in (self.segs if net.deep_supervision_outputs else self.segs[-1:]):
                plan.grad_of(h)
                h.grad_written = True
            plan.grad_of(self.logits)
            self.logits.grad_written = True
        plan.emit_backward()
        self.programs = plan.finalize()
        self.plan = plan
        self.param_ptr = net.flat_p.data_ptr()
        # gradient-readiness order for bucketed all-reduce: (ready_at, name)
        # (parameters no backward op writes -- e.g. the unused deep-supervision heads when deep_supervision is
        # off -- keep a zero gradient, which Adam turns into a zero update, same as torch skipping grad=None)
        self.ready = sorted((max(s.ready_at, 0), s.name) for s in net.slots.values())
        # ... snapshotted per compiled step: net.slots is plan-time scratch that the NEXT StepPlan resets and rewrites
        # (another batch size, an evaluation step), and backward op indices are shape-dependent
        self.slot_ready = [(net.slots[n].offset, net.slots[n].numel, max(net.slots[n].ready_at, 0)) for n in net._order]
        self.buckets = None                     # trainer.plan_buckets(...) of this step, built on first use


class HipMultiTaskNet(nn.Module):
    architecture = ""

    def __init__(self, named_params: List[Tuple[str, torch.Tensor]], in_channels: int, deep_supervision: bool,
                 graph: Callable, force_direct: bool = False):
        super().__init__()
        self.compute = 0          # 3x3-conv MFMA operand type: 0 fp32 (reference arithmetic), 1 bf16, 2 fp16; set_compute()
        self.loss_scale = 1.0     # fused step only: dL is multiplied by this, Adam's grad_scale divides it out (fp16: 65536)
        self.coop_reserve_cus = 0  # CUs the cooperative InstanceNorm grids leave free (set by a data-parallel FusedTrainStep)
        self.in_channels = in_channels
        self.deep_supervision = deep_supervision
        self.deep_supervision_outputs = deep_supervision
        self._graph = graph
        self.force_direct = force_direct
        self.slots: Dict[str, ParamSlot] = {}
        self._order: List[str] = []
        off = 0
        for name, t in named_params:
            self.slots[name] = ParamSlot(name, tuple(t.shape), off)
            self._order.append(name)
            off += (t.numel() + 3) // 4 * 4          # 16-byte aligned slots (float4 Adam, 16-B weight loads)
            node = self
            parts = name.split(".")
            for part in parts[:-1]:
                if not hasattr(node, part):
                    node.add_module(part, _Node())
                node = getattr(node, part)
            node.register_parameter(parts[-1], nn.Parameter(t.clone().float().contiguous()))
        self.flat_numel = off
        self.flat_p: Optional[torch.Tensor] = None
        self.flat_g: Optional[torch.Tensor] = None
        self._steps: Dict[Tuple, CompiledStep] = {}

    def set_compute(self, dtype) -> "HipMultiTaskNet":
        """'f32' | 'bf16' | 'f16' (or 0/1/2): operand type of the conv3x3 MFMAs.  Master weights, activations in HBM,
        accumulation, norm, losses and Adam stay fp32 in every mode."""
        table = {"f32": 0, "fp32": 0, "bf16": 1, "f16": 2, "fp16": 2, 0: 0, 1: 1, 2: 2}
        if dtype not in table:
            raise ValueError(f"unknown compute dtype {dtype!r}")
        self.compute = table[dtype]
        # fp16 operands: a static loss scale keeps the back-propagated dz out of fp16's subnormals when the MFMA operands are rounded (storage of the
        # sums and accumulation are fp32).  2^16 since round 4 (2^12 before): the design's gradient error against the exact gradient, tensor by tensor
        # (tests/studies/design_error_cpu.py, profiles/r04_fp16_design_error.txt): at initialisation the scale does not matter, but dz shrinks as
        # training goes on -- after 1500 steps 2^12 leaves 13 - 23 % error in the gradients of conv_4_0 (bf16 mode: 1.5 - 3.4 %) and a median of 0.65 %
        # over all tensors, 2^16 leaves 0.7 - 2.3 % and 0.07 % (2^20: 0.3 - 2.0 %, 0.07 %: nothing left to gain, and 16 x less room below 65504).
        # No overflow in 4 x 12000 training steps on the hard task at 2^16 (an inf would reach the NaN guard).  The real fix is a dynamic scale
        # with a found-inf skip in the Adam launch.
        self.loss_scale = 65536.0 if self.compute == 2 else 1.0
        self._steps.clear()
        return self

    # ---- flat storage -------------------------------------------------------------------------
    def _named(self) -> Dict[str, nn.Parameter]:
        return dict(self.named_parameters())

    def _param_view(self, name: str) -> torch.Tensor:
        s = self.slots[name]
        return self.flat_p[s.offset:s.offset + s.numel].view(s.shape)

    def _grad_view(self, name: str) -> torch.Tensor:
        s = self.slots[name]
        return self.flat_g[s.offset:s.offset + s.numel].view(s.shape)

    def ensure_flat(self) -> None:
        """(Re)build the flat buffers when parameters were moved (.to(dev)) or replaced; idempotent."""
        params = self._named()
        p0 = params[self._order[0]]
        dev = p0.device
        if dev.type != "cuda":
            L.require_gpu()
            raise L.MtbcError("model parameters are on the CPU: call model.to('cuda:0') first (no CPU path)")
        ok = self.flat_p is not None and self.flat_p.device == dev
        if ok:
            for name in self._order:
                if params[name].data_ptr() != self.flat_p.data_ptr() + 4 * self.slots[name].offset:
                    ok = False
                    break
        if ok:
            return
        flat = torch.zeros(self.flat_numel, dtype=torch.float32, device=dev)
        for name in self._order:
            s = self.slots[name]
            flat[s.offset:s.offset + s.numel].copy_(params[name].detach().reshape(-1))
        self.flat_p = flat
        self.flat_g = torch.zeros_like(flat)
        for name in self._order:
            params[name].data = self._param_view(name)
        self._steps.clear()

    def grads_as_views(self) -> None:
        """Point every p.grad at its slot of the flat gradient buffer (fused-step path)."""
        params = self._named()
        for name in self._order:
            params[name].grad = self._grad_view(name)

    def _coop_state(self) -> torch.Tensor:
        """The cooperative InstanceNorm kernels' mailbox / error-word block of THIS model on its current device: every
        compiled step shares it, so a mailbox timeout in any of them (the short last batch, an evaluation batch of
        another size) is seen by the next `FusedTrainStep.check_nan()` / `FusedEvalStep.result()` whatever ran since."""
        dev = self.flat_p.device
        buf = getattr(self, "_coop_buf", None)
        if buf is None or buf.device != dev:
            buf = torch.zeros(L.load().mtbc_instnorm_coop_state_bytes() // 4, dtype=torch.int32, device=dev)
            self._coop_buf = buf
        return buf

    def coop_error_word(self) -> Optional[torch.Tensor]:
        """Device view (1 x int32) of the model's sticky cooperative-kernel error word (None: no plan has used one)."""
        buf = getattr(self, "_coop_buf", None)
        if buf is None:
            return None
        i = L.load().mtbc_instnorm_coop_error_offset() // 4
        return buf[i:i + 1]

    # ---- compiled steps ---------------------------------------------------------------------
    def compiled(self, N: int, H: int, W: int, fused_loss: Optional[dict] = None) -> CompiledStep:
        self.ensure_flat()
        fkey = None
        if fused_loss is not None:
            fw = fused_loss.get("focal_weight")
            if fw is not None:
                # its raw pointer is baked into the FOCAL op: validate once, key on identity, keep it alive with the plan
                if fw.dtype != torch.float32 or not fw.is_contiguous() or fw.device != self.flat_p.device or fw.numel() != self.n_classes:
                    raise ValueError("focal_weight must be a contiguous float32 tensor of n_classes values on the model's device")
            fkey = (fused_loss["alpha"], fused_loss["inversely_weighted"], self.loss_scale,
                    None if fw is None else (fw.data_ptr(), fw._version), bool(fused_loss.get("binary")), float(fused_loss.get("cls_gamma", 2.0)))
        key = (N, H, W, self.compute, self.coop_reserve_cus, fkey)
        st = self._steps.get(key)
        if st is None or st.param_ptr != self.flat_p.data_ptr():
            st = CompiledStep(self, N, H, W, fused_loss)
            self._steps[key] = st
        return st

    # ---- drop-in forward (autograd-visible) ---------------------------------------------------
    def forward(self, x: torch.Tensor):
        if x.device.type != "cuda":
            L.require_gpu()
            raise L.MtbcError("input is on the CPU: the HIP path needs device tensors")
        if x.dim() != 4 or x.shape[1] != self.in_channels:
            raise ValueError(f"expected (N,{self.in_channels},H,W) input, got {tuple(x.shape)}")
        st = self.compiled(x.shape[0], x.shape[2], x.shape[3])
        params = [self._named()[n] for n in self._order]
        outs = _NetFunction.apply(self, st, x, *params)
        logits, segs = outs[0], list(outs[1:])
        if self.deep_supervision_outputs:
            return [logits], segs
        return logits, segs[-1]


class _NetFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net: HipMultiTaskNet, st: CompiledStep, x: torch.Tensor, *params):
        st.x.data.copy_(x.detach().to(torch.float32))
        st.programs["pack"].run()
        st.programs["fwd"].run()
        ctx.net, ctx.st = net, st
        logits = st.logits.data.view(st.N, -1).clone()
        segs = [s.data.clone() for s in st.segs]
        return (logits, *segs)

    @staticmethod
    def backward(ctx, dlogits, *dsegs):
        net, st = ctx.net, ctx.st
        heads = st.segs if net.deep_supervision_outputs else st.segs[-1:]
        dhead = dsegs if net.deep_supervision_outputs else dsegs[-1:]
        for h, g in zip(heads, dhead):
            if g is None:
                h.grad.zero_()
            else:
                h.grad.copy_(g)
        if dlogits is None:
            st.logits.grad.zero_()
        else:
            st.logits.grad.view(st.N, -1).copy_(dlogits)
        st.programs["bwd"].run()
        grads = [net._grad_view(n) for n in net._order]
        return (None, None, None, *grads)


class MTnnUNet(HipMultiTaskNet):
    """src/models/multitask/MTnnUNet.py:64-183 -- always returns lists (`:183`)."""
    architecture = "MTnnUNet"

    def __init__(self, sequences: int, regions: int, n_classes: int = 3, force_direct: bool = False):
        ncls = 1 if n_classes == 2 else n_classes
        super().__init__(_mtnnunet_params(sequences, regions, ncls), sequences, True, _graph_mtnnunet, force_direct)
        self.n_classes = ncls


class MTUNetPlusPlus(HipMultiTaskNet):
    """src/models/multitask/MTUNetPlusPlus.py:11-136 (MONAI 1.3.0 TwoConv / Down / UpCat semantics)."""
    architecture = "MTUNetPlusPlus"

    def __init__(self, spatial_dims: int = 2, in_channels: int = 1, out_channels: int = 1, n_classes: int = 3,
                 deep_supervision: bool = False, force_direct: bool = False):
        if spatial_dims != 2:
            raise ValueError("only spatial_dims=2 is on the hot path")
        ncls = 1 if n_classes == 2 else n_classes
        super().__init__(_unetpp_params(in_channels, out_channels, ncls), in_channels, deep_supervision,
                         _graph_unetpp, force_direct)
        self.n_classes = ncls

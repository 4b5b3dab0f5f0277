"""Pure-PyTorch (CPU, fp32) restatement of the reference hot path.

TEST INFRASTRUCTURE -- see oracle/__init__.py.  Every function cites the
reference file:line it follows (paths relative to the upstream repository).

The two networks are described by small tables and assembled from stock
torch.nn layers, keeping the reference's module *names* (so state_dicts
interchange) and its construction *order* (so that seeding the global torch
RNG yields the same initial weights).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import List, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------
# MTnnUNet   (src/models/multitask/MTnnUNet.py:19-183)
# --------------------------------------------------------------------------

NNUNET_WIDTHS = (32, 64, 128, 256, 320)      # MTnnUNet.py:72


def _cell(cin: int, cout: int) -> nn.Sequential:
    """conv3x3(pad 1, no bias) -> InstanceNorm2d(eps 1e-5, no affine) -> LeakyReLU(0.01).
    MTnnUNet.py:12-16 (conv factory), :30-38 (cell)."""
    return nn.Sequential(OrderedDict(
        Conv=nn.Conv2d(cin, cout, kernel_size=3, padding=1, bias=False),
        InNorm=nn.InstanceNorm2d(cout),
        LeReLU=nn.LeakyReLU(inplace=True),
    ))


def _level(cin: int, cmid: int, cout: int) -> nn.Sequential:
    """Two stacked cells.  MTnnUNet.py:42-61."""
    return nn.Sequential(OrderedDict(
        ConvInNormLRelu1=_cell(cin, cmid),
        ConvInNormLRelu2=_cell(cmid, cout),
    ))


class OracleMTnnUNet(nn.Module):
    """nnU-Net style 5-level encoder/decoder + pooled classification head."""

    def __init__(self, sequences: int = 1, regions: int = 1, n_classes: int = 3):
        super().__init__()
        w = NNUNET_WIDTHS
        self.n_classes = 1 if n_classes == 2 else n_classes           # :74-76
        # creation order == MTnnUNet.py:79-118
        enc_io = [(sequences, w[0]), (w[0], w[1]), (w[1], w[2]), (w[2], w[3]), (w[3], w[4])]
        for i, (a, b) in enumerate(enc_io, start=1):
            setattr(self, f"encoder{i}", _level(a, b, b))
        self.bottleneck = _level(w[4], w[4], w[4])
        dec_io = {5: (2 * w[4], w[3], w[3]), 4: (2 * w[3], w[2], w[2]), 3: (2 * w[2], w[1], w[1]),
                  2: (2 * w[1], w[0], w[0]), 1: (2 * w[0], w[0], w[0] // 2)}
        for i in (5, 4, 3, 2, 1):
            setattr(self, f"decoder{i}", _level(*dec_io[i]))
        for i in (5, 4, 3, 2, 1):
            c = w[i - 1]
            setattr(self, f"upsample{i}", nn.ConvTranspose2d(c, c, kernel_size=2, stride=2))
        self.downsample = nn.MaxPool2d(2, 2)
        self.output4 = nn.Sequential(nn.ConvTranspose2d(w[2], w[2], kernel_size=8, stride=8),
                                     nn.Conv2d(w[2], regions, kernel_size=1))
        self.output3 = nn.Sequential(nn.ConvTranspose2d(w[1], w[1], kernel_size=4, stride=4),
                                     nn.Conv2d(w[1], regions, kernel_size=1))
        self.output2 = nn.Sequential(nn.ConvTranspose2d(w[0], w[0], kernel_size=2, stride=2),
                                     nn.Conv2d(w[0], regions, kernel_size=1))
        self.output1 = nn.Conv2d(w[0] // 2, regions, kernel_size=1)

        # MTnnUNet.py:120 + :134-140 -- kaiming-normal over every Conv2d that
        # exists *so far*; the classification head built below keeps torch's
        # default init.
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, nonlinearity="leaky_relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

        # MTnnUNet.py:123-132
        self.process_encoder_5 = _cell(w[4], w[4])
        self.process_decoder_5 = _cell(w[3], w[4])
        self.classifier = nn.Sequential(
            _cell(3 * w[4], 512), nn.AdaptiveAvgPool2d(1), nn.Flatten(),
            nn.Linear(512, 256), nn.ReLU(), nn.Linear(256, self.n_classes))

    def forward(self, x):                                            # :142-183
        skips = []
        t = x
        for i in range(1, 6):
            e = getattr(self, f"encoder{i}")(t)
            skips.append(e)
            t = self.downsample(e)
        bott = self.bottleneck(t)
        d = bott
        dec = {}
        for i in (5, 4, 3, 2, 1):
            up = getattr(self, f"upsample{i}")(d)
            d = getattr(self, f"decoder{i}")(torch.cat([skips[i - 1], up], dim=1))
            dec[i] = d
        feats = torch.cat([self.process_encoder_5(skips[4]),
                           self.upsample5(bott),                    # second use, F10
                           self.process_decoder_5(dec[5])], dim=1)
        logits = self.classifier(feats)
        outs = [self.output4(dec[4]), self.output3(dec[3]), self.output2(dec[2]), self.output1(dec[1])]
        return [logits], outs


class OracleSegNnUNet(nn.Module):
    """Single-task segmentation net of BASELINE.json configs[0] (the CPU plumbing case): `nnUNet2021`
    (src/models/segmentation/nnUNet.py:64-168) -- the encoder / decoder / deep-supervision heads of MTnnUNet without the
    classification head; `weights_initialization` (:124-130) runs after every module exists, so ALL Conv2d weights
    (1x1 heads included) are kaiming-normal and their biases zero.  Returns the list [out4, out3, out2, out1] (:163).
    Pinned: tests/golden/seg_nnunet_step.npz (oracle/make_goldens.py imports the reference's class)."""

    def __init__(self, sequences: int = 1, regions: int = 1):
        super().__init__()
        w = NNUNET_WIDTHS
        enc_io = [(sequences, w[0]), (w[0], w[1]), (w[1], w[2]), (w[2], w[3]), (w[3], w[4])]
        for i, (a, b) in enumerate(enc_io, start=1):
            setattr(self, f"encoder{i}", _level(a, b, b))
        self.bottleneck = _level(w[4], w[4], w[4])
        dec_io = {5: (2 * w[4], w[3], w[3]), 4: (2 * w[3], w[2], w[2]), 3: (2 * w[2], w[1], w[1]),
                  2: (2 * w[1], w[0], w[0]), 1: (2 * w[0], w[0], w[0] // 2)}
        for i in (5, 4, 3, 2, 1):
            setattr(self, f"decoder{i}", _level(*dec_io[i]))
        for i in (5, 4, 3, 2, 1):
            c = w[i - 1]
            setattr(self, f"upsample{i}", nn.ConvTranspose2d(c, c, kernel_size=2, stride=2))
        self.downsample = nn.MaxPool2d(2, 2)
        self.output4 = nn.Sequential(nn.ConvTranspose2d(w[2], w[2], kernel_size=8, stride=8),
                                     nn.Conv2d(w[2], regions, kernel_size=1))
        self.output3 = nn.Sequential(nn.ConvTranspose2d(w[1], w[1], kernel_size=4, stride=4),
                                     nn.Conv2d(w[1], regions, kernel_size=1))
        self.output2 = nn.Sequential(nn.ConvTranspose2d(w[0], w[0], kernel_size=2, stride=2),
                                     nn.Conv2d(w[0], regions, kernel_size=1))
        self.output1 = nn.Conv2d(w[0] // 2, regions, kernel_size=1)
        for m in self.modules():                                      # nnUNet.py:124-130
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, nonlinearity="leaky_relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x):                                             # nnUNet.py:133-165
        skips, t = [], x
        for i in range(1, 6):
            e = getattr(self, f"encoder{i}")(t)
            skips.append(e)
            t = self.downsample(e)
        d = self.bottleneck(t)
        dec = {}
        for i in (5, 4, 3, 2, 1):
            d = getattr(self, f"decoder{i}")(torch.cat([skips[i - 1], getattr(self, f"upsample{i}")(d)], dim=1))
            dec[i] = d
        return [self.output4(dec[4]), self.output3(dec[3]), self.output2(dec[2]), self.output1(dec[1])]


def seg_train_step(model: nn.Module, optimizer, image: torch.Tensor, mask: torch.Tensor, inversely_weighted: bool = True):
    """One step of src/training_segmentation.py:29-62: zero_grad -> model -> apply_criterion_binary_segmentation
    (src/utils/criterions.py:27-49: deep-supervision heads reversed, 1/(n+1) weights when inversely_weighted) -> backward
    -> Adam.  Returns (loss, batch Dice of sigmoid(last head) > .5)."""
    optimizer.zero_grad(set_to_none=True)
    outputs = model(image)
    if isinstance(outputs, (list, tuple)):
        terms = [dice_loss_sigmoid_sq(s, mask) / ((j + 1) if inversely_weighted else 1) for j, s in enumerate(reversed(list(outputs)))]
        loss = torch.sum(torch.stack(terms))
        last = outputs[-1]
    else:
        loss, last = dice_loss_sigmoid_sq(outputs, mask), outputs
    if torch.isnan(loss):
        raise SystemExit(1)
    loss.backward()
    optimizer.step()
    return loss.detach(), dice_score_from_tensor(mask, torch.sigmoid(last.detach()) > .5)


# --------------------------------------------------------------------------
# MTUNetPlusPlus (src/models/multitask/MTUNetPlusPlus.py:12-136) on MONAI 1.3.0
# blocks (monai/networks/nets/basic_unet.py TwoConv/Down/UpCat,
# monai/networks/blocks/convolutions.py Convolution + ADN "NDA",
# monai/networks/blocks/upsample.py UpSample mode "deconv").  PARITY UNPINNED.
# --------------------------------------------------------------------------

UNETPP_FEATURES = (24, 48, 96, 192, 384, 24)   # MTUNetPlusPlus.py:18
UNETPP_SLOPE = 0.1                             # MTUNetPlusPlus.py:20


def _monai_convolution(cin: int, cout: int) -> nn.Sequential:
    """MONAI Convolution(k3, s1, pad1, bias) + ADN ordering N-D-A:
    InstanceNorm(affine) -> Dropout(p=0) -> LeakyReLU(0.1)."""
    adn = nn.Sequential(OrderedDict(
        N=nn.InstanceNorm2d(cout, affine=True),
        D=nn.Dropout(0.0),
        A=nn.LeakyReLU(negative_slope=UNETPP_SLOPE, inplace=True),
    ))
    return nn.Sequential(OrderedDict(
        conv=nn.Conv2d(cin, cout, kernel_size=3, stride=1, padding=1, bias=True),
        adn=adn,
    ))


def _two_conv(cin: int, cout: int) -> nn.Sequential:
    return nn.Sequential(OrderedDict(conv_0=_monai_convolution(cin, cout),
                                     conv_1=_monai_convolution(cout, cout)))


def _down(cin: int, cout: int) -> nn.Sequential:
    return nn.Sequential(OrderedDict(max_pooling=nn.MaxPool2d(kernel_size=2),
                                     convs=_two_conv(cin, cout)))


class _UpCat(nn.Module):
    """MONAI UpCat with upsample='deconv': ConvTranspose(k2,s2,bias) then
    cat([skip, up]) then TwoConv.  Channel order: skip first."""

    def __init__(self, in_chns: int, cat_chns: int, out_chns: int, halves: bool = True):
        super().__init__()
        up_chns = in_chns // 2 if halves else in_chns
        self.upsample = nn.Sequential(OrderedDict(
            deconv=nn.ConvTranspose2d(in_chns, up_chns, kernel_size=2, stride=2, bias=True)))
        self.convs = _two_conv(cat_chns + up_chns, out_chns)

    def forward(self, x, x_e):
        x_0 = self.upsample(x)
        # MONAI replicate-pads odd sizes; all hot-path sizes are even so the
        # branch never fires -- guarded here so misuse is loud, not silent.
        if x_e.shape[-2:] != x_0.shape[-2:]:
            raise ValueError("oracle UpCat: spatial sizes must match (H, W % 16 == 0)")
        return self.convs(torch.cat([x_e, x_0], dim=1))


class OracleMTUNetPlusPlus(nn.Module):
    def __init__(self, in_channels: int = 1, out_channels: int = 1, n_classes: int = 3,
                 deep_supervision: bool = False, features: Sequence[int] = UNETPP_FEATURES):
        super().__init__()
        self.deep_supervision = deep_supervision
        self.n_classes = 1 if n_classes == 2 else n_classes
        f = tuple(features)
        # creation order == MTUNetPlusPlus.py:47-87
        self.conv_0_0 = _two_conv(in_channels, f[0])
        self.conv_1_0 = _down(f[0], f[1])
        self.conv_2_0 = _down(f[1], f[2])
        self.conv_3_0 = _down(f[2], f[3])
        self.conv_4_0 = _down(f[3], f[4])
        self.upcat_0_1 = _UpCat(f[1], f[0], f[0], halves=False)
        self.upcat_1_1 = _UpCat(f[2], f[1], f[1])
        self.upcat_2_1 = _UpCat(f[3], f[2], f[2])
        self.upcat_3_1 = _UpCat(f[4], f[3], f[3])
        self.upcat_0_2 = _UpCat(f[1], f[0] * 2, f[0], halves=False)
        self.upcat_1_2 = _UpCat(f[2], f[1] * 2, f[1])
        self.upcat_2_2 = _UpCat(f[3], f[2] * 2, f[2])
        self.upcat_0_3 = _UpCat(f[1], f[0] * 3, f[0], halves=False)
        self.upcat_1_3 = _UpCat(f[2], f[1] * 3, f[1])
        self.upcat_0_4 = _UpCat(f[1], f[0] * 4, f[5], halves=False)
        self.final_conv_0_1 = nn.Conv2d(f[0], out_channels, kernel_size=1)
        self.final_conv_0_2 = nn.Conv2d(f[0], out_channels, kernel_size=1)
        self.final_conv_0_3 = nn.Conv2d(f[0], out_channels, kernel_size=1)
        self.final_conv_0_4 = nn.Conv2d(f[5], out_channels, kernel_size=1)
        self.process_level_3 = _down(f[3], f[4])
        self.classifier = nn.Sequential(
            _two_conv(f[4] * 3, 512), nn.AdaptiveAvgPool2d(1), nn.Flatten(),
            nn.Linear(512, 256), nn.ReLU(), nn.Linear(256, self.n_classes))

    def forward(self, x):                                            # :101-136
        cat = lambda *ts: torch.cat(ts, dim=1)
        x00 = self.conv_0_0(x)
        x10 = self.conv_1_0(x00)
        x01 = self.upcat_0_1(x10, x00)
        x20 = self.conv_2_0(x10)
        x11 = self.upcat_1_1(x20, x10)
        x02 = self.upcat_0_2(x11, cat(x00, x01))
        x30 = self.conv_3_0(x20)
        x21 = self.upcat_2_1(x30, x20)
        x12 = self.upcat_1_2(x21, cat(x10, x11))
        x03 = self.upcat_0_3(x12, cat(x00, x01, x02))
        x40 = self.conv_4_0(x30)
        x31 = self.upcat_3_1(x40, x30)
        x22 = self.upcat_2_2(x31, cat(x20, x21))
        x13 = self.upcat_1_3(x22, cat(x10, x11, x12))
        x04 = self.upcat_0_4(x13, cat(x00, x01, x02, x03))
        o1, o2, o3, o4 = (self.final_conv_0_1(x01), self.final_conv_0_2(x02),
                          self.final_conv_0_3(x03), self.final_conv_0_4(x04))
        feats = cat(self.process_level_3(x30), x40, self.process_level_3(x31))   # shared weights, F10
        logits = self.classifier(feats)
        if self.deep_supervision:
            return [logits], [o1, o2, o3, o4]
        return logits, o4


def build_oracle_model(architecture: str, sequences: int = 1, regions: int = 1, n_classes: int = 3,
                       deep_supervision: bool = True) -> nn.Module:
    """src/utils/experiment_init.py:154-159."""
    if architecture == "MTnnUNet":
        return OracleMTnnUNet(sequences, regions, n_classes)
    if architecture == "MTUNetPlusPlus":
        return OracleMTUNetPlusPlus(sequences, regions, n_classes, deep_supervision)
    raise ValueError(f"unknown architecture {architecture!r}")


# --------------------------------------------------------------------------
# Losses
# --------------------------------------------------------------------------

def dice_loss_sigmoid_sq(logits: torch.Tensor, target: torch.Tensor,
                         smooth_nr: float = 1.0, smooth_dr: float = 1.0) -> torch.Tensor:
    """MONAI 1.3.0 DiceLoss(include_background=True, sigmoid=True, squared_pred=True,
    smooth_nr=1, smooth_dr=1, reduction='mean', batch=False) as built at
    src/utils/experiment_init.py:210-211.  PARITY UNPINNED (monai absent).

    p = sigmoid(x);  per (n, c):  I = sum_hw p*t,  D = sum_hw p^2 + sum_hw t^2
    f = 1 - (2 I + nr) / (D + dr);  loss = mean_{n,c} f
    """
    p = torch.sigmoid(logits)
    dims = tuple(range(2, logits.dim()))
    inter = torch.sum(p * target, dim=dims)
    denom = torch.sum(p * p, dim=dims) + torch.sum(target * target, dim=dims)
    f = 1.0 - (2.0 * inter + smooth_nr) / (denom + smooth_dr)
    return torch.mean(f)


def focal_loss_soft(logits: torch.Tensor, targets: torch.Tensor, alpha: float = 1.0,
                    gamma: float = 2.0, weight: torch.Tensor | None = None) -> torch.Tensor:
    """src/utils/criterions.py:14-20 with reduction='mean':
    ce_i = -sum_c w_c t_ic log_softmax(x_i)_c ; pt = exp(-ce) ; mean(alpha (1-pt)^gamma ce)."""
    logp = F.log_softmax(logits, dim=1)
    if weight is not None:
        logp = logp * weight.view(1, -1)
    ce = -(targets * logp).sum(dim=1)
    pt = torch.exp(-ce)
    return torch.mean(alpha * (1.0 - pt) ** gamma * ce)


def multitask_losses(seg_outputs, mask, cls_outputs, onehot, inversely_weighted: bool = True):
    """src/utils/criterions.py:52-76 (list branch and tensor branch); NaN -> SystemExit(1)."""
    if isinstance(seg_outputs, (list, tuple)):
        seg_terms = []
        for j, s in enumerate(reversed(list(seg_outputs))):
            d = dice_loss_sigmoid_sq(s, mask)
            seg_terms.append(d / (j + 1) if inversely_weighted else d)
        seg = torch.sum(torch.stack(seg_terms))
        cls = torch.sum(torch.stack([focal_loss_soft(c, onehot) for c in reversed(list(cls_outputs))]))
    else:
        seg = dice_loss_sigmoid_sq(seg_outputs, mask)
        cls = focal_loss_soft(cls_outputs, onehot)
    if torch.isnan(seg) or torch.isnan(cls):
        raise SystemExit(1)
    return seg, cls


def dice_score_from_tensor(gt: torch.Tensor, seg: torch.Tensor):
    """src/utils/metrics.py:255-267 (whole-batch TP/FP/FN in float64)."""
    gt = gt.double()
    seg = seg.double()
    tp = torch.sum(torch.logical_and(seg, gt)).double()
    fp = torch.sum(torch.logical_and(seg, torch.logical_not(gt))).double()
    fn = torch.sum(torch.logical_and(torch.logical_not(seg), gt)).double()
    if torch.sum(gt) == 0:
        return 1 if torch.sum(seg) == 0 else 0
    return 2 * tp / (2 * tp + fp + fn)


# --------------------------------------------------------------------------
# One optimisation step  (src/training_multitask.py:82-103)
# --------------------------------------------------------------------------

def make_adam(model: nn.Module, lr: float = 1e-4) -> torch.optim.Optimizer:
    """src/utils/experiment_init.py:186-187 -- note eps=1e-4."""
    return torch.optim.Adam(model.parameters(), lr=lr, eps=1e-4)


def train_step(model: nn.Module, optimizer, image: torch.Tensor, mask: torch.Tensor,
               label: torch.Tensor, alpha: float, inversely_weighted: bool = True, n_classes: int = 3,
               loss_scale: float = 1.0):
    """Returns (total, seg, cls) python floats plus the raw outputs of the forward pass.  loss_scale != 1 mirrors the
    product's fp16 mode: the loss is scaled before backward and the gradients unscaled before the optimizer."""
    optimizer.zero_grad(set_to_none=True)
    logits, outputs = model(image)
    if n_classes == 2:
        # the binary head (MTUNetPlusPlus.py:39-41: ONE logit): training_multitask.py:83-84 leaves the label (N, 1) float, the criterion is
        # torch.nn.BCEWithLogitsLoss() (experiment_init.py:241-242), aggregated like any other (criterions.py:61-66)
        bce = torch.nn.BCEWithLogitsLoss()
        if isinstance(outputs, (list, tuple)):
            seg = torch.sum(torch.stack([dice_loss_sigmoid_sq(s_, mask) / (j + 1) if inversely_weighted else dice_loss_sigmoid_sq(s_, mask)
                                         for j, s_ in enumerate(reversed(list(outputs)))]))
        else:
            seg = dice_loss_sigmoid_sq(outputs, mask)
        tgt = label.view(-1, 1).to(image.dtype)
        cls = (torch.sum(torch.stack([bce(c, tgt) for c in reversed(list(logits))])) if isinstance(logits, (list, tuple)) else bce(logits, tgt))
        if torch.isnan(seg) or torch.isnan(cls):
            raise SystemExit(1)
    else:
        onehot = F.one_hot(label.flatten().to(torch.int64), num_classes=n_classes).to(torch.float)
        seg, cls = multitask_losses(outputs, mask, logits, onehot, inversely_weighted)
    total = alpha * seg + (1.0 - alpha) * cls
    (total * loss_scale).backward()
    if loss_scale != 1.0:
        for p in model.parameters():
            if p.grad is not None:
                p.grad.div_(loss_scale)
    optimizer.step()
    return total.detach(), seg.detach(), cls.detach(), logits, outputs


def seed_everything(seed: int) -> None:
    """src/utils/miscellany.py:78-96 (the parts that matter on CPU)."""
    import os
    import random
    import numpy as np
    os.environ["PYTHONHASHSEED"] = str(seed)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def synthetic_batch(n: int, h: int, w: int, seed: int = 0) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Curated-BUSI-shaped synthetic batch: image U[0,255) smooth speckle,
    one filled ellipse per non-normal sample, labels 0/1/2.  Deterministic in `seed`."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(n, 1, h // 8 + 1, w // 8 + 1, generator=g)
    low = F.interpolate(base, size=(h, w), mode="bilinear", align_corners=True)
    img = torch.clamp(128.0 + 48.0 * low + 32.0 * torch.randn(n, 1, h, w, generator=g), 0.0, 255.0)
    label = torch.randint(0, 3, (n, 1), generator=g).to(torch.float)
    yy = torch.arange(h).view(1, h, 1).float()
    xx = torch.arange(w).view(1, 1, w).float()
    cy = (0.25 + 0.5 * torch.rand(n, 1, 1, generator=g)) * h
    cx = (0.25 + 0.5 * torch.rand(n, 1, 1, generator=g)) * w
    ry = (0.08 + 0.17 * torch.rand(n, 1, 1, generator=g)) * h
    rx = (0.08 + 0.17 * torch.rand(n, 1, 1, generator=g)) * w
    ell = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0).float()
    mask = (ell * (label.view(n, 1, 1) != 2).float()).view(n, 1, h, w)
    return img.contiguous(), mask.contiguous(), label


# ---------------------------------------------------------------------------------------------------------------
# Emulation of the product's optional 16-bit MFMA compute modes (NOT reference behaviour: the reference would use
# torch.autocast; this restates what the HIP kernels do so the mode has an oracle of its own).  Every 3x3 conv rounds
# its MFMA operands to bf16 / fp16 (round-to-nearest-even) and accumulates in the tensor dtype: forward rounds x and
# w, dgrad rounds dy and w, wgrad rounds x and dy; everything else stays in the tensor dtype.
# Storage (round 2, what torch.autocast keeps in 16 bits too): the conv OUTPUT z of a conv cell is stored rounded (the
# InstanceNorm statistics are those of the stored values), and the gradient a conv-cell activation receives from ALL its
# 3x3 consumers is summed in the accumulation dtype and stored rounded once; what its other readers (max-pool, ConvT, 1x1
# head, average pool) send back is added to that un-rounded.
def _z16_plane_ok(H: int, W: int) -> bool:
    """Plane sizes the channel-group InstanceNorm kernels take (norm_coop.hip): one workgroup up to 64 x 64, teams of 512-thread
    workgroups above."""
    return H * W <= 4096 or (H * W) % 512 == 0


class _LowpConv3x3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, lp, zt=None):
        """zt: the type the conv output is STORED in (None: fp32, not rounded).  A type other than the operands' (fp16 storage
        of the bf16 mode) saturates at its largest finite value, like the kernel's epilogue."""
        ctx.save_for_backward(x, w)
        ctx.lp = lp
        ctx.has_b = b is not None
        r = lambda t: t.to(lp).to(t.dtype)
        z = torch.conv2d(r(x), r(w), b, 1, 1)      # not F.conv2d: that name is patched inside lowp_conv3x3
        if zt is None:
            return z
        if zt != lp and zt == torch.float16:
            z = z.clamp(-65504.0, 65504.0)
        return z.to(zt).to(z.dtype)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        r = lambda t: t.to(ctx.lp).to(t.dtype)
        dyr = r(dy)
        dx = torch.nn.grad.conv2d_input(x.shape, r(w), dyr, padding=1)
        dw = torch.nn.grad.conv2d_weight(r(x), w.shape, dyr, padding=1)
        db = dy.sum(dim=(0, 2, 3)) if ctx.has_b else None
        return dx, dw, db, None, None


class _StoreRounded(torch.autograd.Function):
    """A tensor stored in a 16-bit type: rounded in forward, straight-through in backward."""

    @staticmethod
    def forward(ctx, z, zt):
        return z.to(zt).to(z.dtype)

    @staticmethod
    def backward(ctx, g):
        return g, None


class _Detour(torch.autograd.Function):
    """Identity whose backward parks the gradient in `stash` and sends zeros upstream: how a NON-conv reader of a conv-cell
    activation (pool, ConvT, 1x1 head, average pool) keeps its contribution out of the 16-bit sum of the conv readers."""

    @staticmethod
    def forward(ctx, x, stash):
        ctx.stash = stash
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        ctx.stash.append(g)
        return torch.zeros_like(g), None


class _LowpConv1x1(torch.autograd.Function):
    """1x1 head of the 16-bit modes (c8_ops.hip): reads the stored (rounded) activation; weights, products and sums
    stay in the tensor dtype -- forward and weight gradient see r(x), the input gradient is exact."""

    @staticmethod
    def forward(ctx, x, w, b, lp):
        ctx.save_for_backward(x, w)
        ctx.lp, ctx.has_b = lp, b is not None
        return torch.conv2d(x.to(lp).to(x.dtype), w, b)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx = torch.nn.grad.conv2d_input(x.shape, w, dy)
        dw = torch.nn.grad.conv2d_weight(x.to(ctx.lp).to(x.dtype), w.shape, dy)
        db = dy.sum(dim=(0, 2, 3)) if ctx.has_b else None
        return dx, dw, db, None


class _LowpConvT2Bwd(torch.autograd.Function):
    """ConvTranspose2d(k = s = 2) of the 16-bit modes: forward MFMA on rounded x, w (fp32 accumulate + bias), backward
    MFMAs on rounded operands (dgrad: dy, w; wgrad: x, dy), bias gradient from the rounded dy -- convt2.hip."""

    @staticmethod
    def forward(ctx, x, w, b, lp, fwd_lp, bwd_lp):
        ctx.save_for_backward(x, w)
        ctx.lp, ctx.has_b, ctx.bwd_lp = lp, b is not None, bwd_lp
        r = (lambda t: t.to(lp).to(t.dtype)) if fwd_lp else (lambda t: t)
        return torch.conv_transpose2d(r(x), r(w), b, 2)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        r = (lambda t: t.to(ctx.lp).to(t.dtype)) if ctx.bwd_lp else (lambda t: t)
        dyr = r(dy)
        dx = torch.conv2d(dyr, r(w), None, 2)
        with torch.enable_grad():
            wz = torch.zeros_like(w, requires_grad=True)
            (dw,) = torch.autograd.grad(torch.conv_transpose2d(r(x).detach(), wz, None, 2), wz, dyr)
        # the bias gradient: the product hands the ConvT backward a 16-bit dy whenever the up-sampled tensor has a single
        # reader (every UpCat of the U-Net++; all but upsample5 of MTnnUNet) -> sum of the rounded values
        db = dyr.sum(dim=(0, 2, 3)) if ctx.has_b else None
        return dx, dw, db, None, None, None


class lowp_conv3x3:
    """Context manager: inside it every 3x3 / padding-1 F.conv2d (hence every nn.Conv2d of the oracle nets) runs
    through _LowpConv3x3 with operands rounded to `mode` ('bf16' | 'f16'), and the backward of every k = s = 2
    F.conv_transpose2d through _LowpConvT2Bwd.  The small consumers of a conv-cell activation read the stored (rounded)
    tensor too: F.max_pool2d(2, 2) pools r(x) (forward unchanged -- max commutes with rounding -- but the backward routes
    on the rounded values) and a 1x1 F.conv2d with <= 8 outputs runs through _LowpConv1x1.  `model`: its
    ConvTranspose2d -> 1x1 Conv2d heads stay exact (the product fuses them into one fp32 transposed conv,
    engine.convT_head)."""

    def __init__(self, mode: str, model=None, z16: bool = True, da16: bool = False, fold_partials: bool = False, z_fp16: bool = True,
                 stem16: bool = True):
        self.lp = {"bf16": torch.bfloat16, "f16": torch.float16}[mode]
        self.z16, self.da16 = z16, z16 and da16       # (the product's MTBC_NO_Z16 arm / its gathered 16-bit activation gradients: da16=True is the product's default since round 3)
        # the conv outputs are stored as fp16 in BOTH modes (bf16 mode: same bytes, 11 instead of 8 significant bits; the
        # MTBC_Z_BF16 arm stores bf16)
        self.zt = torch.float16 if (z_fp16 or self.lp == torch.float16) else self.lp
        # False (the product's default): the other readers' fp32 partial gradient is added, un-rounded, to the rounded sum of the
        # 3x3 consumers (inside the InstanceNorm backward); True (the MTBC_EPI_BSTATS arm): the gathered dgrad's epilogue adds
        # it BEFORE the one rounding
        self.fold = fold_partials
        self.stem16 = stem16       # the 1-channel stem stores its conv output in 16 bits too (MTBC_NO_STEM16 arm: False)
        self.exempt = set()
        models = [] if model is None else (list(model) if isinstance(model, (list, tuple)) else [model])
        for mod in models:
            for m in mod.modules():
                if (isinstance(m, torch.nn.Sequential) and len(m) >= 2 and isinstance(m[0], torch.nn.ConvTranspose2d)
                            and isinstance(m[1], torch.nn.Conv2d) and tuple(m[1].kernel_size) == (1, 1)):
                        self.exempt.add(id(m[0].weight))
                        self.exempt.add(id(m[1].weight))

    def __enter__(self):
        self._orig = F.conv2d
        self._orig_t = F.conv_transpose2d
        self._orig_p = F.max_pool2d
        orig, lp = self._orig, self.lp
        orig_t, exempt = self._orig_t, self.exempt
        orig_p = self._orig_p
        self._orig_in, self._orig_lr, self._orig_ap, self._orig_do = F.instance_norm, F.leaky_relu, F.adaptive_avg_pool2d, F.dropout
        orig_in, orig_lr, orig_ap, orig_do = self._orig_in, self._orig_lr, self._orig_ap, self._orig_do
        z16, da16, fold, zt, stem16 = self.z16, self.da16, self.fold, self.zt, self.stem16

        def detour(t):
            """a conv-cell activation on its way into a non-conv reader"""
            stash = getattr(t, "_mtbc_stash", None)
            return _Detour.apply(t, stash) if stash is not None else t

        def instance_norm(input, *a, **k):
            out = orig_in(input, *a, **k)
            if getattr(input, "_mtbc_z16", False):
                out._mtbc_z16 = True
            return out

        def dropout(input, p=0.5, training=True, inplace=False):      # MONAI's ADN puts Dropout(0) between the norm and the activation
            out = orig_do(input, p, training, inplace)
            if getattr(input, "_mtbc_z16", False) and p == 0.0:
                out._mtbc_z16 = True
            return out

        def leaky_relu(input, negative_slope=0.01, inplace=False):
            flagged = getattr(input, "_mtbc_z16", False)
            out = orig_lr(input, negative_slope, inplace)
            if flagged and da16 and out.requires_grad:
                stash = []
                out._mtbc_stash = stash
                # fires once every reader has sent its gradient: `g` holds the 3x3 convs' sum (the detours sent zeros)
                rr = lambda t: t.to(lp).to(t.dtype)

                def hook(g, stash=stash):
                    if not stash:
                        return rr(g)
                    if not bool(g.any()):           # no 3x3 consumer (the detours sent zeros): no gathered launch, the gradient stays as it is
                        return sum(stash)
                    return rr(g + sum(stash)) if fold else rr(g) + sum(stash)

                out.register_hook(hook)
            return out

        def adaptive_avg_pool2d(input, output_size):
            return orig_ap(detour(input), output_size)

        F.instance_norm, F.leaky_relu, F.adaptive_avg_pool2d, F.dropout = instance_norm, leaky_relu, adaptive_avg_pool2d, dropout

        def max_pool2d(input, kernel_size, stride=None, padding=0, dilation=1, ceil_mode=False, return_indices=False):
            # same eligibility as the product (engine.maxpool): channel groups of 8, and a pooled map the 3x3 convs take
            # in the channel-blocked layout
            H, W = input.shape[-2:]
            if (kernel_size in (2, (2, 2)) and stride in (None, 2, (2, 2)) and padding in (0, (0, 0)) and not return_indices
                    and input.dim() == 4 and input.shape[1] % 8 == 0 and (H * W) % 4 == 0 and H % 2 == 0 and W % 2 == 0
                    and (W // 2) % 4 == 0 and H // 2 >= 8 and W // 2 >= 8):
                input = detour(input).to(lp).to(input.dtype)
            else:
                input = detour(input)
            return orig_p(input, kernel_size, stride, padding, dilation, ceil_mode, return_indices)

        F.max_pool2d = max_pool2d

        def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):
            # same eligibility as the product (engine.convT): the direct-to-fragment backward kernels of convt2.hip
            H, W = input.shape[-2:]
            input = detour(input)
            if (tuple(weight.shape[-2:]) == (2, 2) and stride in (2, (2, 2)) and padding in (0, (0, 0)) and groups == 1
                    and (H * W) % 32 == 0 and id(weight) not in exempt):
                fwd_lp = weight.shape[0] % 8 == 0 and weight.shape[1] % 8 == 0      # forward: 16-bit MFMA on channel-blocked x
                bwd_lp = W % 8 == 0 and weight.shape[1] % 2 == 0
                if fwd_lp or bwd_lp:
                    return _LowpConvT2Bwd.apply(input, weight, bias, lp, fwd_lp, bwd_lp)
            return orig_t(input, weight, bias, stride, padding, output_padding, groups, dilation)

        F.conv_transpose2d = conv_transpose2d

        def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
            # same eligibility as the product's MFMA path (conv3x3.hip mfma_ok): maps >= 8x8 with W % 4 == 0 and
            # channel counts in multiples of 8; everything else (the 1-channel stem, 4x4 maps) stays in fp32 there
            H, W = input.shape[-2:]
            if (tuple(weight.shape[-2:]) == (3, 3) and padding in (1, (1, 1)) and stride in (1, (1, 1)) and groups == 1
                    and H >= 8 and W >= 8 and W % 4 == 0 and weight.shape[1] % 8 == 0 and weight.shape[0] % 8 == 0):
                cell = z16 and _z16_plane_ok(H, W)
                out = _LowpConv3x3.apply(input, weight, bias, lp, zt if cell else None)
                if cell:
                    out._mtbc_z16 = True      # InstanceNorm + LeakyReLU behind it make a conv-cell activation
                return out
            if (tuple(weight.shape[-2:]) == (3, 3) and padding in (1, (1, 1)) and stride in (1, (1, 1)) and groups == 1 and z16 and stem16
                    and weight.shape[1] == 1 and weight.shape[0] % 8 == 0 and H >= 8 and W >= 8 and W % 4 == 0 and _z16_plane_ok(H, W)):
                # the stem: exact fp32 operands (a 1-channel image has no 16-bit operand tensor), the output stored like every other
                z = orig(input, weight, bias, stride, padding, dilation, groups)
                if zt == torch.float16 and lp != torch.float16:
                    z = z.clamp(-65504.0, 65504.0)
                out = _StoreRounded.apply(z, zt)
                out._mtbc_z16 = True
                return out
            if tuple(weight.shape[-2:]) == (1, 1):
                input = detour(input)
            if (tuple(weight.shape[-2:]) == (1, 1) and padding in (0, (0, 0)) and stride in (1, (1, 1)) and groups == 1
                    and weight.shape[1] % 8 == 0 and weight.shape[0] <= 8 and (H * W) % 4 == 0 and id(weight) not in exempt):
                return _LowpConv1x1.apply(input, weight, bias, lp)
            return orig(input, weight, bias, stride, padding, dilation, groups)

        F.conv2d = conv2d
        return self

    def __exit__(self, *exc):
        F.conv2d = self._orig
        F.conv_transpose2d = self._orig_t
        F.max_pool2d = self._orig_p
        F.instance_norm, F.leaky_relu, F.adaptive_avg_pool2d, F.dropout = self._orig_in, self._orig_lr, self._orig_ap, self._orig_do
        return False


# ---------------------------------------------------------------------------------------------------------------
# Augmentation oracle: torchvision is absent here, so this restates what
# torchvision.transforms.functional.{hflip, vflip, rotate} do for tensors (v0.15 _functional_tensor.py: rotate =
# _get_inverse_affine_matrix(center 0, -angle) -> _gen_affine_grid -> grid_sample(nearest, zeros, align_corners=False)).
# Parity unpinned (no torchvision to run); the grid construction and the sampler are torch's own.
def tv_flip_rotate(stack: torch.Tensor, angles_deg, hflip, vflip) -> torch.Tensor:
    import math
    out = []
    for i in range(stack.shape[0]):
        img = stack[i:i + 1]
        if hflip[i]:
            img = img.flip(-1)
        if vflip[i]:
            img = img.flip(-2)
        a = math.radians(float(angles_deg[i]))
        # rotate(img, angle): matrix = inverse affine of (-angle) = [cos a, -sin a, 0, sin a, cos a, 0]
        theta = torch.tensor([[math.cos(a), -math.sin(a), 0.0], [math.sin(a), math.cos(a), 0.0]], dtype=torch.float32).view(1, 2, 3)
        h, w = img.shape[-2:]
        d = 0.5
        base = torch.empty(1, h, w, 3, dtype=torch.float32)
        base[..., 0].copy_(torch.linspace(-w * 0.5 + d, w * 0.5 + d - 1, steps=w))
        base[..., 1].copy_(torch.linspace(-h * 0.5 + d, h * 0.5 + d - 1, steps=h).unsqueeze(-1))
        base[..., 2].fill_(1)
        rescaled = theta.transpose(1, 2) / torch.tensor([0.5 * w, 0.5 * h], dtype=torch.float32)
        grid = base.view(1, h * w, 3).bmm(rescaled).view(1, h, w, 2)
        out.append(F.grid_sample(img, grid, mode="nearest", padding_mode="zeros", align_corners=False))
    return torch.cat(out, dim=0)

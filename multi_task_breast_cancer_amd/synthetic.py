"""Curated-BUSI-shaped synthetic batches generated on the device: raw 0-255 float32 speckle images
(the reference feeds un-normalised pixels, BUSI_dataloader.py:352), one lesion per non-normal sample, labels with the
oversampled class proportions 444:492:448 (benign : malignant : normal).  The task is learnable: a lesion is a darker
(hypoechoic) region; benign = smooth ellipse, malignant = irregular (harmonic-perturbed) boundary with lower contrast,
normal = no lesion and an empty mask.  Deterministic in (seed, rank).

`hard=True` (round 4): a task that can FAIL, shaped after what makes Curated-BUSI hard (the reference's published level is DSC 0.751 /
ACC 0.802, README.md:77): lesions under multiplicative speckle with structure at lesion scale (excursions of +-25 %: a lesion of 30 - 50 %
contrast is one dark region among several),
dark distractor regions (acoustic shadows) that are NOT lesions in every class, an annotation that is not the lesion's exact outline
(the mask is the lesion with its radii and centre jittered, as a second reader would draw it), benign / malignant told apart by
boundary irregularity alone, and 12 % of the benign / malignant labels swapped.  No arithmetic reaches Dice 0.99 / accuracy 1.0 on it."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def synthetic_batch(n: int, h: int, w: int, seed: int, device, rank: int = 0, hard: bool = False):
    if hard:
        return _hard_batch(n, h, w, seed, device, rank)
    g = torch.Generator(device="cpu").manual_seed(seed * 7919 + rank)
    base = torch.randn(n, 1, h // 8 + 1, w // 8 + 1, generator=g).to(device)
    low = F.interpolate(base, size=(h, w), mode="bilinear", align_corners=True)
    noise = torch.randn(n, 1, h, w, generator=g).to(device)
    probs = torch.tensor([444.0, 492.0, 448.0])
    label = torch.multinomial(probs, n, replacement=True, generator=g).to(torch.float32).view(n, 1)
    yy = torch.arange(h, device=device).view(1, h, 1).float()
    xx = torch.arange(w, device=device).view(1, 1, w).float()
    r = torch.rand(4, n, 1, 1, generator=g).to(device)
    cy, cx = (0.3 + 0.4 * r[0]) * h, (0.3 + 0.4 * r[1]) * w
    ry, rx = (0.08 + 0.14 * r[2]) * h, (0.08 + 0.14 * r[3]) * w
    dy, dx = (yy - cy) / ry, (xx - cx) / rx
    rad = torch.sqrt(dy * dy + dx * dx)
    theta = torch.atan2(dy, dx)
    lab_d = label.to(device).view(n, 1, 1)
    amp = (lab_d == 1).float() * 0.22                       # malignant: irregular boundary
    coef = torch.randn(5, n, 1, 1, generator=g).to(device)
    phase = (torch.rand(5, n, 1, 1, generator=g) * 2 * math.pi).to(device)
    bound = torch.ones_like(rad)
    for k in range(5):
        bound = bound + amp * 0.45 * coef[k] * torch.cos((k + 3) * theta + phase[k])
    lesion = (rad <= bound).float() * (lab_d != 2).float()
    mask = lesion.view(n, 1, h, w).contiguous()
    contrast = torch.where(lab_d == 1, torch.tensor(0.50, device=device), torch.tensor(0.38, device=device)).view(n, 1, 1, 1)
    soft = F.avg_pool2d(mask, 5, stride=1, padding=2)        # slightly blurred edge
    img = (128.0 + 48.0 * low + 30.0 * noise) * (1.0 - contrast * soft)
    img = torch.clamp(img, 0.0, 255.0).contiguous()
    return img, mask, label.to(device)


def _blobs(n, h, w, g, device, cy, cx, ry, rx, amp, coef, phase):
    """harmonic-perturbed ellipses (rad <= bound) as float masks; every argument (n,1,1) or (5,n,1,1)"""
    yy = torch.arange(h, device=device).view(1, h, 1).float()
    xx = torch.arange(w, device=device).view(1, 1, w).float()
    dy, dx = (yy - cy) / ry, (xx - cx) / rx
    rad = torch.sqrt(dy * dy + dx * dx)
    theta = torch.atan2(dy, dx)
    bound = torch.ones_like(rad)
    for k in range(5):
        bound = bound + amp * 0.45 * coef[k] * torch.cos((k + 3) * theta + phase[k])
    return (rad <= bound).float()


HARD_CONTRAST = (0.30, 0.20)       # lesion contrast = lo + span * U(0, 1); calibrated with tools/train_parity.py --hard-contrast (profiles/r04_quality_hard.md)


def _hard_batch(n: int, h: int, w: int, seed: int, device, rank: int = 0):
    g = torch.Generator(device="cpu").manual_seed(seed * 104729 + rank + 17)
    dev = device
    rnd = lambda *shape: torch.rand(*shape, generator=g).to(dev)
    rndn = lambda *shape: torch.randn(*shape, generator=g).to(dev)
    probs = torch.tensor([444.0, 492.0, 448.0])
    true = torch.multinomial(probs, n, replacement=True, generator=g).view(n, 1)          # 0 benign, 1 malignant, 2 normal
    lab_d = true.to(dev).view(n, 1, 1)
    # the lesion in the IMAGE
    r = rnd(4, n, 1, 1)
    cy, cx = (0.3 + 0.4 * r[0]) * h, (0.3 + 0.4 * r[1]) * w
    ry, rx = (0.07 + 0.13 * r[2]) * h, (0.07 + 0.13 * r[3]) * w
    amp = (lab_d == 1).float() * 0.20 + (lab_d == 0).float() * 0.05          # malignant: irregular boundary; benign: nearly smooth
    coef, phase = rndn(5, n, 1, 1), rnd(5, n, 1, 1) * 2 * math.pi
    lesion = _blobs(n, h, w, g, dev, cy, cx, ry, rx, amp, coef, phase) * (lab_d != 2).float()
    # the ANNOTATION: the same outline drawn by another hand -- radii x U(0.88, 1.12), centre moved by up to 3 % of the image
    j = rnd(4, n, 1, 1)
    mask = _blobs(n, h, w, g, dev, cy + (j[0] - 0.5) * 0.06 * h, cx + (j[1] - 0.5) * 0.06 * w, ry * (0.88 + 0.24 * j[2]), rx * (0.88 + 0.24 * j[3]),
                  amp, coef, phase) * (lab_d != 2).float()
    # distractors: one or two dark regions per image that are not lesions (any class)
    d = rnd(8, n, 1, 1)
    shadow = _blobs(n, h, w, g, dev, (0.15 + 0.7 * d[0]) * h, (0.15 + 0.7 * d[1]) * w, (0.05 + 0.10 * d[2]) * h, (0.05 + 0.12 * d[3]) * w,
                    torch.full_like(amp, 0.12), rndn(5, n, 1, 1), rnd(5, n, 1, 1) * 2 * math.pi)
    shadow2 = _blobs(n, h, w, g, dev, (0.15 + 0.7 * d[4]) * h, (0.15 + 0.7 * d[5]) * w, (0.04 + 0.07 * d[6]) * h, (0.04 + 0.09 * d[7]) * w,
                     torch.full_like(amp, 0.12), rndn(5, n, 1, 1), rnd(5, n, 1, 1) * 2 * math.pi) * (d[6] > 0.5).float()
    # tissue: smooth background + speckle with structure at three scales (pixel, ~6 px, ~20 px), multiplicative
    def field(scale):
        b = rndn(n, 1, h // scale + 2, w // scale + 2)
        return F.interpolate(b, size=(h, w), mode="bilinear", align_corners=True)
    low = field(32)
    # (the pixel-scale field comes from a generator ON the device: 2 M normals per batch from the CPU generator cost more than a training step)
    if torch.device(dev).type == "cuda":
        gd = torch.Generator(device=dev).manual_seed(seed * 104729 + rank + 23)
        white = torch.randn(n, 1, h, w, generator=gd, device=dev)
    else:
        white = rndn(n, 1, h, w)
    speckle = 0.55 * white + 0.35 * field(6) + 0.30 * field(20)
    contrast = (HARD_CONTRAST[0] + HARD_CONTRAST[1] * rnd(n, 1, 1, 1))           # 30 - 50 % under speckle whose own excursions are +-25 %
    sc = (0.10 + 0.15 * rnd(n, 1, 1, 1))
    soft = lambda m: F.avg_pool2d(m.view(n, 1, h, w), 9, stride=1, padding=4)
    tissue = (118.0 + 40.0 * low) * (1.0 + 0.33 * speckle)
    img = tissue * (1.0 - contrast * soft(lesion)) * (1.0 - sc * soft(torch.clamp(shadow + shadow2, 0, 1)))
    img = torch.clamp(img, 0.0, 255.0).contiguous()
    # label noise: 12 % of the benign / malignant labels swapped (normal stays normal: its mask is empty)
    flip = (torch.rand(n, 1, generator=g) < 0.12) & (true != 2)
    label = torch.where(flip, 1 - true, true).to(torch.float32)
    return img, mask.view(n, 1, h, w).contiguous(), label.to(dev)

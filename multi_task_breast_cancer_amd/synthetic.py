"""Curated-BUSI-shaped synthetic batches generated on the device: raw 0-255 float32 speckle images
(the reference feeds un-normalised pixels, BUSI_dataloader.py:352), one lesion per non-normal sample, labels with the
oversampled class proportions 444:492:448 (benign : malignant : normal).  The task is learnable: a lesion is a darker
(hypoechoic) region; benign = smooth ellipse, malignant = irregular (harmonic-perturbed) boundary with lower contrast,
normal = no lesion and an empty mask.  Deterministic in (seed, rank)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


def synthetic_batch(n: int, h: int, w: int, seed: int, device, rank: int = 0):
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

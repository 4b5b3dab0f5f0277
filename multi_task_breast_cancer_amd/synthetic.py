"""Curated-BUSI-shaped synthetic batches generated on the device: raw 0-255 float32 speckle images
(the reference feeds un-normalised pixels, BUSI_dataloader.py:352), one filled ellipse per non-normal sample,
labels with the oversampled class proportions 444:492:448.  Deterministic in (seed, rank)."""
from __future__ import annotations

import torch
import torch.nn.functional as F


def synthetic_batch(n: int, h: int, w: int, seed: int, device, rank: int = 0):
    g = torch.Generator(device="cpu").manual_seed(seed * 7919 + rank)
    base = torch.randn(n, 1, h // 8 + 1, w // 8 + 1, generator=g).to(device)
    low = F.interpolate(base, size=(h, w), mode="bilinear", align_corners=True)
    noise = torch.randn(n, 1, h, w, generator=g).to(device)
    img = torch.clamp(128.0 + 48.0 * low + 32.0 * noise, 0.0, 255.0).contiguous()
    probs = torch.tensor([444.0, 492.0, 448.0])
    label = torch.multinomial(probs, n, replacement=True, generator=g).to(torch.float32).view(n, 1)
    yy = torch.arange(h, device=device).view(1, h, 1).float()
    xx = torch.arange(w, device=device).view(1, 1, w).float()
    r = torch.rand(4, n, 1, 1, generator=g).to(device)
    cy, cx = (0.25 + 0.5 * r[0]) * h, (0.25 + 0.5 * r[1]) * w
    ry, rx = (0.08 + 0.17 * r[2]) * h, (0.08 + 0.17 * r[3]) * w
    ell = (((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0).float()
    lab_d = label.to(device)
    mask = (ell * (lab_d.view(n, 1, 1) != 2).float()).view(n, 1, h, w).contiguous()
    return img, mask, lab_d

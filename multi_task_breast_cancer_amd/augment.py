"""GPU augmentation of the joint (mask, image) stack -- SURVEY 8(f) row N2.

Mirror of the reference's `transforms = Sequential(RandomHorizontalFlip(.5), RandomVerticalFlip(.5),
RandomRotation(degrees=360))` (training_multitask.py:193-197), which BUSI_dataset.py:142-147 applies to
`torch.cat([mask, image], dim=0)` so that both receive the SAME flip / angle.  Here the whole batch is transformed by
one HIP gather kernel (csrc/augment.hip) from per-sample parameters drawn on the host.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib as L


def random_params(n: int, rng: np.random.Generator, p_hflip: float = 0.5, p_vflip: float = 0.5,
                  degrees: float = 360.0) -> torch.Tensor:
    """(n, 4) float32 rows {cos a, sin a, flip_h, flip_v}: a ~ U(-degrees, degrees) (RandomRotation.get_params), the
    flips Bernoulli(p) -- cos / sin evaluated in float64 on the host like torchvision's `math.cos(math.radians(a))`."""
    ang = rng.uniform(-degrees, degrees, size=n)
    out = np.empty((n, 4), dtype=np.float32)
    out[:, 0] = [math.cos(math.radians(a)) for a in ang]
    out[:, 1] = [math.sin(math.radians(a)) for a in ang]
    out[:, 2] = rng.random(n) < p_hflip
    out[:, 3] = rng.random(n) < p_vflip
    return torch.from_numpy(out)


def params_from(angles_deg, hflip, vflip) -> torch.Tensor:
    out = np.empty((len(angles_deg), 4), dtype=np.float32)
    out[:, 0] = [math.cos(math.radians(float(a))) for a in angles_deg]
    out[:, 1] = [math.sin(math.radians(float(a))) for a in angles_deg]
    out[:, 2] = np.asarray(hflip, dtype=np.float32)
    out[:, 3] = np.asarray(vflip, dtype=np.float32)
    return torch.from_numpy(out)


def flip_rotate(stack: torch.Tensor, params: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """stack (N, C, H, W) float32 on the GPU (channel 0 = mask, 1.. = image planes); params from `random_params`."""
    if not stack.is_cuda:
        raise L.MtbcError("flip_rotate needs CUDA/HIP tensors (no CPU fallback)")
    stack = stack.contiguous().float()
    params = params.to(stack.device, torch.float32).contiguous()
    if out is None:
        out = torch.empty_like(stack)
    n, c, h, w = stack.shape
    L.check(L.load().mtbc_augment_flip_rotate(stack.data_ptr(), out.data_ptr(), params.data_ptr(), n, c, h, w,
                                              C.c_void_p(torch.cuda.current_stream().cuda_stream)), "augment_flip_rotate")
    return out


def augment_batch(image: torch.Tensor, mask: torch.Tensor, rng: np.random.Generator) -> Tuple[torch.Tensor, torch.Tensor]:
    """(image, mask) -> augmented (image, mask): the same flips / rotation for both, as BUSI_dataset.py:142-147."""
    joined = torch.cat([mask, image], dim=1)
    res = flip_rotate(joined, random_params(joined.shape[0], rng))
    return res[:, mask.shape[1]:], res[:, :mask.shape[1]]

"""Segmentation / classification criteria on HIP + the reference's loss-aggregation glue.

  DiceLoss   replaces monai.losses.DiceLoss(include_background=True, sigmoid=True, smooth_dr=1, smooth_nr=1,
             squared_pred=True) built at src/utils/experiment_init.py:210-211
  FocalLoss  replaces src/utils/criterions.py:6-24
  apply_criterion_multitask_segmentation_classification mirrors criterions.py:52-76 (same signature, NaN -> exit 1)

Both criteria are torch.autograd.Functions over the C-ABI kernels (mtbc_dice_fwd/_bwd, mtbc_focal_fwd_bwd);
they raise when handed CPU tensors -- there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import logging
import sys

import torch

from . import _lib as L


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_cuda(*ts: torch.Tensor) -> None:
    for t in ts:
        if t is not None and t.device.type != "cuda":
            L.require_gpu()
            raise L.MtbcError("criterion called with a CPU tensor: the HIP path needs device tensors")


class _DiceFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits: torch.Tensor, target: torch.Tensor, smooth_nr: float, smooth_dr: float):
        _need_cuda(logits, target)
        x = logits.detach().contiguous().float()
        t = target.detach().contiguous().float()
        if x.dim() < 3 or x.shape != t.shape:
            raise ValueError(f"DiceLoss: logits {tuple(x.shape)} vs target {tuple(t.shape)}")
        n, c = x.shape[0], x.shape[1]
        hw = x[0, 0].numel()
        a = L.DiceArgs()
        a.n_heads, a.N, a.C, a.H, a.W = 1, n, c, hw, 1
        a.smooth_nr, a.smooth_dr = smooth_nr, smooth_dr
        stats = torch.empty(n * c * 3, dtype=torch.float32, device=x.device)
        loss = torch.empty(2, dtype=torch.float32, device=x.device)
        a.x[0], a.target, a.head_weight[0] = x.data_ptr(), t.data_ptr(), 1.0
        a.stats, a.loss = stats.data_ptr(), loss.data_ptr()
        L.check(L.load().mtbc_dice_fwd(C.byref(a), _stream()), "dice_fwd")
        ctx.save_for_backward(x, t, stats)
        ctx.smooth = (smooth_nr, smooth_dr)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, gout):
        x, t, stats = ctx.saved_tensors
        n, c = x.shape[0], x.shape[1]
        a = L.DiceArgs()
        a.n_heads, a.N, a.C, a.H, a.W = 1, n, c, x[0, 0].numel(), 1
        a.smooth_nr, a.smooth_dr = ctx.smooth
        dx = torch.empty_like(x)
        g = gout.detach().contiguous().float()
        a.x[0], a.target, a.head_weight[0] = x.data_ptr(), t.data_ptr(), 1.0
        a.stats, a.dx[0] = stats.data_ptr(), dx.data_ptr()
        a.gscale, a.gscale_dev = 1.0, g.data_ptr()
        L.check(L.load().mtbc_dice_bwd(C.byref(a), _stream()), "dice_bwd")
        return dx, None, None, None


class DiceLoss(torch.nn.Module):
    """Only the configuration the reference builds is on the hot path; anything else raises."""

    def __init__(self, include_background: bool = True, sigmoid: bool = True, smooth_dr: float = 1.0,
                 smooth_nr: float = 1.0, squared_pred: bool = True):
        super().__init__()
        if not (include_background and sigmoid and squared_pred):
            raise ValueError("only DiceLoss(include_background=True, sigmoid=True, squared_pred=True) is supported")
        self.smooth_nr, self.smooth_dr = float(smooth_nr), float(smooth_dr)

    def forward(self, input: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return _DiceFn.apply(input, target, self.smooth_nr, self.smooth_dr)


class _FocalFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, inputs, targets, alpha, gamma, weight):
        _need_cuda(inputs, targets, weight)
        x = inputs.detach().contiguous().float()
        t = targets.detach().contiguous().float()
        if x.dim() != 2 or x.shape != t.shape:
            raise ValueError(f"FocalLoss expects (N,C) logits and float (N,C) targets, got {tuple(x.shape)} / {tuple(t.shape)}")
        loss = torch.empty(1, dtype=torch.float32, device=x.device)
        dx = torch.empty_like(x)
        a = L.FocalArgs()
        a.N, a.C, a.alpha, a.gamma = x.shape[0], x.shape[1], alpha, gamma
        a.x, a.target = x.data_ptr(), t.data_ptr()
        a.weight = weight.contiguous().float().data_ptr() if weight is not None else None
        a.loss, a.dx, a.gscale = loss.data_ptr(), dx.data_ptr(), 1.0
        L.check(L.load().mtbc_focal_fwd_bwd(C.byref(a), _stream()), "focal")
        ctx.save_for_backward(dx)
        return loss[0].clone()

    @staticmethod
    def backward(ctx, gout):
        (dx,) = ctx.saved_tensors
        return dx * gout, None, None, None, None


class FocalLoss(torch.nn.Module):
    def __init__(self, alpha=1, gamma=2, reduction="mean", weight=None):
        super().__init__()
        if reduction != "mean":
            raise ValueError("only reduction='mean' is on the hot path (experiment_init.py:259)")
        self.alpha, self.gamma, self.reduction, self.weight = float(alpha), float(gamma), reduction, weight

    def forward(self, inputs, targets):
        return _FocalFn.apply(inputs, targets, self.alpha, self.gamma, self.weight)


def apply_criterion_multitask_segmentation_classification(criterion_seg, ground_truth, segmentation, criterion_class,
                                                          label, predicted_class, inversely_weighted=False):
    """criterions.py:52-76: deep-supervision heads weighted 1/(n+1) from the LAST head backwards."""
    if isinstance(segmentation, list):
        heads = list(reversed(segmentation))
        if inversely_weighted:
            segmentation_loss = torch.sum(torch.stack([criterion_seg(s, ground_truth) / (n + 1) for n, s in enumerate(heads)]))
        else:
            segmentation_loss = torch.sum(torch.stack([criterion_seg(s, ground_truth) for s in heads]))
        classification_loss = torch.sum(torch.stack([criterion_class(c, label) for c in reversed(predicted_class)]))
    else:
        segmentation_loss = criterion_seg(segmentation, ground_truth)
        classification_loss = criterion_class(predicted_class, label)
    if not torch.isnan(segmentation_loss) and not torch.isnan(classification_loss):
        return segmentation_loss, classification_loss
    logging.info("NaN in model loss!!")
    sys.exit(1)

"""Test-time prediction refining of the multi-task model (second half).

`inference_multitask_multiclass_classification_segmentation` (src/utils/models.py:270-400) runs the model twice
over the batch-1 test loader and applies two cross-task rules, each on the RAW prediction of the other task:

  * overlap_seg_based_on_class (:325-332): predicted class == 2 ("normal") -> the predicted mask is cleared;
  * overlap_class_based_on_seg (:366-376): no tumour pixel in sigmoid(last head) > .5 -> the predicted class becomes 2.

Here both rules are tensor ops on the device for a whole batch (the forward pass is the HIP step program); Hausdorff /
sensitivity tables and PNG dumps stay the reference's code.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple, Union

import torch

NORMAL_CLASS = 2


def _last(x):
    return x[-1] if isinstance(x, (list, tuple)) else x


def _mean_logits(logits: Union[torch.Tensor, Sequence[torch.Tensor]]) -> torch.Tensor:
    if isinstance(logits, (list, tuple)):                       # deep supervision: average the heads (:327, :358)
        return torch.mean(torch.stack(list(logits), dim=0), dim=0)
    return logits


def refine_predictions(cls_logits, seg_logits, overlap_seg_based_on_class: bool = True,
                       overlap_class_based_on_seg: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
    """(cls_logits, seg_logits) as returned by the model -> (binary masks (N,1,H,W) float, class ids (N,) int64)."""
    lg = _mean_logits(cls_logits)
    lg = lg.view(lg.shape[0], -1)
    raw_cls = lg.argmax(dim=1)
    raw_seg = (torch.sigmoid(_last(seg_logits)) > .5).float()
    seg, cls = raw_seg, raw_cls
    if overlap_seg_based_on_class:
        seg = raw_seg * (raw_cls != NORMAL_CLASS).view(-1, 1, 1, 1).to(raw_seg.dtype)
    if overlap_class_based_on_seg:
        empty = raw_seg.flatten(1).sum(dim=1) == 0
        cls = torch.where(empty, torch.full_like(raw_cls, NORMAL_CLASS), raw_cls)
    return seg, cls


@torch.no_grad()
def predict(model, images: torch.Tensor, overlap_seg_based_on_class: bool = True,
            overlap_class_based_on_seg: bool = True) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """One forward pass + refining: (masks, class ids, class probabilities)."""
    logits, segs = model(images)
    seg, cls = refine_predictions(logits, segs, overlap_seg_based_on_class, overlap_class_based_on_seg)
    lg = _mean_logits(logits)
    return seg, cls, torch.softmax(lg.view(lg.shape[0], -1), dim=1)

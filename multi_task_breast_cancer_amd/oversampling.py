"""`deterministic_oversampling` of src/dataset/BUSI_dataloader.py:320-340 -- integer/index work on the host.

Follows the pandas-1.5 meaning of the reference code (`value_counts().reset_index()` columns 'index' = class,
'class' = proportion), works on a pandas DataFrame with a 'class' column or on a plain sequence of
class labels, and needs no particular pandas version.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import List, Sequence

import numpy as np


def compute_scaling_factor(classes: Sequence) -> "OrderedDict[str, int]":
    labels = list(classes)
    total = len(labels)
    counts: "OrderedDict[str, int]" = OrderedDict()
    for c in labels:
        counts[c] = counts.get(c, 0) + 1
    names = list(counts.keys())
    order = sorted(range(len(names)), key=lambda i: (-counts[names[i]], i))     # descending, ties first-seen
    out: "OrderedDict[str, int]" = OrderedDict()
    for i in order:
        prop = counts[names[i]] / total
        out[names[i]] = int(np.round(1.0 / prop, 0))                            # half-to-even, as Series.round
    return out


def oversampled_positions(classes: Sequence) -> np.ndarray:
    """Row positions of the oversampled frame: originals, then per class (value_counts order) its rows repeated
    factor-1 times -- or once when factor == 1 (reference quirk, :334-336)."""
    labels = np.asarray(list(classes), dtype=object)
    pieces: List[np.ndarray] = [np.arange(len(labels), dtype=np.int64)]
    for name, factor in compute_scaling_factor(labels.tolist()).items():
        rows = np.nonzero(labels == name)[0].astype(np.int64)
        pieces.extend([rows] * (factor - 1 if factor > 1 else 1))
    return np.concatenate(pieces) if pieces else np.zeros(0, dtype=np.int64)


def deterministic_oversampling(mapping_df):
    """DataFrame in, DataFrame out (`ignore_index=True`), same name as the reference function."""
    pos = oversampled_positions(mapping_df["class"].tolist())
    return mapping_df.iloc[pos].reset_index(drop=True)

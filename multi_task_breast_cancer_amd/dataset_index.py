"""Index service for the training loop: cross-validation folds, deterministic oversampling and per-rank batches.

Integer-only restatement of the index work of `BUSI_dataloader_CV` (src/dataset/BUSI_dataloader.py:79-150) -- SURVEY
8(f) row N1.  The reference builds three `BUSI` datasets per fold from DataFrame slices; what defines *which samples*
land where is:

    StratifiedKFold(n_splits, shuffle=True, random_state=seed).split(mapping, mapping['class'])         (:104-105)
    train_test_split(train_val, train_size, random_state=seed, shuffle=True, stratify=train_val.class)  (:110-111)
    deterministic_oversampling(train_mapping)                                                           (:124, :320-340)

and that is what `cv_fold_positions` returns, as positions into the class-filtered mapping.  The two sklearn calls are
the reference's own calls (same arguments), so the folds equal the reference's whenever the sklearn version is the
same; they are pure functions of (class labels, seed).  The reference then shuffles with `DataLoader(shuffle=True)` on
torch's global RNG (:146); under data parallelism every rank must see the SAME order, so `EpochIndex` replaces that
with one seeded permutation per epoch, cut into global batches and equal contiguous per-rank shards.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, Iterator, List, Optional, Sequence

import numpy as np

from .oversampling import oversampled_positions


def cv_fold_positions(classes: Sequence[str], seed: int, n_folds: int = 5, train_size: float = 0.8,
                      oversampling: bool = True, keep_classes: Optional[Sequence[str]] = None) -> List[Dict[str, np.ndarray]]:
    """One dict per fold: 'train' (oversampled, in the reference's row order), 'val', 'test' -- positions into the
    mapping *after* the `mapping['class'].isin(classes)` filter (:99), plus 'kept' = positions of the filtered rows in
    the unfiltered mapping."""
    from sklearn.model_selection import StratifiedKFold, train_test_split

    labels = np.asarray(list(classes), dtype=object)
    kept = np.arange(len(labels)) if keep_classes is None else np.nonzero(np.isin(labels, list(keep_classes)))[0]
    y = labels[kept]
    folds = []
    kfold = StratifiedKFold(n_splits=n_folds, shuffle=True, random_state=int(seed))
    for train_ix, test_ix in kfold.split(np.zeros(len(y)), y):
        tr_rel, va_rel = train_test_split(np.arange(len(train_ix)), train_size=train_size, random_state=int(seed),
                                          shuffle=True, stratify=y[train_ix])
        train, val = train_ix[tr_rel], train_ix[va_rel]
        if oversampling:
            train = train[oversampled_positions(y[train].tolist())]
        folds.append({"train": train.astype(np.int64), "val": val.astype(np.int64), "test": test_ix.astype(np.int64),
                      "kept": kept.astype(np.int64)})
    return folds


@dataclass
class EpochIndex:
    """Per-rank batches of one training set.  Every rank builds the same object (same seed) and reads its own shard:
    batch b of epoch e is rows [b*G, (b+1)*G) of ONE permutation of `positions`, and rank r owns the contiguous slice
    [r*G/world, (r+1)*G/world) of it -- the union over ranks is exactly the single-process batch.

    drop_last=False keeps the reference's `DataLoader(drop_last=False)` (:146): the final short batch is then split as
    evenly as contiguity allows and `weights()` gives each rank's share n_local / n_batch for exact gradient
    averaging (FusedTrainStep assumes equal shards: use drop_last=True with it)."""
    positions: np.ndarray
    global_batch: int
    seed: int
    rank: int = 0
    world: int = 1
    drop_last: bool = False

    def __post_init__(self):
        self.positions = np.asarray(self.positions, dtype=np.int64)
        if self.global_batch % self.world:
            raise ValueError("global batch must divide evenly across ranks")

    def __len__(self) -> int:
        n = len(self.positions)
        return n // self.global_batch if self.drop_last else -(-n // self.global_batch)

    def permutation(self, epoch: int) -> np.ndarray:
        order = np.random.Generator(np.random.PCG64(self.seed * 1_000_003 + epoch)).permutation(len(self.positions))
        return self.positions[order]

    def _bounds(self, n_batch: int, rank: int):
        per, extra = divmod(n_batch, self.world)
        lo = rank * per + min(rank, extra)
        return lo, lo + per + (1 if rank < extra else 0)

    def batches(self, epoch: int) -> Iterator[np.ndarray]:
        perm = self.permutation(epoch)
        for b in range(len(self)):
            chunk = perm[b * self.global_batch:(b + 1) * self.global_batch]
            lo, hi = self._bounds(len(chunk), self.rank)
            yield chunk[lo:hi]

    def weights(self, epoch: int) -> List[float]:
        """n_local / n_batch per batch for this rank (1/world for every full batch)."""
        n = len(self.positions)
        out = []
        for b in range(len(self)):
            nb = min(self.global_batch, n - b * self.global_batch)
            lo, hi = self._bounds(nb, self.rank)
            out.append((hi - lo) / nb)
        return out

"""Plain-Python restatement of `deterministic_oversampling`
(src/dataset/BUSI_dataloader.py:320-340) with **pandas-1.5** semantics.

TEST INFRASTRUCTURE -- see oracle/__init__.py.

pandas 1.5 meaning of `value_counts(normalize=True).reset_index()` (:323): column
'index' holds the class name, column 'class' the proportion, rows sorted by
descending count with ties left in first-appearance order (insertion sort for
n <= 16 in numpy's quicksort).  `round(1 / proportion, 0)` (:324) is
`Series.round` = numpy round-half-to-even.  Under the pandas 2.x installed in the
build container the reference itself raises TypeError on that line,
so this restatement is pinned on hand-derived known answers and on the class
counts of the reference's own data/mapping_curated_BUSI.csv.

The result is expressed as *source row positions*: entry k of the returned list
is the position (0-based, in the input order) of the row that lands at index k
of the oversampled frame (`ignore_index=True`, :338).
"""
from __future__ import annotations

from typing import Dict, List, Sequence


def _round_half_even(x: float) -> int:
    # Python's round() on floats is round-half-to-even, same as numpy.round.
    return int(round(x))


def scaling_factors(classes: Sequence[str]) -> Dict[str, int]:
    """compute_scaling_factor, :322-325.  Dict order == value_counts order."""
    n = len(classes)
    counts: Dict[str, int] = {}
    for c in classes:                       # first-appearance order
        counts[c] = counts.get(c, 0) + 1
    ordered = sorted(counts.items(), key=lambda kv: -kv[1])   # stable: ties keep first-seen order
    return {name: _round_half_even(1.0 / (cnt / n)) for name, cnt in ordered}


def deterministic_oversampling_positions(classes: Sequence[str]) -> List[int]:
    """:327-338.  Quirk kept: factor == 1 still appends the class once (:334-336)."""
    factors = scaling_factors(classes)
    out = list(range(len(classes)))
    for name, factor in factors.items():
        rows = [i for i, c in enumerate(classes) if c == name]
        reps = factor - 1 if factor > 1 else 1
        for _ in range(reps):
            out.extend(rows)
    return out

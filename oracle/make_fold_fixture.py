#!/usr/bin/env python3
"""Write tests/golden/cv_folds_sklearn_<version>.npz: the row positions that `BUSI_dataloader_CV`
(src/dataset/BUSI_dataloader.py:79-150) selects per fold on the reference's curated mapping, computed with the reference's
own call sequence on a DataFrame -- StratifiedKFold(shuffle, random_state=seed).split(mapping, mapping['class']) (:104),
train_test_split(train_val, train_size, random_state=seed, shuffle=True, stratify=class) (:110), then
deterministic_oversampling (:124) through the ORACLE's pandas-1.5 restatement (oracle/oversampling_oracle.py; the
reference's own function raises under pandas >= 2).

The folds are sklearn-version-sensitive in principle (the reference pins 1.3.0): the file name and the `sklearn` field
carry the version that generated them, and the test that reads the fixture skips under any other version.

    python oracle/make_fold_fixture.py
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle.oversampling_oracle import deterministic_oversampling_positions      # noqa: E402


def main() -> None:
    import pandas as pd
    import sklearn
    from sklearn.model_selection import StratifiedKFold, train_test_split
    gold = os.path.join(ROOT, "tests", "golden")
    classes = [str(c) for c in np.load(os.path.join(gold, "curated_busi_classes.npz"))["classes"]]
    mapping = pd.DataFrame({"id": np.arange(len(classes)), "class": classes})
    out = {"sklearn": np.array(sklearn.__version__), "seed": np.array(1993), "n_folds": np.array(5), "train_size": np.array(0.8)}
    kfold = StratifiedKFold(n_splits=5, shuffle=True, random_state=1993)
    for n, (train_ix, test_ix) in enumerate(kfold.split(mapping, mapping["class"])):
        train_val, test = mapping.iloc[train_ix], mapping.iloc[test_ix]
        train, val = train_test_split(train_val, train_size=0.8, random_state=1993, shuffle=True, stratify=train_val["class"])
        over = train.iloc[deterministic_oversampling_positions(train["class"].tolist())]
        out[f"train{n}"] = over["id"].to_numpy().astype(np.int64)
        out[f"val{n}"] = val["id"].to_numpy().astype(np.int64)
        out[f"test{n}"] = test["id"].to_numpy().astype(np.int64)
    path = os.path.join(gold, f"cv_folds_sklearn_{sklearn.__version__}.npz")
    np.savez_compressed(path, **out)
    print("written", path, {k: len(v) for k, v in out.items() if k.startswith("train") and v.ndim == 1})


if __name__ == "__main__":
    main()

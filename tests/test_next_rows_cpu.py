"""SURVEY 8(f) rows N1 / N4 on the host: CV fold positions + per-rank epoch index, metrics-file / early-stop helpers."""
import os

import numpy as np
import pytest

from multi_task_breast_cancer_amd import checkpoint as CK
from multi_task_breast_cancer_amd.dataset_index import EpochIndex, cv_fold_positions
from multi_task_breast_cancer_amd.oversampling import oversampled_positions
from oracle.oversampling_oracle import deterministic_oversampling_positions


def _classes(golden_dir):
    return [str(c) for c in np.load(os.path.join(golden_dir, "curated_busi_classes.npz"))["classes"]]


def test_cv_folds_equal_the_reference_call_sequence(golden_dir):
    """BUSI_dataloader_CV (:79-150) verbatim on a DataFrame -- StratifiedKFold.split(mapping, mapping['class']) then
    train_test_split(train_val_mapping, ..., stratify=train_val_mapping['class']) then oversampling -- must select the
    same rows as the index-only restatement (the sklearn calls are the reference's own; pinned to the sklearn here)."""
    import pandas as pd
    from sklearn.model_selection import StratifiedKFold, train_test_split
    classes = _classes(golden_dir)
    seed, n_folds, train_size = 1993, 5, 0.8
    mapping = pd.DataFrame({"id": np.arange(len(classes)), "class": classes})
    folds = cv_fold_positions(classes, seed, n_folds, train_size, oversampling=True)
    kfold = StratifiedKFold(n_splits=n_folds, shuffle=True, random_state=int(seed))
    for n, (train_ix, test_ix) in enumerate(kfold.split(mapping, mapping["class"])):
        train_val_mapping, test_mapping = mapping.iloc[train_ix], mapping.iloc[test_ix]
        train_mapping, val_mapping = train_test_split(train_val_mapping, train_size=train_size, random_state=int(seed),
                                                      shuffle=True, stratify=train_val_mapping["class"])
        # the ORACLE's restatement of deterministic_oversampling (:320-340), not the product's own function
        over = train_mapping.iloc[deterministic_oversampling_positions(train_mapping["class"].tolist())]
        assert np.array_equal(folds[n]["test"], test_mapping["id"].to_numpy())
        assert np.array_equal(folds[n]["val"], val_mapping["id"].to_numpy())
        assert np.array_equal(folds[n]["train"], over["id"].to_numpy())


def test_cv_folds_equal_the_committed_fixture(golden_dir):
    """tests/golden/cv_folds_sklearn_<version>.npz (oracle/make_fold_fixture.py: the reference's call sequence + the
    oracle's oversampling on the curated mapping).  sklearn's splitters are version-sensitive in principle, so the
    fixture is labelled with the version that generated it and only compared under that version."""
    import sklearn
    path = os.path.join(golden_dir, f"cv_folds_sklearn_{sklearn.__version__}.npz")
    if not os.path.exists(path):
        pytest.skip(f"no fold fixture generated with sklearn {sklearn.__version__}")
    g = np.load(path)
    folds = cv_fold_positions(_classes(golden_dir), int(g["seed"]), int(g["n_folds"]), float(g["train_size"]), oversampling=True)
    assert len(folds) == 5
    for n, f in enumerate(folds):
        for k in ("train", "val", "test"):
            assert np.array_equal(f[k], g[f"{k}{n}"]), (n, k)
    assert [len(g[f"train{n}"]) for n in range(5)] == [len(f["train"]) for f in folds]


def test_cv_folds_partition_and_stratify(golden_dir):
    classes = np.asarray(_classes(golden_dir), dtype=object)
    folds = cv_fold_positions(classes.tolist(), 7, 5, 0.8, oversampling=False)
    tests = np.concatenate([f["test"] for f in folds])
    assert sorted(tests.tolist()) == list(range(len(classes)))                 # test folds tile the data set
    for f in folds:
        assert not set(f["train"]) & set(f["val"]) and not set(f["train"]) & set(f["test"]) and not set(f["val"]) & set(f["test"])
        assert len(f["train"]) + len(f["val"]) + len(f["test"]) == len(classes)
        for c in np.unique(classes):                                           # stratified: class share within 3 points
            assert abs((classes[f["test"]] == c).mean() - (classes == c).mean()) < 0.03
    again = cv_fold_positions(classes.tolist(), 7, 5, 0.8, oversampling=False)
    assert all(np.array_equal(a[k], b[k]) for a, b in zip(folds, again) for k in ("train", "val", "test"))
    other = cv_fold_positions(classes.tolist(), 8, 5, 0.8, oversampling=False)
    assert any(not np.array_equal(a["test"], b["test"]) for a, b in zip(folds, other))
    two = cv_fold_positions(classes.tolist(), 7, 5, 0.8, keep_classes=["benign", "malignant"])
    assert set(classes[two[0]["kept"]]) == {"benign", "malignant"}


@pytest.mark.parametrize("world", [1, 2, 8])
def test_epoch_index_shards_reassemble_the_global_batch(world):
    pos = np.arange(1000, 1000 + 203)
    full = EpochIndex(pos, 32, seed=5)
    ranks = [EpochIndex(pos, 32, seed=5, rank=r, world=world) for r in range(world)]
    for epoch in (0, 1):
        per_rank = [list(r.batches(epoch)) for r in ranks]
        for b, ref in enumerate(full.batches(epoch)):
            assert np.array_equal(np.concatenate([per_rank[r][b] for r in range(world)]), ref)
        assert sorted(np.concatenate(list(full.batches(epoch))).tolist()) == sorted(pos.tolist())      # a permutation
        w = np.array([r.weights(epoch) for r in ranks])
        assert np.allclose(w.sum(axis=0), 1.0)
    assert not np.array_equal(full.permutation(0), full.permutation(1))
    assert len(EpochIndex(pos, 32, seed=5, drop_last=True)) == 6 and len(full) == 7
    with pytest.raises(ValueError):
        EpochIndex(pos, 30, seed=1, rank=0, world=4)


def test_metrics_file_and_early_stopping(tmp_path):
    f = tmp_path / "metrics.csv"
    CK.write_metrics_file(str(f), CK.METRICS_HEADER)
    CK.write_metrics_file(str(f), CK.metrics_row(0, 1e-4, 1.23456, 1.5, 0.5, 0.25, 0.75, 0.7, 0.6, 0.55))
    lines = f.read_text().splitlines()
    assert lines[0] == "epoch,LR,Train_loss,Validation_loss,Train_dice,Validation_dice,Train_acc,Train_F1,Validation_acc,Validation_F1"
    assert lines[1] == "0,0.00010000,1.2346,1.5000,0.5000, 0.2500,0.7500,0.7000,0.6000,0.5500"
    es = CK.EarlyStopping(max_patience=2)
    assert [es.update(v) for v in (1.0, 0.9, 0.95, 0.97, 0.99)] == [True, True, False, False, False]
    assert es.patience == 3 and es.should_stop and es.best == 0.9
    with pytest.raises(ValueError):
        CK.load_pretrained_model(None, str(tmp_path / "missing.tar"))


def test_prediction_refining_rules():
    """utils/models.py:325-332 / :366-376 restated literally (batch-1 loops) against the batched tensor version."""
    import torch
    from multi_task_breast_cancer_amd.inference import refine_predictions
    g = torch.Generator().manual_seed(0)
    n = 12
    seg_logits = [torch.randn(n, 1, 16, 16, generator=g) for _ in range(4)]
    seg_logits[-1][3] = -5.0                                   # no tumour pixel predicted
    seg_logits[-1][7] = -5.0
    cls_logits = [torch.randn(n, 3, generator=g)]
    cls_logits[0][5] = torch.tensor([0.0, 0.0, 9.0])           # predicted "normal" with a non-empty mask
    for rule_seg in (False, True):
        for rule_cls in (False, True):
            seg, cls = refine_predictions(cls_logits, seg_logits, rule_seg, rule_cls)
            for i in range(n):                                 # the reference's per-patient logic
                features_map = seg_logits[-1][i:i + 1]
                test_outputs = (torch.sigmoid(features_map) > .5).float().numpy()
                pred_class = torch.mean(torch.stack([c[i:i + 1] for c in cls_logits], dim=0), dim=0)
                pc = [pl.argmax() for pl in pred_class]
                if rule_seg and pc[0].item() == 2:
                    test_outputs[test_outputs > 0] = 0
                tumour = int(((torch.sigmoid(seg_logits[-1][i:i + 1]) > .5).float().numpy() == 1).sum())
                want_cls = 2 if (rule_cls and tumour == 0) else int(pc[0].item())
                assert np.array_equal(seg[i:i + 1].numpy(), test_outputs), (i, rule_seg, rule_cls)
                assert int(cls[i].item()) == want_cls, (i, rule_seg, rule_cls)

"""CPU oracle for the multi-task conv encoder-decoder training path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package
(`multi_task_breast_cancer_amd/`) may import from here; only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg do, and only as
the checker / the timed CPU baseline.

Parity status:
  * MTnnUNet, FocalLoss, loss aggregation, dice_score_from_tensor, seeding:
    PINNED against the reference imported in the build container
    (`oracle/make_goldens.py` -> `tests/golden/*.npz`).
  * MTUNetPlusPlus blocks and DiceLoss live in un-vendored monai==1.3.0
    (requirements.txt:4): restated from the published MONAI 1.3.0 semantics,
    anchored on closed-form known answers and on the MTnnUNet cell where the
    math coincides -- "parity unpinned" for those two pieces.
  * deterministic_oversampling: restated with pandas-1.5 semantics
    (the reference raises under the pandas 2.x present here), pinned on the
    reference's own data/mapping_curated_BUSI.csv class counts.
"""

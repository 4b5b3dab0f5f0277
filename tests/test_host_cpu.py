"""Host-side logic that needs no GPU: parameter naming / seeding parity of the product models, the config and
factory surface, oversampling (product vs oracle vs golden), bucket planning and batch sharding."""
import hashlib
import os

import numpy as np
import pytest
import torch

from multi_task_breast_cancer_amd import experiment_init as EI
from multi_task_breast_cancer_amd import miscellany as M
from multi_task_breast_cancer_amd import oversampling as OS
from multi_task_breast_cancer_amd.nets import MTnnUNet, MTUNetPlusPlus
from multi_task_breast_cancer_amd.trainer import (Bucket, dice_score_from_counts, global_permutation, plan_buckets,
                                                  shard_positions)
from oracle import torch_oracle as O
from oracle.oversampling_oracle import deterministic_oversampling_positions, scaling_factors


def _sha(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def test_product_mtnnunet_seeded_weights_bit_exact_vs_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "mtnnunet_seed1993_forward.npz"))
    M.seed_everything(1993)
    m = MTnnUNet(1, 1, 3)
    sd = m.state_dict()
    assert list(sd.keys()) == [str(n) for n in g["names"]]          # same keys, same order (checkpoints interchange)
    assert _sha(sd) == str(g["sha256"])


def test_product_unetpp_matches_oracle_restatement():
    M.seed_everything(7)
    m = MTUNetPlusPlus(in_channels=1, out_channels=1, n_classes=3, deep_supervision=True)
    O.seed_everything(7)
    o = O.OracleMTUNetPlusPlus(1, 1, 3, True)
    sd, od = m.state_dict(), o.state_dict()
    assert list(sd.keys()) == list(od.keys())
    assert _sha(sd) == _sha(od)
    assert sum(p.numel() for p in m.parameters()) == 14_927_455
    o.load_state_dict(sd)                                              # interchangeable checkpoints


def test_flat_slots_are_16_byte_aligned_and_disjoint():
    m = MTUNetPlusPlus(in_channels=1, out_channels=1, n_classes=3, deep_supervision=True)
    end = 0
    for name in m._order:
        s = m.slots[name]
        assert s.offset % 4 == 0 and s.offset >= end
        end = s.offset + s.numel
    assert m.flat_numel >= end and m.flat_numel % 4 == 0


def test_config_and_factory_surface(tmp_path):
    import yaml
    cfg = M.default_config()
    p = tmp_path / "config.yaml"
    p.write_text(yaml.safe_dump(cfg))
    cm, co, cl, ct, cd = M.load_config_file(str(p))
    assert cm["architecture"] == "MTnnUNet" and co["lr"] == 1e-4 and ct["alpha"] == 0.35 and cl["function"] == "DICE"
    model, opt, seg_c, cls_c, sched = EI.load_multitask_experiment_artefacts(cd, cm, co, cl, 0, str(tmp_path / "run"))
    assert (tmp_path / "run" / "model.txt").exists()
    assert opt.param_groups[0]["lr"] == 1e-4 and opt.param_groups[0]["eps"] == 1e-4      # experiment_init.py:187
    assert type(sched).__name__ == "ReduceLROnPlateau"
    assert type(seg_c).__name__ == "DiceLoss" and type(cls_c).__name__ == "FocalLoss"
    sched.step(1.0)
    cm2 = dict(cm, architecture="MTUNetPlusPlus")
    co2 = dict(co, scheduler="cosine")
    model2, _, _, _, sched2 = EI.load_multitask_experiment_artefacts(cd, cm2, co2, cl, 0, None)
    assert type(model2).__name__ == "MTUNetPlusPlus" and model2.deep_supervision
    assert type(sched2).__name__ == "CosineAnnealingLR"
    with pytest.raises(ValueError):
        EI.init_multitask_model("NoSuchNet")
    with pytest.raises(SystemExit):
        EI.init_criterion_segmentation("Hausdorff")


def test_oversampling_product_matches_oracle_and_golden(golden_dir):
    import pandas as pd
    classes = [str(c) for c in np.load(os.path.join(golden_dir, "curated_busi_classes.npz"))["classes"]]
    pos = OS.oversampled_positions(classes)
    assert pos.dtype == np.int64 and pos.tolist() == deterministic_oversampling_positions(classes)   # bit-exact indices
    assert dict(OS.compute_scaling_factor(classes)) == scaling_factors(classes) == {"benign": 2, "malignant": 3, "normal": 7}
    df = pd.DataFrame({"class": classes, "id": np.arange(len(classes))})
    out = OS.deterministic_oversampling(df)
    assert len(out) == 1384 and out["id"].tolist() == pos.tolist() and list(out.index) == list(range(1384))
    rng = np.random.default_rng(0)
    for _ in range(50):                                             # ragged / random class mixes
        k = int(rng.integers(1, 5))
        n = int(rng.integers(1, 60))
        cl = [f"c{int(i)}" for i in rng.integers(0, k, n)]
        assert OS.oversampled_positions(cl).tolist() == deterministic_oversampling_positions(cl)
    assert OS.oversampled_positions([]).tolist() == []


def test_plan_buckets_tile_the_flat_buffer():
    m = MTnnUNet(1, 1, 3)
    slots = [(m.slots[n].offset, m.slots[n].numel, i) for i, n in enumerate(reversed(m._order))]
    slots = [(m.slots[n].offset, m.slots[n].numel, len(m._order) - i) for i, n in enumerate(m._order)]
    for nb in (1, 3, 4, 8):
        bk = plan_buckets(slots, m.flat_numel, nb)
        assert 1 <= len(bk) <= nb
        spans = sorted((b.start, b.end) for b in bk)
        assert spans[0][0] == 0 and spans[-1][1] == m.flat_numel
        for (a0, a1), (b0, b1) in zip(spans[:-1], spans[1:]):
            assert a1 == b0
        assert [b.ready_op for b in bk] == sorted(b.ready_op for b in bk)
        for b in bk:       # ready_op covers every parameter inside the bucket
            assert b.ready_op == max(r for off, n, r in slots if b.start <= off < b.end)


def test_shards_reassemble_the_global_batch():
    perm = global_permutation(1384, seed=1993, epoch=3)
    assert sorted(perm.tolist()) == list(range(1384))
    assert (perm == global_permutation(1384, 1993, 3)).all() and not (perm == global_permutation(1384, 1993, 4)).all()
    G = 64
    for step in (0, 5):
        whole = shard_positions(perm, 0, 1, G, step)
        for world in (2, 4, 8):
            parts = [shard_positions(perm, r, world, G, step) for r in range(world)]
            assert np.concatenate(parts).tolist() == whole.tolist()
    with pytest.raises(ValueError):
        shard_positions(perm, 0, 3, 64, 0)


def test_dice_score_from_counts_semantics():
    assert dice_score_from_counts(torch.tensor([1.0, 1.0, 1.0], dtype=torch.float64)) == 0.5
    assert dice_score_from_counts(torch.tensor([0.0, 0.0, 0.0], dtype=torch.float64)) == 1.0
    assert dice_score_from_counts(torch.tensor([0.0, 4.0, 0.0], dtype=torch.float64)) == 0.0


def test_every_plan_switch_names_its_gpu_test():
    """switches.py: a plan switch is kept only while a -m gpu test runs its non-default arm against the oracle / emulation; the test it
    names must exist (file and function), and every switch the engine / nets / trainer read must be declared."""
    import os
    import re
    from multi_task_breast_cancer_amd import switches
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert len(switches.PLAN_SWITCHES) <= 10
    for name, (default, what, test) in switches.PLAN_SWITCHES.items():
        path, func = test.split("::")
        func = func.split("[")[0]
        src = open(os.path.join(root, path)).read()
        assert re.search(r"^def %s\(" % re.escape(func), src, re.M), (name, test)
    used = set()
    for fn in ("engine.py", "nets.py", "trainer.py"):
        used |= set(re.findall(r'_sw\.(?:flag|get)\("(MTBC_[A-Z0-9_]+)"\)', open(os.path.join(root, "multi_task_breast_cancer_amd", fn)).read()))
    assert used <= set(switches.PLAN_SWITCHES), used - set(switches.PLAN_SWITCHES)
    assert not (set(switches.REMOVED) & set(switches.PLAN_SWITCHES))

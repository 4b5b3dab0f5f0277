#!/usr/bin/env python3
"""Generate tests/golden/*.npz by importing the *reference's own* modules.

Runs ONLY in the build container (needs /root/reference).  The fixtures it
writes are data (inputs + expected outputs); nothing of the reference's source
is copied.  Importable there: src.models.multitask.MTnnUNet, src.utils.criterions,
src.utils.metrics, src.utils.miscellany.  NOT importable (monai / cv2 /
torchvision absent): MTUNetPlusPlus, experiment_init, the dataset package --
those pieces stay "parity unpinned" (oracle/__init__.py).

    python oracle/make_goldens.py            # rewrites tests/golden/
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np
import torch

REF = os.environ.get("MTBC_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

from src.models.multitask.MTnnUNet import MTnnUNet, LevelBlock          # noqa: E402
from src.utils.criterions import (FocalLoss,                            # noqa: E402
                                  apply_criterion_multitask_segmentation_classification)
from src.utils.metrics import dice_score_from_tensor                    # noqa: E402
from src.utils.miscellany import seed_everything                        # noqa: E402

from oracle.torch_oracle import dice_loss_sigmoid_sq                    # noqa: E402  (MONAI DiceLoss stand-in)


def state_sha256(sd) -> str:
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


def seg_nnunet_goldens() -> None:
    """(8) BASELINE.json configs[0], the single-task CPU plumbing case: the reference's nnUNet2021
    (src/models/segmentation/nnUNet.py) + apply_criterion_binary_segmentation (src/utils/criterions.py:27-49) + Adam
    (eps 1e-4, src/utils/experiment_init.py:187): seeded weights, forward, loss, weights after one step."""
    from src.models.segmentation.nnUNet import nnUNet2021
    from src.utils.criterions import apply_criterion_binary_segmentation
    seed_everything(1993)
    model = nnUNet2021(sequences=1, regions=1)
    sha0 = state_sha256(model.state_dict())
    torch.manual_seed(0)
    x = torch.rand(2, 1, 64, 64) * 255
    mask = (torch.rand(2, 1, 64, 64) > 0.7).float()
    model.train(True)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, eps=1e-4)
    opt.zero_grad(set_to_none=True)
    outs = model(x)
    loss = apply_criterion_binary_segmentation(dice_loss_sigmoid_sq, mask, outs, True)
    loss.backward()
    opt.step()
    sd = model.state_dict()
    probe = ["encoder1.ConvInNormLRelu1.Conv.weight", "bottleneck.ConvInNormLRelu2.Conv.weight", "upsample3.weight",
             "output4.1.weight", "output1.weight", "output1.bias"]
    np.savez_compressed(
        os.path.join(OUT, "seg_nnunet_step.npz"), sha256_init=np.array(sha0), x=x.numpy(), mask=mask.numpy(),
        out_means=np.array([o.mean().item() for o in outs]), out1=outs[-1].detach().numpy(), loss=np.array(loss.item()),
        probe_names=np.array(probe), **{f"after_{i}": sd[k].flatten()[:32].numpy() for i, k in enumerate(probe)},
        **{f"aftersum_{i}": np.array(sd[k].double().sum().item()) for i, k in enumerate(probe)})
    print("seg_nnunet_step.npz written; loss", loss.item())


def main() -> None:
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if "--only-seg" in sys.argv:
        seg_nnunet_goldens()
        return

    # ---- (1) MTnnUNet: seeded weights + forward ------------------------------------------
    seed_everything(1993)
    model = MTnnUNet(sequences=1, regions=1, n_classes=3)
    sd = model.state_dict()
    names = list(sd.keys())
    heads = np.stack([sd[k].flatten()[:4].numpy() if sd[k].numel() >= 4
                      else np.pad(sd[k].flatten().numpy(), (0, 4 - sd[k].numel())) for k in names])
    torch.manual_seed(0)
    x = torch.rand(2, 1, 64, 64) * 255
    model.train(True)
    logits, segs = model(x)
    np.savez_compressed(
        os.path.join(OUT, "mtnnunet_seed1993_forward.npz"),
        sha256=np.array(state_sha256(sd)), names=np.array(names), first4=heads,
        shapes=np.array([str(tuple(sd[k].shape)) for k in names]),
        x=x.numpy(), logits=logits[0].detach().numpy(),
        **{f"seg{i}": s.detach().numpy() for i, s in enumerate(segs)})

    # ---- (5) one full optimisation step on the same model/input ------------------------
    torch.manual_seed(1)
    mask = (torch.rand(2, 1, 64, 64) > 0.7).float()
    label = torch.tensor([[0.0], [2.0]])
    onehot = torch.nn.functional.one_hot(label.flatten().long(), 3).float()
    opt = torch.optim.Adam(model.parameters(), lr=1e-4, eps=1e-4)      # experiment_init.py:187
    focal = FocalLoss(alpha=1, gamma=2, reduction="mean")              # experiment_init.py:259
    opt.zero_grad(set_to_none=True)
    logits, segs = model(x)
    seg_l, cls_l = apply_criterion_multitask_segmentation_classification(
        dice_loss_sigmoid_sq, mask, segs, focal, onehot, logits, True)
    alpha = 0.35
    total = alpha * seg_l + (1 - alpha) * cls_l
    total.backward()
    probe = ["encoder1.ConvInNormLRelu1.Conv.weight", "decoder1.ConvInNormLRelu2.Conv.weight",
             "upsample5.weight", "upsample5.bias", "output4.0.weight", "output1.weight", "output1.bias",
             "classifier.0.Conv.weight", "classifier.3.weight", "classifier.5.bias",
             "process_encoder_5.Conv.weight", "bottleneck.ConvInNormLRelu2.Conv.weight"]
    params = dict(model.named_parameters())
    grads = {f"grad::{k}": params[k].grad.flatten()[:16].clone().numpy() for k in probe}
    gnorm = {f"gnorm::{k}": np.array(params[k].grad.double().norm().item()) for k in probe}
    opt.step()
    after = {f"after::{k}": params[k].detach().flatten()[:16].clone().numpy() for k in probe}
    np.savez_compressed(
        os.path.join(OUT, "mtnnunet_step.npz"), mask=mask.numpy(), label=label.numpy(),
        alpha=np.array(alpha), total=np.array(total.item()), seg=np.array(seg_l.item()),
        cls=np.array(cls_l.item()), probe=np.array(probe), **grads, **gnorm, **after)

    # ---- (2) FocalLoss known answers ------------------------------------------------------
    fx = torch.tensor([[2.0, -1.0, 0.5], [0.1, 0.2, 0.3]])
    ft = torch.tensor([[1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    g = torch.Generator().manual_seed(7)
    fx2 = torch.randn(16, 3, generator=g) * 3
    ft2 = torch.nn.functional.one_hot(torch.randint(0, 3, (16,), generator=g), 3).float()
    ft3 = torch.softmax(torch.randn(16, 3, generator=g), dim=1)       # soft targets
    w = torch.tensor([0.2, 0.3, 0.5])
    np.savez_compressed(
        os.path.join(OUT, "focal.npz"), x1=fx.numpy(), t1=ft.numpy(), y1=np.array(focal(fx, ft).item()),
        x2=fx2.numpy(), t2=ft2.numpy(), y2=np.array(focal(fx2, ft2).item()),
        t3=ft3.numpy(), y3=np.array(focal(fx2, ft3).item()),
        w=w.numpy(), y4=np.array(FocalLoss(weight=w)(fx2, ft2).item()),
        y5=np.array(FocalLoss(gamma=2, reduction="sum")(fx2, ft2).item()))

    # ---- (3) loss aggregation ------------------------------------------------------------
    g = torch.Generator().manual_seed(11)
    seg_list = [torch.randn(3, 1, 16, 16, generator=g) for _ in range(4)]
    cls_list = [torch.randn(3, 3, generator=g)]
    gt = (torch.rand(3, 1, 16, 16, generator=g) > 0.5).float()
    oh = torch.nn.functional.one_hot(torch.tensor([0, 1, 2]), 3).float()
    res = {}
    for iw in (True, False):
        s, c = apply_criterion_multitask_segmentation_classification(
            dice_loss_sigmoid_sq, gt, seg_list, focal, oh, cls_list, iw)
        res[f"seg_iw{int(iw)}"] = np.array(s.item())
        res[f"cls_iw{int(iw)}"] = np.array(c.item())
    s, c = apply_criterion_multitask_segmentation_classification(
        dice_loss_sigmoid_sq, gt, seg_list[3], focal, oh, cls_list[0], True)
    res["seg_tensor"] = np.array(s.item())
    res["cls_tensor"] = np.array(c.item())
    np.savez_compressed(os.path.join(OUT, "criterion_aggregation.npz"), gt=gt.numpy(), onehot=oh.numpy(),
                        cls0=cls_list[0].numpy(), **{f"seg{i}": t.numpy() for i, t in enumerate(seg_list)}, **res)

    # ---- (4) dice_score_from_tensor --------------------------------------------------------
    g = torch.Generator().manual_seed(3)
    gts = (torch.rand(4, 1, 32, 32, generator=g) > 0.6)
    sgs = (torch.rand(4, 1, 32, 32, generator=g) > 0.5)
    np.savez_compressed(
        os.path.join(OUT, "dice_score.npz"),
        k1=np.array(float(dice_score_from_tensor(torch.tensor([[1., 1.], [0., 0.]]),
                                                 torch.tensor([[True, False], [True, False]])))),
        k_empty=np.array(float(dice_score_from_tensor(torch.zeros(2, 2), torch.zeros(2, 2).bool()))),
        k_fp_only=np.array(float(dice_score_from_tensor(torch.zeros(2, 2), torch.ones(2, 2).bool()))),
        gt=gts.numpy(), seg=sgs.numpy(), k_rand=np.array(float(dice_score_from_tensor(gts.float(), sgs))))

    # ---- (6) LevelBlock cell golden ------------------------------------------------------
    torch.manual_seed(5)
    blk = LevelBlock(3, 8, 8)
    xb = torch.randn(2, 3, 16, 16)
    np.savez_compressed(os.path.join(OUT, "levelblock.npz"), x=xb.numpy(),
                        w1=blk.ConvInNormLRelu1.Conv.weight.detach().numpy(),
                        w2=blk.ConvInNormLRelu2.Conv.weight.detach().numpy(), y=blk(xb).detach().numpy())

    # ---- (7) oversampling: class counts of the reference's curated mapping ----------------
    import csv
    with open(os.path.join(REF, "data", "mapping_curated_BUSI.csv")) as fh:
        rows = list(csv.reader(fh, delimiter=";"))
    header, body = rows[0], rows[1:]
    ci = header.index("class")
    classes = [r[ci] for r in body]
    np.savez_compressed(os.path.join(OUT, "curated_busi_classes.npz"), classes=np.array(classes))
    seg_nnunet_goldens()
    print("goldens written to", OUT)


if __name__ == "__main__":
    main()

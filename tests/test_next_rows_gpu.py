"""SURVEY 8(f) rows N2 / N3 / N4 on the GPU: augmentation gather, validation epoch, checkpoint interchange."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from multi_task_breast_cancer_amd import augment as AUG  # noqa: E402
from multi_task_breast_cancer_amd import checkpoint as CK  # noqa: E402
from multi_task_breast_cancer_amd.miscellany import seed_everything  # noqa: E402
from multi_task_breast_cancer_amd.nets import MTnnUNet, MTUNetPlusPlus  # noqa: E402
from multi_task_breast_cancer_amd.optim import FusedAdam  # noqa: E402
from multi_task_breast_cancer_amd.trainer import FusedEvalStep, FusedTrainStep, validate_one_epoch  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

DEV = torch.device("cuda:0")


@pytest.mark.parametrize("H,W", [(64, 64), (48, 80), (256, 256)])
def test_flip_rotate_matches_torchvision_restatement(H, W):
    g = torch.Generator().manual_seed(3)
    n = 9
    stack = torch.rand(n, 2, H, W, generator=g)
    stack[:, 0] = (stack[:, 0] > 0.6).float()                       # a mask plane and an image plane
    angles = [0.0, 90.0, -90.0, 180.0, 37.5, -123.4, 359.0, 12.0, -300.25]
    hf = [0, 1, 0, 1, 1, 0, 1, 0, 1]
    vf = [0, 0, 1, 1, 0, 1, 1, 0, 0]
    want = O.tv_flip_rotate(stack, angles, hf, vf)
    got = AUG.flip_rotate(stack.to(DEV), AUG.params_from(angles, hf, vf)).cpu()
    # identical gather except where the source coordinate sits on a .5 tie and the fp32 products round differently
    # (bmm vs fused multiply-add): a handful of pixels per image at most
    diff = (got != want).float().mean(dim=(1, 2, 3))
    assert diff.max().item() < 2e-3, diff
    exact = [0, 3]                                                  # angle 0 / 180 with flips: pure index permutations
    for i in exact:
        assert torch.equal(got[i], want[i]), i
    # mask and image move together: both planes come from the same source pixel
    idx = AUG.flip_rotate(torch.arange(n * H * W, dtype=torch.float32).view(n, 1, H, W).repeat(1, 2, 1, 1).to(DEV),
                          AUG.params_from(angles, hf, vf)).cpu()
    assert torch.equal(idx[:, 0], idx[:, 1])
    assert set(torch.unique(got[:, 0]).tolist()) <= {0.0, 1.0}     # nearest: a binary mask stays binary


def test_random_params_distribution():
    p = AUG.random_params(4000, np.random.default_rng(0))
    assert abs(p[:, 2].mean().item() - 0.5) < 0.03 and abs(p[:, 3].mean().item() - 0.5) < 0.03
    assert torch.allclose(p[:, 0] ** 2 + p[:, 1] ** 2, torch.ones(4000), atol=1e-6)


def test_validation_epoch_matches_oracle():
    """validate_one_epoch (training_multitask.py:119-159): losses, batch Dice, accuracy and weighted F1."""
    from sklearn.metrics import accuracy_score, f1_score
    seed_everything(11)
    prod = MTUNetPlusPlus(in_channels=1, out_channels=1, n_classes=3, deep_supervision=True)
    O.seed_everything(11)
    ref = O.build_oracle_model("MTUNetPlusPlus", 1, 1, 3, True)
    ref.load_state_dict(prod.state_dict())
    prod = prod.to(DEV)
    batches = []
    for s in range(3):
        img, mask, label = O.synthetic_batch(4, 64, 64, seed=40 + s)
        batches.append({"image": img, "mask": mask, "label": label})
    step = FusedEvalStep(prod, alpha=0.5, inversely_weighted=True)
    got = validate_one_epoch(step, batches, DEV)
    # oracle: the reference's loop
    tot = seg_s = cls_s = dice_s = 0.0
    gts, preds = [], []
    ref.train(False)
    with torch.no_grad():
        for b in batches:
            onehot = torch.nn.functional.one_hot(b["label"].flatten().long(), 3).float()
            logits, outs = ref(b["image"])
            seg, cls = O.multitask_losses(outs, b["mask"], logits, onehot, True)
            tot += (0.5 * seg + 0.5 * cls).item(); seg_s += seg.item(); cls_s += cls.item()
            d = O.dice_score_from_tensor(b["mask"], torch.sigmoid(outs[-1]) > .5)
            dice_s += float(d)
            lg = torch.mean(torch.stack(logits, dim=0), dim=0)
            preds += torch.softmax(lg, dim=1).argmax(dim=1).tolist()
            gts += onehot.argmax(dim=1).tolist()
    want = (tot / 3, dice_s / 3, accuracy_score(gts, preds), f1_score(gts, preds, labels=[0, 1, 2], average="weighted"),
            seg_s / 3, cls_s / 3)
    for g, w in zip(got, want):
        assert abs(g - w) < 1e-4, (got, want)


def test_checkpoint_interchanges_with_torch_adam(tmp_path):
    """A checkpoint written here resumes under torch.optim.Adam on the oracle model (and vice versa): same model keys,
    optimizer state in torch.optim.Adam's layout; the next step then agrees."""
    seed_everything(5)
    prod = MTnnUNet(1, 1, 3).to(DEV)
    opt = FusedAdam(prod, lr=1e-4, eps=1e-4)
    step = FusedTrainStep(prod, opt, alpha=0.5)
    img, mask, label = O.synthetic_batch(2, 64, 64, seed=1)
    step(img.to(DEV), mask.to(DEV), label.to(DEV))
    path = str(tmp_path / "model_fold_0")
    CK.save_checkpoint(path, 3, prod, opt, 0.123)
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler", "val_loss"}
    # -> into the CPU oracle + torch.optim.Adam
    ref = O.build_oracle_model("MTnnUNet", 1, 1, 3, True)
    ref.load_state_dict(ck["model_state_dict"])
    ropt = O.make_adam(ref, 1e-4)
    ropt.load_state_dict(ck["optimizer_state_dict"])
    img2, mask2, label2 = O.synthetic_batch(2, 64, 64, seed=2)
    O.train_step(ref, ropt, img2, mask2, label2, 0.5, True, 3)
    step(img2.to(DEV), mask2.to(DEV), label2.to(DEV))
    for (k, a), (_, b) in zip(prod.state_dict().items(), ref.state_dict().items()):
        assert (a.cpu() - b).abs().max().item() < 2.5e-4, k            # second Adam step from the restored moments
    # <- and back: a torch.optim.Adam checkpoint into a fresh HIP model / FusedAdam
    torch.save({"epoch": 4, "model_state_dict": ref.state_dict(), "optimizer_state_dict": ropt.state_dict(),
                "scheduler": "scheduler", "val_loss": 0.1}, path)
    seed_everything(99)
    fresh = MTnnUNet(1, 1, 3).to(DEV)
    fopt = FusedAdam(fresh, lr=1e-4, eps=1e-4)
    CK.load_pretrained_model(fresh, path, optimizer=fopt)
    assert fopt.step_count == 2
    for (k, a), (_, b) in zip(fresh.state_dict().items(), ref.state_dict().items()):
        assert torch.equal(a.cpu(), b), k
    sd = fopt.state_dict()["state"]
    rsd = ropt.state_dict()["state"]
    for i in rsd:
        assert torch.allclose(sd[i]["exp_avg"].cpu(), rsd[i]["exp_avg"]) and torch.allclose(sd[i]["exp_avg_sq"].cpu(), rsd[i]["exp_avg_sq"])


@pytest.mark.parametrize("n_classes,criterion", [(3, "CE"), (2, "CE")])
def test_dropin_loop_with_torch_classification_criteria(n_classes, criterion):
    """experiment_init.py:235-263: binary -> BCEWithLogitsLoss, otherwise CE / Focal.  The torch criteria run on the HIP
    model's outputs through autograd in the reference's loop (training_multitask.py:87-103); one step vs the oracle."""
    from multi_task_breast_cancer_amd import criterions as CR
    from multi_task_breast_cancer_amd.experiment_init import init_criterion_classification, init_criterion_segmentation
    seed_everything(3)
    prod = MTnnUNet(1, 1, n_classes)
    ref = O.build_oracle_model("MTnnUNet", 1, 1, n_classes, True)
    ref.load_state_dict(prod.state_dict())
    prod = prod.to(DEV)
    cls_c = init_criterion_classification(n_classes=n_classes, classes_weighted=None, classification_criterion=criterion)
    assert isinstance(cls_c, torch.nn.BCEWithLogitsLoss if n_classes == 2 else torch.nn.CrossEntropyLoss)
    seg_c = init_criterion_segmentation("DICE")
    img, mask, label = O.synthetic_batch(2, 64, 64, seed=8)
    if n_classes == 2:
        label = (label > 0).float()
        target_dev, target_cpu = label.to(DEV), label
    else:
        onehot = torch.nn.functional.one_hot(label.flatten().long(), 3).float()
        target_dev, target_cpu = onehot.to(DEV), onehot
    opt = FusedAdam(prod, lr=1e-4, eps=1e-4)
    opt.zero_grad(set_to_none=True)
    logits, outs = prod(img.to(DEV))
    seg, cls = CR.apply_criterion_multitask_segmentation_classification(seg_c, mask.to(DEV), outs, cls_c, target_dev, logits, True)
    total = 0.5 * seg + 0.5 * cls
    total.backward()
    opt.step()
    rl, ro = ref(img)
    rseg = sum(O.dice_loss_sigmoid_sq(o, mask) / (j + 1) for j, o in enumerate(reversed(ro)))
    rcls = sum((torch.nn.BCEWithLogitsLoss() if n_classes == 2 else torch.nn.CrossEntropyLoss())(l, target_cpu) for l in reversed(rl))
    rtot = 0.5 * rseg + 0.5 * rcls
    ropt = O.make_adam(ref, 1e-4)
    ropt.zero_grad()
    rtot.backward()
    ropt.step()
    assert abs(total.item() - rtot.item()) < 1e-4
    for (k, a), (_, b) in zip(prod.state_dict().items(), ref.state_dict().items()):
        assert (a.cpu() - b).abs().max().item() < 2.0e-4, k


def test_factory_torch_optimizers_step_the_flat_parameters():
    """experiment_init.py:186-195: SGD / AdamW are torch optimizers over the HIP model's parameters (views of the flat
    buffer): one drop-in step moves the weights exactly as the same optimizer moves the oracle's."""
    from multi_task_breast_cancer_amd import criterions as CR
    from multi_task_breast_cancer_amd.experiment_init import init_optimizer
    for name, make_ref in (("SGD", lambda ps: torch.optim.SGD(ps, lr=1e-3, momentum=0.9, nesterov=True)),
                           ("AdamW", lambda ps: torch.optim.AdamW(ps, lr=1e-3))):
        seed_everything(4)
        prod = MTnnUNet(1, 1, 3)
        ref = O.build_oracle_model("MTnnUNet", 1, 1, 3, True)
        ref.load_state_dict(prod.state_dict())
        prod = prod.to(DEV)
        opt = init_optimizer(prod, name, 1e-3)
        assert type(opt).__name__ == name
        img, mask, label = O.synthetic_batch(2, 64, 64, seed=2)
        onehot = torch.nn.functional.one_hot(label.flatten().long(), 3).float()
        opt.zero_grad(set_to_none=True)
        logits, outs = prod(img.to(DEV))
        seg, cls = CR.apply_criterion_multitask_segmentation_classification(CR.DiceLoss(), mask.to(DEV), outs, CR.FocalLoss(), onehot.to(DEV), logits, True)
        (0.5 * seg + 0.5 * cls).backward()
        opt.step()
        ropt = make_ref(ref.parameters())
        rl, ro = ref(img)
        rseg, rcls = O.multitask_losses(ro, mask, rl, onehot, True)
        ropt.zero_grad()
        (0.5 * rseg + 0.5 * rcls).backward()
        ropt.step()
        for (k, a), (_, b) in zip(prod.state_dict().items(), ref.state_dict().items()):
            d = (a.cpu() - b).abs()
            if name == "SGD":                       # update = lr * (1.9 g): as exact as the gradient
                assert d.max().item() < 5e-5, (name, k, d.max().item())
            else:                                   # AdamW, eps 1e-8, step 1: update = lr * sign(g) -- an element whose
                assert d.max().item() <= 2.1e-3     # gradient is rounding noise may go the other way (2 lr), few do
                assert (d > 1e-4).float().mean().item() < 0.02, (name, k)
        # the step landed in the flat buffer (parameters are views of it)
        assert torch.equal(prod._param_view(prod._order[0]), dict(prod.named_parameters())[prod._order[0]].detach())


def test_hip_criterions_reproduce_the_reference_aggregation_goldens(golden_dir):
    """tests/golden/criterion_aggregation.npz holds the reference's apply_criterion_multitask_segmentation_classification
    (criterions.py:52-76) results for lists (inversely_weighted True AND False) and for the tensor branch; here the HIP
    DiceLoss / FocalLoss modules go through the package's mirror of that function -- forward values and, by autograd,
    the gradients of the weighted sum against the oracle's."""
    import os
    from multi_task_breast_cancer_amd import criterions as CR
    g = np.load(os.path.join(golden_dir, "criterion_aggregation.npz"))
    t = lambda k: torch.from_numpy(g[k]).to(DEV)
    segs = [t(f"seg{i}").requires_grad_(True) for i in range(4)]
    cls0, gt, onehot = t("cls0").requires_grad_(True), t("gt"), t("onehot")
    dice, focal = CR.DiceLoss(), CR.FocalLoss(alpha=1, gamma=2)
    for iw in (True, False):
        s, c = CR.apply_criterion_multitask_segmentation_classification(dice, gt, segs, focal, onehot, [cls0], iw)
        assert abs(s.item() - float(g[f"seg_iw{int(iw)}"])) < 1e-5, (iw, s.item())
        assert abs(c.item() - float(g[f"cls_iw{int(iw)}"])) < 1e-5, (iw, c.item())
        for x in segs + [cls0]:
            x.grad = None
        (0.35 * s + 0.65 * c).backward()
        rs = [x.detach().cpu().clone().requires_grad_(True) for x in segs]
        rc = cls0.detach().cpu().clone().requires_grad_(True)
        so, co = O.multitask_losses(rs, gt.cpu(), [rc], onehot.cpu(), iw)
        (0.35 * so + 0.65 * co).backward()
        for a, b in zip(segs + [cls0], rs + [rc]):
            assert (a.grad.cpu() - b.grad).abs().max().item() < 1e-7 + 1e-4 * b.grad.abs().max().item()
    s, c = CR.apply_criterion_multitask_segmentation_classification(dice, gt, segs[3], focal, onehot, cls0, True)
    assert abs(s.item() - float(g["seg_tensor"])) < 1e-5 and abs(c.item() - float(g["cls_tensor"])) < 1e-5


@pytest.mark.parametrize("arch", ["MTnnUNet", "MTUNetPlusPlus"])
def test_fused_step_without_inverse_weighting_matches_oracle(arch):
    """config.yaml `loss.inversely_weighted: False` (criterions.py:64-66: every deep-supervision head weighs 1) on the
    fused HIP step against the oracle's step."""
    seed_everything(3)
    prod = MTnnUNet(1, 1, 3) if arch == "MTnnUNet" else MTUNetPlusPlus(in_channels=1, out_channels=1, n_classes=3, deep_supervision=True)
    O.seed_everything(3)
    ref = O.build_oracle_model(arch, 1, 1, 3, True)
    ref.load_state_dict(prod.state_dict())
    prod = prod.to(DEV)
    img, mask, label = O.synthetic_batch(2, 64, 64, seed=8)
    step = FusedTrainStep(prod, FusedAdam(prod, lr=1e-4, eps=1e-4), alpha=0.35, inversely_weighted=False)
    got = step(img.to(DEV), mask.to(DEV), label.to(DEV)).cpu()
    total, seg, cls, _, _ = O.train_step(ref, O.make_adam(ref, 1e-4), img, mask, label, 0.35, False, 3)
    assert abs(got[0].item() - total.item()) < 1e-4 and abs(got[1].item() - seg.item()) < 1e-4 and abs(got[2].item() - cls.item()) < 1e-4
    for (k, a), (_, b) in zip(prod.state_dict().items(), ref.state_dict().items()):
        assert (a.cpu() - b).abs().max().item() < 2.0e-4, k


def test_validation_epoch_with_the_binary_head_matches_oracle():
    """n_classes == 2: ONE logit, BCEWithLogits (experiment_init.py:242), predictions sigmoid > .5 against the {0,1} label
    (training_multitask.py:53-61), f1 still asked for labels [0,1,2] (:155)."""
    from sklearn.metrics import accuracy_score, f1_score
    seed_everything(13)
    prod = MTUNetPlusPlus(in_channels=1, out_channels=1, n_classes=2, deep_supervision=True)
    O.seed_everything(13)
    ref = O.build_oracle_model("MTUNetPlusPlus", 1, 1, 2, True)
    ref.load_state_dict(prod.state_dict())
    prod = prod.to(DEV)
    batches = []
    for s in range(3):
        img, mask, label = O.synthetic_batch(4, 64, 64, seed=70 + s)
        batches.append({"image": img, "mask": mask, "label": (label > 0).float()})      # benign / malignant -> {0, 1}
    step = FusedEvalStep(prod, alpha=0.35, inversely_weighted=True, n_classes=2)
    got = validate_one_epoch(step, batches, DEV)
    tot = seg_s = cls_s = dice_s = 0.0
    gts, preds = [], []
    ref.train(False)
    bce = torch.nn.BCEWithLogitsLoss()
    with torch.no_grad():
        for b in batches:
            logits, outs = ref(b["image"])
            seg = torch.sum(torch.stack([O.dice_loss_sigmoid_sq(s_, b["mask"]) / (j + 1) for j, s_ in enumerate(reversed(outs))]))
            cls = torch.sum(torch.stack([bce(c_, b["label"].view(-1, 1)) for c_ in reversed(logits)]))
            tot += (0.35 * seg + 0.65 * cls).item(); seg_s += seg.item(); cls_s += cls.item()
            dice_s += float(O.dice_score_from_tensor(b["mask"], torch.sigmoid(outs[-1]) > .5))
            lg = torch.mean(torch.stack(logits, dim=0), dim=0)
            preds += (torch.sigmoid(lg[:, 0]) > .5).double().tolist()
            gts += b["label"].flatten().tolist()
    want = (tot / 3, dice_s / 3, accuracy_score(gts, preds), f1_score(gts, preds, labels=[0, 1, 2], average="weighted", zero_division=0),
            seg_s / 3, cls_s / 3)
    for g_, w_ in zip(got, want):
        assert abs(g_ - w_) < 1e-4, (got, want)
    with pytest.raises(NotImplementedError):
        FusedEvalStep(prod, alpha=0.35, n_classes=5)


def test_validation_epoch_with_cross_entropy_matches_oracle_and_shares_the_training_plan():
    """`classification_criterion: CE` (experiment_init.py:232-262): the reference validates with the criterion it trains with, and
    `scheduler.step(val_loss)` runs on that value (training_multitask.py:119-159, 234-237).  FusedEvalStep(cls_criterion="CE") reports
    CrossEntropy -- not Focal -- losses, and with the training step's (alpha, weighting, criterion) it runs on the SAME compiled plan at equal
    (N, H, W): no second activation arena."""
    seed_everything(19)
    prod = MTUNetPlusPlus(in_channels=1, out_channels=1, n_classes=3, deep_supervision=True)
    O.seed_everything(19)
    ref = O.build_oracle_model("MTUNetPlusPlus", 1, 1, 3, True)
    ref.load_state_dict(prod.state_dict())
    prod = prod.to(DEV)
    batches = []
    for s in range(2):
        img, mask, label = O.synthetic_batch(4, 64, 64, seed=90 + s)
        batches.append({"image": img, "mask": mask, "label": label})
    step = FusedEvalStep(prod, alpha=0.5, inversely_weighted=True, cls_criterion="CE")
    got = validate_one_epoch(step, batches, DEV)
    tot = seg_s = cls_s = 0.0
    ref.train(False)
    ce = torch.nn.CrossEntropyLoss()
    with torch.no_grad():
        for b in batches:
            onehot = torch.nn.functional.one_hot(b["label"].flatten().long(), 3).float()
            logits, outs = ref(b["image"])
            seg = torch.sum(torch.stack([O.dice_loss_sigmoid_sq(s_, b["mask"]) / (j + 1) for j, s_ in enumerate(reversed(outs))]))
            cls = torch.sum(torch.stack([ce(c_, onehot) for c_ in reversed(logits)]))
            tot += (0.5 * seg + 0.5 * cls).item(); seg_s += seg.item(); cls_s += cls.item()
    assert abs(got[0] - tot / 2) < 1e-4 and abs(got[4] - seg_s / 2) < 1e-4 and abs(got[5] - cls_s / 2) < 1e-4, (got, tot / 2, cls_s / 2)
    # the Focal evaluation of the same batches reports another classification loss (gamma 2 down-weights easy samples)
    focal = validate_one_epoch(FusedEvalStep(prod, alpha=0.5, inversely_weighted=True), batches, DEV)
    assert focal[5] < got[5] - 1e-3
    # one compiled plan for training and evaluation with the same criterion
    n_before = len(prod._steps)
    train = FusedTrainStep(prod, FusedAdam(prod, lr=1e-4, eps=1e-4), alpha=0.5, inversely_weighted=True, cls_criterion="CE")
    b = batches[0]
    train(b["image"].to(DEV), b["mask"].to(DEV), b["label"].to(DEV))
    assert len(prod._steps) == n_before
    with pytest.raises(NotImplementedError):
        FusedEvalStep(prod, alpha=0.5, cls_criterion="CE", focal_weight=torch.ones(3, device=DEV))
    with pytest.raises(ValueError):
        FusedEvalStep(prod, alpha=0.5, cls_criterion="BCE")

"""Pin the CPU oracle against fixtures generated from the reference's own modules
(oracle/make_goldens.py).  CPU only."""
import hashlib
import os

import numpy as np
import pytest
import torch

from oracle import torch_oracle as O
from oracle.oversampling_oracle import deterministic_oversampling_positions, scaling_factors


def _sha(sd):
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(v.detach().cpu().contiguous().numpy().tobytes())
    return h.hexdigest()


@pytest.fixture(scope="module")
def seeded_nnunet():
    O.seed_everything(1993)
    return O.OracleMTnnUNet(1, 1, 3)


def test_nnunet_seeded_state_dict_matches_reference(golden_dir, seeded_nnunet):
    g = np.load(os.path.join(golden_dir, "mtnnunet_seed1993_forward.npz"))
    sd = seeded_nnunet.state_dict()
    assert list(sd.keys()) == [str(n) for n in g["names"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in g["shapes"]]
    assert _sha(sd) == str(g["sha256"])          # bit-exact initial weights
    assert sum(p.numel() for p in seeded_nnunet.parameters()) == 15_819_799


def test_nnunet_forward_matches_reference(golden_dir, seeded_nnunet):
    g = np.load(os.path.join(golden_dir, "mtnnunet_seed1993_forward.npz"))
    seeded_nnunet.train(True)
    with torch.no_grad():
        logits, segs = seeded_nnunet(torch.from_numpy(g["x"]))
    np.testing.assert_allclose(logits[0].numpy(), g["logits"], rtol=0, atol=1e-6)
    for i, s in enumerate(segs):
        np.testing.assert_allclose(s.numpy(), g[f"seg{i}"], rtol=0, atol=2e-6)


def test_nnunet_full_step_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "mtnnunet_step.npz"))
    f = np.load(os.path.join(golden_dir, "mtnnunet_seed1993_forward.npz"))
    O.seed_everything(1993)
    model = O.OracleMTnnUNet(1, 1, 3)
    opt = O.make_adam(model, lr=1e-4)
    total, seg, cls, _, _ = O.train_step(model, opt, torch.from_numpy(f["x"]), torch.from_numpy(g["mask"]),
                                         torch.from_numpy(g["label"]), float(g["alpha"]), True, 3)
    assert abs(total.item() - float(g["total"])) < 1e-6
    assert abs(seg.item() - float(g["seg"])) < 1e-6
    assert abs(cls.item() - float(g["cls"])) < 1e-6
    params = dict(model.named_parameters())
    for k in [str(p) for p in g["probe"]]:
        np.testing.assert_allclose(params[k].grad.flatten()[:16].numpy(), g[f"grad::{k}"], rtol=1e-4, atol=1e-7)
        np.testing.assert_allclose(params[k].detach().flatten()[:16].numpy(), g[f"after::{k}"], rtol=0, atol=1e-7)


def test_focal_known_answers(golden_dir):
    g = np.load(os.path.join(golden_dir, "focal.npz"))
    t = torch.from_numpy
    assert abs(O.focal_loss_soft(t(g["x1"]), t(g["t1"])).item() - 0.20617523789405823) < 1e-7   # SURVEY A8 KAT
    assert abs(O.focal_loss_soft(t(g["x1"]), t(g["t1"])).item() - float(g["y1"])) < 1e-7
    assert abs(O.focal_loss_soft(t(g["x2"]), t(g["t2"])).item() - float(g["y2"])) < 1e-6
    assert abs(O.focal_loss_soft(t(g["x2"]), t(g["t3"])).item() - float(g["y3"])) < 1e-6
    assert abs(O.focal_loss_soft(t(g["x2"]), t(g["t2"]), weight=t(g["w"])).item() - float(g["y4"])) < 1e-6


def test_loss_aggregation_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "criterion_aggregation.npz"))
    t = torch.from_numpy
    segs = [t(g[f"seg{i}"]) for i in range(4)]
    for iw in (True, False):
        s, c = O.multitask_losses(segs, t(g["gt"]), [t(g["cls0"])], t(g["onehot"]), iw)
        assert abs(s.item() - float(g[f"seg_iw{int(iw)}"])) < 1e-6
        assert abs(c.item() - float(g[f"cls_iw{int(iw)}"])) < 1e-6
    s, c = O.multitask_losses(segs[3], t(g["gt"]), t(g["cls0"]), t(g["onehot"]), True)
    assert abs(s.item() - float(g["seg_tensor"])) < 1e-6
    assert abs(c.item() - float(g["cls_tensor"])) < 1e-6


def test_loss_nan_exits():
    bad = [torch.full((1, 1, 4, 4), float("nan"))] * 4
    with pytest.raises(SystemExit):
        O.multitask_losses(bad, torch.zeros(1, 1, 4, 4), [torch.zeros(1, 3)], torch.tensor([[1., 0., 0.]]), True)


def test_dice_score_known_answers(golden_dir):
    g = np.load(os.path.join(golden_dir, "dice_score.npz"))
    assert float(O.dice_score_from_tensor(torch.tensor([[1., 1.], [0., 0.]]),
                                          torch.tensor([[True, False], [True, False]]))) == float(g["k1"]) == 0.5
    assert float(O.dice_score_from_tensor(torch.zeros(2, 2), torch.zeros(2, 2).bool())) == float(g["k_empty"]) == 1.0
    assert float(O.dice_score_from_tensor(torch.zeros(2, 2), torch.ones(2, 2).bool())) == float(g["k_fp_only"]) == 0.0
    v = float(O.dice_score_from_tensor(torch.from_numpy(g["gt"]).float(), torch.from_numpy(g["seg"])))
    assert abs(v - float(g["k_rand"])) < 1e-12


def test_dice_loss_closed_form():
    # all-zero logits & zero target: p=.5, I=0, D=.25*HW -> f = 1 - 1/(.25 HW + 1)   
    for hw in (4, 16, 64):
        x = torch.zeros(2, 1, hw, hw)
        want = 1.0 - 1.0 / (0.25 * hw * hw + 1.0)
        assert abs(O.dice_loss_sigmoid_sq(x, torch.zeros_like(x)).item() - want) < 1e-6
    # saturated perfect prediction: p->1 on target, 0 elsewhere -> f = 1 - (2A+1)/(2A+1) = 0
    tgt = torch.zeros(1, 1, 8, 8)
    tgt[..., 2:6, 2:6] = 1
    x = (tgt * 2 - 1) * 40.0
    assert abs(O.dice_loss_sigmoid_sq(x, tgt).item()) < 1e-6


def test_levelblock_cell_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "levelblock.npz"))
    blk = O._level(3, 8, 8)
    with torch.no_grad():
        blk.ConvInNormLRelu1.Conv.weight.copy_(torch.from_numpy(g["w1"]))
        blk.ConvInNormLRelu2.Conv.weight.copy_(torch.from_numpy(g["w2"]))
        y = blk(torch.from_numpy(g["x"]))
    np.testing.assert_allclose(y.numpy(), g["y"], rtol=0, atol=1e-6)


def test_unetpp_restatement_shape_and_param_count():
    O.seed_everything(1993)
    m = O.OracleMTUNetPlusPlus(1, 1, 3, deep_supervision=True)
    assert sum(p.numel() for p in m.parameters()) == 14_927_455        # SURVEY §6 / BASELINE.md
    sd = m.state_dict()
    for k in ("conv_0_0.conv_0.conv.weight", "conv_0_0.conv_0.adn.N.weight", "conv_1_0.convs.conv_1.conv.bias",
              "upcat_0_4.upsample.deconv.weight", "upcat_2_2.convs.conv_0.adn.N.bias", "final_conv_0_4.bias",
              "process_level_3.convs.conv_0.conv.weight", "classifier.0.conv_1.conv.weight", "classifier.5.weight"):
        assert k in sd, k
    assert tuple(sd["upcat_0_1.upsample.deconv.weight"].shape) == (48, 48, 2, 2)      # halves=False
    assert tuple(sd["upcat_1_1.upsample.deconv.weight"].shape) == (96, 48, 2, 2)      # halves=True
    assert tuple(sd["upcat_0_4.convs.conv_0.conv.weight"].shape) == (24, 144, 3, 3)
    with torch.no_grad():
        logits, segs = m(torch.rand(1, 1, 32, 32) * 255)
    assert logits[0].shape == (1, 3) and all(s.shape == (1, 1, 32, 32) for s in segs)
    m2 = O.OracleMTUNetPlusPlus(1, 1, 3, deep_supervision=False)
    with torch.no_grad():
        lg, sg = m2(torch.rand(1, 1, 32, 32))
    assert lg.shape == (1, 3) and sg.shape == (1, 1, 32, 32)


def test_oversampling_curated_busi(golden_dir):
    classes = [str(c) for c in np.load(os.path.join(golden_dir, "curated_busi_classes.npz"))["classes"]]
    assert len(classes) == 450
    assert scaling_factors(classes) == {"benign": 2, "malignant": 3, "normal": 7}
    pos = deterministic_oversampling_positions(classes)
    assert len(pos) == 1384                                  # 444 / 492 / 448  
    out = [classes[i] for i in pos]
    assert (out.count("benign"), out.count("malignant"), out.count("normal")) == (444, 492, 448)
    assert pos[:450] == list(range(450))
    ben = [i for i, c in enumerate(classes) if c == "benign"]
    mal = [i for i, c in enumerate(classes) if c == "malignant"]
    nor = [i for i, c in enumerate(classes) if c == "normal"]
    assert pos[450:] == ben + mal * 2 + nor * 6


def test_oversampling_edge_cases():
    # proportion 0.4 -> 2.5 -> half-to-even -> 2 ; proportion 0.6 -> 1.67 -> 2
    cl = ["a"] * 6 + ["b"] * 4
    assert scaling_factors(cl) == {"a": 2, "b": 2}
    # dominant class (> 2/3): factor 1 -> still duplicated once (reference quirk :334-336)
    cl = ["a"] * 8 + ["b"] * 2
    assert scaling_factors(cl) == {"a": 1, "b": 5}
    pos = deterministic_oversampling_positions(cl)
    assert pos == list(range(10)) + list(range(8)) + [8, 9] * 4
    # ties keep first-seen order
    cl = ["x", "y", "y", "x"]
    assert list(scaling_factors(cl).keys()) == ["x", "y"]
    assert deterministic_oversampling_positions(cl) == [0, 1, 2, 3, 0, 3, 1, 2]
    assert deterministic_oversampling_positions([]) == []


def test_16bit_emulation_contexts_are_transparent_without_rounding():
    """oracle.lowp_conv3x3 restates where the 16-bit modes round (3x3 conv operands, k=2 ConvT backward operands): with a
    'rounding' type that does not round (float64) its custom backward passes must reproduce autograd exactly, and the
    ConvT -> 1x1 heads named by `model` must stay untouched."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from oracle import torch_oracle as O
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 8, 8, 8, generator=g, dtype=torch.float64, requires_grad=True)
    up = nn.ConvTranspose2d(8, 8, 2, 2).double()
    conv = nn.Conv2d(8, 8, 3, padding=1).double()
    head = nn.Sequential(nn.ConvTranspose2d(8, 8, 2, 2), nn.Conv2d(8, 1, 1)).double()
    net = nn.ModuleList([up, conv, head])

    def run():
        for p in list(net.parameters()) + [x]:
            p.grad = None
        y = conv(up(x))
        (y.square().sum() + head(x).square().sum()).backward()
        return [x.grad.clone()] + [p.grad.clone() for p in net.parameters()]

    want = run()
    ctx = O.lowp_conv3x3("bf16", model=net)
    ctx.lp = ctx.zt = torch.float64
    with ctx:
        assert F.conv2d is not ctx._orig and F.conv_transpose2d is not ctx._orig_t
        got = run()
    assert F.conv2d is ctx._orig and id(head[0].weight) in ctx.exempt and id(up.weight) not in ctx.exempt
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-12)
    # with real rounding the up-conv runs on rounded x and w (forward) and its backward differs; the head stays exact
    r = lambda t: t.bfloat16().to(t.dtype)
    with O.lowp_conv3x3("bf16", model=net):
        assert torch.equal(up(x), torch.conv_transpose2d(r(x), r(up.weight), up.bias, 2))
        assert torch.equal(head[0](x), torch.conv_transpose2d(x, head[0].weight, head[0].bias, 2))
        got_r = run()
    assert not torch.equal(got_r[1], want[1])


def test_single_task_seg_nnunet_step_matches_reference(golden_dir):
    """BASELINE.json configs[0] (the CPU plumbing case): the oracle's nnUNet2021 restatement and its segmentation-only
    step against the reference's own class / criterion glue (oracle/make_goldens.py:seg_nnunet_goldens)."""
    g = np.load(os.path.join(golden_dir, "seg_nnunet_step.npz"))
    O.seed_everything(1993)
    model = O.OracleSegNnUNet(1, 1)
    assert _sha(model.state_dict()) == str(g["sha256_init"])          # bit-exact initial weights, heads included
    model.train(True)
    opt = O.make_adam(model, lr=1e-4)
    x, mask = torch.from_numpy(g["x"]), torch.from_numpy(g["mask"])
    with torch.no_grad():
        outs = model(x)
    np.testing.assert_allclose([o.mean().item() for o in outs], g["out_means"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(outs[-1].numpy(), g["out1"], rtol=0, atol=2e-6)
    loss, dice = O.seg_train_step(model, opt, x, mask, True)
    assert abs(loss.item() - float(g["loss"])) < 1e-6 and 0.0 <= float(dice) <= 1.0
    sd = model.state_dict()
    for i, k in enumerate(str(n) for n in g["probe_names"]):
        np.testing.assert_allclose(sd[k].flatten()[:32].numpy(), g[f"after_{i}"], rtol=0, atol=2e-6)
        assert abs(sd[k].double().sum().item() - float(g[f"aftersum_{i}"])) < 1e-3


def test_16bit_emulation_of_pool_and_1x1_head_is_transparent_without_rounding_and_rounds_with_it():
    """The small consumers of a conv-cell activation (max-pool, 1x1 head <= 8 outputs) read the stored (rounded) tensor in
    the 16-bit modes: with a non-rounding type the patched functions must reproduce autograd exactly; with bf16 the pooled
    values equal the rounded pooled values and the gradient is routed by the rounded values."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from oracle import torch_oracle as O
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 8, 16, 16, generator=g, dtype=torch.float64, requires_grad=True)
    pool, head = nn.MaxPool2d(2, 2), nn.Conv2d(8, 1, 1).double()

    def run():
        for p in list(head.parameters()) + [x]:
            p.grad = None
        (pool(x).square().sum() + head(x).square().sum()).backward()
        return [x.grad.clone()] + [p.grad.clone() for p in head.parameters()]

    want = run()
    ctx = O.lowp_conv3x3("bf16")
    ctx.lp = ctx.zt = torch.float64
    with ctx:
        assert F.max_pool2d is not ctx._orig_p
        got = run()
    assert F.max_pool2d is ctx._orig_p
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=1e-12, atol=1e-12)
    r = lambda t: t.bfloat16().to(t.dtype)
    with O.lowp_conv3x3("bf16"):
        assert torch.equal(pool(x), r(pool(x.detach())))                       # max commutes with rounding
        assert torch.equal(head(x), torch.conv2d(r(x), head.weight, head.bias))
        x.grad = None
        pool(x).sum().backward()
        # every window routes its gradient to exactly one element, a maximal one of the ROUNDED window
        gx = x.grad
        win = gx.view(2, 8, 8, 2, 8, 2).permute(0, 1, 2, 4, 3, 5).reshape(2, 8, 8, 8, 4)
        assert torch.equal(win.sum(-1), torch.ones(2, 8, 8, 8, dtype=torch.float64))
        xr = r(x.detach()).view(2, 8, 8, 2, 8, 2).permute(0, 1, 2, 4, 3, 5).reshape(2, 8, 8, 8, 4)
        assert torch.equal((xr * win).sum(-1), xr.max(-1).values)
    small = torch.randn(1, 8, 8, 8, dtype=torch.float64)      # pooled map 4x4: the product keeps fp32 planes there
    with O.lowp_conv3x3("bf16"):
        assert torch.equal(pool(small), F.max_pool2d(small, 2, 2)) and not torch.equal(pool(small), r(pool(small)))


def test_16bit_emulation_of_stored_conv_outputs_and_gathered_gradients():
    """Round-2 storage of the 16-bit modes: the conv output of a conv cell is stored rounded, and the gradient its activation
    receives from ALL 3x3 consumers is summed, then rounded once, the other readers' contributions added un-rounded.  With a
    non-rounding type the patched functions reproduce autograd exactly; with bf16 the two rounding points are where they are
    said to be."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from oracle import torch_oracle as O
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 8, 16, 16, generator=g, dtype=torch.float64)
    cell = nn.Sequential(nn.Conv2d(8, 8, 3, padding=1), nn.InstanceNorm2d(8, affine=True), nn.Dropout(0.0), nn.LeakyReLU(0.1, inplace=True)).double()
    c1, c2 = nn.Conv2d(8, 8, 3, padding=1).double(), nn.Conv2d(16, 8, 3, padding=1).double()
    pool, up, head, gap = nn.MaxPool2d(2, 2), nn.ConvTranspose2d(8, 8, 2, 2).double(), nn.Conv2d(8, 1, 1).double(), nn.AdaptiveAvgPool2d(1)
    net = nn.ModuleList([cell, c1, c2, up, head])
    seen = {}

    def run():
        for p in net.parameters():
            p.grad = None
        y = cell(x)
        y.register_hook(lambda gr: seen.__setitem__("dy", gr.clone()))
        other = pool(y).square().sum() + up(y).sum() + head(y).square().sum() + gap(y).sum()
        convs = c1(y).square().sum() + c2(torch.cat([y, 2 * y], 1)).sum()
        (other + convs).backward()
        seen["y"] = y.detach().clone()
        return [p.grad.clone() for p in net.parameters()]

    want = run()
    dy_exact = seen["dy"]
    ctx = O.lowp_conv3x3("bf16", da16=True)
    ctx.lp = ctx.zt = torch.float64
    with ctx:
        got = run()
    assert F.instance_norm is ctx._orig_in and F.leaky_relu is ctx._orig_lr and F.dropout is ctx._orig_do and F.adaptive_avg_pool2d is ctx._orig_ap
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=1e-11, atol=1e-11)
    assert torch.allclose(seen["dy"], dy_exact, rtol=1e-11, atol=1e-11)
    r = lambda t: t.bfloat16().to(t.dtype)
    with O.lowp_conv3x3("bf16"):
        z = cell[0](x)
        # the stored conv output: fp16 (11 significant bits in the same 2 bytes), the MTBC_Z_BF16 arm bf16
        assert torch.equal(z, torch.conv2d(r(x), r(cell[0].weight), cell[0].bias, 1, 1).half().to(z.dtype))
    with O.lowp_conv3x3("bf16", z_fp16=False):
        assert torch.equal(cell[0](x), r(torch.conv2d(r(x), r(cell[0].weight), cell[0].bias, 1, 1)))
    with O.lowp_conv3x3("bf16", da16=True):
        run()
        y = seen["y"]
        # the conv readers' share, recomputed: dgrads on rounded dy and w, summed, rounded once
        yl = y.clone().requires_grad_(True)
        (c1(yl).square().sum() + c2(torch.cat([yl, 2 * yl], 1)).sum()).backward()
        conv_part = r(yl.grad)
        yo = y.clone().requires_grad_(True)
        (pool(yo).square().sum() + up(yo).sum() + head(yo).square().sum() + gap(yo).sum()).backward()
        # default: the other readers' fp32 partial is added un-rounded to the rounded sum of the 3x3 consumers
        assert torch.allclose(seen["dy"], conv_part + yo.grad, rtol=1e-12, atol=1e-12)
        assert not torch.allclose(seen["dy"], r(seen["dy"]), rtol=0, atol=0)                      # the total itself is not rounded
    with O.lowp_conv3x3("bf16", da16=True, fold_partials=True):                                   # the MTBC_EPI_BSTATS arm
        run()
        assert torch.equal(seen["dy"], r(yl.grad + yo.grad))                                      # the partial joins the sum BEFORE the one rounding
    with O.lowp_conv3x3("bf16", z16=False):
        assert torch.equal(cell[0](x), torch.conv2d(r(x), r(cell[0].weight), cell[0].bias, 1, 1))

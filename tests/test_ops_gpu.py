"""Per-op parity of the HIP kernels (through the C-ABI) against the same torch op on the CPU in fp32.
Tolerances are written per test; integer results are bit-exact."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from multi_task_breast_cancer_amd import ops  # noqa: E402
from oracle import torch_oracle as O  # noqa: E402

DEV = "cuda:0"


def _g(seed):
    return torch.Generator().manual_seed(seed)


def _close(got, want, rtol, atol, what=""):
    got = got.detach().cpu()
    err = (got - want).abs()
    tol = atol + rtol * want.abs()
    assert bool((err <= tol).all()), f"{what}: max err {err.max().item():.3e}, max |want| {want.abs().max().item():.3e}"


# (N, segs, Cout, H, W): geometry wide / 16-wide / 8x8, multi-segment virtual concat, ragged channel tiles
CONV_CASES = [
    (2, [8], 16, 32, 32),
    (1, [24, 48], 24, 40, 64),          # UNet++ level-0 style node, H not a multiple of 8
    (2, [24, 24, 24, 48], 24, 16, 32),
    (3, [16], 40, 16, 16),              # 16-wide geometry, Cout not a multiple of 16
    (2, [96, 96], 96, 16, 16),          # MT=3 x 2 blocks
    (5, [32], 80, 8, 8),                # 8x8 geometry: 4 images / block, ragged batch
    (6, [64, 64], 320, 8, 8),
    (1, [8], 8, 24, 48),                # width not a multiple of 32
    (2, [128], 64, 32, 32),
    (2, [320, 320, 320], 512, 16, 16),  # MTnnUNet classifier conv at 256^2 input
    (2, [320], 320, 8, 8),              # bottleneck, fewer images than a block holds
    (2, [320, 320], 256, 16, 16),
    (1, [384, 384, 384], 512, 16, 16),  # UNet++ classifier conv
    (2, [48], 48, 32, 32),              # 48-channel output blocks (3 tiles, 384-thread wgrad blocks)
    (2, [24, 24], 48, 16, 16),
    (3, [64], 144, 8, 8),
]


@pytest.mark.parametrize("N,segs,Cout,H,W", CONV_CASES)
def test_conv3x3_mfma_fwd_dgrad_wgrad(N, segs, Cout, H, W):
    g = _g(N * 1000 + Cout + H)
    Cin = sum(segs)
    xs = [torch.randn(N, c, H, W, generator=g) for c in segs]
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g)
    dz = torch.randn(N, Cout, H, W, generator=g)
    xcat = torch.cat(xs, 1).requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    z = F.conv2d(xcat, wr, br, padding=1)
    z.backward(dz)
    dxs_ref = torch.split(xcat.grad, segs, dim=1)

    xd = [x.to(DEV) for x in xs]
    wd, bd, dzd = w.to(DEV), b.to(DEV), dz.to(DEV)
    pf, pd = ops.conv3x3_pack(wd)
    zg = ops.conv3x3_fwd(xd, wd, bd, packed=pf)
    _close(zg, z.detach(), 2e-5, 2e-5, "fwd")
    # dgrad: first segment overwritten, the others accumulate onto a preset value
    pre = [torch.randn(N, c, H, W, generator=g) for c in segs]
    dxd = [p.to(DEV).clone() for p in pre]
    acc = [0] + [1] * (len(segs) - 1)
    ops.conv3x3_dgrad(dzd, wd, dxd, acc, packed=pd)
    for i, (got, want) in enumerate(zip(dxd, dxs_ref)):
        _close(got, want + (pre[i] if acc[i] else 0), 2e-5, 5e-5, f"dgrad seg {i}")
    dw, db = ops.conv3x3_wgrad(xd, dzd, tuple(w.shape), want_bias=True)
    scale = max(1.0, wr.grad.abs().max().item())
    _close(dw, wr.grad, 1e-4, 2e-5 * scale, "wgrad")
    _close(db, br.grad, 1e-4, 1e-4 * max(1.0, br.grad.abs().max().item()), "dbias")
    # shared-module accumulation (F10)
    dw2, _ = ops.conv3x3_wgrad(xd, dzd, tuple(w.shape), dw=dw.clone(), accumulate=True)
    _close(dw2, 2 * wr.grad, 1e-4, 4e-5 * scale, "wgrad accumulate")


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 1, 32, 64, 64), (1, 3, 5, 7, 9), (2, 12, 8, 4, 4), (1, 8, 8, 2, 2),
                                             (3, 1, 24, 40, 24), (2, 2, 5, 8, 12)])
def test_conv3x3_direct_path(N, Cin, Cout, H, W):
    g = _g(Cin * 7 + H)
    x = (torch.rand(N, Cin, H, W, generator=g) * 255.0)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.1
    dz = torch.randn(N, Cout, H, W, generator=g)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    z = F.conv2d(xr, wr, None, padding=1)
    z.backward(dz)
    xd, wd, dzd = x.to(DEV), w.to(DEV), dz.to(DEV)
    zg = ops.conv3x3_fwd([xd], wd, None, packed=None)
    _close(zg, z.detach(), 1e-5, 1e-3, "direct fwd")
    dx = torch.zeros_like(xd)
    ops.conv3x3_dgrad(dzd, wd, [dx], [0], packed=None)
    _close(dx, xr.grad, 1e-5, 1e-5, "direct dgrad")
    dw, _ = ops.conv3x3_wgrad([xd], dzd, tuple(w.shape))
    _close(dw, wr.grad, 1e-4, 1e-4 * max(1.0, wr.grad.abs().max().item()), "direct wgrad")


@pytest.mark.parametrize("N,Cout,H,W", [(2, 24, 64, 64), (1, 5, 8, 12), (3, 32, 40, 24)])
def test_stem_kernel_equals_direct_kernel_bitwise(N, Cout, H, W):
    """Cin == 1: the vectorised stem kernel keeps the direct kernel's fmaf order -> identical bits (bias included)."""
    g = _g(Cout + H)
    x = (torch.rand(N, 1, H, W, generator=g) * 255.0).to(DEV)
    w = (torch.randn(Cout, 1, 3, 3, generator=g) * 0.1).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    assert torch.equal(ops.conv3x3_fwd([x], w, b), ops.conv3x3_fwd([x], w, b, force_direct=True))
    assert torch.equal(ops.conv3x3_fwd([x], w, None), ops.conv3x3_fwd([x], w, None, force_direct=True))


def test_conv3x3_mfma_equals_direct_kernel():
    g = _g(11)
    x = torch.randn(2, 16, 32, 32, generator=g).to(DEV)
    w = (torch.randn(24, 16, 3, 3, generator=g) * 0.1).to(DEV)
    pf, _ = ops.conv3x3_pack(w)
    a = ops.conv3x3_fwd([x], w, None, packed=pf)
    b = ops.conv3x3_fwd([x], w, None, packed=pf, force_direct=True)
    _close(a, b.cpu(), 1e-5, 1e-5, "mfma vs direct")


@pytest.mark.parametrize("compute", [0, 1, 2])
def test_pack_many_equals_single_packs(compute):
    """The batched weight-image launch must write bit-for-bit what the per-tensor entry points write (110 tensors:
    more than one kernel-argument batch of 96)."""
    g = torch.Generator().manual_seed(5)
    shapes = [(24, 1), (24, 24), (48, 72), (40, 16), (96, 288), (7, 13), (384, 192)] * 8
    ws = [torch.randn(co, ci, 3, 3, generator=g).to(DEV) for co, ci in shapes[:55]]
    many = ops.conv3x3_pack_many(ws, compute)
    for w, (pf, pd) in zip(ws, many):
        rf, rd = ops.conv3x3_pack_lp(w, compute) if compute else ops.conv3x3_pack(w)
        assert torch.equal(pf, rf) and torch.equal(pd, rd), tuple(w.shape)


@pytest.mark.parametrize("N,C,H,W,affine,slope", [(2, 24, 256, 256, True, 0.1), (3, 5, 64, 64, False, 0.01),
                                                      (1, 3, 512, 512, True, 0.1), (2, 2, 300, 304, False, 0.01),
                                                   (2, 7, 32, 32, True, 0.1), (4, 320, 8, 8, False, 0.01),
                                                   (2, 3, 16, 16, True, 0.1), (1, 2, 6, 5, True, 0.1),
                                                   (1, 2, 2, 2, False, 0.01)])
def test_instnorm_lrelu_fwd_bwd(N, C, H, W, affine, slope):
    g = _g(C + H)
    z = torch.randn(N, C, H, W, generator=g) * 3.0 + 1.5
    gamma = (torch.rand(C, generator=g) + 0.5) if affine else None
    beta = (torch.randn(C, generator=g) * 0.3) if affine else None
    dy = torch.randn(N, C, H, W, generator=g)
    zr = z.clone().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True) if affine else None
    br = beta.clone().requires_grad_(True) if affine else None
    y = F.leaky_relu(F.instance_norm(zr, weight=gr, bias=br, eps=1e-5), slope)
    y.backward(dy)
    zd, dyd = z.to(DEV), dy.to(DEV)
    gd, bd = (gamma.to(DEV), beta.to(DEV)) if affine else (None, None)
    yg, mean, rstd = ops.instnorm_lrelu_fwd(zd, gd, bd, 1e-5, slope)
    _close(yg, y.detach(), 1e-5, 2e-5, "in fwd")
    _close(mean.view(N, C), z.mean(dim=(2, 3)), 1e-5, 1e-5, "mean")
    dz, dg, db = ops.instnorm_lrelu_bwd(zd, dyd, mean, rstd, gd, bd, 1e-5, slope)
    _close(dz, zr.grad, 1e-4, 2e-5, "in bwd")
    if affine:
        _close(dg, gr.grad, 1e-4, 1e-3, "dgamma")
        _close(db, br.grad, 1e-4, 1e-3, "dbeta")
    dbp = torch.empty(C, device=DEV)
    dz2, _, _ = ops.instnorm_lrelu_bwd(zd, dyd.clone(), mean, rstd, gd, bd, 1e-5, slope, inplace=True, dbias_pre=dbp)
    assert torch.equal(dz2, dz)                       # in-place form is bit-identical
    # gradient fan-in: dy split into 3 private contributions must give the same dz as their sum
    parts = [torch.randn(N, C, H, W, generator=g) for _ in range(2)]
    first = dy - parts[0] - parts[1]
    dz3, _, _ = ops.instnorm_lrelu_bwd(zd, first.to(DEV), mean, rstd, gd, bd, 1e-5, slope, dy_extra=[p_.to(DEV) for p_ in parts])
    _close(dz3, zr.grad, 1e-4, 3e-5, "in bwd with fan-in extras")
    want = dz.double().sum(dim=(0, 2, 3)).cpu()       # fused conv-bias gradient = sum of the dz it just wrote
    assert (dbp.cpu().double() - want).abs().max().item() <= 1e-5 * dz.abs().sum(dim=(0, 2, 3)).max().item() + 1e-6


@pytest.mark.parametrize("N,C,H,W", [(2, 3, 8, 8), (1, 5, 6, 10), (2, 24, 64, 64)])
def test_maxpool_fwd_bwd(N, C, H, W):
    g = _g(H * W)
    x = torch.randn(N, C, H, W, generator=g)
    x[0, 0, :2, :2] = 1.0                               # tie: gradient must go to the first element
    dy = torch.randn(N, C, H // 2, W // 2, generator=g)
    xr = x.clone().requires_grad_(True)
    y = F.max_pool2d(xr, 2, 2)
    y.backward(dy)
    xd = x.to(DEV)
    assert torch.equal(ops.maxpool2_fwd(xd).cpu(), y.detach())
    assert torch.equal(ops.maxpool2_bwd(xd, dy.to(DEV)).cpu(), xr.grad)
    pre = torch.randn(N, C, H, W, generator=g)
    got = ops.maxpool2_bwd(xd, dy.to(DEV), dx=pre.to(DEV).clone(), accumulate=True)
    _close(got, xr.grad + pre, 0, 1e-6, "pool bwd accumulate")


@pytest.mark.parametrize("N,Cin,Cout,H,W,k", [(2, 48, 48, 16, 16, 2), (1, 96, 48, 8, 8, 2), (2, 20, 12, 6, 10, 2),
                                               (2, 64, 64, 8, 8, 4), (1, 128, 128, 4, 4, 8), (3, 32, 32, 5, 3, 4),
                                               (3, 40, 6, 8, 16, 2), (2, 100, 52, 16, 8, 2), (5, 48, 48, 32, 32, 2),
                                               (2, 16, 2, 4, 8, 2), (1, 192, 96, 8, 8, 2)])
def test_convT_fwd_dgrad_wgrad(N, Cin, Cout, H, W, k):
    g = _g(Cin + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, k, k, generator=g) * 0.1
    b = torch.randn(Cout, generator=g)
    dy = torch.randn(N, Cout, H * k, W * k, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.conv_transpose2d(xr, wr, br, stride=k)
    y.backward(dy)
    xd, wd, bd, dyd = x.to(DEV), w.to(DEV), b.to(DEV), dy.to(DEV)
    _close(ops.convT_fwd(xd, wd, bd, k), y.detach(), 2e-5, 2e-5, "convT fwd")
    _close(ops.convT_dgrad(xd, wd, dyd, k), xr.grad, 2e-5, 5e-5, "convT dgrad")
    pre = torch.randn(N, Cin, H, W, generator=g)
    _close(ops.convT_dgrad(xd, wd, dyd, k, dx=pre.to(DEV).clone(), accumulate=True), xr.grad + pre, 2e-5, 5e-5, "convT dgrad acc")
    dw, db = ops.convT_wgrad(xd, wd, dyd, k)
    _close(dw, wr.grad, 1e-4, 2e-5 * max(1.0, wr.grad.abs().max().item()), "convT wgrad")
    _close(db, br.grad, 1e-4, 1e-4 * max(1.0, br.grad.abs().max().item()), "convT dbias")
    if k == 2 and (H * W) % 32 == 0 and W % 8 == 0 and Cout % 2 == 0:
        # 16-bit operand modes of the k == 2 backward kernels (shapes the generic fp32 kernel takes ignore `compute`):
        # the same sums over rounded operands
        for mode, lp in ((1, torch.bfloat16), (2, torch.float16)):
            r = lambda t: t.to(lp).float()
            xq, wq, dq = r(x).requires_grad_(True), r(w).requires_grad_(True), r(dy)
            F.conv_transpose2d(xq, wq, None, stride=k).backward(dq)
            _close(ops.convT_dgrad(xd, wd, dyd, k, compute=mode), xq.grad, 2e-5, 5e-5, f"convT dgrad lp{mode}")
            dwq, _ = ops.convT_wgrad(xd, wd, dyd, k, compute=mode)
            _close(dwq, wq.grad, 1e-4, 2e-5 * max(1.0, wq.grad.abs().max().item()), f"convT wgrad lp{mode}")


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 16, 1, 32, 32), (1, 24, 1, 64, 64), (2, 12, 3, 8, 8), (1, 20, 10, 4, 4)])
def test_conv1x1(N, Cin, Cout, H, W):
    g = _g(Cin)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) * 0.2
    b = torch.randn(Cout, generator=g)
    dy = torch.randn(N, Cout, H, W, generator=g)
    xr, wr, br = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y = F.conv2d(xr, wr, br)
    y.backward(dy)
    xd, wd = x.to(DEV), w.to(DEV)
    _close(ops.conv1x1_fwd(xd, wd, b.to(DEV)), y.detach(), 1e-5, 1e-5, "1x1 fwd")
    dx, dw, db = ops.conv1x1_bwd(xd, wd, dy.to(DEV))
    _close(dx, xr.grad, 1e-5, 1e-5, "1x1 dgrad")
    _close(dw, wr.grad, 1e-4, 1e-4 * max(1.0, wr.grad.abs().max().item()), "1x1 wgrad")
    _close(db, br.grad, 1e-4, 1e-4 * max(1.0, br.grad.abs().max().item()), "1x1 dbias")


def test_gap_linear_head():
    g = _g(5)
    x = torch.randn(4, 512, 16, 16, generator=g)
    w1, b1 = torch.randn(256, 512, generator=g) * 0.05, torch.randn(256, generator=g) * 0.1
    w2, b2 = torch.randn(3, 256, generator=g) * 0.05, torch.randn(3, generator=g) * 0.1
    dlog = torch.randn(4, 3, generator=g)
    xr = x.clone().requires_grad_(True)
    ws = [t.clone().requires_grad_(True) for t in (w1, b1, w2, b2)]
    pooled = F.adaptive_avg_pool2d(xr, 1).flatten(1)
    h = F.relu(F.linear(pooled, ws[0], ws[1]))
    out = F.linear(h, ws[2], ws[3])
    out.backward(dlog)
    xd = x.to(DEV)
    pg = ops.gap_fwd(xd)
    _close(pg, pooled.detach(), 1e-5, 1e-6, "gap")
    hg = ops.linear_fwd(pg, w1.to(DEV), b1.to(DEV), relu=True)
    og = ops.linear_fwd(hg, w2.to(DEV), b2.to(DEV), relu=False)
    _close(og, out.detach(), 1e-5, 1e-5, "linear fwd")
    dh, dw2, db2 = ops.linear_bwd(hg, w2.to(DEV), og, dlog.to(DEV), relu=False)
    dp, dw1, db1 = ops.linear_bwd(pg, w1.to(DEV), hg, dh, relu=True)
    _close(dw2, ws[2].grad, 1e-4, 1e-5, "dw2"); _close(db2, ws[3].grad, 1e-4, 1e-5, "db2")
    _close(dw1, ws[0].grad, 1e-4, 1e-5, "dw1"); _close(db1, ws[1].grad, 1e-4, 1e-5, "db1")
    _close(ops.gap_bwd(dp, 16, 16), xr.grad, 1e-4, 1e-7, "gap bwd")


def test_dice_multihead_matches_oracle():
    g = _g(9)
    N, H, W = 3, 64, 64
    xs = [torch.randn(N, 1, H, W, generator=g) * 2 for _ in range(4)]
    t = (torch.rand(N, 1, H, W, generator=g) > 0.7).float()
    t[1] = 0                                              # empty mask (class "normal")
    weights = [1 / 4, 1 / 3, 1 / 2, 1.0]
    xr = [x.clone().requires_grad_(True) for x in xs]
    each = [O.dice_loss_sigmoid_sq(x, t) for x in xr]
    total = sum(wi * e for wi, e in zip(weights, each))
    (0.35 * total).backward()
    loss, dxs = ops.dice_multihead([x.to(DEV) for x in xs], t.to(DEV), weights, gscale=0.35)
    loss = loss.cpu()
    for i in range(4):
        assert abs(loss[i].item() - each[i].item()) < 2e-6
        _close(dxs[i], xr[i].grad, 1e-4, 1e-9, f"dice dx head {i}")
    assert abs(loss[4].item() - total.item()) < 5e-6
    # closed form: zero logits, zero target -> 1 - 1/(.25 HW + 1)
    z = torch.zeros(2, 1, 16, 16, device=DEV)
    l0, _ = ops.dice_multihead([z], z.clone(), [1.0])
    assert abs(l0[0].item() - (1 - 1 / (0.25 * 256 + 1))) < 1e-6


def test_focal_matches_reference_goldens(golden_dir):
    import os
    gold = np.load(os.path.join(golden_dir, "focal.npz"))
    t = lambda k: torch.from_numpy(gold[k]).to(DEV)
    l, _ = ops.focal(t("x1"), t("t1"))
    assert abs(l.item() - 0.20617523789405823) < 1e-6          # SURVEY A8 known answer
    l, _ = ops.focal(t("x2"), t("t2"))
    assert abs(l.item() - float(gold["y2"])) < 1e-6
    l, _ = ops.focal(t("x2"), t("t3"))
    assert abs(l.item() - float(gold["y3"])) < 1e-6
    l, _ = ops.focal(t("x2"), t("t2"), weight=t("w"))
    assert abs(l.item() - float(gold["y4"])) < 1e-6
    for tk in ("t2", "t3"):
        x = torch.from_numpy(gold["x2"]).clone().requires_grad_(True)
        (0.65 * O.focal_loss_soft(x, torch.from_numpy(gold[tk]))).backward()
        _, dx = ops.focal(t("x2"), t(tk), gscale=0.65)
        _close(dx, x.grad, 1e-4, 1e-7, "focal dx")


def test_adam_matches_torch():
    g = _g(21)
    n = 10_003 // 4 * 4
    p0, grads = torch.randn(n, generator=g), [torch.randn(n, generator=g) * 10 ** float(e) for e in (-6, -3, 0, -2)]
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([pr], lr=1e-4, eps=1e-4)
    pd, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    for step, gr in enumerate(grads, start=1):
        pr.grad = gr.clone()
        opt.step()
        ops.adam_step(pd, gr.to(DEV), m, v, lr=1e-4, step=step, eps=1e-4)
        _close(pd, pr.detach(), 0, 2e-7, f"adam step {step}")
    # grad_scale (data-parallel averaging) and zero_grad
    gd = (grads[0] * 8).to(DEV)
    p1, p2 = p0.to(DEV), p0.to(DEV)
    ops.adam_step(p1, gd, torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), 1e-4, 1, grad_scale=0.125, zero_grad=True)
    ops.adam_step(p2, grads[0].to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), 1e-4, 1)
    assert torch.allclose(p1, p2, atol=1e-8) and float(gd.abs().max()) == 0.0


def test_dice_counts_exact(golden_dir):
    import os
    from multi_task_breast_cancer_amd.trainer import dice_counts, dice_score_from_counts
    gold = np.load(os.path.join(golden_dir, "dice_score.npz"))
    gt = torch.from_numpy(gold["gt"]).float()
    seg = torch.from_numpy(gold["seg"])
    logits = torch.where(seg, torch.tensor(3.0), torch.tensor(-3.0))
    c = dice_counts(logits.to(DEV), gt.to(DEV))
    assert abs(dice_score_from_counts(c) - float(gold["k_rand"])) < 1e-12
    tp = float((seg & (gt > 0)).sum()); fp = float((seg & (gt == 0)).sum()); fn = float((~seg & (gt > 0)).sum())
    assert c.tolist() == [tp, fp, fn]                                # integer counts: bit-exact
    z = torch.zeros(1, 1, 8, 8, device=DEV)
    assert dice_score_from_counts(dice_counts(z - 1, z)) == 1.0       # empty / empty -> 1 (metrics.py:262-263)


# ------------------------------------------------------------------ full-size properties (BASELINE.json configs[1] shapes)
@pytest.mark.parametrize("compute,tol", [(0, 2e-5), (1, 5e-3)])
@pytest.mark.parametrize("segs,Cout,H", [([24, 24, 48], 24, 256), ([48], 48, 128)])
def test_conv3x3_adjoint_identities_at_bench_size(compute, tol, segs, Cout, H):
    """At N=32 the CPU oracle is too slow to be the checker; the bilinear form is: for y = conv(x; w),
    <y, dz> = <x, dgrad(dz; w)> = <w, wgrad(x, dz)>.  Any indexing / padding / segment bug breaks one of the three.
    (16-bit mode: each side rounds its own operands, so the identities hold to the rounding level only.)"""
    N, W = 32, H
    g = _g(H + Cout)
    xs = [torch.randn(N, c, H, W, generator=g).to(DEV) for c in segs]
    cin = sum(segs)
    w = (torch.randn(Cout, cin, 3, 3, generator=g) * 0.05).to(DEV)
    dz = torch.randn(N, Cout, H, W, generator=g).to(DEV)
    if compute:
        pf, pd = ops.conv3x3_pack_lp(w, compute)
    else:
        pf, pd = ops.conv3x3_pack(w)
    y = ops.conv3x3_fwd(xs, w, None, packed=pf, compute=compute)
    dxs = [torch.zeros_like(x) for x in xs]
    ops.conv3x3_dgrad(dz, w, dxs, [0] * len(xs), packed=pd, compute=compute)
    dw, _ = ops.conv3x3_wgrad(xs, dz, tuple(w.shape), compute=compute)
    s_y = (y.double() * dz.double()).sum().item()
    s_x = sum((x.double() * d.double()).sum().item() for x, d in zip(xs, dxs))
    s_w = (w.double() * dw.double()).sum().item()
    scale = y.double().norm().item() * dz.double().norm().item()
    assert abs(s_y - s_x) < tol * scale and abs(s_y - s_w) < tol * scale, (s_y, s_x, s_w, scale)
    # linearity in x (forward), exact-fp32 mode only: conv(2x) = 2 conv(x) bit for bit (a power of two commutes with RNE)
    if compute == 0:
        y2 = ops.conv3x3_fwd([2.0 * x for x in xs], w, None, packed=pf)
        assert torch.equal(y2, 2.0 * y)


def test_instnorm_and_convT_properties_at_bench_size():
    N, C, H = 32, 24, 256
    g = _g(77)
    z = (torch.randn(N, C, H, H, generator=g) * 3.0 + 1.5).to(DEV)
    y, mean, rstd = ops.instnorm_lrelu_fwd(z, None, None, slope=1.0)   # slope 1: plain instance norm
    m = y.double().mean(dim=(2, 3))
    v = y.double().var(dim=(2, 3), unbiased=False)
    assert m.abs().max().item() < 1e-5 and (v - 1.0).abs().max().item() < 1e-4
    # ConvT k=2: <convT(x; w), dy> = <x, dgrad(dy; w)> = <w, wgrad(x, dy)>
    x = torch.randn(N, 48, 128, 128, generator=g).to(DEV)
    w = (torch.randn(48, 48, 2, 2, generator=g) * 0.1).to(DEV)
    dy = torch.randn(N, 48, 256, 256, generator=g).to(DEV)
    yt = ops.convT_fwd(x, w, None, 2)
    dx = ops.convT_dgrad(x, w, dy, 2)
    dw, _ = ops.convT_wgrad(x, w, dy, 2, want_bias=False)
    s_y = (yt.double() * dy.double()).sum().item()
    s_x = (x.double() * dx.double()).sum().item()
    s_w = (w.double() * dw.double()).sum().item()
    scale = yt.double().norm().item() * dy.double().norm().item()
    assert abs(s_y - s_x) < 2e-5 * scale and abs(s_y - s_w) < 2e-5 * scale, (s_y, s_x, s_w)


@pytest.mark.parametrize("N,Cin,Cmid,R,H,W,k", [(2, 16, 12, 1, 8, 8, 8), (1, 24, 20, 3, 6, 10, 4), (3, 32, 32, 1, 16, 16, 2)])
def test_fused_convT_1x1_head_equals_the_two_layers(N, Cin, Cmid, R, H, W, k):
    """MTnnUNet.py:106-116: ConvTranspose2d(k = s) then Conv2d 1x1, no non-linearity in between == one transposed conv
    with combined weights; forward and every parameter gradient against autograd of the two layers."""
    g = _g(Cin + k)
    x = torch.randn(N, Cin, H, W, generator=g)
    wT = torch.randn(Cin, Cmid, k, k, generator=g) * 0.2
    bT = torch.randn(Cmid, generator=g)
    w1 = torch.randn(R, Cmid, 1, 1, generator=g) * 0.3
    b1 = torch.randn(R, generator=g)
    dout = torch.randn(N, R, H * k, W * k, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x, wT, bT, w1, b1)]
    y = F.conv2d(F.conv_transpose2d(leaves[0], leaves[1], leaves[2], stride=k), leaves[3], leaves[4])
    y.backward(dout)
    got = ops.convT_head_fwd_bwd(*(t.to(DEV) for t in (x, wT, bT, w1, b1)), k, dout.to(DEV))
    want = [y.detach()] + [t.grad for t in leaves]
    for name, a, b in zip(("y", "dx", "dwT", "dbT", "dw1", "db1"), got, want):
        _close(a, b, 1e-4, 2e-5 * max(1.0, b.abs().max().item()), "head " + name)


# ------------------------------------------------------------------ 16-bit channel-blocked operands (MTBC_LAYOUT_C8)
def _round16(t, compute):
    return (t.bfloat16() if compute == 1 else t.half()).float()


@pytest.mark.parametrize("compute", [1, 2])
def test_c8_pack_is_rne_and_unpack_is_exact(compute):
    x = torch.randn(3, 24, 10, 12, generator=_g(5)) * 3
    c8 = ops.C8.pack(x.to(DEV), compute)
    assert c8.data.shape == (3, 3, 120, 8)
    back = c8.unpack().cpu()
    assert torch.equal(back, _round16(x, compute))
    # layout: piece (n, g, px) holds channels 8g..8g+7 of pixel px
    raw = c8.data.cpu().view(torch.bfloat16 if compute == 1 else torch.float16).float()
    assert torch.equal(raw[1, 2, 37], _round16(x, compute)[1, 16:24].reshape(8, -1)[:, 37])


C8_CASES = [
    (2, [8], 16, 32, 32),
    (1, [24, 48], 24, 40, 64),          # two segments, H not a multiple of the tile
    (2, [24, 24, 24, 48], 24, 16, 32),  # 4 segments, 120 channels: the last 32-channel chunk is ragged
    (3, [16], 40, 16, 16),              # 16-wide geometry, Cout not a multiple of 16
    (5, [32], 80, 8, 8),                # 8x8 geometry: 4 images per block, ragged batch
    (1, [8], 8, 24, 48),                # width not a multiple of 32
    (2, [144], 24, 36, 64),             # >1 chunk, rows not a multiple of 4
    (2, [96, 96], 96, 16, 16),
    (2, [48], 48, 32, 32),
]


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,segs,Cout,H,W", C8_CASES)
def test_conv3x3_c8_operands(N, segs, Cout, H, W, compute):
    """fwd / dgrad read the SAME rounded values in the SAME MFMA order as the planar 16-bit path -> bit-identical;
    wgrad (another split and tile order) and dbias are checked against fp64 on the rounded operands."""
    g = _g(N * 77 + Cout + H + compute)
    Cin = sum(segs)
    xs = [torch.randn(N, c, H, W, generator=g) for c in segs]
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g)
    dz = torch.randn(N, Cout, H, W, generator=g)
    xd = [x.to(DEV) for x in xs]
    wd, bd, dzd = w.to(DEV), b.to(DEV), dz.to(DEV)
    pf, pd = ops.conv3x3_pack_lp(wd, compute)
    x8 = [ops.C8.pack(x, compute) for x in xd]
    dz8 = ops.C8.pack(dzd, compute)

    z_planar = ops.conv3x3_fwd(xd, wd, bd, packed=pf, compute=compute)
    z_c8 = ops.conv3x3_fwd_c8(x8, wd, bd, pf)
    assert torch.equal(z_c8, z_planar), f"fwd differs: {(z_c8 - z_planar).abs().max().item():.3e}"

    pre = [torch.randn(N, c, H, W, generator=g) for c in segs]
    acc = [0] + [1] * (len(segs) - 1)
    d_planar = [p.to(DEV).clone() for p in pre]
    d_c8 = [p.to(DEV).clone() for p in pre]
    ops.conv3x3_dgrad(dzd, wd, d_planar, acc, packed=pd, compute=compute)
    ops.conv3x3_dgrad_c8(dz8, wd, d_c8, acc, pd)
    for i in range(len(segs)):
        assert torch.equal(d_c8[i], d_planar[i]), f"dgrad seg {i} differs"

    xr = _round16(torch.cat(xs, 1), compute).double().requires_grad_(False)
    dzr = _round16(dz, compute).double()
    wref = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wref, padding=1).backward(dzr)
    dw, db = ops.conv3x3_wgrad_c8(x8, dz8, tuple(w.shape), want_bias=True)
    scale = max(1.0, wref.grad.abs().max().item())
    _close(dw, wref.grad.float(), 1e-5, 2e-5 * scale, "wgrad c8")
    dbr = dzr.sum((0, 2, 3)).float()
    _close(db, dbr, 1e-5, 2e-5 * max(1.0, dbr.abs().max().item()), "dbias c8")
    dw2, db2 = ops.conv3x3_wgrad_c8(x8, dz8, tuple(w.shape), want_bias=False, dw=dw.clone(), accumulate=True)
    _close(dw2, 2 * wref.grad.float(), 1e-5, 4e-5 * scale, "wgrad c8 accumulate")


# shapes that take the wide-block / rolling-row weight-gradient kernel (conv3x3_wgrad_c8w_kernel: Cin >= 64 on maps >= 128 wide)
C8W_CASES = [
    (8, [24, 24, 24], 24, 20, 128),            # image -> XCD block mapping (N % 8 == 0); Cin = 72: a half-empty last input tile; 5 row steps
    (2, [24, 24, 24, 24, 48], 24, 256, 256),   # 144 -> 24 @256x256: two input-channel blocks (5 + 4 tiles), 64 steps per strip
    (1, [48, 48], 48, 130, 128),               # 48 outputs = 3 tiles in one block; H not a multiple of the 4-row step; one image
    (2, [96, 96], 96, 32, 160),                # 2 output-channel blocks x 3 input-channel blocks; 5 strips
    (2, [64], 40, 36, 144),                    # Cout = 40: blocks of 32 + 8 channels; W not a multiple of the 32-column strip
    (3, [96, 48, 48], 48, 12, 256),            # 192 -> 48: three input-channel blocks; rows < one ring (3 steps); N % 8 != 0
]


# ... and the image-tile kernel of the 16 x 16 maps (conv3x3_wgrad_c8i_kernel: Cin >= 64, Cout >= 32)
C8I_CASES = [
    (5, [64], 40, 16, 16),                     # Cout = 40: blocks of 32 + 8 channels; 5 image ranges of one image
    (9, [96, 48, 48], 48, 16, 16),             # three segments, 192 -> 48: three input-channel blocks of 4 tiles; 48 outputs in one block
    (16, [264], 64, 16, 16),                   # 16.5 input tiles: four blocks, the last tile half empty; two images per range
    (3, [72], 96, 16, 16),                     # 4.5 input tiles in one block; 2 output-channel blocks of 48
    (1, [64], 48, 16, 16),                     # ONE image range: without bias / accumulation the blocks store straight into dw (no reduction launch)
]


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,segs,Cout,H,W", C8W_CASES + C8I_CASES)
def test_conv3x3_wgrad_c8_wide_blocks(N, segs, Cout, H, W, compute):
    """conv3x3_wgrad_c8w_kernel (all output channels x <= 80 input channels per block, rows staged once in an LDS ring, DMA of the
    next rows under the MFMAs) against fp64 on the rounded operands: weight and bias gradient, the no-bias kernel instance,
    accumulation into dw -- ragged rows / columns / channel tiles / batch."""
    g = _g(N * 31 + Cout + H + W + compute)
    Cin = sum(segs)
    xs = [torch.randn(N, c, H, W, generator=g) for c in segs]
    dz = torch.randn(N, Cout, H, W, generator=g)
    x8 = [ops.C8.pack(x.to(DEV), compute) for x in xs]
    dz8 = ops.C8.pack(dz.to(DEV), compute)
    xr = _round16(torch.cat(xs, 1), compute).double()
    dzr = _round16(dz, compute).double()
    ref = torch.nn.grad.conv2d_weight(xr, (Cout, Cin, 3, 3), dzr, padding=1).float()
    dbr = dzr.sum((0, 2, 3)).float()
    scale = max(1.0, ref.abs().max().item())
    dw, db = ops.conv3x3_wgrad_c8(x8, dz8, (Cout, Cin, 3, 3), want_bias=True)
    _close(dw, ref, 1e-5, 2e-5 * scale, "wgrad c8w")
    _close(db, dbr, 1e-5, 2e-5 * max(1.0, dbr.abs().max().item()), "dbias c8w")
    dw1, _ = ops.conv3x3_wgrad_c8(x8, dz8, (Cout, Cin, 3, 3), want_bias=False)
    _close(dw1, ref, 1e-5, 2e-5 * scale, "wgrad c8w (no-bias instance)")
    dw2, _ = ops.conv3x3_wgrad_c8(x8, dz8, (Cout, Cin, 3, 3), want_bias=False, dw=dw1.clone(), accumulate=True)
    _close(dw2, 2 * ref, 1e-5, 4e-5 * scale, "wgrad c8w accumulate")
    again, _ = ops.conv3x3_wgrad_c8(x8, dz8, (Cout, Cin, 3, 3), want_bias=False)
    assert torch.equal(again, dw1), "not deterministic"


# the in-kernel split-K reduction across its tree shapes: (N, segs, Cout, H, W) -> splits per (co, ci) tile / levels
FIXUP_CASES = [
    (16, [24], 24, 256, 256),                  # 32 x 32 kernel, ONE tile, 1024 splits: three levels of fan-in 11
    (8, [48], 48, 128, 128),                   # 4 tiles x 256 splits: three levels of 7
    (4, [96], 96, 64, 64),                     # 9 tiles x 113 splits (not a multiple of 8: ragged (split % 8) classes)
    (8, [96], 96, 128, 128),                   # 1024 pixel tiles, 9 channel blocks: 112 splits (a multiple of 8: one XCD per split) of 9 or 10 tiles each
    (2, [192], 192, 32, 32),                   # 36 tiles x 28 splits: two levels of 6, ragged last group
    (2, [384], 192, 32, 32),                   # 72 tiles x 14 splits
    (8, [24, 24, 24], 24, 256, 256),           # wide-block kernel: 512 row-segment splits of image-major leaves (N % 8 == 0)
    (3, [96, 48, 48], 48, 128, 128),           # wide-block kernel, N % 8 != 0: leaf = split
    (32, [192], 384, 16, 16),                  # image-tile kernel: 24 channel blocks x 8 image ranges = one level
    (6, [384], 384, 16, 16),                   # image-tile kernel: 6 ranges of one image
    (1, [64], 48, 16, 16),                     # ONE split: with bias / accumulation the single row goes through the top level
]


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,segs,Cout,H,W", FIXUP_CASES)
def test_conv3x3_wgrad_c8_in_kernel_split_k_reduction(N, segs, Cout, H, W, compute):
    """Round 4: the split-K partials of the channel-blocked weight gradients are summed INSIDE the launch by the last-arriving block of
    each group of rows (mtbc_conv3x3_args.wgrad_sync; conv3x3.hip splitk_fixup) instead of by a reduction launch.  Against fp64 on the
    rounded operands, like the reduction-launch path (which must agree with it to fp32 re-association); bit-identical from run to run
    (sums are taken in row order, not arrival order); with bias, and accumulating into dw; the counter buffer is all zeros afterwards
    (every group's winner resets its counter), so one buffer serves every launch of a stream.  Backward of nn.Conv2d,
    MTUNetPlusPlus.py:47-81 / training_multitask.py:102."""
    g = _g(N * 17 + Cout + H + compute)
    Cin = sum(segs)
    xs = [torch.randn(N, c, H, W, generator=g) for c in segs]
    dz = torch.randn(N, Cout, H, W, generator=g)
    x8 = [ops.C8.pack(x.to(DEV), compute) for x in xs]
    dz8 = ops.C8.pack(dz.to(DEV), compute)
    xr = _round16(torch.cat(xs, 1), compute).double()
    dzr = _round16(dz, compute).double()
    ref = torch.nn.grad.conv2d_weight(xr, (Cout, Cin, 3, 3), dzr, padding=1).float()
    dbr = dzr.sum((0, 2, 3)).float()
    scale = max(1.0, ref.abs().max().item())
    shape = (Cout, Cin, 3, 3)
    dw, db = ops.conv3x3_wgrad_c8(x8, dz8, shape, want_bias=True)
    sync = ops.wgrad_sync_buffer(DEV, 4)
    assert int(sync.abs().sum().item()) == 0, "counters not back at zero"
    _close(dw, ref, 1e-5, 2e-5 * scale, "wgrad, in-kernel reduction")
    _close(db, dbr, 1e-5, 2e-5 * max(1.0, dbr.abs().max().item()), "dbias, in-kernel reduction")
    old, dbo = ops.conv3x3_wgrad_c8(x8, dz8, shape, want_bias=True, in_kernel_reduce=False)
    _close(old, ref, 1e-5, 2e-5 * scale, "wgrad, reduction launch")
    _close(dw, old.cpu(), 1e-5, 1e-5 * scale, "the two reductions agree")
    _close(db, dbo.cpu(), 1e-5, 1e-5 * max(1.0, dbr.abs().max().item()), "the two bias reductions agree")
    for _ in range(3):
        again, dba = ops.conv3x3_wgrad_c8(x8, dz8, shape, want_bias=True)
        assert torch.equal(again, dw) and torch.equal(dba, db), "not deterministic"
    dw1, _ = ops.conv3x3_wgrad_c8(x8, dz8, shape, want_bias=False)
    _close(dw1, ref, 1e-5, 2e-5 * scale, "no-bias instance")
    dw2, _ = ops.conv3x3_wgrad_c8(x8, dz8, shape, want_bias=False, dw=dw1.clone(), accumulate=True)
    _close(dw2, 2 * ref, 1e-5, 4e-5 * scale, "accumulate")
    assert int(sync.abs().sum().item()) == 0, "counters not back at zero"


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,segs,Cout,H,W", C8_CASES + [(2, [24, 24, 24, 24, 24, 24], 24, 256, 256),   # 144->24 @256x256: the bench's widest level-0 node
                                                        (1, [384, 384, 384], 512, 16, 16),                 # K = 1152 x 9: the longest accumulation of the step
                                                        (16, [24, 24], 24, 256, 256)])                     # enough 16 x 32 tiles for the 8-wave (512-pixel) blocks
def test_conv3x3_16bit_fwd_dgrad_match_fp64_on_rounded_operands(N, segs, Cout, H, W, compute):
    """The oracle of the 16-bit forward / dgrad kernels (channel-blocked AND planar staging): fp64 conv2d / conv2d_input
    on operands rounded (RNE) to the MFMA's 16-bit type -- x and w forward, dz and w backward -- i.e. exact products,
    fp32 accumulation being the only difference.  Tolerance 1e-5 relative to the output scale (fp32 accumulation over
    K <= 10368 terms: ~ sqrt(K) * 2^-24 * |terms|), the standard wgrad_c8 already meets."""
    g = _g(N * 131 + Cout + H + compute)
    Cin = sum(segs)
    xs = [torch.randn(N, c, H, W, generator=g) for c in segs]
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g)
    dz = torch.randn(N, Cout, H, W, generator=g)
    xr, wr, dzr = _round16(torch.cat(xs, 1), compute).double(), _round16(w, compute).double(), _round16(dz, compute).double()
    z_ref = F.conv2d(xr, wr, b.double(), padding=1)
    dx_ref = torch.nn.grad.conv2d_input(xr.shape, wr, dzr, padding=1)
    dxs_ref = torch.split(dx_ref, segs, dim=1)

    xd = [x.to(DEV) for x in xs]
    wd, bd, dzd = w.to(DEV), b.to(DEV), dz.to(DEV)
    pf, pd = ops.conv3x3_pack_lp(wd, compute)
    x8 = [ops.C8.pack(x, compute) for x in xd]
    dz8 = ops.C8.pack(dzd, compute)
    zs = z_ref.abs().max().item()
    for name, z in (("c8", ops.conv3x3_fwd_c8(x8, wd, bd, pf)), ("planar", ops.conv3x3_fwd(xd, wd, bd, packed=pf, compute=compute))):
        err = (z.cpu().double() - z_ref).abs().max().item()
        assert err <= 1e-5 * zs, f"fwd {name}: max err {err:.3e} vs scale {zs:.3e}"
    pre = [torch.randn(N, c, H, W, generator=g) for c in segs]
    acc = [0] + [1] * (len(segs) - 1)
    ds = dx_ref.abs().max().item()
    for name in ("c8", "planar"):
        d = [p.to(DEV).clone() for p in pre]
        if name == "c8":
            ops.conv3x3_dgrad_c8(dz8, wd, d, acc, pd)
        else:
            ops.conv3x3_dgrad(dzd, wd, d, acc, packed=pd, compute=compute)
        for i in range(len(segs)):
            want = dxs_ref[i] + (pre[i].double() if acc[i] else 0)
            err = (d[i].cpu().double() - want).abs().max().item()
            assert err <= 1e-5 * max(ds, want.abs().max().item()), f"dgrad {name} seg {i}: max err {err:.3e} vs scale {ds:.3e}"


def test_conv3x3_c8_rejects_what_it_cannot_run():
    from multi_task_breast_cancer_amd import _lib as L
    x = torch.randn(1, 8, 8, 10, generator=_g(1)).to(DEV)      # W % 4 != 0: the fp32 output needs 16-byte rows
    w = torch.randn(8, 8, 3, 3, generator=_g(2)).to(DEV)
    pf, _ = ops.conv3x3_pack_lp(w, 1)
    with pytest.raises(L.MtbcError):
        ops.conv3x3_fwd_c8([ops.C8.pack(x, 1)], w, None, pf)


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 48, 48, 16, 16), (3, 24, 24, 8, 12), (1, 64, 40, 16, 8), (2, 16, 8, 4, 8)])
def test_convT_forward_into_channel_blocked_output_equals_planar_then_pack(N, Cin, Cout, H, W, compute):
    g = _g(N + Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cin, Cout, 2, 2, generator=g) * 0.2).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    want = ops.C8.pack(ops.convT_fwd(x, w, b, 2), compute)
    got = ops.convT_fwd_c8(x, w, b, 2, compute)
    assert got.shape == want.shape and torch.equal(got.data, want.data)


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,C,H,W,affine", [(2, 24, 256, 256, True), (3, 8, 64, 64, False), (2, 16, 12, 20, True), (1, 48, 128, 128, True)])
def test_instnorm_16bit_planar_outputs_are_the_rounded_fp32_outputs(N, C, H, W, affine, compute):
    """y16 / dz16 = RNE of the fp32 y / dz bit for bit, the fp32 side results (statistics, parameter gradients) do not
    change, and pack16 of the planes equals pack of the fp32 tensor."""
    g = _g(N + C + H + compute)
    z = (torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(DEV)
    dy = torch.randn(N, C, H, W, generator=g).to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV) if affine else None
    beta = (torch.randn(C, generator=g) * 0.1).to(DEV) if affine else None
    dt = torch.bfloat16 if compute == 1 else torch.float16
    y, mean, rstd = ops.instnorm_lrelu_fwd(z, gamma, beta, slope=0.1)
    y16, mean2, rstd2 = ops.instnorm_lrelu_fwd(z, gamma, beta, slope=0.1, out16=compute)
    assert torch.equal(y16.view(dt), y.to(dt)) and torch.equal(mean, mean2) and torch.equal(rstd, rstd2)
    db1, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dz, dg, dbt = ops.instnorm_lrelu_bwd(z, dy, mean, rstd, gamma, beta, slope=0.1, dbias_pre=db1)
    dz16, dg2, dbt2 = ops.instnorm_lrelu_bwd(z, dy, mean, rstd, gamma, beta, slope=0.1, dbias_pre=db2, out16=compute)
    assert torch.equal(dz16.view(dt), dz.to(dt)) and torch.equal(db1, db2)
    if affine:
        assert torch.equal(dg, dg2) and torch.equal(dbt, dbt2)
    if C % 8 == 0:
        assert torch.equal(ops.C8.pack16(y16, compute).data, ops.C8.pack(y, compute).data)


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,C,H,W,affine", [(2, 24, 256, 256, True), (3, 48, 128, 128, True), (2, 96, 64, 64, False),
                                            (5, 16, 32, 32, True), (2, 8, 16, 16, True), (3, 8, 8, 8, False), (32, 24, 256, 256, True),
                                            (4, 24, 512, 512, True)])
def test_cooperative_instnorm_into_channel_blocked_layout(N, C, H, W, affine, compute):
    """The split-plane cooperative kernels (teams of workgroups exchanging partial statistics through a mailbox) against
    the one-plane-per-workgroup kernels: statistics to fp32 re-association, outputs to one 16-bit ulp of the rounded
    fp32 result, parameter gradients to fp32 re-association; launched repeatedly (mailbox epochs) and never giving up
    a poll."""
    g = _g(N + C + H + compute)
    z = (torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(DEV)
    dy = torch.randn(N, C, H, W, generator=g).to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV) if affine else None
    beta = (torch.randn(C, generator=g) * 0.1).to(DEV) if affine else None
    ulp = 2.0 ** -7 if compute == 1 else 2.0 ** -10
    y, mean, rstd = ops.instnorm_lrelu_fwd(z, gamma, beta, slope=0.1)
    for rep in range(3):
        y8, mean2, rstd2, yp = ops.instnorm_lrelu_fwd_c8(z, gamma, beta, slope=0.1, compute=compute, want_planar=rep == 1)
    assert torch.allclose(mean2, mean, rtol=1e-5, atol=1e-6) and torch.allclose(rstd2, rstd, rtol=1e-5, atol=1e-6)
    got = y8.unpack()
    assert bool(((got - y).abs() <= ulp * y.abs() + 1e-5).all()), (got - y).abs().max().item()
    db1, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dz, dg, dbt = ops.instnorm_lrelu_bwd(z, dy, mean, rstd, gamma, beta, slope=0.1, dbias_pre=db1)
    for rep in range(2):
        dz8, dg2, dbt2 = ops.instnorm_lrelu_bwd_c8(z, dy, mean, rstd, gamma, beta, slope=0.1, dbias_pre=db2, compute=compute)
    gotz = dz8.unpack()
    scale = dz.abs().max().item()
    assert bool(((gotz - dz).abs() <= ulp * dz.abs() + 2e-6 * scale).all()), (gotz - dz).abs().max().item()
    # the bias gradient of a conv in front of InstanceNorm is mathematically zero: both kernels return the rounding noise
    # of a sum over N*H*W terms, which scales with the sum of |dz|
    noise = 1e-6 * dz.abs().sum((0, 2, 3)).max().item()
    assert (db2 - db1).abs().max().item() <= max(1e-3, noise)
    if affine:
        assert torch.allclose(dg2, dg, rtol=1e-4, atol=1e-3 * max(1.0, dg.abs().max().item()))
        assert torch.allclose(dbt2, dbt, rtol=1e-4, atol=1e-3 * max(1.0, dbt.abs().max().item()))
    assert ops.coop_error(DEV) == 0


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 48, 48, 16, 16), (3, 24, 24, 8, 12), (1, 64, 40, 16, 8), (2, 96, 48, 8, 8),
                                            (2, 384, 192, 4, 8), (1, 512, 512, 8, 4), (2, 16, 8, 4, 8)])
def test_convT_forward_on_the_16bit_mfma_with_channel_blocked_tensors(N, Cin, Cout, H, W, compute):
    """x and w rounded to the 16-bit type, fp32 accumulation, bias, ONE rounding of the result: against fp64 on the
    rounded operands, to one 16-bit ulp."""
    g = _g(N + Cin + Cout + H + compute)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cin, Cout, 2, 2, generator=g) * (1.0 / Cin) ** 0.5
    b = torch.randn(Cout, generator=g)
    want = F.conv_transpose2d(_round16(x, compute).double(), _round16(w, compute).double(), b.double(), stride=2).float()
    got = ops.convT_fwd_c8_lp(ops.C8.pack(x.to(DEV), compute), w.to(DEV), b.to(DEV), 2).unpack().cpu()
    ulp = 2.0 ** -7 if compute == 1 else 2.0 ** -10
    assert bool(((got - want).abs() <= ulp * want.abs() + 1e-5).all()), (got - want).abs().max().item()


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,C,H,W", [(2, 24, 256, 256), (3, 8, 16, 24), (1, 48, 32, 32), (2, 16, 64, 16)])
def test_maxpool_on_channel_blocked_tensors(N, C, H, W, compute):
    """forward == fp32 pool of the stored values, re-packed, bit for bit; backward == the fp32 kernel run on the stored
    (unpacked) values, bit for bit, overwrite and accumulate (ties of rounded values included: the test data has many)."""
    g = _g(N + C + H + compute)
    x = torch.randn(N, C, H, W, generator=g).to(DEV)
    x8 = ops.C8.pack(x, compute)
    xs = x8.unpack()                                         # the stored values
    y8 = ops.maxpool2_fwd_c8(x8)
    want = ops.C8.pack(ops.maxpool2_fwd(xs), compute)
    assert y8.shape == want.shape and torch.equal(y8.data, want.data)
    assert torch.equal(y8.data, ops.C8.pack(ops.maxpool2_fwd(x), compute).data)      # max commutes with the rounding
    dy = torch.randn(N, C, H // 2, W // 2, generator=g).to(DEV)
    assert torch.equal(ops.maxpool2_bwd_c8(x8, dy), ops.maxpool2_bwd(xs, dy))
    pre = torch.randn(N, C, H, W, generator=g).to(DEV)
    assert torch.equal(ops.maxpool2_bwd_c8(x8, dy, dx=pre.clone(), accumulate=True), ops.maxpool2_bwd(xs, dy, dx=pre.clone(), accumulate=True))
    # a window with a tie among its rounded maxima, first-maximum routing checked explicitly
    t = torch.zeros(1, 8, 8, 8, device=DEV)
    t[0, :, 0, 1] = 1.0; t[0, :, 1, 0] = 1.0 + 2.0 ** -12        # both round to 1.0 in bf16 AND fp16: (0,1) comes first
    g1 = ops.maxpool2_bwd_c8(ops.C8.pack(t, compute), torch.ones(1, 8, 4, 4, device=DEV))
    assert g1[0, 0, 0, 1].item() == 1.0 and g1[0, 0, 1, 0].item() == 0.0


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 24, 1, 256, 256), (3, 16, 1, 32, 32), (2, 24, 3, 64, 48), (1, 64, 8, 16, 16)])
def test_conv1x1_head_on_channel_blocked_input(N, Cin, Cout, H, W, compute):
    """forward == the fp32 1x1 kernel on the stored values bit for bit (same fmaf chain); weight / bias gradient against
    fp64 on the stored values (another summation order)."""
    g = _g(N + Cin + Cout + H + compute)
    x = torch.randn(N, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cout, Cin, 1, 1, generator=g) * 0.3).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    dy = torch.randn(N, Cout, H, W, generator=g).to(DEV)
    x8 = ops.C8.pack(x, compute)
    xs = x8.unpack()
    assert torch.equal(ops.conv1x1_fwd_c8(x8, w, b), ops.conv1x1_fwd(xs, w, b))
    dw, db = ops.conv1x1_wgrad_c8(x8, w, dy)
    wr = torch.zeros(Cout, Cin, 1, 1, dtype=torch.float64, requires_grad=True)
    F.conv2d(xs.cpu().double(), wr).backward(dy.cpu().double())
    _close(dw, wr.grad.float(), 1e-5, 2e-5 * max(1.0, wr.grad.abs().max().item()), "conv1x1 c8 wgrad")
    dbr = dy.cpu().double().sum((0, 2, 3)).float()
    _close(db, dbr, 1e-5, 2e-5 * max(1.0, dbr.abs().max().item()), "conv1x1 c8 dbias")
    dw2, db2 = ops.conv1x1_wgrad_c8(x8, w, dy, accumulate=True, dw=dw.clone(), db=db.clone())
    _close(dw2, 2 * wr.grad.float(), 1e-5, 4e-5 * max(1.0, wr.grad.abs().max().item()), "conv1x1 c8 wgrad accumulate")
    _close(db2, 2 * dbr, 1e-5, 4e-5 * max(1.0, dbr.abs().max().item()), "conv1x1 c8 dbias accumulate")


def test_weight_view_many_equals_single_views():
    """The batched weight-view launch (one per step) writes what the per-view entry point writes."""
    import ctypes as C
    from multi_task_breast_cancer_amd import _lib as L
    g = _g(17)
    lib = L.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    descs, outs_many, outs_single = [], [], []
    cases = [(24, 144, 24, 24, 0), (24, 144, 48, 96, 0), (48, 96, 0, 48, 1), (24, 72, 24, 24, 1), (96, 288, 96, 96, 1)] * 11      # 55 > one kernel-argument batch
    keep = []
    for Cout, Cin, off, cnt, mode in cases:
        w = torch.randn(Cout, Cin, 3, 3, generator=g).to(DEV)
        K = Cout + 8
        shape = (cnt, K, 3, 3) if mode else (Cout, cnt, 3, 3)
        a, b = torch.zeros(shape, device=DEV), torch.zeros(shape, device=DEV)
        keep.append(w)
        d = L.WViewDesc()
        d.w, d.dst, d.Cout, d.Cin, d.ci_off, d.ci_cnt, d.mode, d.k_off, d.K = w.data_ptr(), a.data_ptr(), Cout, Cin, off, cnt, mode, 8 if mode else 0, K if mode else 0
        descs.append(d)
        L.check(lib.mtbc_conv3x3_weight_view(w.data_ptr(), b.data_ptr(), Cout, Cin, off, cnt, mode, 8 if mode else 0, K if mode else 0, st), "wview")
        outs_many.append(a); outs_single.append(b)
    arr = (L.WViewDesc * len(descs))(*descs)
    L.check(lib.mtbc_conv3x3_weight_view_many(arr, len(descs), st), "wview_many")
    for a, b in zip(outs_many, outs_single):
        assert torch.equal(a, b) and a.abs().sum().item() > 0


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,segs,Cout,H,W", [(2, [24, 48], 24, 40, 64), (2, [48], 24, 256, 256), (3, [16, 16], 40, 16, 16), (5, [32, 8], 80, 8, 8)])
def test_conv3x3_dgrad_into_16bit_planar_segment(N, segs, Cout, H, W, compute):
    """mtbc_seg.accumulate = 2: the LAST dx segment is written as 16-bit planes = the fp32 result of the same launch rounded
    to nearest even, bit for bit; the other segments are untouched by the mode."""
    g = _g(N + Cout + H + compute)
    Cin = sum(segs)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(DEV)
    dz8 = ops.C8.pack(torch.randn(N, Cout, H, W, generator=g).to(DEV), compute)
    _, pd = ops.conv3x3_pack_lp(w, compute)
    ref = [torch.zeros(N, c, H, W, device=DEV) for c in segs]
    ops.conv3x3_dgrad_c8(dz8, w, ref, [0] * len(segs), pd)
    got = [torch.zeros(N, c, H, W, device=DEV) for c in segs[:-1]] + [torch.zeros(N, segs[-1], H, W, dtype=torch.int16, device=DEV)]
    ops.conv3x3_dgrad_c8(dz8, w, got, [0] * (len(segs) - 1) + [2], pd)
    dt = torch.bfloat16 if compute == 1 else torch.float16
    for a, b in zip(got[:-1], ref[:-1]):
        assert torch.equal(a, b)
    assert torch.equal(got[-1].view(dt), ref[-1].to(dt))


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 48, 48, 32, 32), (3, 96, 48, 8, 16), (1, 384, 192, 8, 8), (2, 48, 24, 16, 8), (2, 56, 40, 16, 16),
                                            (2, 512, 256, 8, 8), (32, 96, 48, 64, 64), (2, 24, 12, 16, 16)])
def test_convT_backward_reads_16bit_planar_dy(N, Cin, Cout, H, W, compute):
    """dy_type16: the k = 2 ConvT dgrad / wgrad on a 16-bit planar dy must equal the same kernels fed the fp32 tensor with
    the same (representable) values, bit for bit -- they rounded an fp32 dy to exactly these operands while loading it."""
    g = _g(N + Cin + Cout + H + compute)
    dt = torch.bfloat16 if compute == 1 else torch.float16
    x = torch.randn(N, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cin, Cout, 2, 2, generator=g) * 0.1).to(DEV)
    dy16 = torch.randn(N, Cout, 2 * H, 2 * W, generator=g).to(DEV).to(dt)
    dy = dy16.float()
    dy16i = dy16.view(torch.int16)
    dx_ref = ops.convT_dgrad(x, w, dy, 2, compute=compute)
    dx = ops.convT_dgrad(x, w, dy16i, 2, compute=compute, dy16=True)
    assert torch.equal(dx, dx_ref)
    pre = torch.randn(N, Cin, H, W, generator=g).to(DEV)
    assert torch.equal(ops.convT_dgrad(x, w, dy16i, 2, dx=pre.clone(), accumulate=True, compute=compute, dy16=True),
                       ops.convT_dgrad(x, w, dy, 2, dx=pre.clone(), accumulate=True, compute=compute))
    dw_ref, db_ref = ops.convT_wgrad(x, w, dy, 2, compute=compute)
    dw, db = ops.convT_wgrad(x, w, dy16i, 2, compute=compute, dy16=True)
    assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
    # and against fp64 on the rounded operands (the oracle of the mode)
    r = lambda t: t.to(dt).double()
    want = F.conv2d(dy.cpu().double(), r(w.cpu()), stride=2)
    assert (dx.cpu().double() - want).abs().max().item() <= 1e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,segs,Cout,H,W", C8_CASES + [(2, [24, 24, 24, 24, 24, 24], 24, 256, 256), (16, [24, 24], 24, 256, 256),
                                                        (1, [384, 384, 384], 512, 16, 16), (2, [192], 192, 32, 32)])
def test_conv3x3_channel_blocked_16bit_output_is_the_rounded_fp32_output(N, segs, Cout, H, W, compute):
    """out_layout = C8 (forward) / segment mode 3 (dgrad): the conv output / input gradient of the 16-bit modes stored as a
    channel-blocked 16-bit tensor = the fp32 planar result of the same launch, + bias, rounded once (RNE) -- the MFMAs only
    run transposed (channels on the rows)."""
    if Cout % 8:
        pytest.skip("channel-blocked outputs hold multiples of 8 channels")
    g = _g(N * 53 + Cout + H + compute)
    Cin = sum(segs)
    xs = [torch.randn(N, c, H, W, generator=g) for c in segs]
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, generator=g)
    dz = torch.randn(N, Cout, H, W, generator=g)
    xd = [x.to(DEV) for x in xs]
    wd, bd, dzd = w.to(DEV), b.to(DEV), dz.to(DEV)
    pf, pd = ops.conv3x3_pack_lp(wd, compute)
    x8 = [ops.C8.pack(x, compute) for x in xd]
    dz8 = ops.C8.pack(dzd, compute)
    z = ops.conv3x3_fwd_c8(x8, wd, bd, pf)
    z8 = ops.conv3x3_fwd_c8(x8, wd, bd, pf, out_c8=True)
    assert z8.data.shape == (N, Cout // 8, H * W, 8)
    want = ops.C8.pack(z, compute)
    assert torch.equal(z8.data, want.data), f"fwd: {(z8.unpack() - want.unpack()).abs().max().item():.3e}"
    if all(c % 8 == 0 for c in segs):
        d32 = [torch.empty(N, c, H, W, device=DEV) for c in segs]
        ops.conv3x3_dgrad_c8(dz8, wd, d32, [0] * len(segs), pd)
        d8 = [ops.C8(torch.empty(N, c // 8, H * W, 8, dtype=torch.int16, device=DEV), (N, c, H, W), compute) for c in segs]
        ops.conv3x3_dgrad_c8(dz8, wd, d8, [3] * len(segs), pd)
        for i in range(len(segs)):
            assert torch.equal(d8[i].data, ops.C8.pack(d32[i], compute).data), f"dgrad seg {i}"


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,C,H,W,affine", [(2, 24, 256, 256, True), (3, 48, 128, 128, True), (2, 96, 64, 64, False), (5, 16, 32, 32, True),
                                            (2, 8, 16, 16, True), (3, 8, 8, 8, False), (2, 16, 24, 40, True), (1, 24, 96, 96, True)])
def test_instnorm_on_channel_blocked_16bit_inputs(N, C, H, W, affine, compute):
    """z_layout / dy_layout = C8: InstanceNorm + LeakyReLU forward / backward reading the conv output (and the gradient)
    as 16-bit channel-blocked tensors = the same kernels on the unpacked fp32 planes (the statistics are those of the stored
    values): bit for bit where one workgroup owns a plane group; the cooperative kernels may split a plane differently per
    variant (team size follows the variant's register count), so there to fp32 re-association / one 16-bit ulp.  A planar
    fp32 partial gradient (dy_extra) is added in fp32 while loading."""
    g = _g(N + C + H + 7 * compute)
    z = (torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(DEV)
    dy = torch.randn(N, C, H, W, generator=g).to(DEV)
    ex = torch.randn(N, C, H, W, generator=g).to(DEV)
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV) if affine else None
    beta = (torch.randn(C, generator=g) * 0.1).to(DEV) if affine else None
    z8, dy8 = ops.C8.pack(z, compute), ops.C8.pack(dy, compute)
    zr, dyr = z8.unpack(), dy8.unpack()
    ya, mean_a, rstd_a, ypa = ops.instnorm_lrelu_fwd_c8(zr, gamma, beta, slope=0.1, compute=compute, want_planar=True)
    yb, mean_b, rstd_b, ypb = ops.instnorm_lrelu_fwd_c8(z8, gamma, beta, slope=0.1, want_planar=True)
    solo = H * W <= 4096
    ulp = 2.0 ** -7 if compute == 1 else 2.0 ** -10

    def same8(a8, b8, exact=solo):
        if exact:
            return torch.equal(a8.data, b8.data)
        a_, b_ = a8.unpack(), b8.unpack()
        return bool(((a_ - b_).abs() <= ulp * b_.abs() + 2e-6 * b_.abs().max()).all())

    def same(a_, b_):
        return torch.equal(a_, b_) if solo else torch.allclose(a_, b_, rtol=1e-5, atol=1e-5 * max(1.0, b_.abs().max().item()))

    assert same(mean_a, mean_b) and same(rstd_a, rstd_b)
    assert same8(ya, yb) and same(ypa, ypb)
    # ... and they ARE the statistics of the stored values
    ref_mean = zr.double().mean((2, 3)).reshape(-1)
    assert torch.allclose(mean_b.double(), ref_mean, rtol=1e-5, atol=1e-5)
    for extra in (None, ex):
        db1, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        da, dga, dba = ops.instnorm_lrelu_bwd_c8(zr, dyr + (extra if extra is not None else 0), mean_b, rstd_b, gamma, beta, slope=0.1, dbias_pre=db1, compute=compute)
        db, dgb, dbb = ops.instnorm_lrelu_bwd_c8(z8, dy8, mean_b, rstd_b, gamma, beta, slope=0.1, dbias_pre=db2, dy_extra=extra)
        # (backward: the variants are separate instantiations whose fused multiply-adds may be contracted differently)
        assert same8(da, db, False), (da.unpack() - db.unpack()).abs().max().item()
        noise = 1e-6 * da.unpack().abs().sum((0, 2, 3)).max().item()
        assert (db1 - db2).abs().max().item() <= max(1e-3, noise)
        if affine:
            assert torch.allclose(dga, dgb, rtol=1e-4, atol=1e-3 * max(1.0, dga.abs().max().item())) and torch.allclose(dba, dbb, rtol=1e-4, atol=1e-3 * max(1.0, dba.abs().max().item()))
    # mixed: channel-blocked z with a planar fp32 gradient
    dc, _, _ = ops.instnorm_lrelu_bwd_c8(z8, dyr, mean_b, rstd_b, gamma, beta, slope=0.1)
    dd, _, _ = ops.instnorm_lrelu_bwd_c8(zr, dyr, mean_b, rstd_b, gamma, beta, slope=0.1, compute=compute)
    assert same8(dc, dd, False)
    assert ops.coop_error(DEV) == 0


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,segs,Cout,H,W", [(2, [24], 24, 256, 256), (16, [24, 24], 24, 256, 256), (2, [48], 48, 128, 128), (3, [16], 40, 16, 16),
                                             (5, [32], 80, 8, 8), (2, [96, 96], 96, 16, 16), (1, [24, 48], 24, 40, 64), (2, [8], 16, 36, 64)])
def test_instnorm_statistics_from_the_conv_epilogue(N, segs, Cout, H, W, compute):
    """mtbc_conv3x3_args.stats_partial: the forward conv's epilogue leaves {sum, sum of squares} of its STORED (rounded)
    outputs per image, pixel subset and channel; summed they are the plane sums of the stored tensor, and the InstanceNorm
    forward fed with them (finalize + one streaming pass) equals the channel-group kernels that reduce the tensor
    themselves: statistics to fp32 re-association, outputs to one 16-bit ulp."""
    g = _g(N * 19 + Cout + H + compute)
    Cin = sum(segs)
    xs = [(torch.randn(N, c, H, W, generator=g) + 0.3).to(DEV) for c in segs]
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    gamma, beta = (torch.rand(Cout, generator=g) + 0.5).to(DEV), (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    pf, _ = ops.conv3x3_pack_lp(w, compute)
    x8 = [ops.C8.pack(x, compute) for x in xs]
    z8, part = ops.conv3x3_fwd_c8(x8, w, b, pf, out_c8=True, stats=True)
    assert torch.equal(z8.data, ops.conv3x3_fwd_c8(x8, w, b, pf, out_c8=True).data)      # the statistics change nothing else
    assert not torch.isnan(part).any(), "a (image, subset, channel) entry was not written"
    zr = z8.unpack().double()
    tot = part.double().sum(1)                                                            # (N, Cout, 2)
    s_ref, q_ref = zr.sum((2, 3)), (zr * zr).sum((2, 3))
    assert torch.allclose(tot[..., 0].cpu(), s_ref.cpu(), rtol=1e-5, atol=1e-5 * q_ref.abs().max().item() ** 0.5 * H * W ** 0.5)
    assert torch.allclose(tot[..., 1].cpu(), q_ref.cpu(), rtol=1e-5, atol=1e-3)
    ya, mean_a, rstd_a, ypa = ops.instnorm_lrelu_fwd_c8(z8, gamma, beta, slope=0.1, want_planar=True)
    yb, mean_b, rstd_b, ypb = ops.instnorm_lrelu_fwd_c8(z8, gamma, beta, slope=0.1, want_planar=True, stats=part)
    assert torch.allclose(mean_a, mean_b, rtol=1e-5, atol=1e-5) and torch.allclose(rstd_a, rstd_b, rtol=2e-5, atol=1e-6)
    ulp = 2.0 ** -7 if compute == 1 else 2.0 ** -10
    ua, ub = ya.unpack(), yb.unpack()
    assert bool(((ua - ub).abs() <= ulp * ua.abs() + 1e-5).all()), (ua - ub).abs().max().item()
    assert torch.allclose(ypa, ypb, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("extra", [False, True])
@pytest.mark.parametrize("N,K,C,H,W,affine", [(2, 48, 24, 256, 256, True), (2, 96, 48, 128, 128, True), (3, 96, 96, 64, 64, False), (2, 40, 16, 16, 16, True),
                                              (5, 32, 80, 8, 8, False), (1, 72, 24, 40, 64, True), (2, 192, 192, 32, 32, True)])
def test_gathered_dgrad_prepares_the_instnorm_backward(N, K, C, H, W, affine, extra, compute):
    """mtbc_conv3x3_args.norm_z / out_partial: a forward-type launch over the consumers' dz (the gathered dgrad) writes the
    tensor's gradient once in 16 bits -- its fp32 sum plus the other readers' fp32 partial, one RNE -- and leaves the two
    reductions of the InstanceNorm + LeakyReLU backward in stats_partial; the backward fed with them (finalize + one streaming
    pass) equals the channel-group kernel that reduces the tensors itself."""
    g = _g(N * 23 + K + C + H + compute)
    dzs = ops.C8.pack((torch.randn(N, K, H, W, generator=g)).to(DEV), compute)             # the consumers' dz, side by side
    wg = (torch.randn(C, K, 3, 3, generator=g) * (2.0 / (9 * K)) ** 0.5).to(DEV)             # their gathered weights
    z8 = ops.C8.pack((torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(DEV), compute)      # the tensor's own conv output
    gamma = (torch.rand(C, generator=g) + 0.5).to(DEV) if affine else None
    beta = (torch.randn(C, generator=g) * 0.1).to(DEV) if affine else None
    ex = torch.randn(N, C, H, W, generator=g).to(DEV) if extra else None
    pf, _ = ops.conv3x3_pack_lp(wg, compute)
    _, mean, rstd, _ = ops.instnorm_lrelu_fwd_c8(z8, gamma, beta, slope=0.1)
    dy32 = ops.conv3x3_fwd_c8([dzs], wg, None, pf)                                           # fp32 planar result of the same launch
    dy8, part = ops.conv3x3_fwd_c8([dzs], wg, None, pf, out_c8=True, norm=(z8, mean, rstd, gamma, beta, 0.1), out_partial=ex)
    want8 = ops.C8.pack(dy32 + ex if extra else dy32, compute)
    assert torch.equal(dy8.data, want8.data)
    assert not torch.isnan(part).any()
    # the reductions, recomputed from the stored tensors
    zr, dyr = z8.unpack().double(), dy8.unpack().double()
    xh = (zr - mean.double().view(N, C, 1, 1)) * rstd.double().view(N, C, 1, 1)
    pre = xh * (gamma.double().view(1, C, 1, 1) if affine else 1.0) + (beta.double().view(1, C, 1, 1) if affine else 0.0)
    gg = dyr * torch.where(pre > 0, 1.0, 0.1)
    tot = part.double().sum(1)
    s1, s2 = gg.sum((2, 3)), (gg * xh).sum((2, 3))
    scale = gg.abs().sum((2, 3)).max().item()
    assert (tot[..., 0] - s1).abs().max().item() <= 1e-5 * scale and (tot[..., 1] - s2).abs().max().item() <= 3e-5 * scale * max(1.0, xh.abs().max().item())
    db1, db2 = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    da, dga, dba = ops.instnorm_lrelu_bwd_c8(z8, dy8, mean, rstd, gamma, beta, slope=0.1, dbias_pre=db1)
    db, dgb, dbb = ops.instnorm_lrelu_bwd_c8(z8, dy8, mean, rstd, gamma, beta, slope=0.1, dbias_pre=db2, stats=part)
    ulp = 2.0 ** -7 if compute == 1 else 2.0 ** -10
    ua, ub = da.unpack(), db.unpack()
    assert bool(((ua - ub).abs() <= ulp * ua.abs() + 2e-6 * ua.abs().max()).all()), (ua - ub).abs().max().item()
    assert torch.equal(db2, torch.zeros_like(db2))                     # a conv bias in front of a norm: its gradient IS zero
    if affine:
        assert torch.allclose(dga, dgb, rtol=1e-4, atol=1e-3 * max(1.0, dga.abs().max().item()))
        assert torch.allclose(dba, dbb, rtol=1e-4, atol=1e-3 * max(1.0, dba.abs().max().item()))


@pytest.mark.parametrize("N,segs,Cout,H,W", [(2, [24], 24, 256, 256), (2, [48], 48, 128, 128), (3, [96], 96, 64, 64), (2, [96, 96], 96, 16, 16), (5, [32], 80, 8, 8)])
def test_bf16_mode_conv_output_stored_as_fp16(N, segs, Cout, H, W):
    """out_type = 2 / z_type = 2: in the bf16 mode the conv output z is stored as fp16 (same 2 bytes, 11 significant bits,
    saturated at +-65504) while the MFMA operands and the activation stay bf16: the stored tensor = the fp32 planar result of the
    same launch, clamped and rounded once to fp16; its InstanceNorm (statistics from the epilogue, and the channel-group kernels
    in both directions) = the same kernels fed with that fp16 tensor unpacked to fp32 planes."""
    g = _g(N * 29 + Cout + H)
    Cin = sum(segs)
    xs = [(torch.randn(N, c, H, W, generator=g) + 0.2).to(DEV) for c in segs]
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    b[0] = 1.0e5                                          # one channel past the fp16 range: it must saturate, not overflow
    gamma, beta = (torch.rand(Cout, generator=g) + 0.5).to(DEV), (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    pf, _ = ops.conv3x3_pack_lp(w, 1)
    x8 = [ops.C8.pack(x, 1) for x in xs]
    z32 = ops.conv3x3_fwd_c8(x8, w, b, pf)
    z8, part = ops.conv3x3_fwd_c8(x8, w, b, pf, out_c8=True, out_fp16=True, stats=True)
    assert z8.compute == 2
    want = ops.C8.pack(z32.clamp(-65504.0, 65504.0), 2)
    assert torch.equal(z8.data, want.data)
    zr = z8.unpack()
    assert bool(torch.isfinite(zr).all()) and zr[:, 0].abs().max().item() == 65504.0
    tot = part.double().sum(1)
    assert torch.allclose(tot[..., 0].cpu(), zr.double().sum((2, 3)).cpu(), rtol=1e-5, atol=1e-2)
    # InstanceNorm forward: statistics from the epilogue, bf16 activation
    y_a, mean_a, rstd_a, _ = ops.instnorm_lrelu_fwd_c8(z8, gamma, beta, slope=0.1, compute=1, stats=part)
    y_b, mean_b, rstd_b, _ = ops.instnorm_lrelu_fwd_c8(zr, gamma, beta, slope=0.1, compute=1)
    y_c, mean_c, rstd_c, _ = ops.instnorm_lrelu_fwd_c8(z8, gamma, beta, slope=0.1, compute=1)
    assert y_a.compute == 1
    for m_, r_, y_ in ((mean_a, rstd_a, y_a), (mean_c, rstd_c, y_c)):
        assert torch.allclose(m_[Cout > 1:], mean_b[Cout > 1:], rtol=1e-5, atol=1e-3) and torch.allclose(r_, rstd_b, rtol=3e-5, atol=1e-7)
        ua, ub = y_.unpack(), y_b.unpack()
        assert bool(((ua - ub).abs() <= 2.0 ** -7 * ub.abs() + 1e-5).all()), (ua - ub).abs().max().item()
    # backward: fp16 z, fp32 planar dy (the default plan) and bf16 channel-blocked dy
    dy = torch.randn(N, Cout, H, W, generator=g).to(DEV)
    dz_a, _, _ = ops.instnorm_lrelu_bwd_c8(z8, dy, mean_b, rstd_b, gamma, beta, slope=0.1, compute=1)
    dz_b, _, _ = ops.instnorm_lrelu_bwd_c8(zr, dy, mean_b, rstd_b, gamma, beta, slope=0.1, compute=1)
    dy8 = ops.C8.pack(dy, 1)
    dz_c, _, _ = ops.instnorm_lrelu_bwd_c8(z8, dy8, mean_b, rstd_b, gamma, beta, slope=0.1)
    dz_d, _, _ = ops.instnorm_lrelu_bwd_c8(zr, dy8.unpack(), mean_b, rstd_b, gamma, beta, slope=0.1, compute=1)
    for a_, b_ in ((dz_a, dz_b), (dz_c, dz_d)):
        assert a_.compute == 1
        ua, ub = a_.unpack(), b_.unpack()
        assert bool(((ua - ub).abs() <= 2.0 ** -7 * ub.abs() + 2e-6 * ub.abs().max()).all()), (ua - ub).abs().max().item()


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,C,H,W,with_dy", [(2, 24, 256, 256, True), (2, 24, 256, 256, False), (3, 48, 64, 64, True), (2, 16, 16, 16, False), (1, 24, 96, 96, True)])
def test_instnorm_backward_forms_the_rank1_head_gradient(N, C, H, W, with_dy, compute):
    """mtbc_instnorm_args.dy_rank1 / dy_rank1_w: the input gradient of a one-output 1x1 head, w[c] * dyhead[n, pixel], is formed
    inside the InstanceNorm backward = the same backward with that tensor written out and added to dy (or standing for a dy
    nothing else wrote)."""
    g = _g(N + C + H + 3 * compute)
    z = ops.C8.pack((torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(DEV), compute)
    dy = torch.randn(N, C, H, W, generator=g).to(DEV) if with_dy else None
    dyh = torch.randn(N, 1, H, W, generator=g).to(DEV)
    wh = torch.randn(C, generator=g).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.1).to(DEV)
    _, mean, rstd, _ = ops.instnorm_lrelu_fwd_c8(z, gamma, beta, slope=0.1)
    full = wh.view(1, C, 1, 1) * dyh + (dy if with_dy else 0)
    db1, db2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    a_, dga, dba = ops.instnorm_lrelu_bwd_c8(z, full.contiguous(), mean, rstd, gamma, beta, slope=0.1, dbias_pre=db1, compute=compute)
    b_, dgb, dbb, hdw, hdb = ops.instnorm_lrelu_bwd_c8(z, dy, mean, rstd, gamma, beta, slope=0.1, dbias_pre=db2, compute=compute, rank1=(dyh, wh),
                                                       rank1_grads=True)
    # ... and the head's own gradients: the weight gradient contracts the STORED activation (what the head's forward read)
    y8, _, _, _ = ops.instnorm_lrelu_fwd_c8(z, gamma, beta, slope=0.1, compute=compute)
    yr = y8.unpack().double()
    dw_ref = (yr * dyh.double()).sum((0, 2, 3))
    assert torch.allclose(hdw.double(), dw_ref, rtol=2e-4, atol=2e-4 * max(1.0, dw_ref.abs().max().item())), (hdw.double() - dw_ref).abs().max().item()
    assert abs(hdb.item() - dyh.double().sum().item()) <= 1e-4 * max(1.0, dyh.abs().sum().item() ** 0.5 * 10)
    ulp = 2.0 ** -7 if compute == 1 else 2.0 ** -10
    ua, ub = a_.unpack(), b_.unpack()
    assert bool(((ua - ub).abs() <= ulp * ua.abs() + 2e-6 * ua.abs().max()).all()), (ua - ub).abs().max().item()
    assert torch.allclose(dga, dgb, rtol=1e-4, atol=1e-3 * max(1.0, dga.abs().max().item()))
    assert torch.allclose(dba, dbb, rtol=1e-4, atol=1e-3 * max(1.0, dba.abs().max().item()))


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,C,H,W,with_dy", [(2, 24, 256, 256, True), (2, 48, 128, 128, False), (3, 96, 64, 64, True), (2, 16, 16, 16, False), (1, 24, 96, 80, True)])
def test_instnorm_backward_routes_the_maxpool_gradient(N, C, H, W, with_dy, compute):
    """mtbc_maxpool_args.argmax + mtbc_instnorm_args.dy_pool: the backward of MaxPool2d(2,2) formed inside the InstanceNorm
    backward of the tensor it pooled = the pool's own backward kernel writing its fp32 tensor, added to dy (or standing for a dy
    nothing else wrote).  Ties included (equal activations in a window: the FIRST maximum gets the gradient)."""
    g = _g(N + C + H + 5 * compute)
    z = ops.C8.pack((torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(DEV), compute)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.1).to(DEV)
    y8, mean, rstd, _ = ops.instnorm_lrelu_fwd_c8(z, gamma, beta, slope=0.1)
    yp8, arg = ops.maxpool2_fwd_c8(y8, want_argmax=True)
    assert torch.equal(yp8.data, ops.maxpool2_fwd_c8(y8).data)
    dyp = torch.randn(N, C, H // 2, W // 2, generator=g).to(DEV)
    dy = torch.randn(N, C, H, W, generator=g).to(DEV) if with_dy else None
    routed = ops.maxpool2_bwd_c8(y8, dyp)                                     # the pool's own backward (fp32 planar, 3/4 zeros)
    assert int((routed != 0).sum()) <= dyp.numel()
    full = routed + dy if with_dy else routed
    a_, dga, dba = ops.instnorm_lrelu_bwd_c8(z, full.contiguous(), mean, rstd, gamma, beta, slope=0.1, compute=compute)
    b_, dgb, dbb = ops.instnorm_lrelu_bwd_c8(z, dy, mean, rstd, gamma, beta, slope=0.1, compute=compute, pool=(dyp, arg))
    ulp = 2.0 ** -7 if compute == 1 else 2.0 ** -10
    ua, ub = a_.unpack(), b_.unpack()
    assert bool(((ua - ub).abs() <= ulp * ua.abs() + 2e-6 * ua.abs().max()).all()), (ua - ub).abs().max().item()
    assert torch.allclose(dga, dgb, rtol=1e-4, atol=1e-3 * max(1.0, dga.abs().max().item()))
    assert torch.allclose(dba, dbb, rtol=1e-4, atol=1e-3 * max(1.0, dba.abs().max().item()))


@pytest.mark.parametrize("N,C,H,W", [(2, 48, 128, 128), (3, 96, 64, 64), (2, 16, 16, 16)])
def test_instnorm_backward_adds_a_second_planar_gradient(N, C, H, W):
    """fp32 planar dy + n_dy_extra = 1: the gathered dgrad's own output buffer added while loading = the backward of the sum."""
    g = _g(N + C + H)
    z = ops.C8.pack((torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(DEV), 2)
    dy, ex = torch.randn(N, C, H, W, generator=g).to(DEV), torch.randn(N, C, H, W, generator=g).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.1).to(DEV)
    _, mean, rstd, _ = ops.instnorm_lrelu_fwd_c8(z, gamma, beta, slope=0.1, compute=1)
    a_, _, _ = ops.instnorm_lrelu_bwd_c8(z, (dy + ex).contiguous(), mean, rstd, gamma, beta, slope=0.1, compute=1)
    b_, _, _ = ops.instnorm_lrelu_bwd_c8(z, dy, mean, rstd, gamma, beta, slope=0.1, compute=1, dy_extra=ex)
    ua, ub = a_.unpack(), b_.unpack()
    assert bool(((ua - ub).abs() <= 2.0 ** -7 * ua.abs() + 2e-6 * ua.abs().max()).all()), (ua - ub).abs().max().item()


@pytest.mark.parametrize("compute,out_fp16", [(1, True), (1, False), (2, False)])
@pytest.mark.parametrize("N,Cout,H,W", [(2, 24, 256, 256), (3, 32, 40, 24), (1, 8, 8, 12), (2, 24, 96, 96)])
def test_stem_conv_on_the_16bit_path(N, Cout, H, W, compute, out_fp16):
    """The 1-channel stem in the 16-bit modes: fp32 operands and the stem kernel's fmaf order, output channel-blocked 16-bit
    (= the fp32 stem output, clamped for fp16 storage, rounded once) + InstanceNorm statistics of the stored values; its weight
    gradient from a channel-blocked dz = the fp32 weight-gradient kernel on the unpacked dz."""
    g = _g(N + Cout + H + compute)
    x = (torch.rand(N, 1, H, W, generator=g) * 255.0).to(DEV)
    w = (torch.randn(Cout, 1, 3, 3, generator=g) * 0.05).to(DEV)
    b = torch.randn(Cout, generator=g).to(DEV)
    z32 = ops.conv3x3_fwd([x], w, b)
    z8, part = ops.conv3x3_stem_fwd_c8(x, w, b, compute, out_fp16=out_fp16, stats=True)
    t = 2 if out_fp16 else compute
    want = ops.C8.pack(z32.clamp(-65504.0, 65504.0) if t == 2 else z32, t)
    assert z8.compute == t and torch.equal(z8.data, want.data)
    assert not torch.isnan(part).any()
    zr = z8.unpack().double()
    tot = part.double().sum(1)
    assert torch.allclose(tot[..., 0].cpu(), zr.sum((2, 3)).cpu(), rtol=1e-5, atol=1e-2)
    assert torch.allclose(tot[..., 1].cpu(), (zr * zr).sum((2, 3)).cpu(), rtol=1e-5, atol=1e-1)
    dz = torch.randn(N, Cout, H, W, generator=g).to(DEV)
    dz8 = ops.C8.pack(dz, compute)
    dw_ref, _ = ops.conv3x3_wgrad([x], dz8.unpack(), tuple(w.shape))
    dw, _ = ops.conv3x3_wgrad_c8([x], dz8, tuple(w.shape))
    assert torch.allclose(dw, dw_ref, rtol=1e-4, atol=1e-4 * max(1.0, dw_ref.abs().max().item())), (dw - dw_ref).abs().max().item()


@pytest.mark.parametrize("N,C,H,W", [(2, 24, 256, 256), (3, 96, 64, 64), (2, 16, 16, 16)])
def test_deferred_instnorm_parameter_gradients_equal_the_immediate_ones(N, C, H, W):
    """defer_dparams + mtbc_instnorm_dparam_many: the per-plane partials left in the workspace, reduced by the batched entry point
    = the reduction inside mtbc_instnorm_lrelu_bwd, bit for bit (same summation order)."""
    g = _g(N + C + H)
    z = ops.C8.pack((torch.randn(N, C, H, W, generator=g) * 2 + 0.5).to(DEV), 2)
    dy = torch.randn(N, C, H, W, generator=g).to(DEV)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(DEV), (torch.randn(C, generator=g) * 0.1).to(DEV)
    _, mean, rstd, _ = ops.instnorm_lrelu_fwd_c8(z, gamma, beta, slope=0.1, compute=1)
    b1, b2 = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    a_, dga, dba = ops.instnorm_lrelu_bwd_c8(z, dy, mean, rstd, gamma, beta, slope=0.1, compute=1, dbias_pre=b1)
    b_, dgb, dbb = ops.instnorm_lrelu_bwd_c8(z, dy, mean, rstd, gamma, beta, slope=0.1, compute=1, dbias_pre=b2, defer_dparams=True)
    assert torch.equal(a_.data, b_.data) and torch.equal(dga, dgb) and torch.equal(dba, dbb) and torch.equal(b1, b2)


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,segs,Cout,H,W", [(2, [24], 24, 256, 256), (2, [48], 48, 128, 128), (3, [96], 96, 64, 64), (2, [32], 16, 16, 24)])
def test_streaming_instnorm_also_writes_the_maxpool(N, segs, Cout, H, W, compute):
    """mtbc_instnorm_args.pool_y8 / pool_arg: the streaming normalisation (statistics from the conv epilogue) writes the 2x2
    max-pool of the activation it stores and the pool's argmax codes = the pool kernel run on that activation, bit for bit."""
    g = _g(N * 31 + Cout + H + compute)
    Cin = sum(segs)
    xs = [(torch.randn(N, c, H, W, generator=g) + 0.3).to(DEV) for c in segs]
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(DEV)
    gamma, beta = (torch.rand(Cout, generator=g) + 0.5).to(DEV), (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    pf, _ = ops.conv3x3_pack_lp(w, compute)
    z8, part = ops.conv3x3_fwd_c8([ops.C8.pack(x, compute) for x in xs], w, None, pf, out_c8=True, stats=True)
    y_a, mean_a, rstd_a, yp_a = ops.instnorm_lrelu_fwd_c8(z8, gamma, beta, slope=0.1, stats=part, want_planar=True)
    y_b, mean_b, rstd_b, yp_b, pool8, parg = ops.instnorm_lrelu_fwd_c8(z8, gamma, beta, slope=0.1, stats=part, want_planar=True, want_pool=True)
    assert torch.equal(y_a.data, y_b.data) and torch.equal(yp_a, yp_b) and torch.equal(mean_a, mean_b) and torch.equal(rstd_a, rstd_b)
    ref8, refarg = ops.maxpool2_fwd_c8(y_b, want_argmax=True)
    assert torch.equal(pool8.data, ref8.data) and torch.equal(parg, refarg)


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("pool", [False, True])
@pytest.mark.parametrize("N,segs,Cout,H,W", [(2, [24], 24, 256, 256), (2, [48], 48, 128, 128), (3, [96], 96, 64, 64), (2, [32], 16, 16, 24),
                                             (2, [192], 192, 8, 8)])
def test_streaming_instnorm_writes_16bit_planes(N, segs, Cout, H, W, compute, pool):
    """mtbc_instnorm_args.y16 beside y8: the planar copy of the activation as 16-bit planes of the output type = the fp32
    planes rounded once (RNE) = the values of the channel-blocked copy, bit for bit; y8 / statistics / pooled outputs as
    without it."""
    g = _g(N * 37 + Cout + H + compute)
    Cin = sum(segs)
    dt = torch.bfloat16 if compute == 1 else torch.float16
    xs = [(torch.randn(N, c, H, W, generator=g) + 0.3).to(DEV) for c in segs]
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (9 * Cin)) ** 0.5).to(DEV)
    gamma, beta = (torch.rand(Cout, generator=g) + 0.5).to(DEV), (torch.randn(Cout, generator=g) * 0.1).to(DEV)
    pf, _ = ops.conv3x3_pack_lp(w, compute)
    z8, part = ops.conv3x3_fwd_c8([ops.C8.pack(x, compute) for x in xs], w, None, pf, out_c8=True, stats=True)
    ref = ops.instnorm_lrelu_fwd_c8(z8, gamma, beta, slope=0.1, stats=part, want_planar=True, want_pool=pool)
    got = ops.instnorm_lrelu_fwd_c8(z8, gamma, beta, slope=0.1, stats=part, planar16=True, want_pool=pool)
    assert torch.equal(got[0].data, ref[0].data) and torch.equal(got[1], ref[1]) and torch.equal(got[2], ref[2])
    assert got[3].dtype == torch.int16 and torch.equal(got[3].view(dt), ref[3].to(dt))
    assert torch.equal(got[3].view(dt).float(), got[0].unpack())
    if pool:
        assert torch.equal(got[4].data, ref[4].data) and torch.equal(got[5], ref[5])


def test_instnorm_16bit_planes_need_the_streaming_pass():
    """y16 beside y8 without conv-epilogue statistics is refused (MTBC_E_UNSUPPORTED), not silently ignored."""
    from multi_task_breast_cancer_amd import _lib as L
    z = torch.randn(2, 16, 16, 16, generator=_g(5)).to(DEV)
    with pytest.raises(L.MtbcError):
        ops.instnorm_lrelu_fwd_c8(z, compute=1, planar16=True)


@pytest.mark.parametrize("compute", [1, 2])
@pytest.mark.parametrize("N,Cin,Cout,H,W", [(2, 48, 48, 32, 32), (3, 96, 48, 8, 16), (1, 384, 192, 8, 8), (2, 40, 24, 16, 8), (32, 96, 48, 64, 64)])
def test_convT_wgrad_reads_16bit_planar_x(N, Cin, Cout, H, W, compute):
    """x_type16: the k = 2 ConvT weight gradient on 16-bit planes of x (and dy) = the same kernel fed the fp32 tensors with the
    same (representable) values, bit for bit; fp32 x that is NOT representable gives the same result as its rounded planes."""
    from multi_task_breast_cancer_amd import _lib as L
    g = _g(N + Cin + Cout + H + compute)
    dt = torch.bfloat16 if compute == 1 else torch.float16
    x = torch.randn(N, Cin, H, W, generator=g).to(DEV)
    w = (torch.randn(Cin, Cout, 2, 2, generator=g) * 0.1).to(DEV)
    dy16 = torch.randn(N, Cout, 2 * H, 2 * W, generator=g).to(DEV).to(dt)
    x16 = x.to(dt)
    dw_ref, db_ref = ops.convT_wgrad(x, w, dy16.view(torch.int16), 2, compute=compute, dy16=True)
    dw, db = ops.convT_wgrad(x16.view(torch.int16), w, dy16.view(torch.int16), 2, compute=compute, dy16=True, x16=True)
    if N * H * W <= 4096:       # both launches split the pixels the same way (8 steps per split): the same sums in the same order
        assert torch.equal(dw, dw_ref) and torch.equal(db, db_ref)
    else:                       # the 16-bit-planes launch takes fewer, longer splits: fp32 re-association of the split sums
        assert (dw - dw_ref).abs().max().item() <= 1e-5 * dw_ref.abs().max().item()
        assert (db - db_ref).abs().max().item() <= 1e-5 * db_ref.abs().max().item()
    want = torch.einsum("nchw,ndhawb->cdab", x16.double().cpu(), dy16.double().cpu().view(N, Cout, H, 2, W, 2))
    assert (dw.cpu().double() - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())
    with pytest.raises(L.MtbcError):        # 16-bit x with an fp32 dy: no kernel, refused
        ops.convT_wgrad(x16.view(torch.int16), w, dy16.float(), 2, compute=compute, x16=True)


def test_program_run_on_two_streams_orders_a_forked_section():
    """mtbc_program_run_ms (MTBC_OP_SET_STREAM / EVENT_RECORD / EVENT_WAIT): ops between a fork and its join run on the second stream,
    ordered behind what the main stream had issued before the fork, and the main stream waits at the join.  (The step plan no longer
    emits forked sections -- two-stream backward measured slower, DESIGN.md -- so the entry point is exercised here.)"""
    import ctypes as C
    from multi_task_breast_cancer_amd import _lib as L
    from multi_task_breast_cancer_amd.engine import _mk
    lib = L.load()
    N, Cc, HW = 2, 16, 4096
    g = _g(5)
    x = torch.randn(N, Cc, 64, 64, generator=g).to(DEV)
    a8 = torch.zeros(N, Cc // 8, HW, 8, dtype=torch.int16, device=DEV)
    back = torch.zeros_like(x)
    ev = [C.c_void_p(), C.c_void_p()]
    for h in ev:
        L.check(lib.mtbc_event_create(C.byref(h)), "event_create")

    def sync(kind, event=0, index=0):
        op = _mk(kind)
        op.u.sync.event, op.u.sync.index = (ev[event] if kind != L.OP_SET_STREAM else None), index
        return op

    pack = _mk(L.OP_C8_PACK)
    pack.u.c8pack.src, pack.u.c8pack.src_batch_stride, pack.u.c8pack.dst = x.data_ptr(), Cc * HW, a8.data_ptr()
    pack.u.c8pack.N, pack.u.c8pack.C, pack.u.c8pack.HW, pack.u.c8pack.compute = N, Cc, HW, 1
    # main: pack | fork -> side: (waits for the pack) nothing else | join ; then unpack on the caller's side through the ordinary entry
    ops_ = [pack, sync(L.OP_EVENT_RECORD, 0), sync(L.OP_SET_STREAM, index=1), sync(L.OP_EVENT_WAIT, 0),
            sync(L.OP_EVENT_RECORD, 1), sync(L.OP_SET_STREAM, index=0), sync(L.OP_EVENT_WAIT, 1)]
    arr = (L.Op * len(ops_))(*ops_)
    side = torch.cuda.Stream()
    streams = (C.c_void_p * 2)(torch.cuda.current_stream().cuda_stream, side.cuda_stream)
    failed = C.c_int32(-1)
    L.check(lib.mtbc_program_run_ms(arr, 0, len(ops_), streams, 2, C.byref(failed)), "program_run_ms")
    L.check(lib.mtbc_c8_unpack(a8.data_ptr(), back.data_ptr(), N, Cc, HW, 1, C.c_void_p(torch.cuda.current_stream().cuda_stream)), "unpack")
    torch.cuda.synchronize()
    assert torch.equal(back, x.to(torch.bfloat16).float())
    for h in ev:
        lib.mtbc_event_destroy(h)


@pytest.mark.parametrize("gamma", [0.0, 2.0])
def test_focal_one_logit_is_bce_with_logits(gamma):
    """mtbc_focal_fwd_bwd with C == 1: the binary head's criterion.  gamma = 0, alpha = 1 = torch.nn.BCEWithLogitsLoss() (the reference's
    choice for n_classes == 2, experiment_init.py:241-242), loss and gradient; gamma = 2 = the focal modulation of the same per-sample term."""
    g = _g(12)
    x = (torch.randn(37, 1, generator=g) * 4.0)
    x[0, 0], x[1, 0] = 30.0, -30.0                      # saturated logits: the stable form
    t = torch.randint(0, 2, (37, 1), generator=g).float()
    xr = x.clone().double().requires_grad_(True)
    ce = torch.nn.functional.binary_cross_entropy_with_logits(xr, t.double(), reduction="none")
    ref = ((1.0 - torch.exp(-ce)) ** gamma * ce).mean() if gamma else torch.nn.BCEWithLogitsLoss()(xr, t.double())
    ref.backward()
    loss, dx = ops.focal(x.to(DEV), t.to(DEV), alpha=1.0, gamma=gamma, gscale=0.65)
    assert abs(loss.item() - ref.item()) < 1e-6 * max(1.0, abs(ref.item()))
    _close(dx, (0.65 * xr.grad).float(), 1e-5, 1e-7, "d BCE / d logit")


def test_focal_gamma_zero_is_cross_entropy():
    """`classification_criterion: CE` (experiment_init.py:257-261): torch.nn.CrossEntropyLoss(reduction='mean') on the one-hot float target =
    the focal kernel with gamma = 0 (what FusedTrainStep(cls_criterion="CE") runs), with and without class weights."""
    g = _g(13)
    x = torch.randn(29, 3, generator=g) * 3.0
    t = torch.nn.functional.one_hot(torch.randint(0, 3, (29,), generator=g), 3).float()
    for w in (None, torch.tensor([0.2, 0.5, 0.3])):
        xr = x.clone().double().requires_grad_(True)
        # FocalLoss.forward's reduction (criterions.py:14-24 with gamma = 0): mean over samples of the weighted per-sample CE
        ce = -(torch.log_softmax(xr, 1) * t.double() * (w.double() if w is not None else 1.0)).sum(1)
        ref = ce.mean()
        ref.backward()
        loss, dx = ops.focal(x.to(DEV), t.to(DEV), alpha=1.0, gamma=0.0, weight=None if w is None else w.to(DEV))
        assert abs(loss.item() - ref.item()) < 1e-6 * max(1.0, abs(ref.item()))
        _close(dx, xr.grad.float(), 1e-5, 1e-7, "d CE / d logits")
        if w is None:
            assert abs(loss.item() - torch.nn.CrossEntropyLoss()(x, t).item()) < 1e-6

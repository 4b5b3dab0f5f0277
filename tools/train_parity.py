"""End-to-end quality parity of the compute modes on Curated-BUSI-shaped synthetic data: trains the same network from
the same seed on the same batch stream in fp32 (the reference-parity arithmetic) and in the 16-bit MFMA modes, and
evaluates hard Dice (metrics.py:255-267 semantics) and 3-class accuracy on held-out batches.

    python tools/train_parity.py --steps 600 --batch 32 --size 256 --out profiles/r01_train_parity.json
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multi_task_breast_cancer_amd.experiment_init import init_multitask_model, init_optimizer
from multi_task_breast_cancer_amd.miscellany import seed_everything
from multi_task_breast_cancer_amd.synthetic import synthetic_batch
from multi_task_breast_cancer_amd.trainer import FusedTrainStep, dice_counts, dice_score_from_counts

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="MTUNetPlusPlus"); ap.add_argument("--steps", type=int, default=600)
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--size", type=int, default=256)
ap.add_argument("--lr", type=float, default=5e-4); ap.add_argument("--eval-every", type=int, default=100)
ap.add_argument("--eval-batches", type=int, default=8); ap.add_argument("--dtypes", default="f32,bf16")
ap.add_argument("--seed", type=int, default=1993); ap.add_argument("--out", default=""); ap.add_argument("--hard", action="store_true", help="the task that can fail: synthetic.synthetic_batch(hard=True)"); ap.add_argument("--hard-contrast", default="", help="lo,span of the hard task's lesion contrast (calibration)"); ap.add_argument("--loss-scale", type=float, default=0.0, help="fp16 mode: static loss scale instead of the library default (nets.HipMultiTaskNet.set_compute)"); ap.add_argument("--cosine", action="store_true", help="CosineAnnealingLR(T_max=steps, eta_min=1e-6) as config.yaml scheduler: cosine")
args = ap.parse_args()
dev = torch.device("cuda:0")
if args.hard_contrast:
    from multi_task_breast_cancer_amd import synthetic as _syn
    _syn.HARD_CONTRAST = tuple(float(v) for v in args.hard_contrast.split(","))
val = [synthetic_batch(args.batch, args.size, args.size, seed=10_000 + i, device=dev, hard=args.hard) for i in range(args.eval_batches)]

def evaluate(model):
    tot = torch.zeros(3, dtype=torch.float64, device=dev); correct = 0; n = 0
    with torch.no_grad():
        for img, mask, label in val:
            logits, segs = model(img)
            tot += dice_counts(segs[-1], mask)
            correct += int((logits[0].argmax(dim=1) == label.flatten().long()).sum().item()); n += label.numel()
    return dice_score_from_counts(tot), correct / n

report = {"config": vars(args), "runs": {}}
for dtype in args.dtypes.split(","):
    seed_everything(args.seed)
    model = init_multitask_model(args.arch, 1, 1, 3, deep_supervision=True).to(dev)
    model.set_compute(dtype)
    if args.loss_scale and dtype in ("f16", "fp16"):
        model.loss_scale = float(args.loss_scale)
    opt = init_optimizer(model, "Adam", args.lr)
    step = FusedTrainStep(model, opt, alpha=0.35, inversely_weighted=True)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=args.steps, eta_min=1e-6) if args.cosine else None
    curve = []; t0 = time.time(); run = 0.0
    for s in range(1, args.steps + 1):
        batch = synthetic_batch(args.batch, args.size, args.size, seed=s, device=dev, hard=args.hard)
        losses = step(*batch)
        if sched is not None:
            sched.step()
        if s % 20 == 0:
            run = float(losses[0].item())
        if s % args.eval_every == 0 or s == args.steps:
            step.check_nan()
            d, a = evaluate(model)
            curve.append({"step": s, "train_loss": run, "val_dice": d, "val_acc": a})
            print(f"[{dtype}] step {s:5d} loss {run:.4f} val dice {d:.4f} acc {a:.4f} ({time.time()-t0:.0f}s)", flush=True)
    report["runs"][dtype] = curve
base = report["runs"].get("f32")
if base:
    for k, c in report["runs"].items():
        if k != "f32":
            report[f"final_delta_{k}_vs_f32"] = {"dice_pt": 100 * (c[-1]["val_dice"] - base[-1]["val_dice"]),
                                                "acc_pt": 100 * (c[-1]["val_acc"] - base[-1]["val_acc"])}
print(json.dumps({k: v for k, v in report.items() if k.startswith("final")}))
if args.out:
    json.dump(report, open(args.out, "w"), indent=1)

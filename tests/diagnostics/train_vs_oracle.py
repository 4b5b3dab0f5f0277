"""north_star: "Dice/accuracy within +-0.2 pt of the reference on Curated-BUSI-shaped synthetic data".  Trains the CPU
oracle (pure-torch restatement of the reference step) and the HIP fp32 path from the SAME initial weights on the SAME
batch stream, then evaluates both on the same held-out batches (hard Dice of metrics.py:255-267, 3-class accuracy).
Small problem (the oracle runs at a few images/s): 64x64 images, batch 8.

    python tests/diagnostics/train_vs_oracle.py --steps 400 --out gpurun_out/train_vs_oracle.json
"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from multi_task_breast_cancer_amd.miscellany import seed_everything
from multi_task_breast_cancer_amd.nets import MTUNetPlusPlus, MTnnUNet
from multi_task_breast_cancer_amd.optim import FusedAdam
from multi_task_breast_cancer_amd.synthetic import synthetic_batch
from multi_task_breast_cancer_amd.trainer import FusedTrainStep
from oracle import torch_oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="MTUNetPlusPlus"); ap.add_argument("--steps", type=int, default=400)
ap.add_argument("--batch", type=int, default=8); ap.add_argument("--size", type=int, default=64)
ap.add_argument("--lr", type=float, default=5e-4); ap.add_argument("--eval-batches", type=int, default=16)
ap.add_argument("--out", default="")
args = ap.parse_args()
dev = torch.device("cuda:0")
torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))

seed_everything(1993)
prod = (MTUNetPlusPlus(in_channels=1, out_channels=1, n_classes=3, deep_supervision=True) if args.arch == "MTUNetPlusPlus"
        else MTnnUNet(1, 1, 3))
ref = O.build_oracle_model(args.arch, 1, 1, 3, True)
ref.load_state_dict(prod.state_dict())
prod = prod.to(dev)
opt = FusedAdam(prod, lr=args.lr, eps=1e-4)
step = FusedTrainStep(prod, opt, alpha=0.35, inversely_weighted=True)
ropt = torch.optim.Adam(ref.parameters(), lr=args.lr, eps=1e-4)
val = [synthetic_batch(args.batch, args.size, args.size, seed=50_000 + i, device=torch.device("cpu")) for i in range(args.eval_batches)]


def evaluate(model, device):
    tp = fp = fn = 0.0; correct = n = 0
    model.train(False)
    with torch.no_grad():
        for img, mask, label in val:
            logits, segs = model(img.to(device))
            seg = (torch.sigmoid(segs[-1]) > .5).cpu()
            gt = mask.bool()
            tp += float((seg & gt).sum()); fp += float((seg & ~gt).sum()); fn += float((~seg & gt).sum())
            correct += int((logits[0].cpu().argmax(dim=1) == label.flatten().long()).sum()); n += label.numel()
    model.train(True)
    return 100.0 * 2 * tp / max(2 * tp + fp + fn, 1.0), 100.0 * correct / n


log = []
t0 = time.time()
for s in range(args.steps):
    img, mask, label = synthetic_batch(args.batch, args.size, args.size, seed=s, device=torch.device("cpu"))
    lh = step(img.to(dev), mask.to(dev), label.to(dev))
    lo = O.train_step(ref, ropt, img, mask, label, 0.35, True, 3)
    if (s + 1) % 100 == 0 or s + 1 == args.steps:
        dh, ah = evaluate(prod, dev)
        do, ao = evaluate(ref, torch.device("cpu"))
        rec = {"step": s + 1, "loss_hip": float(lh[0]), "loss_oracle": float(lo[0]), "dice_hip": dh, "dice_oracle": do,
               "acc_hip": ah, "acc_oracle": ao}
        log.append(rec)
        print(json.dumps(rec), f"({time.time() - t0:.0f}s)", flush=True)
res = {"config": vars(args), "log": log, "final_delta_dice_pt": log[-1]["dice_hip"] - log[-1]["dice_oracle"],
       "final_delta_acc_pt": log[-1]["acc_hip"] - log[-1]["acc_oracle"]}
print(json.dumps({k: res[k] for k in ("final_delta_dice_pt", "final_delta_acc_pt")}))
if args.out:
    json.dump(res, open(args.out, "w"), indent=1)

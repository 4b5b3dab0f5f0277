"""Paired summary of the round-3 quality sweep (tools/experiments/tp_r3.sh): fp32 mode vs bf16 mode per seed, mean +- standard error of
the paired differences.   usage: python tools/experiments/tp_r3_summary.py gpurun_out/r3q > profiles/r03_quality_sweep.md"""
import glob, json, math, os, re, sys
d = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(d, "tp_s*.json")), key=lambda s: int(re.search(r"tp_s(\d+)", s).group(1))):
    r = json.load(open(f))
    seed = r["config"]["seed"]
    a, b = r["runs"]["f32"], r["runs"]["bf16"]
    rows.append((seed, a, b))
print("# fp32 mode vs bf16 mode: held-out Dice / accuracy over seeds (round 3)\n")
print("`tools/experiments/tp_r3.sh`: `python tools/train_parity.py --steps 6000 --batch 32 --size 256 --lr 3e-4 --cosine --eval-every 3000 --eval-batches 32 "
      "--dtypes f32,bf16 --seed S` -- U-Net++ MT on synthetic Curated-BUSI-shaped batches, the same batch stream and initial weights in both modes of a "
      "seed, hard Dice (metrics.py:255-267) and 3-class accuracy on 32 fresh batches (1024 images).  6000 steps instead of round 2's 3000: at 3000 steps "
      "8 of 15 runs had not finished converging (Dice 0.81 - 0.97, whatever the arithmetic), which made the seed-to-seed spread (+-4 pt) swamp any "
      "arithmetic effect; at 6000 every run of either mode ends at 0.987 +- 0.001.\n")
print("| seed | fp32 mode: Dice @3000 | @6000 | acc @6000 | bf16 mode: Dice @3000 | @6000 | acc @6000 | bf16 - fp32 @6000 (pt) |")
print("|---|---|---|---|---|---|---|---|")
diffs, adiffs = [], []
for seed, a, b in rows:
    da, db = a[-1]["val_dice"], b[-1]["val_dice"]
    diffs.append(100 * (db - da)); adiffs.append(100 * (b[-1]["val_acc"] - a[-1]["val_acc"]))
    print(f"| {seed} | {a[0]['val_dice']:.4f} | {da:.4f} | {a[-1]['val_acc']:.4f} | {b[0]['val_dice']:.4f} | {db:.4f} | {b[-1]['val_acc']:.4f} | {100 * (db - da):+.3f} |")
n = len(diffs)
if n >= 2:
    m = sum(diffs) / n
    se = math.sqrt(sum((x - m) ** 2 for x in diffs) / (n - 1) / n)
    ma = sum(adiffs) / n
    sea = math.sqrt(sum((x - ma) ** 2 for x in adiffs) / (n - 1) / n)
    mf = sum(r[1][-1]["val_dice"] for r in rows) / n
    mb = sum(r[2][-1]["val_dice"] for r in rows) / n
    print(f"\n**{n} paired seeds: mean Dice fp32 {mf:.4f}, bf16 {mb:.4f}; bf16 - fp32 = {m:+.3f} pt, standard error {se:.3f} pt "
          f"(max |difference| {max(abs(x) for x in diffs):.3f} pt); accuracy difference {ma:+.3f} +- {sea:.3f} pt.**  "
          f"north_star asks for +-0.2 pt: the paired difference is {'inside' if abs(m) + 2 * se < 0.2 else 'NOT shown to be inside'} that band at two standard errors.")

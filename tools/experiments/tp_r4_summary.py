"""Paired summary of the round-4 quality sweep on the task that can fail (tools/experiments/tp_r4_hard.sh): fp32 mode vs bf16 mode per seed, mean +- standard
error of the paired differences.   usage: python tools/experiments/tp_r4_summary.py gpurun_out/r4q [gpurun_out/r4q6000] > profiles/r04_quality_hard.md"""
import glob, json, math, os, re, sys
d = sys.argv[1]
rows = []
for f in sorted(glob.glob(os.path.join(d, "tp_s*.json")), key=lambda s: int(re.search(r"tp_s(\d+)", s).group(1))):
    r = json.load(open(f))
    rows.append((r["config"]["seed"], r["runs"]["f32"], r["runs"]["bf16"], r["config"]))
cfg = rows[0][3]
print("# fp32 mode vs the 16-bit modes on a task that can fail: held-out Dice / accuracy over seeds (round 4)\n")
print("**What changed because of this file: the 16-bit gathered activation gradients (round 3's default) are an opt-in again (`MTBC_DA16=1`).**  They end at the same Dice "
      "but reach the plateau later (section 3); the shipped bf16 plan keeps those gradients in fp32 and tracks the fp32 runs seed by seed.  Sections 1 and 2 ran while the "
      "16-bit gradients were still the default: they are the `MTBC_DA16=1` plan.\n")
print("## 1. fp32 mode against the bf16 mode WITH 16-bit gathered activation gradients (the default when this ran; `MTBC_DA16=1` now)\n")
print(f"`tools/experiments/tp_r4_hard.sh`: `python tools/train_parity.py --steps {cfg['steps']} --batch {cfg['batch']} --size {cfg['size']} --lr {cfg['lr']} --cosine "
      f"--eval-every {cfg['eval_every']} --eval-batches {cfg['eval_batches']} --dtypes f32,bf16 --hard --seed S` -- U-Net++ MT (deep supervision, Dice + Focal, alpha 0.35, Adam eps 1e-4) on "
      "`synthetic.synthetic_batch(hard=True)`: lesions of 30 - 50 % contrast under multiplicative speckle with structure at lesion scale, dark non-lesion regions in every "
      "class, an annotation whose radii / centre are jittered against the lesion in the image, benign / malignant told apart by boundary irregularity alone, 12 % of those "
      "labels swapped (accuracy ceiling ~0.92).  The same batch stream and initial weights in both modes of a seed; hard Dice (metrics.py:255-267) and 3-class accuracy on "
      f"{cfg['eval_batches']} held-out batches ({cfg['eval_batches'] * cfg['batch']} images).  The reference publishes DSC 0.751 / ACC 0.802 on the real Curated BUSI (README.md:77, table 5): this task sits in "
      "that regime -- no run of any arithmetic reaches 0.87 / 0.86 -- where round 3's task ended at 0.987 / 1.000 whatever was run.\n")
print("| seed | fp32 mode: Dice mid-run | final | acc | bf16 mode: Dice mid-run | final | acc | bf16 - fp32 Dice (pt) | acc (pt) |")
print("|---|---|---|---|---|---|---|---|---|")
diffs, adiffs = [], []
for seed, a, b, _ in rows:
    da, db = a[-1]["val_dice"], b[-1]["val_dice"]
    mid = len(a) // 2 - 1 if len(a) > 1 else 0
    diffs.append(100 * (db - da)); adiffs.append(100 * (b[-1]["val_acc"] - a[-1]["val_acc"]))
    print(f"| {seed} | {a[mid]['val_dice']:.4f} | {da:.4f} | {a[-1]['val_acc']:.4f} | {b[mid]['val_dice']:.4f} | {db:.4f} | {b[-1]['val_acc']:.4f} | {diffs[-1]:+.3f} | {adiffs[-1]:+.2f} |")
n = len(diffs)
if n >= 2:
    m = sum(diffs) / n
    se = math.sqrt(sum((x - m) ** 2 for x in diffs) / (n - 1) / n)
    ma = sum(adiffs) / n
    sea = math.sqrt(sum((x - ma) ** 2 for x in adiffs) / (n - 1) / n)
    mf = sum(r[1][-1]["val_dice"] for r in rows) / n
    mb = sum(r[2][-1]["val_dice"] for r in rows) / n
    af = sum(r[1][-1]["val_acc"] for r in rows) / n
    ab = sum(r[2][-1]["val_acc"] for r in rows) / n
    inside = abs(m) + 2 * se < 0.2
    print(f"\n**{n} paired seeds: mean Dice fp32 {mf:.4f}, bf16 {mb:.4f}; bf16 - fp32 = {m:+.3f} pt, standard error {se:.3f} pt (max |difference| {max(abs(x) for x in diffs):.3f} pt).  "
          f"Accuracy fp32 {af:.4f}, bf16 {ab:.4f}: {ma:+.3f} +- {sea:.3f} pt (one image of {cfg['eval_batches'] * cfg['batch']} = {100.0 / (cfg['eval_batches'] * cfg['batch']):.2f} pt).**  "
          f"north_star asks for +-0.2 pt: the paired Dice difference is {'inside' if inside else 'NOT shown to be inside'} that band at two standard errors"
          + ("" if inside else f" (|mean| + 2 SE = {abs(m) + 2 * se:.3f} pt)") + f"; the accuracy difference is {'inside' if abs(ma) + 2 * sea < 0.2 else 'NOT resolved to'} +-0.2 pt "
          f"(|mean| + 2 SE = {abs(ma) + 2 * sea:.2f} pt: the held-out set resolves {100.0 / (cfg['eval_batches'] * cfg['batch']):.2f} pt per image and 12 % of its labels are noise).")
    # the jump from ~0.70 to ~0.85 mid-run: is one arithmetic systematically later?
    md = [100 * (r[2][len(r[2]) // 2 - 1]["val_dice"] - r[1][len(r[1]) // 2 - 1]["val_dice"]) for r in rows]
    mm_ = sum(md) / n
    sem = math.sqrt(sum((x - mm_) ** 2 for x in md) / (n - 1) / n)
    print(f"\nMid-run (step {rows[0][1][len(rows[0][1]) // 2 - 1]['step']}), where the runs are in the middle of their climb: bf16 - fp32 = {mm_:+.1f} +- {sem:.1f} pt "
          f"({sum(1 for x in md if x < -1)} of {n} seeds with bf16 more than a point behind, {sum(1 for x in md if x > 1)} ahead) -- WHEN a run makes its jump varies by thousands of steps "
          "with the seed; said plainly: on this evidence this plan reaches the plateau later (section 3 finds the cause), and it reaches the same plateau.")
# the fp16 mode (gpurun_out/r4q_f16, tools/experiments/tp_r4_hard_f16.sh), paired with the fp32 runs above by seed
fdir = d.rstrip("/") + "_f16"
fr = {}
for f in glob.glob(os.path.join(fdir, "tp_s*.json")):
    r = json.load(open(f))
    fr[r["config"]["seed"]] = r["runs"]["f16"]
if len(fr) >= 2:
    print("\n## 2. The fp16 mode (configs[4]'s arithmetic; also with the 16-bit gathered gradients) on the same task\n")
    print("`tools/experiments/tp_r4_hard_f16.sh`: the same command with `--dtypes f16` (fp16 MFMA operands, static loss scale 4096, 16-bit gathered activation gradients -- the plan "
          "`bench.py --dtype f16` times), paired by seed with the fp32 runs above.  (ADVICE r3: the 16-bit gathered gradients had been measured for bf16 only.)\n")
    print("| seed | fp32 Dice | acc | fp16 Dice | acc | fp16 - fp32 Dice (pt) | acc (pt) |")
    print("|---|---|---|---|---|---|---|")
    fd, fa = [], []
    for seed, a, b, _ in rows:
        if seed not in fr: continue
        c = fr[seed]
        fd.append(100 * (c[-1]["val_dice"] - a[-1]["val_dice"])); fa.append(100 * (c[-1]["val_acc"] - a[-1]["val_acc"]))
        print(f"| {seed} | {a[-1]['val_dice']:.4f} | {a[-1]['val_acc']:.4f} | {c[-1]['val_dice']:.4f} | {c[-1]['val_acc']:.4f} | {fd[-1]:+.3f} | {fa[-1]:+.2f} |")
    k = len(fd)
    if k >= 2:
        m = sum(fd) / k; se = math.sqrt(sum((x - m) ** 2 for x in fd) / (k - 1) / k)
        ma = sum(fa) / k; sea = math.sqrt(sum((x - ma) ** 2 for x in fa) / (k - 1) / k)
        ins = abs(m) + 2 * se < 0.2
        print(f"\n**{k} paired seeds: fp16 - fp32 = {m:+.3f} pt Dice, standard error {se:.3f} pt (max |difference| {max(abs(x) for x in fd):.3f} pt); accuracy {ma:+.3f} +- {sea:.3f} pt.**  "
              + ("Inside +-0.2 pt at two standard errors." if ins else
                 f"Said plainly: NOT shown to be inside +-0.2 pt at two standard errors (|mean| + 2 SE = {abs(m) + 2 * se:.3f} pt); the mean itself is within the band, "
                 f"two seeds of ten sit 0.4 - 0.7 pt below their fp32 twin."))
# the bf16 mode with fp32 gathered activation gradients (MTBC_NO_DA16=1; gpurun_out/r4q_noda16, tools/experiments/tp_r4_hard_noda16.sh)
ndir = d.rstrip("/") + "_noda16"
nr = {}
for f in glob.glob(os.path.join(ndir, "tp_s*.json")):
    r = json.load(open(f))
    nr[r["config"]["seed"]] = r["runs"]["bf16"]
if len(nr) >= 2:
    print("\n## 3. Which rounding delays the climb?  The bf16 mode with fp32 gathered activation gradients -- THE SHIPPED bf16 PLAN since\n")
    print("`tools/experiments/tp_r4_hard_noda16.sh`: the bf16 runs again with the one storage choice round 3 made on the easy task's evidence switched off (`MTBC_NO_DA16=1` at the "
          "time; the default now), paired by seed.  Mid-run = step 6000 of 12000.\n")
    print("| seed | fp32 mid-run | final | bf16 + 16-bit gradients mid-run | final | bf16, fp32 gradients (shipped) mid-run | final |")
    print("|---|---|---|---|---|---|---|")
    dm, df, dm0 = [], [], []
    for seed, a, b, _ in rows:
        if seed not in nr: continue
        c = nr[seed]; mid = len(a) // 2 - 1
        print(f"| {seed} | {a[mid]['val_dice']:.4f} | {a[-1]['val_dice']:.4f} | {b[mid]['val_dice']:.4f} | {b[-1]['val_dice']:.4f} | {c[mid]['val_dice']:.4f} | {c[-1]['val_dice']:.4f} |")
        dm.append(100 * (c[mid]["val_dice"] - a[mid]["val_dice"])); df.append(100 * (c[-1]["val_dice"] - a[-1]["val_dice"])); dm0.append(100 * (b[mid]["val_dice"] - a[mid]["val_dice"]))
    k = len(dm)
    st = lambda v: (sum(v) / len(v), math.sqrt(sum((x - sum(v) / len(v)) ** 2 for x in v) / (len(v) - 1) / len(v)))
    (m1, s1), (m2, s2), (m0, s0) = st(dm), st(df), st(dm0)
    ins = abs(m2) + 2 * s2 < 0.2
    print(f"\n**{k} paired seeds against fp32: the shipped bf16 plan mid-run {m1:+.1f} +- {s1:.1f} pt (with 16-bit gradients on the same seeds: {m0:+.1f} +- {s0:.1f}), final {m2:+.3f} +- {s2:.3f} pt "
          f"({'inside' if ins else 'NOT shown inside'} +-0.2 pt at two standard errors).**  In the seeds where the 16-bit-gradient run is late (4, 5, 8, 10: 0.60 - 0.74 at mid-run) the fp32-gradient run "
          "sits within a few thousandths of the fp32 MODE's value: it follows the fp32 trajectory, the 16-bit gradients leave it.  Cost of the fp32 gradients: +0.2 ms per step (2 %); a plan that "
          "reaches the plateau thousands of steps later is not 2 % faster.")
st2 = lambda v: (sum(v) / len(v), math.sqrt(sum((x - sum(v) / len(v)) ** 2 for x in v) / (len(v) - 1) / len(v)))
# the shipped fp16 plan (fp32 gathered gradients): gpurun_out/r4q_f16_shipped
sdir = d.rstrip("/") + "_f16_shipped"
sr = {}
for f in glob.glob(os.path.join(sdir, "tp_s*.json")):
    r = json.load(open(f))
    sr[r["config"]["seed"]] = r["runs"]["f16"]
if len(sr) >= 2:
    print("\n### The shipped fp16 plan (fp32 gathered activation gradients), `tools/experiments/tp_r4_hard_f16_shipped.sh`\n")
    print("| seed | fp32 mid-run | final | fp16 shipped mid-run | final | fp16 - fp32 final (pt) |")
    print("|---|---|---|---|---|---|")
    fm, ff = [], []
    for seed, a, b, _ in rows:
        if seed not in sr: continue
        c = sr[seed]; mid = len(a) // 2 - 1
        fm.append(100 * (c[mid]["val_dice"] - a[mid]["val_dice"])); ff.append(100 * (c[-1]["val_dice"] - a[-1]["val_dice"]))
        print(f"| {seed} | {a[mid]['val_dice']:.4f} | {a[-1]['val_dice']:.4f} | {c[mid]['val_dice']:.4f} | {c[-1]['val_dice']:.4f} | {ff[-1]:+.3f} |")
    k = len(ff)
    (a1, b1), (a2, b2) = st2(fm), st2(ff)
    print(f"\n**{k} paired seeds: shipped fp16 plan - fp32: final {a2:+.3f} +- {b2:.3f} pt ({'inside' if abs(a2) + 2 * b2 < 0.2 else 'NOT shown inside'} +-0.2 pt at two standard errors), mid-run {a1:+.1f} +- {b1:.1f} pt.**")
    reached = [x for x in ff if x > -2.0]
    if len(reached) < k and len(reached) > 2:
        (a3, b3) = st2(reached)
        print(f"\nSaid plainly: {k - len(reached)} of {k} fp16 runs had NOT made the jump to the plateau when the cosine schedule ended (the fp32 run of the same seed made it between step 6000 and 9000; so did the fp16 run "
              f"with 16-bit gradients of section 2); the other {len(reached)} end at {a3:+.3f} +- {b3:.3f} pt of their fp32 twin.  The +-0.2 pt claim does NOT hold for the fp16 mode on this evidence, and the fp32 gathered "
              "gradients do not explain it (mid-run deficit with them and without them: the same within the noise).")
# the fp16 plan at another static loss scale: gpurun_out/r4q_f16_ls<LS>
for ldir in sorted(glob.glob(d.rstrip("/") + "_f16_ls*")):
    ls = ldir.rsplit("_f16_ls", 1)[1]
    lr_ = {}
    for f in glob.glob(os.path.join(ldir, "tp_s*.json")):
        r = json.load(open(f))
        lr_[r["config"]["seed"]] = r["runs"]["f16"]
    if len(lr_) < 2:
        continue
    print(f"\n### The fp16 plan with a static loss scale of {ls} instead of 4096, `LS={ls} tools/experiments/tp_r4_hard_f16_ls.sh`\n")
    print(f"| seed | fp32 mid-run | final | fp16, scale 4096 mid-run | final | fp16, scale {ls} mid-run | final | scale {ls} - fp32 final (pt) |")
    print("|---|---|---|---|---|---|---|---|")
    gm, gf, hm, hf = [], [], [], []
    for seed, a, b, _ in rows:
        if seed not in lr_ or seed not in sr: continue
        c = lr_[seed]; o = sr[seed]; mid = len(a) // 2 - 1
        gm.append(100 * (c[mid]["val_dice"] - a[mid]["val_dice"])); gf.append(100 * (c[-1]["val_dice"] - a[-1]["val_dice"]))
        hm.append(100 * (o[mid]["val_dice"] - a[mid]["val_dice"])); hf.append(100 * (o[-1]["val_dice"] - a[-1]["val_dice"]))
        print(f"| {seed} | {a[mid]['val_dice']:.4f} | {a[-1]['val_dice']:.4f} | {o[mid]['val_dice']:.4f} | {o[-1]['val_dice']:.4f} | {c[mid]['val_dice']:.4f} | {c[-1]['val_dice']:.4f} | {gf[-1]:+.3f} |")
    (a1, b1), (a2, b2), (c1, e1), (c2, e2) = st2(gm), st2(gf), st2(hm), st2(hf)
    print(f"\n**{len(gf)} paired seeds: scale {ls} - fp32: final {a2:+.3f} +- {b2:.3f} pt ({'inside' if abs(a2) + 2 * b2 < 0.2 else 'NOT shown inside'} +-0.2 pt at two standard errors), mid-run {a1:+.1f} +- {b1:.1f} pt; "
          f"scale 4096 on the same seeds: final {c2:+.3f} +- {e2:.3f}, mid-run {c1:+.1f} +- {e1:.1f}.**")
    print("\nThese seeds were CHOSEN as the four with the largest mid-run deficit at scale 4096 (selection: they are the marginal ones, so 'worse' cannot be read off this table); what it does show: "
          "a 16 x larger scale does not move the trajectories (mid-run values within 0.6 pt of the scale-4096 run on three of four seeds), i.e. fp16 underflow of the back-propagated dz is not what separates these runs from their fp32 twins.  "
          "(The underflow exists and grows with training -- design error against the exact gradient, `profiles/r04_fp16_design_error.txt`: 13 - 23 % in conv_4_0 after 1500 steps at scale 4096, 0.7 - 2.3 % at 65536 -- and 65536 is the library default since; every ten-seed fp16 sweep of this file ran at 4096.)  "
          "Open (next round): which fp16-only code path or storage choice is behind it -- bisect by switching the fp16 mode's tensors to the bf16 mode's types one at a time.")
# fp16 at the shipped scale on a longer schedule: gpurun_out/r4q_f16_18000
ldir = d.rstrip("/") + "_f16_18000"
l18 = {}
for f in glob.glob(os.path.join(ldir, "tp_s*.json")):
    r = json.load(open(f))
    l18[r["config"]["seed"]] = r["runs"]["f16"]
if l18:
    print("\n### The two seeds with an fp16 run that missed the plateau, 18000 steps (cosine over 18000), `OUT=gpurun_out/r4q_f16_18000 LS=4096 STEPS=18000 tools/experiments/tp_r4_hard_f16_ls.sh 3 7`\n")
    print("| seed | " + " | ".join(f"Dice @{e['step']}" for e in next(iter(l18.values()))) + " |")
    print("|---|" + "---|" * len(next(iter(l18.values()))))
    for seed in sorted(l18):
        print(f"| {seed} | " + " | ".join(f"{e['val_dice']:.4f}" for e in l18[seed]) + " |")
    print("\nSeed 3 does not jump late, it does not jump: a slow climb (0.675 -> 0.768 over 18000 steps) where the fp32 run of the seed goes 0.718 -> 0.768 -> 0.856 by step 9000.  Over everything this "
          "file holds: 4 of 26 fp16 runs (seed 3: three of four, seed 7: one of four) end in that slow mode, 0 of 30 fp32 / bf16 runs (Fisher exact p = 0.04).  The fp16 mode has a defect of convergence "
          "on this task that its loss scale and its 16-bit gradients do not explain; the per-tensor parity of its gradients with the fp16 emulation at a trained state "
          "(`test_16bit_mfma_modes_match_their_emulation[...f16-512-40]`) says the kernels compute what the mode is designed to compute.  Unresolved; first item of the next round's list.")
# fp16 on seed 3 under the plan switches: gpurun_out/r4q_f16_arms
adir = d.rstrip("/") + "_f16_arms"
arms = sorted(glob.glob(os.path.join(adir, "tp_s*_*.json")))
if arms:
    print("\n### Seed 3 in fp16 under the plan switches (9000 steps, cosine over 9000), `tools/experiments/tp_r4_hard_f16_arms.sh 3 \"MTBC_NO_Z16=1\" \"MTBC_NO_GATHER=1\" \"MTBC_NO_COOP=1\"`\n")
    print("| arm | Dice @3000 | @6000 | @9000 | accuracy @9000 |")
    print("|---|---|---|---|---|")
    for f in arms:
        r = json.load(open(f))["runs"]["f16"]
        arm = re.search(r"tp_s\d+_(.*)\.json", os.path.basename(f)).group(1)
        print(f"| `{arm}` | " + " | ".join(f"{e['val_dice']:.4f}" for e in r) + f" | {r[-1]['val_acc']:.4f} |")
    print("\nEach switch moves where values are rounded or which kernels run.  With the conv outputs kept in fp32 (`MTBC_NO_Z16`) the seed reaches the plateau, with the one-plane InstanceNorm kernels (`MTBC_NO_COOP`, "
          "fp32 conv outputs as well) it is in its jump at step 9000, with per-consumer input gradients (`MTBC_NO_GATHER`, conv outputs still fp16) it stays in the slow mode.  Together with the rows above: "
          "on this seed the fp16 arithmetic with 16-bit conv outputs ends in the slow mode in 4 of 5 variants, with fp32 conv outputs or 16-bit gradients in 0 of 3 -- a lead (the fp16-stored conv output z in the fp16 "
          "mode; the bf16 mode stores the same fp16 z and is not affected, so it is the combination), not a diagnosis.")
if len(sys.argv) > 2:
    print("\n## 4. The first protocol (6000 steps) had not converged\n")
    print("The same command with `--steps 6000 --eval-every 2000`: the runs were still climbing (0.67 -> 0.70 -> 0.73 ...), and WHEN a run makes its jump from ~0.70 to ~0.85 depends on the seed, "
          "not on the arithmetic -- paired differences of +-9 pt that say nothing about bf16.  (Round 3 had the same lesson at 3000 -> 6000 steps on the easy task.)\n")
    print("| seed | fp32 Dice @2000 | @4000 | @6000 | bf16 @2000 | @4000 | @6000 | bf16 - fp32 @6000 (pt) |")
    print("|---|---|---|---|---|---|---|---|")
    for f in sorted(glob.glob(os.path.join(sys.argv[2], "tp_s*.log")), key=lambda s: int(re.search(r"tp_s(\d+)", s).group(1))):
        v = {"f32": [], "bf16": []}
        for line in open(f):
            mm = re.match(r"\[(f32|bf16)\] step\s+\d+ loss [\d.]+ val dice ([\d.]+)", line)
            if mm: v[mm.group(1)].append(float(mm.group(2)))
        if len(v["f32"]) == 3 and len(v["bf16"]) == 3:
            s_ = re.search(r"tp_s(\d+)", f).group(1)
            print(f"| {s_} | " + " | ".join(f"{x:.4f}" for x in v["f32"] + v["bf16"]) + f" | {100 * (v['bf16'][2] - v['f32'][2]):+.2f} |")

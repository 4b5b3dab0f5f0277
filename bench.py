#!/usr/bin/env python3
"""Headline benchmark: training images/sec of the multi-task U-Net++ step (1-ch 256x256) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` from a bare interpreter (no WORLD_SIZE in the environment) starts the N ranks itself -- as child
processes of `torch.distributed.run`, BEFORE anything in this process touches the GPU -- and relays rank 0's JSON line.

One "step" = one optimisation step (pack -> fwd -> Dice+Focal -> bwd -> [all-reduce] -> Adam) on a per-GPU batch
of 32 synthetic images already resident in HBM (BASELINE.json configs[1]; weak scaling: global batch = 32*N).
Default compute mode is the one configs[1] names: bf16 MFMA operands (3x3 convs, k=2 ConvT), stored once in the MFMA's
16-bit channel-blocked layout; the conv outputs in front of InstanceNorm are stored in 16 bits too (fp16, 11 significant
bits); accumulators, gradient sums, norm statistics, parameters, losses and Adam are fp32.  At N=1 the same step is also
timed in the fp32 parity mode (`fp32_parity_mode`: the reference's arithmetic, the mode the 1e-4 parity tests run in).
Rank 0 prints ONE JSON line; `roofline` is measured live with HIP events around the dominant kernel family (the
implicit-GEMM conv3x3, forward + dgrad launches) and priced against the roof the kernel's arithmetic intensity puts it
under (fp32: MFMA; bf16: HBM); `cpu_baseline` times the CPU oracle on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_16BIT_MFMA_TFLOPS = 2500.0     # bf16 / fp16 dense
PEAK_HBM_BYTES = 8.0e12             # HBM3E spec (6.29 TB/s is the measured copy rate, MI355X_MICROARCH.md)


def conv_flops(op) -> float:
    a = op.u.conv3
    return 2.0 * a.N * a.H * a.W * a.Cin * a.Cout * 9


def cpu_baseline(arch: str, size: int, batch: int, steps: int):
    """The CPU oracle (pure-torch restatement of the reference step, pinned by tests/golden) on the host cores."""
    from oracle import torch_oracle as O
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(avail, 16))      # the GPU box grants 16 cores per GPU; more threads only oversubscribe
    torch.set_num_threads(threads)
    O.seed_everything(1993)
    model = O.build_oracle_model(arch, 1, 1, 3, True)
    opt = O.make_adam(model, 1e-4)
    img, mask, label = O.synthetic_batch(batch, size, size, seed=0)
    O.train_step(model, opt, img, mask, label, 0.5, True, 3)            # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        O.train_step(model, opt, img, mask, label, 0.5, True, 3)
    dt = time.perf_counter() - t0
    out = {"value": round(batch * steps / dt, 3), "unit": "images/sec", "cores": threads, "kind": "port",
           "sample": f"{steps} steps of {arch} B={batch} {size}x{size} fp32 after 1 warm-up ({dt:.1f} s)"}
    # BASELINE.json configs[0]: the reference's own CPU-runnable case (single-task segmentation, B=4, 1x128x128,
    # src/training_segmentation.py) on the oracle's nnUNet2021 restatement (pinned: tests/golden/seg_nnunet_step.npz) --
    # plumbing, not a kernel target
    O.seed_everything(1993)
    seg = O.OracleSegNnUNet(1, 1)
    sopt = O.make_adam(seg, 1e-4)
    simg, smask, _ = O.synthetic_batch(4, 128, 128, seed=1)
    O.seg_train_step(seg, sopt, simg, smask, True)
    t0 = time.perf_counter()
    for _ in range(5):
        sl, _ = O.seg_train_step(seg, sopt, simg, smask, True)
    sdt = time.perf_counter() - t0
    out["configs0_plumbing"] = {"value": round(4 * 5 / sdt, 3), "unit": "images/sec", "cores": threads, "kind": "port",
                                "sample": f"5 steps of single-task nnUNet2021 seg (Dice, deep supervision, Adam) B=4 128x128 fp32 "
                                          f"after 1 warm-up ({sdt:.1f} s), loss {float(sl):.4f}"}
    return out


def note(msg: str) -> None:
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def conv_bytes(op, kind, L) -> float:
    """Algorithmic HBM bytes of one conv3x3 launch: every operand plane once, in the storage the launch is
    handed -- fp32 planes, or 2-byte elements for the tensors an operand_layout = C8 launch reads."""
    a = op.u.conv3
    px = float(a.N * a.H * a.W)
    rd = 2.0 if a.operand_layout == L.LAYOUT_C8 else 4.0
    w = 9.0 * a.Cin * a.Cout * (2.0 if a.compute else 4.0)
    if kind == L.OP_CONV3_FWD:          # (a gathered dgrad is a forward-type launch that may add to its output)
        return (a.Cin * rd + a.Cout * (2.0 if a.out_layout == L.LAYOUT_C8 else (8.0 if a.out_accumulate else 4.0))) * px + w
    if kind == L.OP_CONV3_WGRAD:
        return (a.Cin + a.Cout) * rd * px + 9.0 * a.Cin * a.Cout * 4.0
    out = 0.0
    for i in range(a.n_in):          # dx segments: fp32 written (+ re-read when accumulated), or 16-bit (accumulate = 2 planes, 3 channel-blocked)
        m = a.in_[i].accumulate
        out += a.in_[i].channels * (2.0 if m >= 2 else (8.0 if m == 1 else 4.0))
    return (a.Cout * rd + out) * px + w


def time_op(prog, i, reps=3) -> float:
    """Seconds per launch of op i of a step program, HIP events on the stream the program runs on."""
    ms = 0.0
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); prog.run(i, 1); e.record(); e.synchronize()
        ms += s.elapsed_time(e)
    return ms / reps / 1e3


def run_mode(args, dtype, dev, rank, world, dist, want_roofline):
    """Build the model in one compute mode, time `steps` optimisation steps, optionally price the kernels."""
    from multi_task_breast_cancer_amd import _lib
    from multi_task_breast_cancer_amd.experiment_init import init_multitask_model, init_optimizer
    from multi_task_breast_cancer_amd.miscellany import seed_everything
    from multi_task_breast_cancer_amd.synthetic import synthetic_batch
    from multi_task_breast_cancer_amd.trainer import FusedTrainStep

    seed_everything(1993)                                          # identical initial weights on every rank
    model = init_multitask_model(args.arch, sequences=1, regions=1, n_classes=3, deep_supervision=True).to(dev)
    model.set_compute(dtype)
    opt = init_optimizer(model, "Adam", 1e-4)
    step = FusedTrainStep(model, opt, alpha=0.5, inversely_weighted=True, distributed=world > 1)
    batches = [synthetic_batch(args.batch, args.size, args.size, seed=s, device=dev, rank=rank) for s in range(2)]
    if args.host_input:
        batches = [tuple(t.cpu().pin_memory() for t in b) for b in batches]

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    note(f"[{dtype}] rank {rank}/{world}: model + {len(batches)} synthetic batches ready; warm-up x{args.warmup}")
    for i in range(args.warmup):
        step(*batches[i % 2])
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(*batches[i % 2])
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    step.check_nan()
    losses = step.losses.cpu().tolist()
    note(f"[{dtype}] timed {args.steps} steps in {dt:.3f} s; loss {losses[0]:.5f}")
    res = {"value": round(args.batch * world * args.steps / dt, 2), "ms_per_step": round(dt / args.steps * 1e3, 3),
           "final_loss": round(losses[0], 6)}
    if not want_roofline:
        return res

    st = step._st
    L = _lib
    peak = PEAK_F32_MFMA_TFLOPS if dtype == "f32" else PEAK_16BIT_MFMA_TFLOPS
    # ---- the 3x3 convolutions: every MFMA launch of a step, timed one by one
    fam = {"igemm": [0.0, 0.0, 0.0, 0, 0.0, 0, 0.0], "wgrad": [0.0, 0.0, 0.0, 0, 0.0, 0, 0.0]}   # flops, bytes, seconds, launches, roofline seconds, HBM-bound launches, their seconds
    for prog in (st.programs["fwd"], st.programs["bwd"]):
        for i in range(prog.n):
            op = prog.array[i]
            c = op.u.conv3
            if op.kind in (L.OP_CONV3_FWD, L.OP_CONV3_DGRAD) and c.w_packed:          # Cin=1 stem runs the direct kernel
                f = fam["igemm"]
            elif op.kind == L.OP_CONV3_WGRAD and c.Cin >= 8:
                f = fam["wgrad"]
            else:
                continue
            fl_, by_, t_ = conv_flops(op), conv_bytes(op, op.kind, L), time_op(prog, i)
            f[0] += fl_; f[1] += by_; f[2] += t_; f[3] += 1
            # each launch against ITS OWN roof: the level-0 launches sit under the HBM roof, the deep levels (K = 9 x 384 .. 1152 on
            # 16 x 16 maps) under the MFMA roof -- one byte rate over both families prices the deep levels against the wrong roof
            t_hbm, t_mfma = by_ / PEAK_HBM_BYTES, fl_ / (peak * 1e12)
            f[4] += max(t_hbm, t_mfma)
            if t_hbm >= t_mfma:
                f[5] += 1; f[6] += t_
    fl = fam["igemm"][0] + fam["wgrad"][0]
    by = fam["igemm"][1] + fam["wgrad"][1]
    sec = fam["igemm"][2] + fam["wgrad"][2]
    balance = peak * 1e12 / PEAK_HBM_BYTES                       # FLOP per byte at which the two roofs cross
    bound = "mfma" if fl / by > balance else "hbm"
    names = {"f32": ("conv3x3_igemm_dma_kernel", "conv3x3_wgrad_mfma_kernel"),
             "bf16": ("conv3x3_igemm_c8_kernel", "conv3x3_wgrad_c8 / c8w / c8i kernels"),
             "f16": ("conv3x3_igemm_c8_kernel", "conv3x3_wgrad_c8 / c8w / c8i kernels")}[dtype]
    # HBM traffic per launch comes from PMC counters, which cannot be collected inside a timed run: the figure is read
    # from the committed summary of the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this same command
    # (tools/profile_round.sh); `traffic_source` names the file, and the field is null when no summary fits the build
    traffic, traffic_source, traffic_step = None, None, None
    fam_prefix = {"f32": ("conv3x3_igemm_dma_kernel", "conv3x3_igemm_kernel"),
                  "bf16": ("conv3x3_igemm_c8_kernel", "conv3x3_igemm_c8_ring_kernel"),
                  "f16": ("conv3x3_igemm_c8_kernel", "conv3x3_igemm_c8_ring_kernel")}[dtype]
    for tag in ("r04", "r03", "r02c", "r02b", "r02", "r01"):
        tpath = os.path.join(ROOT, "profiles", f"{tag}_hbm_traffic_{dtype}.json")
        if not os.path.exists(tpath):
            continue
        ks = json.load(open(tpath))["kernels"]
        # the SAME launch set as the algorithmic bytes above: every template instance of every kernel of the family (the ring kernel of the
        # deep levels included -- round 3 selected `startswith("conv3x3_igemm_c8_kernel")` only and compared it with bytes over all launches)
        sel = [v for k, v in ks.items() if k.split("<")[0] in fam_prefix]
        if sel:
            w = sum(v["launches_in_trace"] for v in sel)
            traffic = round(sum(v["hbm_bytes_per_launch"] * v["launches_in_trace"] for v in sel) / w)
            traffic_step = traffic * fam["igemm"][3]          # launch-weighted mean x this step's launches of the family
            traffic_source = f"profiles/{tag}_hbm_traffic_{dtype}.json (separate rocprofv3 --pmc passes, FETCH_SIZE x2 + WRITE_SIZE; not measured in this run)"
            break
    ig = fam["igemm"]
    roof = {"bound": bound, "kernel": f"{names[0]} (fwd + dgrad launches of the 3x3 convs)",
            "arithmetic_intensity_flop_per_byte": round(ig[0] / ig[1], 1), "roof_crossover_flop_per_byte": round(balance, 1)}
    if bound == "mfma":
        roof.update({"achieved": round(ig[0] / ig[2] / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                     "frac": round(ig[0] / ig[2] / 1e12 / peak, 4)})
    else:
        roof.update({"achieved": round(ig[1] / ig[2] / 1e9, 1), "peak": PEAK_HBM_BYTES / 1e9, "unit": "GB/s",
                     "frac": round(ig[1] / ig[2] / PEAK_HBM_BYTES, 4)})
    roof.update({"traffic": traffic, "traffic_source": traffic_source,
                 "traffic_GB_per_step": None if traffic_step is None else round(traffic_step / 1e9, 3),
                 "algorithmic_GB_per_step": round(ig[1] / 1e9, 3),
                 "traffic_over_algorithmic": None if traffic_step is None else round(traffic_step / ig[1], 3),
                 "launches_per_step": ig[3], "avg_launch_ms": round(ig[2] / ig[3] * 1e3, 4),
                 "algorithmic_gflop_per_launch": round(ig[0] / ig[3] / 1e9, 3),
                 "algorithmic_MB_per_launch": round(ig[1] / ig[3] / 1e6, 1),
                 "tflops": round(ig[0] / ig[2] / 1e12, 2), "frac_of_mfma_peak": round(ig[0] / ig[2] / 1e12 / peak, 4),
                 "GBps": round(ig[1] / ig[2] / 1e9, 1), "frac_of_hbm_peak": round(ig[1] / ig[2] / PEAK_HBM_BYTES, 4)})
    roof["per_launch_roofs"] = {"frac": round(ig[4] / ig[2], 4), "hbm_bound_launches": ig[5], "mfma_bound_launches": ig[3] - ig[5],
                                "hbm_bound_share_of_time": round(ig[6] / ig[2], 3),
                                "note": "sum over launches of max(bytes / 8 TB/s, flops / MFMA peak) / sum of measured launch times"}
    wg = fam["wgrad"]
    roof["wgrad"] = {"kernel": names[1] + " + split-K reduce", "tflops": round(wg[0] / wg[2] / 1e12, 2),
                     "frac_of_mfma_peak": round(wg[0] / wg[2] / 1e12 / peak, 4), "GBps": round(wg[1] / wg[2] / 1e9, 1),
                     "frac_of_hbm_peak": round(wg[1] / wg[2] / PEAK_HBM_BYTES, 4), "launches_per_step": wg[3],
                     "avg_launch_ms": round(wg[2] / max(1, wg[3]) * 1e3, 4)}
    roof["all_3x3_conv"] = {"tflops": round(fl / sec / 1e12, 2), "frac_of_mfma_peak": round(fl / sec / 1e12 / peak, 4),
                            "share_of_step": round(sec / (dt / args.steps), 3)}
    res["roofline"] = roof
    # ---- HBM-bound kernels named by north_star (norm / upsample): algorithmic bytes / HIP-event time
    hb = {"in_fwd": [0.0, 0.0, 0], "in_bwd": [0.0, 0.0, 0], "convT_fwd": [0.0, 0.0, 0], "convT_dgrad": [0.0, 0.0, 0],
          "convT_wgrad": [0.0, 0.0, 0], "c8_pack": [0.0, 0.0, 0]}
    for pname in ("fwd", "bwd"):
        prog = st.programs[pname]
        for i in range(prog.n):
            op = prog.array[i]
            if op.kind == L.OP_IN_FWD:
                a = op.u.inorm; e = a.N * a.C * a.H * a.W
                ob = (2 + (2 if a.y16 else 4 if a.y else 0)) if a.y8 else (2 if a.y16 else 4)        # channel-blocked (+ 16-bit / fp32 planes) / 16-bit planes / fp32 planes
                zb = 2 if a.z_layout == L.LAYOUT_C8 else 4                                      # conv output stored in 16 bits (channel-blocked) or fp32 planes
                h = hb["in_fwd"]; h[0] += e * (zb + ob); h[1] += time_op(prog, i); h[2] += 1      # read z, write y
            elif op.kind == L.OP_IN_BWD:
                a = op.u.inorm; e = a.N * a.C * a.H * a.W
                rb = (2 if a.z_layout == L.LAYOUT_C8 else 4) + ((2 + 4 * a.n_dy_extra) if a.dy_layout == L.LAYOUT_C8 else 4)
                h = hb["in_bwd"]; h[0] += e * (rb + (2 if (a.dz16 or a.dz8) else 4)); h[1] += time_op(prog, i); h[2] += 1     # read z, dy (+ fp32 partial); write dz
            elif op.kind in (L.OP_CONVT_FWD, L.OP_CONVT_DGRAD, L.OP_CONVT_WGRAD):
                a = op.u.convT; px = a.N * a.H * a.W
                # bytes per element of the large tensor (y / dy) and of the small one (x / dx): 16-bit where the launch stores them so
                ob = 2 if ((op.kind == L.OP_CONVT_FWD and a.y_layout == L.LAYOUT_C8) or (op.kind != L.OP_CONVT_FWD and a.dy_type16)) else 4
                h = hb[{L.OP_CONVT_FWD: "convT_fwd", L.OP_CONVT_DGRAD: "convT_dgrad", L.OP_CONVT_WGRAD: "convT_wgrad"}[op.kind]]
                xb = 2 if ((op.kind == L.OP_CONVT_FWD and a.x_layout == L.LAYOUT_C8) or (op.kind == L.OP_CONVT_WGRAD and a.x_type16)) else 4
                h[0] += px * (a.Cin * xb + a.Cout * a.k * a.k * ob); h[1] += time_op(prog, i); h[2] += 1
            elif op.kind in (L.OP_C8_PACK, L.OP_C8_PACK16):                               # fp32 / 16-bit planes -> channel-blocked 16-bit
                a = op.u.c8pack; e = a.N * a.C * a.HW
                h = hb["c8_pack"]; h[0] += e * (6 if op.kind == L.OP_C8_PACK else 4); h[1] += time_op(prog, i); h[2] += 1
    res["roofline_hbm"] = {k: {"bound": "hbm", "achieved": round(v[0] / v[1] / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(v[0] / v[1] / 8e12, 4), "launches_per_step": v[2],
                               "algorithmic_MB_per_launch": round(v[0] / max(1, v[2]) / 1e6, 1)}
                           for k, v in hb.items() if v[2]}
    return res


def spawn_ranks(n: int) -> int:
    """Start `n` ranks of this script under torch.distributed.run (127.0.0.1 rendezvous, a free port) and wait."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    note(f"no WORLD_SIZE in the environment: starting {n} ranks: {' '.join(cmd[1:8])} ...")
    return subprocess.call(cmd, env=env)


def workload_labels(args, world: int):
    """(metric, workload tail, BASELINE.json configs index or None) for what was actually run."""
    arch_label = {"MTUNetPlusPlus": "U-Net++ MT", "MTnnUNet": "nnU-Net MT"}.get(args.arch, args.arch)
    metric = f"training images/sec (1-ch {args.size}x{args.size}, {arch_label})"
    idx = None
    if args.arch == "MTUNetPlusPlus" and args.size == 256 and args.batch == 32 and args.dtype == "bf16":
        idx = 1 if world == 1 else (3 if world == 8 else None)
        tail = ("BASELINE.json configs[1]" if world == 1 else
                f"BASELINE.json configs[3] (global batch 256 on 8 GPUs){'' if world == 8 else f' at {world} GPUs: per-GPU shape of configs[1]'}")
    elif args.arch == "MTnnUNet" and args.size == 256 and args.batch == 64 and world == 1:
        idx, tail = 2, "BASELINE.json configs[2]"
    elif args.size == 512 and args.dtype == "f16" and args.batch == 16:
        idx = 4 if world == 8 else None
        tail = "BASELINE.json configs[4]" + ("" if world == 8 else f": its per-GPU shape (global batch 128 on 8 GPUs) at {world} GPU(s)")
    else:
        tail = "not a BASELINE.json configuration"
    return metric, tail, idx


def stub_main(args, rank: int, world: int) -> None:
    """`--cpu-stub`: the launcher and the one-line contract on CPU ranks (gloo), with a stand-in for the step."""
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("gloo")
    g = torch.Generator().manual_seed(rank)
    w = torch.randn(64, 64, generator=g)

    def step():
        y = (w @ w).sum().reshape(1)
        if world > 1:
            dist.all_reduce(y)
        return y

    for _ in range(args.warmup):
        step()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    metric, tail, idx = workload_labels(args, world)
    if rank == 0:
        print(json.dumps({"metric": metric, "value": round(args.batch * world * args.steps / dt, 2), "unit": "images/sec", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
                          "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
                          "config": {"workload": f"CPU stand-in step (launcher rehearsal), {tail}", "baseline_config_index": idx,
                                     "global_batch": args.batch * world, "parallelism": f"dp{world}"},
                          "env": {"not_reportable": ["--cpu-stub"]}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arch", default="MTUNetPlusPlus")
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="bf16", choices=["f32", "bf16", "f16"],
                    help="MFMA operand type of the 3x3 convs / k=2 ConvT (BASELINE.json configs[1] names bf16); in the 16-bit modes "
                         "operand tensors and conv outputs are STORED in 16 bits, accumulation, norm statistics, gradient sums, "
                         "losses and Adam stay fp32; f32 = the reference-arithmetic parity mode")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-parity-mode", action="store_true", help="skip the extra fp32 (parity mode) measurement at N=1")
    ap.add_argument("--host-input", action="store_true",
                    help="feed pinned HOST batches (H2D copy inside the timed step): the PCIe-inclusive rate, never the headline")
    ap.add_argument("--allow-probes", action="store_true",
                    help="A/B experiments only: run although probe variables / MTBC_LIB are set; the JSON line is then marked "
                         "not_reportable and must never be quoted as a result")
    ap.add_argument("--cpu-stub", action="store_true",
                    help="launcher / contract rehearsal WITHOUT a GPU: gloo backend and a stand-in step (tests/test_bench_cli_cpu.py); "
                         "the line is marked not_reportable")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # bare `python bench.py --gpus N`: one fresh process per GPU through torch.distributed.run, started before this
        # process has made any GPU call (a process that has initialised the GPU must never be replaced or forked from);
        # the children inherit stdout, so rank 0's JSON line is this command's output; exit with the launcher's code
        raise SystemExit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node must equal --gpus")
    if args.cpu_stub:
        return stub_main(args, rank, world)
    from multi_task_breast_cancer_amd import _lib as L
    from multi_task_breast_cancer_amd import switches
    gone = switches.removed()
    if gone:
        raise SystemExit(f"bench.py: {gone} name plan arms that were removed in round 3 (switches.py): unset them")
    bad = switches.result_altering()
    if bad and not args.allow_probes:
        raise SystemExit(f"bench.py refuses to run with {bad} set: those select the probes build of the library or a timing "
                         f"hack, so the numbers would not be the product's (unset them; A/B plan switches are reported, not refused)")
    L.require_gpu()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=dev)

    main_res = run_mode(args, args.dtype, dev, rank, world, dist, rank == 0 and not args.no_roofline)
    metric, tail, cfg_idx = workload_labels(args, world)
    out = {
        "metric": metric, "value": main_res["value"],
        "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main_res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic" + (" (pinned host batches, H2D inside the step)" if args.host_input else ""),
        "config": {"workload": f"{args.arch} seg+cls deep-supervision step (Dice+Focal, alpha=0.5, Adam eps 1e-4), "
                               f"per-GPU batch {args.batch}, 1x{args.size}x{args.size}, "
                               + ("fp32 MFMA (parity mode)" if args.dtype == "f32" else
                                  f"{args.dtype} MFMA operands (3x3 convs, ConvT) and 16-bit stored conv outputs / activations; fp32 accumulation, "
                                  f"gradient sums, statistics, losses, parameters and optimizer") +
                               f", random-init weights ({tail})",
                   "baseline_config_index": cfg_idx,
                   "storage": ({"operands": "f32", "conv_outputs": "f32"} if args.dtype == "f32" else
                               {"mfma_operands": args.dtype, "conv_outputs_before_norm": "f16", "accumulators_gradsums_stats_params_adam": "f32"}),
                   "global_batch": args.batch * world, "parallelism": f"dp{world}"},
        "final_loss": main_res["final_loss"],
        "env": {"mtbc_variables_set": switches.active(), "library": os.path.relpath(L.LIB_PATH, ROOT),
                **({"not_reportable": bad} if bad else {})},
    }
    for k in ("roofline", "roofline_hbm"):
        if k in main_res:
            out[k] = main_res[k]
    if rank == 0 and world == 1 and args.dtype != "f32" and not args.no_parity_mode:
        # the same step in the reference-parity arithmetic (exact fp32 MFMA), same run, same box
        torch.cuda.empty_cache()
        r = run_mode(args, "f32", dev, rank, world, dist, not args.no_roofline)
        out["fp32_parity_mode"] = {"value": r["value"], "unit": "images/sec", "ms_per_step": r["ms_per_step"], "dtype": "f32",
                                   "final_loss": r["final_loss"], **({"roofline": r["roofline"]} if "roofline" in r else {})}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        note("cpu baseline (oracle on host cores)")
        out["cpu_baseline"] = cpu_baseline(args.arch, args.size, 8, 3)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

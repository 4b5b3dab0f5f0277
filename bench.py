#!/usr/bin/env python3
"""Headline benchmark: training images/sec of the multi-task U-Net++ step (1-ch 256x256) on N MI355X.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One "step" = one optimisation step (pack -> fwd -> Dice+Focal -> bwd -> [all-reduce] -> Adam) on a per-GPU batch
of 32 synthetic images already resident in HBM (BASELINE.json configs[1]; weak scaling: global batch = 32*N).
Rank 0 prints ONE JSON line; `roofline` is measured live with HIP events around the dominant kernel (the fp32-MFMA
implicit-GEMM conv3x3, forward + dgrad launches), `cpu_baseline` times the CPU oracle on the host cores.
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
    return {"value": round(batch * steps / dt, 3), "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": f"{steps} steps of {arch} B={batch} {size}x{size} fp32 after 1 warm-up ({dt:.1f} s)"}


def note(msg: str) -> None:
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--arch", default="MTUNetPlusPlus")
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "f16"],
                    help="conv3x3 MFMA operand type (storage/accumulation always fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    from multi_task_breast_cancer_amd import _lib as L
    L.require_gpu()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist_
        dist = dist_
        dist.init_process_group("nccl", device_id=dev)

    from multi_task_breast_cancer_amd import _lib
    from multi_task_breast_cancer_amd.experiment_init import init_multitask_model, init_optimizer
    from multi_task_breast_cancer_amd.miscellany import seed_everything
    from multi_task_breast_cancer_amd.synthetic import synthetic_batch
    from multi_task_breast_cancer_amd.trainer import FusedTrainStep

    seed_everything(1993)                                          # identical initial weights on every rank
    model = init_multitask_model(args.arch, sequences=1, regions=1, n_classes=3, deep_supervision=True).to(dev)
    model.set_compute(args.dtype)
    opt = init_optimizer(model, "Adam", 1e-4)
    step = FusedTrainStep(model, opt, alpha=0.5, inversely_weighted=True, distributed=world > 1)
    batches = [synthetic_batch(args.batch, args.size, args.size, seed=s, device=dev, rank=rank) for s in range(2)]

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    note(f"rank {rank}/{world}: model + {len(batches)} synthetic batches ready; warm-up x{args.warmup}")
    for i in range(args.warmup):
        step(*batches[i % 2])
    sync()
    note("warm-up done; timing")
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
    note(f"timed {args.steps} steps in {dt:.3f} s; loss {losses[0]:.5f}")

    out = {
        "metric": "training images/sec (1-ch 256x256, U-Net++ MT)", "value": round(args.batch * world * args.steps / dt, 2),
        "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.arch} seg+cls deep-supervision step (Dice+Focal, alpha=0.5, Adam eps 1e-4), "
                               f"per-GPU batch {args.batch}, 1x{args.size}x{args.size} {args.dtype} MFMA operands / fp32 storage, random-init weights",
                   "global_batch": args.batch * world, "parallelism": f"dp{world}"},
        "final_loss": round(losses[0], 6),
    }
    if rank == 0 and not args.no_roofline:
        st = step._st
        K = {_lib.OP_CONV3_FWD, _lib.OP_CONV3_DGRAD}
        # only the MFMA launches (packed image present); the Cin=1 first conv runs the direct kernel
        def mfma(p):
            return [i for i in range(p.n) if p.array[i].kind in K and p.array[i].u.conv3.w_packed]
        fl, sec, n = 0.0, 0.0, 0
        for prog in (st.programs["fwd"], st.programs["bwd"]):
            keep = set(mfma(prog))
            # time each selected op individually
            for i in sorted(keep):
                f = conv_flops(prog.array[i])
                ms = 0.0
                for _ in range(3):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record(); prog.run(i, 1); e.record(); e.synchronize()
                    ms += s.elapsed_time(e)
                fl += f; sec += ms / 3e3; n += 1
        wg_fl, wg_sec, wg_n = 0.0, 0.0, 0
        prog = st.programs["bwd"]
        for i in range(prog.n):
            if prog.array[i].kind == _lib.OP_CONV3_WGRAD and prog.array[i].u.conv3.Cin >= 8:
                ms = 0.0
                for _ in range(3):
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record(); prog.run(i, 1); e.record(); e.synchronize()
                    ms += s.elapsed_time(e)
                wg_fl += conv_flops(prog.array[i]); wg_sec += ms / 3e3; wg_n += 1
        ach = fl / sec / 1e12
        # HBM traffic per launch of the same kernel from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
        # tools/pmc_traffic.py); launch-count weighted over the template instances of the kernel
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(tpath):
            ks = json.load(open(tpath))["kernels"]
            sel = [v for k, v in ks.items() if k.startswith("conv3x3_igemm_dma_kernel") or k.startswith("conv3x3_igemm_kernel")]
            if sel:
                w = sum(v["launches_in_trace"] for v in sel)
                traffic = round(sum(v["hbm_bytes_per_launch"] * v["launches_in_trace"] for v in sel) / w)
        peak = PEAK_F32_MFMA_TFLOPS if args.dtype == "f32" else PEAK_16BIT_MFMA_TFLOPS
        kname = "conv3x3_igemm_dma_kernel" if args.dtype == "f32" else "conv3x3_igemm_lp_kernel"
        out["roofline"] = {"bound": "mfma", "kernel": kname + " (fwd + dgrad launches)",
                           "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s",
                           "frac": round(ach / peak, 4), "traffic": traffic if args.dtype == "f32" else None,
                           "launches_per_step": n, "avg_launch_ms": round(sec / n * 1e3, 4),
                           "algorithmic_gflop_per_launch": round(fl / n / 1e9, 3),
                           "wgrad": {"kernel": "conv3x3_wgrad_mfma_kernel + split-K reduce",
                                     "achieved": round(wg_fl / wg_sec / 1e12, 2), "launches_per_step": wg_n,
                                     "avg_launch_ms": round(wg_sec / max(1, wg_n) * 1e3, 4)},
                           "conv3x3_share_of_step": round((sec + wg_sec) / (dt / args.steps), 3)}
    if rank == 0 and not args.no_roofline:
        # HBM-bound kernels named by north_star (norm / upsample): algorithmic bytes / HIP-event time
        def timed(prog, i):
            ms = 0.0
            for _ in range(3):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); prog.run(i, 1); e.record(); e.synchronize()
                ms += s.elapsed_time(e)
            return ms / 3e3
        hb = {"in_fwd": [0.0, 0.0, 0], "in_bwd": [0.0, 0.0, 0], "convT_fwd": [0.0, 0.0, 0]}
        for pname in ("fwd", "bwd"):
            prog = st.programs[pname]
            for i in range(prog.n):
                op = prog.array[i]
                if op.kind == _lib.OP_IN_FWD:
                    a = op.u.inorm; e = a.N * a.C * a.H * a.W * 4
                    h = hb["in_fwd"]; h[0] += 2 * e; h[1] += timed(prog, i); h[2] += 1       # read z, write y
                elif op.kind == _lib.OP_IN_BWD:
                    a = op.u.inorm; e = a.N * a.C * a.H * a.W * 4
                    h = hb["in_bwd"]; h[0] += 3 * e; h[1] += timed(prog, i); h[2] += 1       # read z, dy; write dz
                elif op.kind == _lib.OP_CONVT_FWD:
                    a = op.u.convT; px = a.N * a.H * a.W * 4
                    h = hb["convT_fwd"]; h[0] += px * (a.Cin + a.Cout * a.k * a.k); h[1] += timed(prog, i); h[2] += 1
        out["roofline_hbm"] = {k: {"bound": "hbm", "achieved": round(v[0] / v[1] / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                   "frac": round(v[0] / v[1] / 8e12, 4), "launches_per_step": v[2],
                                   "algorithmic_MB_per_launch": round(v[0] / max(1, v[2]) / 1e6, 1)}
                               for k, v in hb.items() if v[2]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        note("cpu baseline (oracle on host cores)")
        out["cpu_baseline"] = cpu_baseline(args.arch, args.size, 4, 2)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""bench.py's command-line contract on CPU: `python bench.py --gpus N` from a bare interpreter starts its own ranks
(torch.distributed.run children, gloo here through --cpu-stub) and rank 0 prints exactly ONE JSON line whose labels
follow the arguments."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*argv, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], capture_output=True, text=True, env=env, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


@pytest.mark.timeout(600)
def test_bare_interpreter_gpus_2_spawns_its_ranks_and_prints_one_line():
    r, lines = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-stub")
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 64 and out["config"]["parallelism"] == "dp2"
    assert out["env"]["not_reportable"] == ["--cpu-stub"]
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data"):
        assert key in out


def test_labels_follow_the_arguments():
    sys.path.insert(0, ROOT)
    import argparse
    import bench
    ns = argparse.Namespace
    m, tail, idx = bench.workload_labels(ns(arch="MTUNetPlusPlus", size=256, batch=32, dtype="bf16"), 1)
    assert m == "training images/sec (1-ch 256x256, U-Net++ MT)" and idx == 1 and "configs[1]" in tail
    m, tail, idx = bench.workload_labels(ns(arch="MTUNetPlusPlus", size=256, batch=32, dtype="bf16"), 8)
    assert idx == 3 and "configs[3]" in tail
    m, tail, idx = bench.workload_labels(ns(arch="MTnnUNet", size=256, batch=64, dtype="bf16"), 1)
    assert "nnU-Net" in m and idx == 2
    m, tail, idx = bench.workload_labels(ns(arch="MTUNetPlusPlus", size=512, batch=16, dtype="f16"), 1)
    assert "512x512" in m and idx is None and "configs[4]" in tail and "per-GPU shape" in tail
    m, tail, idx = bench.workload_labels(ns(arch="MTUNetPlusPlus", size=512, batch=16, dtype="f16"), 8)
    assert idx == 4
    m, tail, idx = bench.workload_labels(ns(arch="MTUNetPlusPlus", size=128, batch=8, dtype="f32"), 1)
    assert idx is None and "not a BASELINE" in tail


def test_mismatched_world_size_is_refused():
    r, lines = _run("--gpus", "2", "--cpu-stub", env_extra={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and not lines and "nproc-per-node" in r.stderr


@pytest.mark.timeout(900)
def test_gpus_8_is_labelled_as_baseline_configs_3():
    """The driver's scaling run: `python bench.py --gpus 8` (bare interpreter -> 8 ranks of torch.distributed.run; gloo + the stand-in step
    here) must label its line BASELINE.json configs[3] -- U-Net++ global batch 256 = 32 per GPU x 8 -- and report the whole-job batch."""
    r, lines = _run("--gpus", "8", "--steps", "2", "--warmup", "1", "--cpu-stub", env_extra={"OMP_NUM_THREADS": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["baseline_config_index"] == 3
    assert out["config"]["global_batch"] == 256 and out["config"]["parallelism"] == "dp8" and out["scaling"] == "weak"
    assert "configs[3]" in out["config"]["workload"]

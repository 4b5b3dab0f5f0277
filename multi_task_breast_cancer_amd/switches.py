"""Every environment variable the package reads, in one place, so that a benchmark or test can report exactly which
ones were set (`active()`) and refuse to run with one that changes results (`result_altering()`).

Plan switches select between two step programs that are BOTH run against the oracle / emulation by a `-m gpu` test (third tuple
field; checked by tests/test_host_cpu.py::test_every_plan_switch_names_its_gpu_test); they change which kernels a step program is
built from, never what it computes beyond the documented 16-bit operand rounding.  The shipped library itself (`libmtbc_hip.so`) reads no environment variable at all: the
kernel-level timing probes (MTBC_DBG load / store skipping, tile-shape overrides) exist only in the separately built
`libmtbc_hip_probes.so` (`make -C csrc probes`), which has to be selected explicitly through MTBC_LIB.
"""
from __future__ import annotations

import os
from typing import Dict, List

# name -> (default, what the non-default arm does, the -m gpu test that runs the non-default arm against the oracle / emulation)
PLAN_SWITCHES: Dict[str, tuple] = {
    "MTBC_NO_COOP": ("0", "InstanceNorm by one-plane workgroups (fp32 conv outputs) + pack instead of the channel-group / cooperative kernels: "
                          "the fallback when their teams cannot be co-resident (the cooperative error message names it)",
                     "tests/test_model_gpu.py::test_16bit_mfma_modes_match_their_emulation[...-no_coop-40]"),
    "MTBC_NO_GATHER": ("0", "every 3x3 conv back-propagates into all its inputs (fan-in by read-modify-write) instead of one gathered launch per dense-skip tensor",
                       "tests/test_model_gpu.py::test_16bit_mfma_modes_match_their_emulation[...-no_gather]"),
    "MTBC_NO_Z16": ("0", "16-bit modes keep the conv outputs z as fp32 planes (InstanceNorm reads 4 bytes per element in both directions): the storage arm of the quality sweeps",
                    "tests/test_model_gpu.py::test_16bit_mfma_modes_match_their_emulation[...-no_z16]"),
    "MTBC_Z_BF16": ("0", "bf16 mode stores the conv outputs as bf16 instead of fp16 (same bytes, 8 instead of 11 significant bits)",
                    "tests/test_model_gpu.py::test_16bit_mfma_modes_match_their_emulation[...-z_bf16]"),
    "MTBC_DA16": ("0", "the gradient a conv-cell activation gets from its 3x3 consumers is written by ONE gathered launch as a 16-bit channel-blocked tensor (fp32 sum in the MFMA "
                       "accumulators, one RNE) instead of fp32 planar fan-in: -0.36 ms per step.  Default in round 3 on the easy task's evidence (-0.010 +- 0.025 pt); OFF again since "
                       "round 4: on the task that can fail it ends at the same Dice (-0.012 +- 0.042 pt against fp32, 10 paired seeds) but reaches the plateau LATER -- mid-run "
                       "-5.6 +- 2.9 pt against fp32 where the fp32-gradient plan is at -0.5 +- 1.4 and tracks the fp32 runs seed by seed (profiles/r04_quality_hard.md)",
                  "tests/test_model_gpu.py::test_16bit_mfma_modes_match_their_emulation[...-da16]"),
    "MTBC_NOFUSE_HEADS": ("0", "MTnnUNet deep-supervision heads as ConvT + 1x1 (the reference's two layers) instead of one combined ConvT",
                          "tests/test_model_gpu.py::test_mtnnunet_two_layer_heads_match_the_fused_heads_and_the_oracle"),
    "MTBC_GRAPH": ("0", "trainer.FusedTrainStep (single process) captures each compiled step -- weight images, forward, losses, backward, Adam -- into ONE hipGraph at its third call "
                        "and replays it from then on: the host issues a step in 0.1 ms instead of 8.4 (profiles/r04_graph_replay.txt); the GPU time is the same, the results are the same bits "
                        "(Adam's per-step scalars travel through device memory, mtbc_adam_args.dynamic)",
                   "tests/test_model_gpu.py::test_graph_replayed_steps_are_the_eager_steps"),
    "MTBC_COOP_RESERVE_CUS": ("64", "CUs kept out of the cooperative InstanceNorm grids under data parallel",
                              "tests/test_coop_safety_gpu.py::test_cooperative_step_beside_a_cu_hogging_kernel"),
}
# Removed in round 3 (arms that were measured slower / no gain and had no diagnostic use; the measurements stay in DESIGN.md "What was tried"):
# MTBC_NO_C8, NO_CT_LP, COOP_MIN_FWD / _BWD, NO_P16, FANIN, NO_C8_SMALL_OPS, NO_X16, NO_G16, NO_EPI_STATS, EPI_BSTATS, NO_R1, NO_POOLFOLD,
# SPLIT_FANIN, NO_STEM16, NO_DEFER_DPARAM, DPARAM_BATCH, NO_POOLFWD_FOLD, BWD_OVERLAP, BWD_OVERLAP_MAX_HW, NO_DA16 (round 4: the opt-in is MTBC_DA16).  Setting one of them
# is an error wherever a step program is planned (`refuse_removed()`, called by engine.py at import), not a silent no-op.
REMOVED = ("MTBC_NO_C8", "MTBC_NO_CT_LP", "MTBC_COOP_MIN_FWD", "MTBC_COOP_MIN_BWD", "MTBC_NO_P16", "MTBC_FANIN", "MTBC_NO_C8_SMALL_OPS", "MTBC_NO_X16",
           "MTBC_NO_G16", "MTBC_NO_EPI_STATS", "MTBC_EPI_BSTATS", "MTBC_NO_R1", "MTBC_NO_POOLFOLD", "MTBC_SPLIT_FANIN", "MTBC_NO_STEM16",
           "MTBC_NO_DEFER_DPARAM", "MTBC_DPARAM_BATCH", "MTBC_NO_POOLFWD_FOLD", "MTBC_BWD_OVERLAP", "MTBC_BWD_OVERLAP_MAX_HW",
           # round 3 made the 16-bit gathered activation gradients the default (switch MTBC_NO_DA16); round 4 took that back on the hard task's evidence:
           # the opt-in is MTBC_DA16 again and the round-3 name is refused
           "MTBC_NO_DA16")
# variables that only the probes build of the library (or removed timing hacks) ever honoured: results are wrong or
# timings are not the product's when one of them takes effect
RESULT_ALTERING = ("MTBC_DBG", "MTBC_NOACC", "MTBC_LOWP", "MTBC_LP_MT", "MTBC_RING", "MTBC_NODMA", "MTBC_C8_BLOCKS_PER_CU",
                   "MTBC_WGRAD_LP1", "MTBC_CT_WG_TASKS", "MTBC_IN_BWD_STREAM", "MTBC_CONVT_GENERIC", "MTBC_C8_NW", "MTBC_C8_RING",
                   "MTBC_CT_DEPTH", "MTBC_CT_WG_CT", "MTBC_CT_WG_XCD", "MTBC_CT_DGRAD_DIRECT", "MTBC_WGRAD_C8W", "MTBC_C8W_BPC", "MTBC_C8W_DEPTH", "MTBC_C8W_HACK",
                   "MTBC_WGRAD_PACK24", "MTBC_WGRAD_C8I", "MTBC_WG_NS8", "MTBC_WG_TS", "MTBC_C8_TS", "MTBC_RING_TS", "MTBC_INB_TS")


def get(name: str) -> str:
    return os.environ.get(name, PLAN_SWITCHES[name][0])


def flag(name: str) -> bool:
    v = get(name)
    return v not in ("", "0")


def active() -> Dict[str, str]:
    """Every MTBC_* variable present in the environment (whether or not anything reads it)."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("MTBC_")}


def removed() -> List[str]:
    """Set variables that named a plan arm which no longer exists: a run that sets one is not what its author thinks it is."""
    return [k for k in REMOVED if k in os.environ]


def refuse_removed() -> None:
    """Called where step programs are planned (engine.py, at import): a removed switch in the environment raises instead of being ignored."""
    bad = removed()
    if bad:
        raise RuntimeError(f"removed plan switch(es) set: {bad} -- these arms no longer exist (switches.py REMOVED; the 16-bit gathered gradients are the opt-in MTBC_DA16=1 since round 4, "
                           "MTBC_NO_DA16 is gone); unset them")


def result_altering() -> List[str]:
    """Set variables under which a measurement is not the product's: probe-build switches, or MTBC_LIB itself."""
    bad = [k for k in RESULT_ALTERING if k in os.environ]
    if "MTBC_LIB" in os.environ:
        bad.append("MTBC_LIB")
    return bad

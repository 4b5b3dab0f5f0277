"""Every environment variable the package reads, in one place, so that a benchmark or test can report exactly which
ones were set (`active()`) and refuse to run with one that changes results (`result_altering()`).

Plan switches select between two paths that are BOTH parity-tested (A/B arms of measured design decisions, DESIGN.md
"What was tried"); they change which kernels a step program is built from, never what it computes beyond the documented
16-bit operand rounding.  The shipped library itself (`libmtbc_hip.so`) reads no environment variable at all: the
kernel-level timing probes (MTBC_DBG load / store skipping, tile-shape overrides) exist only in the separately built
`libmtbc_hip_probes.so` (`make -C csrc probes`), which has to be selected explicitly through MTBC_LIB.
"""
from __future__ import annotations

import os
from typing import Dict, List

# name -> (default, what the non-default arm does)
PLAN_SWITCHES: Dict[str, tuple] = {
    "MTBC_NO_C8": ("0", "16-bit modes stage fp32 planar operands inside every consumer instead of reading channel-blocked tensors"),
    "MTBC_NO_CT_LP": ("0", "k=2 ConvT forward/backward keep fp32 MFMA operands in the 16-bit modes"),
    "MTBC_NO_COOP": ("0", "InstanceNorm by one-plane workgroups + pack instead of the cooperative kernels"),
    "MTBC_COOP_MIN_FWD": ("16384", "smallest plane (pixels) handed to the cooperative InstanceNorm forward"),
    "MTBC_COOP_MIN_BWD": ("65536", "smallest plane (pixels) handed to the cooperative InstanceNorm backward"),
    "MTBC_NO_GATHER": ("0", "every 3x3 conv back-propagates into all its inputs (fan-in by read-modify-write)"),
    "MTBC_NO_P16": ("0", "InstanceNorm keeps writing fp32 y / dz planes in the 16-bit modes"),
    "MTBC_FANIN": ("0", "private gradient fan-in buffers summed by InstanceNorm backward"),
    "MTBC_NOFUSE_HEADS": ("0", "MTnnUNet deep-supervision heads as ConvT + 1x1 (the reference's two layers) instead of one combined ConvT"),
    "MTBC_NO_C8_SMALL_OPS": ("0", "max-pool and the 1x1 heads keep reading fp32 planes in the 16-bit modes (InstanceNorm then writes them too)"),
    "MTBC_NO_X16": ("0", "the input of a k = 2 ConvT stays available as fp32 planes for its weight gradient (else 16-bit planes written by the streaming InstanceNorm pass)"),
    "MTBC_NO_G16": ("0", "the gradient of an up-sampled (ConvT) tensor stays fp32 between the 3x3 conv's dgrad and the ConvT backward"),
    "MTBC_NO_Z16": ("0", "16-bit modes keep the conv outputs z as fp32 planes (InstanceNorm reads 4 bytes per element in both directions)"),
    "MTBC_DA16": ("0", "the gradient a conv-cell activation gets from ALL its 3x3 consumers is one gathered launch writing a 16-bit channel-blocked tensor instead of fp32 planar fan-in (no faster on this workload -- the norm backward that reads it is latency-bound -- and about 1 pt of held-out Dice in the 3000-step sweep: off)"),
    "MTBC_Z_BF16": ("0", "bf16 mode stores the conv outputs as bf16 instead of fp16 (same bytes, 8 instead of 11 significant bits)"),
    "MTBC_NO_EPI_STATS": ("0", "InstanceNorm forward reduces the stored conv output itself (channel-group kernels) instead of taking the statistics from the conv epilogue"),
    "MTBC_EPI_BSTATS": ("0", "the gathered dgrad's epilogue adds the other readers' partial gradient, reads the cell's z and leaves the two sums of the InstanceNorm backward, which becomes one streaming pass (measured slower: the epilogue's VALU work costs what the norm saves)"),
    "MTBC_NO_R1": ("0", "a one-output 1x1 head writes its input gradient as C fp32 planes (mtbc_conv1x1_dgrad) instead of handing (dy, w) to the InstanceNorm backward of the tensor it reads"),
    "MTBC_NO_POOLFOLD": ("0", "the 2x2 max-pool backward is its own launch writing a 4x larger fp32 tensor instead of being routed inside the InstanceNorm backward of the pooled tensor"),
    "MTBC_SPLIT_FANIN": ("0", "the gathered dgrad writes a buffer of its own (added by the InstanceNorm backward while loading) instead of read-modify-writing a gradient something else wrote first (measured 14.75 -> 14.80 ms: off)"),
    "MTBC_NO_STEM16": ("0", "the stem cell (Cin = 1) keeps fp32 conv outputs / fp32 dz (cooperative InstanceNorm forward, fp32 weight-gradient kernel)"),
    "MTBC_NO_DEFER_DPARAM": ("0", "every InstanceNorm backward reduces its parameter-gradient partials right away (one 5 us launch per cell) instead of in batches"),
    "MTBC_DPARAM_BATCH": ("12", "cells per batched InstanceNorm parameter-gradient reduction"),
    "MTBC_NO_POOLFWD_FOLD": ("0", "the 2x2 max-pool forward is its own launch reading the activation back instead of being written by the streaming InstanceNorm pass"),
    "MTBC_BWD_OVERLAP": ("0", "weight gradient and input gradient of a layer run side by side on two streams (measured slower: DESIGN.md)"),
    "MTBC_BWD_OVERLAP_MAX_HW": ("1024", "with MTBC_BWD_OVERLAP=1: largest map (pixels) whose weight / input gradient launches are overlapped"),
    "MTBC_COOP_RESERVE_CUS": ("64", "CUs kept out of the cooperative InstanceNorm grids under data parallel"),
}
# variables that only the probes build of the library (or removed timing hacks) ever honoured: results are wrong or
# timings are not the product's when one of them takes effect
RESULT_ALTERING = ("MTBC_DBG", "MTBC_NOACC", "MTBC_LOWP", "MTBC_LP_MT", "MTBC_RING", "MTBC_NODMA", "MTBC_C8_BLOCKS_PER_CU",
                   "MTBC_WGRAD_LP1", "MTBC_CT_WG_TASKS", "MTBC_IN_BWD_STREAM", "MTBC_CONVT_GENERIC", "MTBC_C8_NW", "MTBC_C8_RING",
                   "MTBC_CT_DEPTH", "MTBC_CT_WG_CT", "MTBC_CT_DGRAD_DIRECT")


def get(name: str) -> str:
    return os.environ.get(name, PLAN_SWITCHES[name][0])


def flag(name: str) -> bool:
    v = get(name)
    return v not in ("", "0")


def active() -> Dict[str, str]:
    """Every MTBC_* variable present in the environment (whether or not anything reads it)."""
    return {k: v for k, v in sorted(os.environ.items()) if k.startswith("MTBC_")}


def result_altering() -> List[str]:
    """Set variables under which a measurement is not the product's: probe-build switches, or MTBC_LIB itself."""
    bad = [k for k in RESULT_ALTERING if k in os.environ]
    if "MTBC_LIB" in os.environ:
        bad.append("MTBC_LIB")
    return bad

#!/usr/bin/env python3
"""Scan every kernel of a HIP source for the serialisation patterns of DESIGN.md ("Compiler-induced serialisation"):
  - self-loops with <= 4 loads and an `s_waitcnt vmcnt(0)` (a memory round trip per round),
  - loads followed within a few instructions by `vmcnt(0)` (count per kernel),
  - scratch (spill) traffic, ds_bpermute / ds_swizzle counts (shuffle butterflies).
usage: tools/isa_scan.py multi_task_breast_cancer_amd/csrc/norm_coop.hip [name filter]"""
import re, subprocess, sys, tempfile, os

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(root, "include"), "-S", "--cuda-device-only",
                    src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
rows = []
for f in re.split(r"\n(?=_Z[\w]+:)", s):
    name = f.split(":")[0]
    if not name.startswith("_Z"):
        continue
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
    if flt not in dn:
        continue
    loops = 0
    for b in re.split(r"\n(?=\.LBB\d+_\d+:)", f):
        m = re.match(r"(\.LBB\d+_\d+):", b)
        if m and re.search(r"s_cbranch_\w+ " + re.escape(m.group(1)) + r"\b", b):
            nl = len(re.findall(r"\n\s*(global_load|buffer_load)", b))
            if 0 < nl <= 4 and "vmcnt(0)" in b and "v_mfma" not in b:
                loops += 1
    toks = []
    for line in f.split("\n"):
        t = line.strip()
        if t.startswith(("global_load", "buffer_load", "scratch_load")): toks.append("L")
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t: toks.append("W0")
        elif t and not t.startswith((";", ".")): toks.append("x")
    chains = 0
    for i, t in enumerate(toks):
        if t == "L":
            for j in range(i + 1, min(i + 9, len(toks))):
                if toks[j] == "L": break
                if toks[j] == "W0": chains += 1; break
    scr = len(re.findall(r"scratch_(load|store)", f))
    bp = len(re.findall(r"ds_bpermute|ds_swizzle", f))
    if loops or chains >= 4 or scr or bp >= 12:
        rows.append((chains, loops, scr, bp, dn))
print(f"{'load->vmcnt(0)':>14s} {'1-trip loops':>12s} {'scratch':>8s} {'bpermute':>9s}  kernel")
for chains, loops, scr, bp, dn in sorted(rows, reverse=True):
    print(f"{chains:14d} {loops:12d} {scr:8d} {bp:9d}  {dn[:110]}")

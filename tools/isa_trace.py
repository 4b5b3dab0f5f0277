#!/usr/bin/env python3
"""Token trace of a kernel's gfx950 ISA: one line per basic block, memory / LDS / MFMA instructions and waits in program order.
  L global/buffer load (incl. LDS-DMA)   S store   A atomic   r ds_read / ds_bpermute   W ds_write   M v_mfma
  wvN s_waitcnt vmcnt(N)   wlN lgkmcnt(N)   BAR s_barrier   zz s_sleep   [cbranch_x LABEL] branches
usage: tools/isa_trace.py multi_task_breast_cancer_amd/csrc/convt2.hip convT2_dgrad_lds_kernelILi1    (substring of the MANGLED name)
What to look for ("Compiler-induced serialisation"): `L wv0` repeated inside a loop (one memory round trip per round),
`wv0` right after a group of loads in a software-pipelined loop (no prefetch), scratch reloads in front of loads, `r wl0` chains."""
import re, subprocess, sys, tempfile, os

src, pat = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.run(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-I", os.path.join(root, "include"), "-S", "--cuda-device-only",
                    src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    s = open(out).read()
for f in re.split(r"\n(?=_Z[\w]+:)", s):
    name = f.split(":")[0]
    if not name.startswith("_Z") or pat not in name:
        continue
    toks = []
    for line in f.split("\n"):
        t = line.strip()
        if t.startswith(("global_load", "buffer_load")): toks.append("L")
        elif t.startswith("scratch_load"): toks.append("Lscr")
        elif t.startswith(("global_store", "buffer_store")): toks.append("S")
        elif t.startswith("scratch_store"): toks.append("Sscr")
        elif t.startswith(("global_atomic", "buffer_atomic")): toks.append("A")
        elif t.startswith(("ds_read", "ds_bpermute", "ds_swizzle")): toks.append("r")
        elif t.startswith("ds_write"): toks.append("W")
        elif "v_mfma" in t: toks.append("M")
        elif t.startswith("s_waitcnt"):
            m, l = re.search(r"vmcnt\((\d+)\)", t), re.search(r"lgkmcnt\((\d+)\)", t)
            toks.append("w" + (("v" + m.group(1)) if m else "") + (("l" + l.group(1)) if l else ""))
        elif t.startswith("s_barrier"): toks.append("BAR")
        elif t.startswith("s_sleep"): toks.append("zz")
        elif re.match(r"\.LBB\d+_\d+:", t): toks.append("\n" + t.split(";")[0].strip())
        elif t.startswith(("s_cbranch", "s_branch")): toks.append("[" + t.split()[0][2:] + " " + t.split()[-1] + "]")
    print(subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip())
    print(" ".join(toks))
    print()

#!/usr/bin/env python3
"""Per-kernel register / LDS / spill / SCRATCH table of a HIP source (a non-zero scratch size costs per LAUNCH, whatever put it there: DESIGN.md section 3, round 4) (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kres.py multi_task_breast_cancer_amd/csrc/conv3x3.hip [name filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["hipcc", "-O3", "-fPIC", "--offload-arch=gfx950", "-std=c++17", "-I", "include", "-I", "../../include", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
rec = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)([A-Za-z \[\]/]+): (.*?) \[-Rpass", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        rec = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
        rows.append(rec)
    elif rec is not None:
        rec[k] = v
for r in rows:
    if flt in r["name"]:
        n = re.sub(r"\(anonymous namespace\)::", "", r["name"]).split("(")[0].replace("void ", "")
        print(f"{n[:70]:70s} VGPR {r.get('VGPRs','?'):>4s} AGPR {r.get('AGPRs','?'):>3s} spillV {r.get('VGPRs Spill','?'):>3s} spillS {r.get('SGPRs Spill','?'):>3s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?')} LDS {r.get('LDS Size [bytes/block]','?')}")

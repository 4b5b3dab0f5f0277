#!/usr/bin/env python3
"""Kernel trace of rocprofv3 (--kernel-trace --output-format csv) in launch order, folded into runs of identical
(kernel, grid) launches: name, grid x block, LDS, launches, average / min duration.  Lets one profiled probe script
(tools/wgrad_probe.py: 13 launches per shape) be read shape by shape.
usage: tools/trace_runs.py DIR_OR_CSV [name filter]"""
import csv, glob, os, re, sys
path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
runs = []
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).replace("void ", "").split("(")[0]
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")), r.get("LDS_Block_Size", ""))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if runs and runs[-1][0] == key:
        runs[-1][1].append(d)
    else:
        runs.append([key, [d]])
for key, ds in runs:
    if flt in key[0]:
        print(f"{key[0][:60]:60s} grid {key[1]:>8s}x{key[2]:>3s} wg {key[3]:>4s} lds {key[4]:>6s}  n={len(ds):3d}  avg {sum(ds)/len(ds):8.1f} us  min {min(ds):8.1f}")

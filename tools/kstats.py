#!/usr/bin/env python3
"""rocprofv3 --stats kernel_stats.csv as a per-step table.  usage: tools/kstats.py CSV [steps_in_trace=7] [top=30]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per step {tot / steps / 1e6:.3f} ms ({steps} steps in the trace)")
for r in rows[:top]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:66]
    print(f"{n:66s} {int(r['Calls']) / steps:6.1f}/step  avg {float(r['AverageNs']) / 1e3:7.1f} us  {float(r['TotalDurationNs']) / steps / 1e6:6.3f} ms/step")

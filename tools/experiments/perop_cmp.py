#!/usr/bin/env python3
"""Side-by-side per-op times of several tools/per_op.py logs (one column per log), filtered by op-kind substring.
usage: tools/experiments/perop_cmp.py CONVT_WGRAD log1 log2 ..."""
import re, sys
kind, logs = sys.argv[1], sys.argv[2:]
rows = {}
for li, path in enumerate(logs):
    for line in open(path):
        m = re.match(r"\s*(\w+)\s+#\s*(\d+)\s+(OP_\w+)\s+(.*?)\s+([\d.]+) ms", line)
        if m and kind in m.group(3):
            rows.setdefault((m.group(1), int(m.group(2)), m.group(3), m.group(4).strip()), {})[li] = float(m.group(5))
tot = [0.0] * len(logs)
for key in sorted(rows):
    vals = rows[key]
    print(f"{key[0]:4s} #{key[1]:3d} {key[2]:16s} {key[3][:40]:40s} " + " ".join(f"{vals.get(i, float('nan')):7.3f}" for i in range(len(logs))))
    for i in range(len(logs)):
        tot[i] += vals.get(i, 0.0)
print(" " * 68 + " ".join(f"{t:7.3f}" for t in tot))

#!/usr/bin/env python3
"""Mean duration of every kernel and mean idle gap in front of it, from a rocprofv3 --kernel-trace CSV (launch-bound
sizes: where does a force pass of a few tens of microseconds go?).   python tools/trace_gaps.py DIR_OR_CSV [skip_first]"""
import csv
import glob
import os
import sys

path = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 200
files = [path] if path.endswith(".csv") else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
rows = rows[skip:]
stats = {}
for k in range(1, len(rows)):
    s, e, name = rows[k]
    short = name.replace("(anonymous namespace)::", "").replace("void nbody::", "").split("(")[0][:70]
    d = stats.setdefault(short, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += (e - s) * 1e-3
    d[2] += max(0, s - rows[k - 1][1]) * 1e-3
print(f"{'kernel':<72} {'calls':>6} {'mean us':>9} {'gap before us':>14}")
for name, (n, dur, gap) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
    print(f"{name:<72} {n:>6} {dur / n:>9.2f} {gap / n:>14.2f}")
if rows:
    print(f"# wall {(rows[-1][1] - rows[0][0]) * 1e-3:.1f} us for {len(rows)} launches; sum of kernel time {sum(e - s for s, e, _ in rows) * 1e-3:.1f} us")

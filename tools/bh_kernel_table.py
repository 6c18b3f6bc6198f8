#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --kernel-trace --stats run of tools/bh_ticks.py, one line per kernel, sorted by share:
    python3 tools/bh_kernel_table.py <kernel_stats.csv> [frames]
With `frames` the table also says how often a kernel runs per frame."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
frames = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
total = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    name = re.sub(r"\(anonymous namespace\)::|nbody::|void ", "", r["Name"])
    name = re.sub(r"\(.*", "", name)
    per = f"  x{int(r['Calls']) / frames:5.2f}/frame" if frames else ""
    print(f"{name[:44]:44s} calls {int(r['Calls']):6d}{per}  avg {float(r['AverageNs']) / 1e3:9.1f} us  {100 * float(r['TotalDurationNs']) / total:5.1f} %")
if frames:
    print(f"sum of kernel time per frame: {total / frames / 1e3:.1f} us")

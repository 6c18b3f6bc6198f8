#!/usr/bin/env python3
"""Per-launch means of the PMC passes tools/evidence_r03.sh collected over the Barnes-Hut frames (one counter set per run,
--kernel-trace only), by kernel.   python3 tools/pmc_bh_summary.py OUTDIR N"""
import csv, glob, os, re, sys
out, n = sys.argv[1], sys.argv[2]
acc, dur = {}, {}
for d in sorted(glob.glob(os.path.join(out, f"bh_pmc_n{n}_*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"(bh_\w+|update_kernel|bounds_kernel)", r["Kernel_Name"])
            if not m:
                continue
            k = (m.group(1), r["Counter_Name"])
            a = acc.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
            t = dur.setdefault(m.group(1), [0, 0.0]); t[0] += 1; t[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
print(f"# rocprofv3 --pmc passes over tools/bh_ticks.py {n} (theta = 1 frames), one counter set per run, MI355X; per-launch means by kernel")
for kern in sorted({k for k, _ in acc}):
    print(f"{kern}: mean duration under the counters {dur[kern][1] / dur[kern][0]:.2f} us")
    vals = {c: v[1] / v[0] for (k, c), v in acc.items() if k == kern}
    for c in sorted(vals):
        print(f"  {c:28s} {vals[c]:.6g}")
    if "FETCH_SIZE" in vals or "WRITE_SIZE" in vals:
        fetch = 2 * vals.get("FETCH_SIZE", 0.0) * 1024; wr = vals.get("WRITE_SIZE", 0.0) * 1024
        print(f"  HBM per launch: FETCH_SIZE x2 (gfx950 correction) {fetch / 1e6:.3f} MB + WRITE_SIZE {wr / 1e6:.3f} MB")
    if "SQ_WAVE_CYCLES" in vals and vals.get("SQ_WAVES"):
        w = vals["SQ_WAVES"]
        print(f"  per wave: {vals.get('SQ_INSTS_VALU', 0) / w:.0f} VALU, {vals.get('SQ_INSTS_SALU', 0) / w:.0f} SALU, {vals.get('SQ_INSTS_LDS', 0) / w:.0f} LDS instructions; "
              f"{vals['SQ_WAVE_CYCLES'] / w:.0f} wave-cycles, of them waiting {vals.get('SQ_WAIT_ANY', 0) / vals['SQ_WAVE_CYCLES'] * 100:.0f} % (any), "
              f"{vals.get('SQ_WAIT_INST_ANY', 0) / vals['SQ_WAVE_CYCLES'] * 100:.0f} % (for an instruction slot)")

#!/usr/bin/env python3
"""Per-launch means of rocprofv3 --pmc passes (one counter set per run, --kernel-trace only) for the kernels whose name
contains a given substring, with the derived per-wave figures.   python3 tools/pmc_kernel_means.py OUTDIR SUBSTRING "COMMAND"
OUTDIR holds one directory per pass (pmc1, pmc2, ...)."""
import csv, glob, os, sys
out, sub = sys.argv[1], sys.argv[2]
cmd = sys.argv[3] if len(sys.argv) > 3 else ""
acc, dur = {}, [0, 0.0]
for d in sorted(glob.glob(os.path.join(out, "pmc[0-9]*"))):
    if not os.path.isdir(d):
        continue
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub not in r["Kernel_Name"]:
                continue
            a = acc.setdefault(r["Counter_Name"], {}); a[(d, r["Dispatch_Id"])] = a.get((d, r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub in r["Kernel_Name"]:
                dur[0] += 1; dur[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
vals = {c: sum(v.values()) / len(v) for c, v in acc.items()}
print(f"# rocprofv3 --pmc passes (one counter set per run, --kernel-trace only), MI355X; command: {cmd}")
print(f"# kernel: *{sub}*; per-launch means over {dur[0]} launches, mean duration under the counters {dur[1] / max(dur[0], 1):.2f} us")
for c in sorted(vals):
    print(f"  {c:28s} {vals[c]:.6g}")
w = vals.get("SQ_WAVES", 0.0)
if w:
    g = lambda c: vals.get(c, 0.0) / w
    print(f"per wave: {g('SQ_INSTS_VALU'):.0f} VALU, {g('SQ_INSTS_SALU'):.0f} SALU, {g('SQ_INSTS_LDS'):.0f} LDS, {g('SQ_INSTS_VMEM_RD'):.0f} VMEM-read instructions; "
          f"{g('SQ_WAVE_CYCLES'):.0f} wave-cycles, of them waiting {100 * vals.get('SQ_WAIT_ANY', 0) / max(vals.get('SQ_WAVE_CYCLES', 1), 1):.0f} % (any), "
          f"{100 * vals.get('SQ_WAIT_INST_ANY', 0) / max(vals.get('SQ_WAVE_CYCLES', 1), 1):.0f} % (for an instruction slot); LDS bank-conflict cycles {vals.get('SQ_LDS_BANK_CONFLICT', 0):.0f}")
if "FETCH_SIZE" in vals or "WRITE_SIZE" in vals:
    print(f"HBM per launch: FETCH_SIZE x2 (gfx950 correction) {2 * vals.get('FETCH_SIZE', 0) * 1024 / 1e6:.3f} MB + WRITE_SIZE {vals.get('WRITE_SIZE', 0) * 1024 / 1e6:.3f} MB")
if "GRBM_GUI_ACTIVE" in vals and dur[0]:
    print(f"GRBM_GUI_ACTIVE per launch {vals['GRBM_GUI_ACTIVE']:.0f} (all XCDs)")

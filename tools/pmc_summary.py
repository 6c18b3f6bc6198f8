#!/usr/bin/env python3
"""Summarise the PMC passes of tools/profile_kernel.sh: per-launch counter means of the dominant kernel (the launches
that ran: the twin launch that returns at once is dropped by duration) and the derived figures DESIGN.md quotes.

    python tools/pmc_summary.py OUTDIR KERNEL_SUBSTRING N BODIES_PER_LANE "COMMAND"
"""
import csv
import glob
import os
import sys

out, kernel = sys.argv[1], sys.argv[2]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1 << 20
BPL = int(sys.argv[4]) if len(sys.argv) > 4 else 16        # bodies per lane of the profiled launch
CMD = sys.argv[5] if len(sys.argv) > 5 else ""
vals, dur = {}, []
min_ms = 0.02 if N < 262144 else 1.0                       # the launch that did the work (its twin returns in ~5 us)
for d in sorted(glob.glob(os.path.join(out, "pmc[0-9]"))):
    tag = os.path.basename(d)
    trace = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                trace[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    long_ids = {k for k, v in trace.items() if v > min_ms}
    if tag == "pmc3":
        dur = sorted(trace[k] for k in long_ids)
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and (not long_ids or r["Dispatch_Id"] in long_ids):
                acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
                acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for name, per in acc.items():
        vals[(tag, name)] = sum(per.values()) / len(per)
print(f"# rocprofv3 --pmc passes (one counter set per run, --kernel-trace only), MI355X, command: {CMD}")
print(f"# kernel: {kernel} (the launch that ran; its guarded twin returns at its first instruction); per-launch means")
for (tag, name), v in sorted(vals.items()):
    print(f"{tag},{name},{v:.6g}")
if dur:
    print("kernel_duration_ms_in_pmc3_run," + str([round(x, 4) for x in dur[:12]]) + (" ..." if len(dur) > 12 else ""))
g = lambda n: next((v for (t, k), v in vals.items() if k == n), None)
steps = N * (N - 1) / 2 / (64.0 * BPL)   # wave-steps: 64 lanes x BPL bodies meet one j
if g("SQ_INSTS_VALU"):
    print(f"# derived: wave-steps = N(N-1)/2/{64 * BPL} = {steps:.4g}; SQ_INSTS_VALU per wave-step = {g('SQ_INSTS_VALU') / steps:.1f}"
          f" (fp32 symmetric strips, general form: {8 * BPL} packed + {BPL} v_rsq_f32 + 6 v_mov_b32_dpp = {9 * BPL + 6};"
          f" equal-mass form: {7 * BPL} packed + {BPL} + 6 = {8 * BPL + 6})")
if g("SQ_WAVE_CYCLES") and g("SQ_BUSY_CYCLES"):
    print(f"# SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = {g('SQ_WAIT_INST_ANY') / g('SQ_WAVE_CYCLES'):.3f} (waves queueing for the VALU), "
          f"SQ_WAIT_ANY / SQ_WAVE_CYCLES = {g('SQ_WAIT_ANY') / g('SQ_WAVE_CYCLES'):.3f} (parked on memory / barriers), "
          f"SQ_ACTIVE_INST_VALU x4 / SQ_BUSY_CYCLES = {4 * g('SQ_ACTIVE_INST_VALU') / g('SQ_BUSY_CYCLES'):.3f} (VALU utilisation of the busy SIMD time)")
if g("GRBM_GUI_ACTIVE") and dur:
    t = sum(dur) / len(dur) * 1e-3
    clk = g("GRBM_GUI_ACTIVE") / 8 / t
    print(f"# clock = GRBM_GUI_ACTIVE/8/duration = {clk / 1e9:.3f} GHz (reads high on launches under ~0.3 ms); SIMD cycles per wave-step = "
          f"{1024 * t * clk / steps:.1f} ({2 * BPL} interactions per lane) = {1024 * t * clk / steps / (2 * BPL):.2f} per interaction-lane")
if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None and dur:
    t = sum(dur) / len(dur) * 1e-3
    rd, wr = g("FETCH_SIZE") * 1024 * 2, g("WRITE_SIZE") * 1024     # KB; gfx950: FETCH_SIZE counts 128-B requests at 64 B
    print(f"# HBM: FETCH_SIZE {g('FETCH_SIZE'):.6g} KB x2 (gfx950 correction) = {rd / 1e9:.3f} GB; WRITE_SIZE {g('WRITE_SIZE'):.6g} KB = {wr / 1e9:.3f} GB"
          f" per launch; {(rd + wr) / 1e9:.3f} GB / {t:.6f} s = {(rd + wr) / t / 1e9:.0f} GB/s = {(rd + wr) / t / 8e12 * 100:.2f} % of HBM bandwidth")
    print(f"traffic_bytes_per_launch,{rd + wr:.0f}")

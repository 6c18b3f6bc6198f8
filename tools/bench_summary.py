#!/usr/bin/env python3
"""The rows of a bench.py line, one per line.   python tools/bench_summary.py LINE.json"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(f"headline {d['value']:.4e} {d['unit']}, {d['ms_per_step']:.2f} ms a step, roofline {d['roofline']['frac']:.4f} at {d['roofline'].get('clock_ghz')} GHz")
for k, v in d.get("configs", {}).items():
    clk = v['roofline'].get('clock_ghz')
    clk = f", {clk:.3f} GHz, {v['roofline']['cycles_per_interaction']:.2f} cycles per interaction" if clk else ""
    print(f"configs.{k}: {v['ms_per_step']:.4f} ms a step, force pass {v['roofline']['frac']:.4f}, whole step {v['roofline']['whole_step_frac']:.4f}, plan {v.get('plan')}{clk}")
for k, v in d.get("mid_sizes", {}).items():
    print(f"mid_sizes.{k}: {v['us_per_step']:.1f} us a step, whole step {v['whole_step_frac']:.4f}, plan {v['plan']}, {v['i_per_lane']} bodies per lane")
for k, v in d.get("bh", {}).items():
    print(f"bh.{k}: {v['us_per_frame']:.1f} us a frame")

#!/usr/bin/env python3
"""Whole steps with and without the per-kernel HIP events (nbody_params.time_kernels): what the event records themselves cost
at launch-bound sizes.   python tools/step_untimed.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parallelnbody_amd as nb
for n in (2000, 8192, 12288, 16384, 20480, 32768, 65536):
    posm, vel = nb.ic_plummer(n, seed=1)
    row = []
    for timed in (True, False):
        with nb.NBodyEngine(n, time_kernels=timed) as e:
            e.set_state(posm, vel)
            e.step(0.002, 400); e.synchronize()
            k = max(200, int(0.3 / (n * n / 6e12 + 1e-5)))
            t = time.perf_counter(); e.step(0.002, k); e.synchronize()
            row.append((time.perf_counter() - t) / k * 1e3)
            kern = e.launch_config()["kernel"]
    print(f"N={n:6d} {kern:22s} step with events {row[0]:.4f} ms   without {row[1]:.4f} ms", flush=True)

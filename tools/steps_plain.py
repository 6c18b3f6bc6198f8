#!/usr/bin/env python3
"""K whole steps at N bodies with the library's defaults and no per-kernel events (what a host runs): for rocprofv3
--kernel-trace + tools/trace_gaps.py.   python tools/steps_plain.py N [K [distinct|equal [precision [eps]]]]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallelnbody_amd as nb
n = int(sys.argv[1]); k = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
posm, vel = nb.ic_plummer(n, seed=1)
if len(sys.argv) > 3 and sys.argv[3] == "distinct":
    posm[:, 3] *= np.random.default_rng(1).uniform(0.5, 1.5, n).astype(np.float32)
prec = sys.argv[4] if len(sys.argv) > 4 else "f32"
eps = float(sys.argv[5]) if len(sys.argv) > 5 else 0.0
with nb.NBodyEngine(n, precision=prec, eps=eps) as e:
    e.set_state(posm, vel)
    e.step(0.002, 300); e.synchronize()
    t = time.perf_counter(); e.step(0.002, k); e.synchronize()
    dt = (time.perf_counter() - t) / k
    print(f"N={n} {prec} {e.launch_config()['kernel']} {dt * 1e3:.4f} ms/step  {n * n * 20 / dt / 157.3e12 * 100:.1f} % of peak (whole step, no events)", flush=True)

#!/usr/bin/env python3
"""(Lives under tests/ because it times the CPU oracle next to the device: only tests/, smoke() and bench.py's
cpu_baseline leg may touch oracle/.)  Barnes-Hut (theta = 1.0, the reference's shipped opening angle) on the device vs the CPU restatement of the
reference's tree (one core): time per force pass."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # repo root
import numpy as np
import parallelnbody_amd as nb
from oracle import oracle as O

for n, box in ((2000, True), (65536, False), (1 << 20, False)):
    posm, vel = nb.ic_reference_box(n, 1000.0, seed=1) if box else nb.ic_plummer(n, seed=1)
    with nb.NBodyEngine(n, theta=1.0, time_kernels=True) as e:
        e.set_state(posm, vel)
        e.compute_forces(); e.synchronize(); e.kernel_time_reset()
        reps = 20 if n <= 65536 else 3
        t0 = time.perf_counter()
        for _ in range(reps): e.compute_forces()
        e.synchronize(); wall = (time.perf_counter() - t0) / reps
        ms, k = e.kernel_time(nb.KERNEL_FORCES)
        st = e.bh_stats()
    t_cpu = None
    if n <= 65536:
        t0 = time.perf_counter(); O.octree_forces_f32(posm[:, :3], posm[:, 3], 1.0); t_cpu = time.perf_counter() - t0
    print(f"N={n:8d} theta=1.0  device pass {ms / k:9.3f} ms (wall {wall * 1e3:9.3f} ms)  nodes {st['nodes']:9d} levels {st['levels']:2d}"
          + (f"  CPU restatement of the reference tree, 1 core: {t_cpu * 1e3:9.2f} ms" if t_cpu else ""), flush=True)

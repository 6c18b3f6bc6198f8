"""Kernel-level picture of one Barnes-Hut frame (run under rocprofv3 --kernel-trace --stats): python3 tools/bh_trace.py N"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parallelnbody_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
posm, vel = nb.ic_reference_box(n, 1000.0, seed=1) if n <= 8192 else nb.ic_plummer(n, seed=1)
with nb.NBodyEngine(n, theta=1.0) as e:
    e.set_state(posm, vel)
    for _ in range(20): e.compute_forces()
    e.synchronize()

"""python3 tools/small_trace.py — 2000 fused small-system steps back to back (run under rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parallelnbody_amd as nb
posm, vel = nb.ic_reference_box(2000, 1000.0, seed=1)
with nb.NBodyEngine(2000) as e:
    e.set_state(posm, vel)
    e.step(0.01, 2000)
    e.synchronize()

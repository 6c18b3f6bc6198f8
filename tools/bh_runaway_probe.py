"""Which frames of a runaway-owned scene does the sort from the previous order give up?  (tuning aid, GPU box)
    python3 tools/bh_runaway_probe.py N runaway_speed runaway_mass frames"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallelnbody_amd as nb

n = int(sys.argv[1]); speed = float(sys.argv[2]); mass = float(sys.argv[3]); frames = int(sys.argv[4])
rng = np.random.default_rng(n)
posm = np.concatenate([rng.uniform(-1.0, 1.0, (n, 3)) + 3.0, 10.0 ** rng.uniform(-12, -9, (n, 1))], 1).astype(np.float32)
posm[5, :3] = (1.0e7, -2.0e6, 3.0e6); posm[5, 3] = mass
vel = np.concatenate([rng.normal(0, 0.05, (n, 3)), np.zeros((n, 1))], 1).astype(np.float32)
vel[5, :3] = (speed, 0.0, 0.0)
with nb.NBodyEngine(n, theta=1.0) as e:
    e.set_state(posm, vel)
    prev = (0, 0)
    for f in range(frames):
        e.step(0.01, 1)
        w, r = ctypes.c_longlong(), ctypes.c_longlong()
        e._L.nbody_debug_bh_sort_counts(e._h, ctypes.byref(w), ctypes.byref(r))
        st = e.bh_stats()
        print(f"frame {f}: warm {w.value - prev[0]} given up {r.value - prev[1]} levels {st['levels']} nodes {st['nodes']} root {st['root_com']} size {e.bounds():.8g}", flush=True)
        prev = (w.value, r.value)

#!/usr/bin/env python3
"""Where an all-pairs actor frame's time goes (N = 2000, theta = 0): nbody_tick with neither Size nor mirror asked for (the
launch and the wait), with Size only (+ a 4-byte copy), with the mirror only (+ an 80 KB copy into pinned caller memory),
with both; and nbody_step + nbody_synchronize for comparison.   python3 tools/tick_parts.py [N]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallelnbody_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
posm, vel = nb.ic_reference_box(n, 1000.0, seed=1)
with nb.NBodyEngine(n) as e:
    e.set_state(posm, vel)
    out = np.zeros(n, nb.PARTICLE_DTYPE); e.pin(out)
    size = ctypes.c_float(0.0)
    L, h, dt = e._L, e._h, ctypes.c_float(0.01)
    cases = {
        "step + synchronize": lambda: (L.nbody_step(h, dt, 1), L.nbody_synchronize(h)),
        "tick, nothing handed over": lambda: L.nbody_tick(h, dt, None, None, 0),
        "tick, Size": lambda: L.nbody_tick(h, dt, ctypes.byref(size), None, 0),
        "tick, mirror (pinned)": lambda: L.nbody_tick(h, dt, None, out.ctypes.data, 40),
        "tick, Size + mirror (pinned)": lambda: L.nbody_tick(h, dt, ctypes.byref(size), out.ctypes.data, 40),
    }
    for name, f in cases.items():
        for _ in range(100): f()
        best = 1e30
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(1000): f()
            best = min(best, (time.perf_counter() - t0) / 1000)
        print(f"N={n} {name:32s} {best * 1e6:6.1f} us per frame", flush=True)

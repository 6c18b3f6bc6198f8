#!/usr/bin/env python3
"""What the host's wait costs an actor-style frame: nbody_tick frames (N = 2000) at theta = 0 and 1 under the device's
scheduling flags (hipSetDeviceFlags before anything else touches the GPU): auto (default), spin, yield, blocking sync.
    python3 tools/tick_sched.py FLAG      FLAG in auto | spin | yield | blocking"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
flag = sys.argv[1] if len(sys.argv) > 1 else "auto"
hip = ctypes.CDLL("libamdhip64.so")
flags = {"auto": 0x0, "spin": 0x1, "yield": 0x2, "blocking": 0x4}[flag]
rc = hip.hipSetDeviceFlags(ctypes.c_uint(flags))
import numpy as np
import parallelnbody_amd as nb
for n, theta in ((2000, 0.0), (2000, 1.0), (8192, 0.0)):
    posm, vel = nb.ic_reference_box(n, 1000.0, seed=1)
    with nb.NBodyEngine(n, theta=theta) as e:
        e.set_state(posm, vel)
        out = np.zeros(n, nb.PARTICLE_DTYPE); e.pin(out)
        for _ in range(50): e.tick(0.01, out=out)
        best = 1e30
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(500): e.tick(0.01, out=out)
            best = min(best, (time.perf_counter() - t0) / 500)
    print(f"hipSetDeviceFlags({flag}) rc={rc}: N={n} theta={theta} nbody_tick {best * 1e6:.1f} us per frame", flush=True)

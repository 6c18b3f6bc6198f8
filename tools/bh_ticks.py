"""The reference's shipped frame on the device: K Ticks (OctreeSearch.cpp:25-31) at theta = 1.0 (OctreeSearch.cpp:85).
    python3 tools/bh_ticks.py N [K [mode [theta [scene]]]]   scene: box | plummer (default: box up to 16384 bodies); mode: step (nbody_step K frames in one call, default) | tick (actor-style
                                                 nbody_tick per frame: bounds + step + FParticle mirror, one host sync each);
                                                 theta: 1.0 (default) or e.g. 0 for the all-pairs frame
Prints wall time per frame; run under `rocprofv3 --kernel-trace --stats` for the per-kernel picture."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallelnbody_amd as nb

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 200
mode = sys.argv[3] if len(sys.argv) > 3 else "step"
theta = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
scene = sys.argv[5] if len(sys.argv) > 5 else ("box" if n <= 16384 else "plummer")      # the shipped kind of scene / BASELINE's kind
posm, vel = nb.ic_reference_box(n, 1000.0, seed=1) if scene == "box" else nb.ic_plummer(n, seed=1)
with nb.NBodyEngine(n, theta=theta) as e:
    e.set_state(posm, vel)
    e.step(0.01, 3); e.synchronize()
    out = np.zeros(n, nb.PARTICLE_DTYPE)
    e.pin(out)
    best = 1e30
    for rep in range(3):
        t0 = time.perf_counter()
        if mode == "tick":
            for _ in range(k):
                e.tick(0.01, out=out)
        else:
            e.step(0.01, k)
        e.synchronize()
        best = min(best, (time.perf_counter() - t0) / k)
    st = e.bh_stats() if theta > 0 else {'nodes': 0, 'levels': 0}
    import ctypes
    warm, retries = ctypes.c_longlong(), ctypes.c_longlong()
    counts = ""
    if theta > 0 and n > 4096 and e._L.nbody_debug_bh_sort_counts(e._h, ctypes.byref(warm), ctypes.byref(retries)) == 0:
        counts = f"; {warm.value} frames sorted from the previous order, {retries.value} times frames were queued again"
    size = e.bounds()
print(f"N={n} theta={theta} mode={mode} scene={scene}: {best * 1e6:.1f} us per frame (best of 3 runs of {k}); nodes {st['nodes']} levels {st['levels']}{counts}; Size at the end {size:.6g}", flush=True)

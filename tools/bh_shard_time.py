#!/usr/bin/env python3
"""What one rank of a P-GPU job does per theta = 1 frame, timed on one GPU: a context that owns the slice [r N/P, (r+1) N/P) builds
the WHOLE tree and walks + integrates its own bodies (nbody_step on the slice; the other slices' bodies stand still here, and the
all-gather between frames — N/P x 16 bytes per rank — is not part of this).  The build is replicated, so the frame shrinks towards
the build's time as P grows: the bound on what sharding the reference's shipped algorithm can buy.
    python3 tools/bh_shard_time.py N FRAMES [scene]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallelnbody_amd as nb
n = int(sys.argv[1]); k = int(sys.argv[2]); scene = sys.argv[3] if len(sys.argv) > 3 else "plummer"
posm, vel = nb.ic_plummer(n, seed=1) if scene == "plummer" else nb.ic_reference_box(n, 1000.0, seed=1)
vel = vel * np.float32(0.0)                                      # (only the own slice moves here: keep it gentle)
base = None
for P in (1, 2, 4, 8):
    ic = n // P
    r = P // 2
    with nb.NBodyEngine(n, i_begin=r * ic, i_count=ic, theta=1.0) as e:
        e.set_state(posm, vel)
        for _ in range(5):
            e.step(1e-6, 1)
        e.synchronize()
        best = 1e30
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(k):
                e.step(1e-6, 1)                                  # a sharded context advances one frame per call (the all-gather goes in between)
            e.synchronize()
            best = min(best, (time.perf_counter() - t0) / k)
    base = base or best
    print(f"N={n} {scene} P={P} rank {r}: {best * 1e6:8.1f} us per frame (own slice of {ic} bodies; one call and one wait per frame)  "
          f"speed-up over P=1: {base / best:.2f}", flush=True)

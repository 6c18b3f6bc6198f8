"""Cold frames of a scene whose Size a runaway body owns (nearly all bodies in ONE cell of level 21: one run of equal first key words):
a cold frame that sorts by the first word places the run's bodies by counting (quadratic in the run); one that knows the run is there
(header word 6 of the frame before) — or knows nothing yet (a new scene) — sorts by both words.  NBODY_BH_SORT_BOTH=0 shows the former.   python3 tools/bh_ties_timing.py N   (GPU box)"""
import os, sys, time
os.environ["NBODY_BH_WARM_SORT"] = "0"                         # every frame cold (read when the theta > 0 state is created)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallelnbody_amd as nb

n = int(sys.argv[1])
rng = np.random.default_rng(n)
posm = np.concatenate([rng.uniform(-1.0, 1.0, (n, 3)) + 3.0, 10.0 ** rng.uniform(-12, -9, (n, 1))], 1).astype(np.float32)
posm[5, :3] = (1.0e7, -2.0e6, 3.0e6); posm[5, 3] = 1e-24
with nb.NBodyEngine(n, theta=1.0) as e:
    e.set_state(posm, np.zeros((n, 4), np.float32))
    out = []
    for k in range(5):
        t0 = time.perf_counter()
        e.compute_forces(); e.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    st = e.bh_stats()
print(f"N={n}: cold force passes, ms: the scene's first frame {out[0]:.3f} (creates the theta > 0 state; round 5's first build sorted it by the first "
      f"word and placed the run by counting, the final one by both words: nothing is known about a new scene); then by both words "
      f"{', '.join(f'{v:.3f}' for v in out[1:])}; levels {st['levels']}, nodes {st['nodes']}", flush=True)

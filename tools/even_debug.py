#!/usr/bin/env python3
"""Where an even-share pass and a guided pass of the same scene differ, and who is right (fp64 direct sum).
   python tools/even_debug.py N IPT"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallelnbody_amd as nb
from oracle import oracle as O
n, ipt = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(n + ipt)
posm = np.concatenate([rng.uniform(-500, 500, (n, 3)), rng.uniform(1, 5000, (n, 1))], 1).astype(np.float32)
vel = np.zeros((n, 4), np.float32)
acc = {}
for even in (1, 0):
    os.environ["NBODY_SYM_EVEN"] = str(even)
    with nb.NBodyEngine(n, algorithm=2, i_per_thread=ipt) as e:
        print(e.launch_config())
        e.set_state(posm, vel); e.compute_forces(); acc[even] = e.accelerations()[:, :3].astype(np.float64)
d = np.linalg.norm(acc[1] - acc[0], axis=1) / np.linalg.norm(acc[0], axis=1)
print("bodies with rel diff > 1e-6:", int((d > 1e-6).sum()), " > 5e-6:", int((d > 5e-6).sum()), " max", d.max())
worst = np.argsort(d)[-12:][::-1]
p64 = posm.astype(np.float64)
for i in worst:
    ref = O.forces_direct_f64(p64[:, :3], p64[:, 3], i0=int(i), i1=int(i) + 1)[0]
    ee = np.linalg.norm(acc[1][i] - ref) / np.linalg.norm(ref); eg = np.linalg.norm(acc[0][i] - ref) / np.linalg.norm(ref)
    print(f"body {i:6d} (granule {i // 64}, block {i // (256 * ipt)}) diff {d[i]:.2e}  even vs fp64 {ee:.2e}  guided vs fp64 {eg:.2e}  |a| {np.linalg.norm(ref):.3e}")
print("median |a|", np.median(np.linalg.norm(acc[0], axis=1)))

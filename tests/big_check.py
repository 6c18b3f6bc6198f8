"""Ad-hoc check of one force pass at N = 2^22 on the GPU (Newton's third law over all bodies, a sample against the fp64
oracle).  Not collected by pytest; run as `python tests/big_check.py`.  Lives under tests/ because it uses the oracle."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import parallelnbody_amd as nb
from oracle import oracle as O
O.build()
n = 1 << 22
posm, vel = nb.ic_plummer(n, seed=22)
with nb.NBodyEngine(n, time_kernels=True) as e:
    print(e.launch_config())
    e.set_state(posm, vel)
    t = time.time(); e.compute_forces(); e.synchronize(); print("pass s", time.time() - t)
    a = e.accelerations()
print(np.isfinite(a).all())
p64 = posm.astype(np.float64)
f = (a.astype(np.float64) * p64[:, 3:4]).sum(0)
print("third law", np.linalg.norm(f) / (np.linalg.norm(a, axis=1) * p64[:, 3]).sum())
rng = np.random.default_rng(0)
for i in rng.choice(n, 4, replace=False):
    ref = O.forces_direct_f64(p64[:, :3], p64[:, 3], i0=int(i), i1=int(i) + 1)
    print(i, np.linalg.norm(a[i, :3] - ref[0]) / np.linalg.norm(ref[0]))

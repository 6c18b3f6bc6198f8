"""Pool phases against the one-pass plan on the same bodies, and the error of the N = 2^23 pass body by body.
(tools/: uses numpy's fp64 direct sum of bench.py, not the oracle.)   python3 tools/check_phases.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallelnbody_amd as nb
import bench

def ref_rows(posm, rows, eps=0.0):
    p = posm.astype(np.float64); out = []
    for i in rows:
        d = p[:, :3] - p[i, :3]; r2 = (d * d).sum(1) + eps * eps
        with np.errstate(divide="ignore", invalid="ignore"):
            s = np.where(r2 > 0, 1.0e4 * p[:, 3] / (r2 * np.sqrt(r2)), 0.0)
        out.append((s[:, None] * d).sum(0))
    return np.array(out)

n = 1 << 20
posm, vel = nb.ic_plummer(n, seed=5)
rows = bench.sample_bodies(0, n, 4096, 16, seed=3)
with nb.NBodyEngine(n) as e:
    e.set_state(posm, vel); e.compute_forces(); a1 = e.accelerations()
os.environ["NBODY_SYM_POOL_BUDGET_MB"] = "512"
with nb.NBodyEngine(n) as e:
    print("N=2^20 forced budget 512 MB: pool, phases", e.sym_pool())
    e.set_state(posm, vel); e.compute_forces(); a2 = e.accelerations()
del os.environ["NBODY_SYM_POOL_BUDGET_MB"]
den = np.linalg.norm(a1, axis=1)
print("phased vs one pass, all bodies: max rel diff", float((np.linalg.norm(a2 - a1, axis=1) / den).max()))
r = ref_rows(posm, rows)
for nm, a in (("one pass", a1), ("phased", a2)):
    print(nm, "vs fp64 on", len(rows), "bodies: max rel err", float((np.linalg.norm(a[rows] - r, axis=1) / np.linalg.norm(r, axis=1)).max()))

n = 1 << 23
posm, vel = nb.ic_plummer(n, seed=23)
with nb.NBodyEngine(n, time_kernels=True) as e:
    print("N=2^23: pool, phases", e.sym_pool(), e.launch_config())
    e.set_state(posm, vel); e.compute_forces(); a = e.accelerations()
    ms, k = e.kernel_time(nb.KERNEL_FORCES)
print(f"force pass {ms / k:.1f} ms  {float(n) * n / (ms / k * 1e-3):.3e} interactions/s")
rows = bench.sample_bodies(0, n, 4096, 16, seed=23)[:24]
r = ref_rows(posm, rows)
err = np.linalg.norm(a[rows] - r, axis=1) / np.linalg.norm(r, axis=1)
absa = np.array([np.abs(r_).sum() for r_ in r])
for i, e_, rr in zip(rows, err, r):
    print(f"  body {i:8d} rel err {e_:.2e}  |a| {np.linalg.norm(rr):.3e}  radius {np.linalg.norm(posm[i, :3]):.2f}")
with nb.NBodyEngine(n, algorithm=1) as e:
    e.set_state(posm, vel); e.compute_forces(); at = e.accelerations()
errt = np.linalg.norm(at[rows] - r, axis=1) / np.linalg.norm(r, axis=1)
print("one-sided kernel on the same bodies: max rel err", float(errt.max()), "symmetric (phased):", float(err.max()))

import os, sys, time
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import numpy as np
import parallelnbody_amd as nb, bench
def ref_rows(posm, rows, eps):
    p = posm.astype(np.float64); out = []
    for i in rows:
        d = p[:, :3] - p[i, :3]; r2 = (d * d).sum(1) + eps * eps
        with np.errstate(divide="ignore", invalid="ignore"):
            s = np.where(r2 > 0, 1.0e4 * p[:, 3] / (r2 * np.sqrt(r2)), 0.0)
        out.append((s[:, None] * d).sum(0))
    return np.array(out)
for n, prec, eps in ((1 << 21, "f32", 0.0), (1 << 21, "f32_kahan", 0.5), (1 << 22, "f32", 0.0)):
    posm, vel = nb.ic_plummer(n, seed=21)
    rows = bench.sample_bodies(0, n, 4096, 16, seed=2)[:20]
    r = ref_rows(posm, rows, eps)
    for cap in ("0", "2048", "1024", "512"):
        os.environ["NBODY_SYM_MAX_SUB"] = cap
        with nb.NBodyEngine(n, precision=prec, eps=eps, time_kernels=True) as e:
            e.set_state(posm, vel); e.compute_forces(); e.synchronize(); e.kernel_time_reset()
            e.compute_forces(); e.compute_forces()
            ms, k = e.kernel_time(nb.KERNEL_FORCES)
            a = e.accelerations()
            err = (np.linalg.norm(a[rows] - r, axis=1) / np.linalg.norm(r, axis=1)).max()
            print(f"N={n} {prec} cap {cap:>5s}: {ms / k:9.2f} ms/pass {float(n) * n / (ms / k * 1e-3):.3e}/s  pool {e.sym_pool()[0] / 1e9:.1f} GB items {e.launch_config()['blocks']}  max rel err {err:.2e}", flush=True)

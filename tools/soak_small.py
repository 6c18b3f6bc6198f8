#!/usr/bin/env python3
"""Create, run and destroy many small contexts (the one-launch step in all three precisions, the fused actor frame with a
pinned and an unpinned mirror, the Barnes-Hut frame): device memory before and after must agree.  python3 tools/soak_small.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import parallelnbody_amd as nb
def free(): return torch.cuda.mem_get_info()[0] / 2**20
torch.cuda.init(); f0 = None
for k in range(400):
    n = [2000, 8192, 3001, 16384][k % 4]
    prec = ["f32", "f32_kahan", "f64", "f32"][k % 4] if k % 8 < 4 else "f32"
    theta = 1.0 if (k % 5 == 0 and prec == "f32" and n <= 8192) else 0.0
    posm, vel = nb.ic_plummer(n, seed=k + 1)
    if prec == "f64": posm, vel = posm.astype(np.float64), vel.astype(np.float64)
    with nb.NBodyEngine(n, precision=prec, theta=theta, eps=0.5 if k % 3 == 0 else 0.0) as e:
        e.set_state(posm, vel)
        buf = np.zeros(n, nb.PARTICLE_DTYPE)
        if k % 2 == 0: e.pin(buf)
        for _ in range(3): e.tick(0.01, out=buf)
        e.step(0.01, 5)
        assert np.isfinite(e.positions()).all() and np.isfinite(buf["Position"]).all()
    if k == 20: f0 = free()
print("free MiB after 20 / after 400 contexts:", round(f0), round(free()))
assert abs(free() - f0) < 64

import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd())
import torch, parallelnbody_amd as nb
def free(): return torch.cuda.mem_get_info()[0] / 2**20
posm, vel = nb.ic_plummer(32768, seed=1)
f0 = free()
for k in range(150):
    prec = ["f32", "f32_kahan", "f64"][k % 3]
    th = 1.0 if (k % 5 == 0 and prec == "f32") else 0.0
    with nb.NBodyEngine(32768, precision=prec, theta=th, time_kernels=(k % 2 == 0)) as e:
        e.set_state(posm.astype(np.float64) if prec == "f64" else posm, vel.astype(np.float64) if prec == "f64" else vel)
        buf = np.zeros(32768, nb.PARTICLE_DTYPE); e.pin(buf)
        e.tick(0.01, out=buf)
        e.step(0.01, 2)
        assert np.isfinite(e.positions()).all()
print("free MiB before/after 150 contexts:", round(f0), round(free()))
# long run
posm, vel = nb.ic_plummer(65536, seed=2)
with nb.NBodyEngine(65536, eps=1.0) as e:
    e.set_state(posm, vel)
    t = time.time(); e.step(0.002, 3000); e.synchronize(); print("3000 steps N=65536:", round(time.time() - t, 2), "s")
    ke, pe = e.energy(); print("energy", ke + pe, np.isfinite(e.positions()).all())

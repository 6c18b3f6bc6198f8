#!/usr/bin/env python3
"""Step latency at small N (the reference's shipped scene is N = 2000): wall time per nbody_step and per actor Tick."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parallelnbody_amd as nb

for n in (1024, 2000, 4096, 8192, 16384):
    posm, vel = nb.ic_reference_box(n, 1000.0, seed=1)
    with nb.NBodyEngine(n, time_kernels=False) as e:
        e.set_state(posm, vel)
        e.step(1e-6, 50); e.synchronize()
        t0 = time.perf_counter(); e.step(1e-6, 2000); e.synchronize(); dt = (time.perf_counter() - t0) / 2000
        cfg = e.launch_config()
    with nb.NBodyEngine(n, time_kernels=True) as e:
        e.set_state(posm, vel)
        e.step(1e-6, 200); e.synchronize()
        f_ms, f_n = e.kernel_time(nb.KERNEL_FORCES); u_ms, u_n = e.kernel_time(nb.KERNEL_UPDATE)
    upd = f"{u_ms / u_n * 1e3:6.1f} us" if u_n else "fused into the force launch"
    print(f"N={n:6d}  step {dt*1e6:8.1f} us  ({n*n/dt:.3e} pairs/s)  force kernel {f_ms/f_n*1e3:7.1f} us  update {upd}  cfg {cfg}")
a = nb.OctreeSearch(); a.set_seed(1); a.CreateSpacePoints(2000, 1000.0); a.PhDeltaTime = 1e-6
cnt = [0]
a.set_draw_callbacks(on_point=None)
for _ in range(20): a.Tick(0.0)
t0 = time.perf_counter()
for _ in range(200): a.Tick(0.0)
print(f"actor Tick N=2000 (with Particles mirror download): {(time.perf_counter()-t0)/200*1e6:.1f} us")

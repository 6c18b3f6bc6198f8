#!/usr/bin/env python3
"""What one rank of a P-GPU job computes, timed on one GPU: the force pass of rank 0's share of N = 2^20 bodies
(symmetric algorithm; the exchange itself is not part of this).  Ideal strong scaling = time(P) * P == time(1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parallelnbody_amd as nb
n = 1 << 20
posm, vel = nb.ic_plummer(n, seed=1)
base = None
for P in (1, 2, 4, 8):
    for r in sorted({0, P - 1}):
        ic = n // P
        with nb.NBodyEngine(n, i_begin=r * ic, i_count=ic, time_kernels=True) as e:
            e.set_state(posm, vel)
            cfg = e.launch_config()
            for _ in range(3 * P):                      # ~0.5 s of passes first: the clock needs sustained load to settle
                e.step_begin(); e.step_end(0.0)
            e.synchronize(); e.kernel_time_reset()
            for _ in range(3 * P):
                e.step_begin(); e.step_end(0.0)
            f_ms, k = e.kernel_time(nb.KERNEL_FORCES); u_ms, ku = e.kernel_time(nb.KERNEL_UPDATE)
        t = f_ms / k
        base = base or t
        print(f"P={P} rank {r}: {cfg['algorithm']:9s} super tile {cfg['super_tile']:5d} workgroups {cfg['blocks']:6d}  force pass {t:8.3f} ms  x P = {t * P:8.3f}  (efficiency {base / (t * P):.3f})  update {u_ms / ku:.3f} ms", flush=True)

#!/usr/bin/env python3
"""Reproducibility soak of the stepping paths: the same initial state stepped twice must end in the same bytes, and the
fused single-device path must end in the same bytes as the two-kernel path (position pointer handed out) — for sizes on
both sides of every kernel-selection boundary, equal and distinct masses, plain / Kahan / fp64, exact and softened, and a
pair of coincident bodies (guarded loops).  A race in the flags, the folds or the detector tables shows up here as a
difference.    python tools/repro_soak.py [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parallelnbody_amd as nb  # noqa: E402



def run(steps=60, sizes=(9000, 12288, 16384, 19000, 20480, 24000, 24576, 33000, 40960, 65536, 88000, 90112, 135000, 139264), out=print):
    """Returns the number of configurations whose end states differ."""
    bad = 0
    for n in sizes:
        for equal in (True, False):
            for prec, eps in (("f32", 0.0), ("f32", 0.3), ("f32_kahan", 0.3), ("f64", 0.0)):
                for dup in (False, True):
                    if dup and (eps > 0 or prec == "f64"):
                        continue
                    posm, vel = nb.ic_plummer(n, seed=n)
                    if not equal:
                        posm[:, 3] *= np.random.default_rng(n).uniform(0.5, 1.5, n).astype(np.float32)
                    if dup:
                        posm[n // 3, :3] = posm[7, :3]; vel[n // 3] = vel[7]          # they stay together
                    if prec == "f64":
                        posm = posm.astype(np.float64); vel = vel.astype(np.float64)
                    ends = []
                    for mode in ("fused", "fused", "pointer"):
                        with nb.NBodyEngine(n, precision=prec, eps=eps) as e:
                            if mode == "pointer":
                                e.device_ptr(nb.BUF_POSM)
                            e.set_state(posm, vel)
                            e.step(0.002, steps)
                            ends.append([a.tobytes() for a in e.state(np.float64 if prec == "f64" else np.float32)])
                            form = e.equal_mass_form(); cfg = e.launch_config()
                            kern = cfg["kernel"] + (" (even shares)" if cfg["plan"] == "even" else "")
                    ok = ends[0] == ends[1] == ends[2]
                    bad += not ok
                    out(f"N={n:6d} {prec:9s} eps={eps} equal={equal!s:5s} dup={dup!s:5s} {kern:36s} equal-mass form {form!s:5s} "
                        f"{'identical' if ok else 'DIFFERENT: rerun %s, pointer path %s' % (ends[0] == ends[1], ends[0] == ends[2])}")
    return bad


if __name__ == "__main__":
    n_bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 60)
    print("soak:", "all identical" if not n_bad else f"{n_bad} configurations differ")
    sys.exit(1 if n_bad else 0)

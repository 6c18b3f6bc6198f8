"""Regenerates tests/golden/*.npz.

Inputs come from the product's seeded host-side generators (nbody_ic_plummer / nbody_ic_reference_box —
no GPU involved); expected outputs come from the CPU oracle (oracle/nbody_oracle.c).  The reference
itself cannot be built in this image (it needs Unreal Engine 4.9 headers), so these fixtures pin the
ORACLE's outputs, not the reference's: parity stays "unpinned" (see DESIGN.md).

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import parallelnbody_amd as nb          # noqa: E402
from oracle import oracle as O          # noqa: E402

DT = 0.01   # OctreeSearch.cpp:8


def make(name, posm, vel):
    pos = np.ascontiguousarray(posm[:, :3])
    mass = np.ascontiguousarray(posm[:, 3])
    acc_direct = O.forces_direct_f32(pos, mass)                    # index-order all-pairs
    acc_tree0, _, _ = O.octree_forces_f32(pos, mass, theta=0.0)    # the reference's walk at theta = 0 (DFS order)
    pos1, vel1 = O.kick_drift_f32(pos, vel[:, :3], acc_direct, DT)
    acc64 = O.forces_direct_f64(pos.astype(np.float64), mass.astype(np.float64))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), posm=posm, vel=vel, acc_direct=acc_direct,
                        acc_tree0=acc_tree0, acc_f64=acc64, pos1=pos1, vel1=vel1, dt=np.float32(DT),
                        bounds=np.float32(O.bounds_f32(pos)))
    print(name, posm.shape, "written")


if __name__ == "__main__":
    O.build()
    make("plummer_n1024_seed1", *nb.ic_plummer(1024, total_mass=1000.0, scale_radius=100.0, G=1.0e4, seed=1))
    make("refbox_n2000_seed1", *nb.ic_reference_box(2000, 1000.0, seed=1))   # the shipped scene: N=2000, Size=1000

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallelnbody_amd as nb
from oracle import oracle as O
g = np.load("tests/golden/refbox_n2000_seed1.npz")
ref = g["acc_direct"]
for ipt in (1, 2, 4):
    for zm in (0, 2):
        for js in (1,):
            with nb.NBodyEngine(2000, i_per_thread=ipt, zero_mode=zm, j_split=js) as e:
                e.set_state(g["posm"], g["vel"]); e.compute_forces(); a = e.accelerations()
            err = np.linalg.norm(a - ref, axis=1) / np.linalg.norm(ref, axis=1)
            bad = np.nonzero(~(err < 2e-5))[0]
            print(f"ipt {ipt} zero {zm} jsplit {js}: max err {np.nanmax(err):.3e} nbad {len(bad)} bad idx {bad[:10]} nan {np.isnan(a).sum()}")
            if len(bad): print("   a", a[bad[0]], "ref", ref[bad[0]], "pos", g["posm"][bad[0]])

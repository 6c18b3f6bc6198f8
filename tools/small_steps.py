import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parallelnbody_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
posm, vel = nb.ic_reference_box(n, 1000.0, seed=1)
with nb.NBodyEngine(n) as e:
    e.set_state(posm, vel)
    e.step(1e-6, 300)
    e.synchronize()

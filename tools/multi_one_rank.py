"""A one-device nbody_create_multi context stepping (one-rank RCCL communicators: the grouped send/recv and the in-place
all-gather still run): for rocprofv3 --kernel-trace.   python3 tools/multi_one_rank.py [N [steps]]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parallelnbody_amd as nb
n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
k = int(sys.argv[2]) if len(sys.argv) > 2 else 10
posm, vel = nb.ic_plummer(n, seed=1)
with nb.NBodyEngine(n, devices=[0]) as e:
    e.set_state(posm, vel)
    e.step(0.001, k)
    e.synchronize()
print("done", n, k)

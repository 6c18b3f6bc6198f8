#!/usr/bin/env python3
"""Two processes sharing ONE GPU, HIP engines on both, collectives over gloo on device tensors (RCCL refuses two
ranks on one device): an end-to-end run of the sharded path with a real process group.
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/two_rank_gloo_gpu.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import parallelnbody_amd as nb

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo")
n = 65536
posm, vel = nb.ic_plummer(n, seed=3)
sim = nb.ShardedSimulation(posm, vel, rank=rank, world_size=world, device="cuda:0")
cfg = sim.engine.launch_config()
sim.warm_collectives()
sim.step(0.001, 3)
torch.cuda.synchronize()
p, v = sim.gather_state()
# the same run with every collective and kernel in ONE stream order (no all-gather under the next force pass): every bit equal
seq = nb.ShardedSimulation(posm, vel, rank=rank, world_size=world, device="cuda:0", overlap=False)
seq.step(0.001, 3)
ps, vs = seq.gather_state()
assert sim.overlap and sim.gather_stream is not None and not seq.overlap
assert np.array_equal(p, ps) and np.array_equal(v, vs), "overlapped and sequential stepping differ"
seq.close()
if rank == 0:
    with nb.NBodyEngine(n, algorithm=1) as e:           # one-sided single-context reference run
        e.set_state(posm, vel); e.step(0.001, 3); pr, vr, _ = e.state()
    err = np.abs(p[:, :3] - np.concatenate([pr[:, :3]])).max() / np.abs(pr[:, :3]).max()
    print(f"two ranks on one GPU over gloo: algorithm {cfg['algorithm']}, exchange ranks {sim.ex_ranks}, max rel position diff vs single context {err:.2e}; "
          "all-gather under the next force pass: bits equal to the sequential order")
    assert err < 1e-6
dist.barrier(); sim.close()
# ... and at the reference's shipped opening angle (OctreeSearch.cpp:85): every rank builds the whole tree from the gathered positions
# and walks + integrates its own slice; EVERY BYTE of positions and velocities equals one context's
n = 30000
posm, vel = nb.ic_reference_box(n, 1000.0, seed=4)
bh = nb.ShardedSimulation(posm, vel, rank=rank, world_size=world, device="cuda:0", theta=1.0)
bh.step(0.01, 4)
pb, vb = bh.gather_state()
if rank == 0:
    with nb.NBodyEngine(n, theta=1.0) as e:
        e.set_state(posm, vel); e.step(0.01, 4); pr, vr, _ = e.state()
    assert np.array_equal(pb, pr) and np.array_equal(vb, vr), "theta = 1 over two ranks differs from one context"
    print("two ranks on one GPU over gloo at theta = 1.0: four frames, every byte of positions and velocities equal to one context's")
dist.barrier(); bh.close(); dist.destroy_process_group()

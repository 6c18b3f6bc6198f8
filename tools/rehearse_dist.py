#!/usr/bin/env python3
"""Rehearsal of the multi-GPU plumbing on ONE GPU: torchrun with a single rank, backend nccl (= RCCL).
Exercises init_process_group(device_id), barrier, all_reduce, the in-place all_gather_into_tensor of a slice of the
replicated tensor and all_to_all_single on device tensors, then a ShardedSimulation step loop and bench.py's timing code."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import parallelnbody_amd as nb

rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); lr = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
dev = torch.device("cuda", lr)
n = 65536
full = torch.arange(n * 4, dtype=torch.float32, device=dev).reshape(n, 4)
ic = n // world
own = full[rank * ic:(rank + 1) * ic]
ref = full.clone()
dist.all_gather_into_tensor(full, own)                      # in place: input is a view of the output
torch.cuda.synchronize()
assert torch.equal(full, ref), "in-place all_gather_into_tensor corrupted the tensor"
send = torch.randn(n, 4, device=dev); recv = torch.empty(world * ic, 4, device=dev)
dist.all_to_all_single(recv, send)
torch.cuda.synchronize()
assert torch.equal(recv[:ic], send[rank * ic:(rank + 1) * ic])
t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=dist.ReduceOp.MAX); dist.barrier()
posm, vel = nb.ic_plummer(n, seed=1)
sim = nb.ShardedSimulation(posm, vel, rank=rank, world_size=world, device=f"cuda:{lr}", time_kernels=True)
assert sim.stream is not None and sim.stream.cuda_stream != 0
sim.step(0.01, 3); torch.cuda.synchronize()
# stream ordering: what the in-place all-gather of step k ships must be step k's positions: issue one more by hand on the
# simulation's stream right behind a step and compare with a copy taken on the same stream.  (A single rank keeps the
# positions inside its engine — sim.posm is None — and gathers nothing.)
if world > 1:
    with torch.cuda.stream(sim.stream):
        sim.wait_for_positions()
        sim.engine.step_begin(); sim.engine.step_end(0.01)
        snap = sim.posm.clone()                                  # enqueued behind the update on the same stream
        dist.all_gather_into_tensor(sim.posm, sim.posm[sim.i_begin:sim.i_begin + sim.i_count])
    torch.cuda.synchronize()
    assert torch.equal(snap, sim.posm) and not torch.equal(snap, torch.from_numpy(posm).to(dev)), "stream ordering broken"
else:
    assert sim.posm is None
p, v = sim.gather_state(); ke, pe = sim.energy()
print("rehearsal ok: rank", rank, "of", world, "algorithm", sim.engine.launch_config()["algorithm"], "finite", bool(np.isfinite(p).all()), "E", ke + pe)
sim.close(); dist.barrier(); dist.destroy_process_group()

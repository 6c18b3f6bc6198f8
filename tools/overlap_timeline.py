#!/usr/bin/env python3
"""Where the all-gather of step k runs relative to step k+1's force pass, from timed events on the two streams.
Two ranks share the one GPU of the box, collectives over gloo on device tensors (RCCL refuses two ranks on one device: only
the transport differs from the 8-GPU job — gloo stages through the host, so its "gather" is far slower than xGMI's).
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/overlap_timeline.py [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import parallelnbody_amd as nb

rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 19
torch.cuda.set_device(0)
dist.init_process_group("gloo")
posm, vel = nb.ic_plummer(n, seed=3)
sim = nb.ShardedSimulation(posm, vel, rank=rank, world_size=world, device="cuda:0", timeline=True)
sim.warm_collectives()
sim.step(0.001, 6)
tl = sim.timeline_ms()
if rank == 0:
    cfg = sim.engine.launch_config()
    print(f"N={n}, {world} ranks on one GPU over gloo, algorithm {cfg['algorithm']}, {cfg['i_per_thread']} bodies per lane; times in ms since the first mark (rank 0)")
    by_step = {}
    for st, name, ms in tl:
        by_step.setdefault(st, {})[name] = ms
        print(f"  step {st}  {name:34s} {ms:10.3f}")
    hidden = 0
    for st in sorted(by_step):
        g = by_step.get(st - 1, {})
        f = by_step[st]
        g_end = g.get("all-gather end", g.get("all-gather end (seen by the host)"))
        if "all-gather begin" in g and g_end is not None and "local strips begin" in f:
            # the gather of step st-1 against the first go of step st
            overlap = min(g_end, f["local strips end"]) - max(g["all-gather begin"], f["local strips begin"])
            print(f"  step {st}: its local strips ran {f['local strips end'] - f['local strips begin']:.3f} ms, the all-gather of step {st - 1} "
                  f"{g_end - g['all-gather begin']:.3f} ms, both at once for {max(overlap, 0.0):.3f} ms")
            hidden += overlap > 0
    assert hidden >= 3, "the all-gather never ran under the local strips"
    print("overlap seen in", hidden, "steps")
dist.barrier(); sim.close(); dist.destroy_process_group()

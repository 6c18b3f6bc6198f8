#!/usr/bin/env python3
"""Stand-in for bench.py's worker process (NBODY_BENCH_WORKER_CMD) in tests/test_bench_watchdog.py: no GPU, a real gloo
process group on whatever rendezvous the watchdog hands over, and a behaviour chosen by WATCHDOG_STUB_MODE:
    ok        every attempt succeeds
    fail1     attempt 1: rank 1 raises after the rendezvous, rank 0 is left inside a collective
    hang1     attempt 1: rank 1 never returns
    late1     attempt 1: rank 0 has printed its line when rank 1 fails in its teardown
    failall   every attempt fails on rank 1
Rank 0 prints one JSON line that says which attempt produced it and with which arguments."""
import json
import os
import sys
import time

import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
attempt = int(os.environ["NBODY_BENCH_ATTEMPT"])
mode = os.environ.get("WATCHDOG_STUB_MODE", "ok")
assert sys.argv[-1] == "--worker" or "--worker" in sys.argv
dist.init_process_group("gloo")
import torch
t = torch.ones(1)
dist.all_reduce(t)
assert int(t[0]) == world
if rank == 1 and ((mode == "fail1" and attempt == 1) or mode == "failall"):
    print("stub: rank 1 fails on purpose", file=sys.stderr, flush=True)
    raise RuntimeError("injected")
if rank == 1 and mode == "hang1" and attempt == 1:
    print("stub: rank 1 hangs on purpose", file=sys.stderr, flush=True)
    time.sleep(1e6)
if not (mode == "late1" and attempt == 1):
    dist.barrier()                                     # fail1 / hang1: rank 0 waits here for a rank that never comes
if rank == 0:
    print(json.dumps({"metric": "stub", "value": 1.0, "attempt": attempt, "argv": sys.argv[1:], "port": os.environ["MASTER_PORT"],
                      "config": {"workload": "stub"}}), flush=True)
if mode == "late1" and attempt == 1:
    if rank == 1:
        time.sleep(0.5)
        raise RuntimeError("injected teardown failure")
    time.sleep(30)                                     # rank 0 dawdles: its line is out, the other rank fails meanwhile
dist.destroy_process_group()

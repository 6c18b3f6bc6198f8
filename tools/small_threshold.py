#!/usr/bin/env python3
"""Round 2 (small_pk_kernel, since replaced by forces_block_pk_kernel — tools/block_by_n.sh is the current measurement): where the one-launch small-system step stops paying: whole steps without events, library default against
the one-sided tile kernel forced (i_per_thread = 4) and the symmetric pass forced.   python tools/small_threshold.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parallelnbody_amd as nb
for n in (2000, 3000, 4096, 5000, 6000, 7000, 8000, 8192, 10240):
    posm, vel = nb.ic_reference_box(n, seed=1) if hasattr(nb, "ic_reference_box") else nb.ic_plummer(n, seed=1)
    out = []
    for kw in ({}, dict(i_per_thread=4), dict(i_per_thread=2), dict(algorithm=2, i_per_thread=2)):
        try:
            with nb.NBodyEngine(n, **kw) as e:
                e.set_state(posm, vel)
                e.step(0.002, 500); e.synchronize()
                t = time.perf_counter(); e.step(0.002, 2000); e.synchronize()
                out.append(f"{e.launch_config()['kernel'][:21]:21s} {(time.perf_counter() - t) / 2000 * 1e3:.4f} ms")
        except nb.NBodyError as err:
            out.append(f"refused ({err.code})")
    print(f"N={n:6d}  default: {out[0]}   ipt4: {out[1]}   ipt2: {out[2]}   symmetric ipt2: {out[3]}", flush=True)

#!/usr/bin/env python3
"""Whole steps under the even-share plan as its cost model's knobs move (distinct masses).
   python tools/even_knobs.py N IPT KNOB=v1,v2,... [KNOB=...]   (knobs: the NBODY_SYM_EVEN_* environment variables' suffixes)"""
import itertools, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallelnbody_amd as nb
n, ipt = int(sys.argv[1]), int(sys.argv[2])
form = "distinct"
args = sys.argv[3:]
if args and args[0] in ("distinct", "equal"):
    form = args.pop(0)
knobs = [(a.split("=")[0], a.split("=")[1].split(",")) for a in args]
posm, vel = nb.ic_plummer(n, seed=1)
if form == "distinct":
    posm[:, 3] *= np.random.default_rng(1).uniform(0.5, 1.5, n).astype(np.float32)
for combo in [None] + list(itertools.product(*[v for _, v in knobs])):
    os.environ["NBODY_SYM_EVEN"] = "0" if combo is None else "1"
    if combo is None:
        combo = ()
    for (k, _), v in zip(knobs, combo):
        os.environ["NBODY_SYM_EVEN_" + k] = v
    with nb.NBodyEngine(n, algorithm=2, i_per_thread=ipt) as e:
        e.set_state(posm, vel)
        e.step(0.002, 200); e.synchronize()
        k = max(100, int(0.25 / (n * n / 6e12 + 1e-5)))
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); e.step(0.002, k); e.synchronize()
            best = min(best, (time.perf_counter() - t) / k)
        print(f"N={n} ipt={ipt} {form} plan {e.launch_config()['plan']:6s} " + " ".join(f"{k}={v}" for (k, _), v in zip(knobs, combo)) + f"  items {e.launch_config()['blocks']}  {best * 1e6:8.1f} us  {n * n * 20 / best / 157.3e12 * 100:5.1f} %", flush=True)

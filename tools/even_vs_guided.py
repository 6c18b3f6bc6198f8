#!/usr/bin/env python3
"""Whole steps (no per-kernel events: what a host runs) of plain fp32 mid-size systems with the guided plan and with the
even-share plan (csrc/sym_plan.h), distinct masses (the general form) and equal masses, by N and bodies per lane.
   python tools/even_vs_guided.py [--kahan] [--default-only] [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import parallelnbody_amd as nb

KAHAN = "--kahan" in sys.argv
DEFAULT_ONLY = "--default-only" in sys.argv or "--before-after" in sys.argv
BEFORE_AFTER = "--before-after" in sys.argv       # same box: guided strips + the detector table of >= 2 N slots (rounds 2-4), then the defaults
sizes = [int(x) for x in sys.argv[1:] if not x.startswith("--")] or [12288, 16384, 20480, 24576, 32768, 40960, 49152, 65536, 98304, 131072]
PEAK = 157.3e12
PREC = dict(precision="f32_kahan") if KAHAN else {}


def run(n, posm, vel, even, ipt):
    if even is None:
        os.environ.pop("NBODY_SYM_EVEN", None)                       # the library's own choice
    else:
        os.environ["NBODY_SYM_EVEN"] = "1" if even else "0"
    kw = dict(algorithm=2, i_per_thread=ipt) if ipt else {}
    with nb.NBodyEngine(n, **kw, **PREC) as e:
        cfg = e.launch_config()
        e.set_state(posm, vel)
        e.step(0.002, 200); e.synchronize()
        k = max(100, int(0.25 / (n * n / 6e12 + 1e-5)))
        best = 1e9
        for _ in range(3):
            t = time.perf_counter(); e.step(0.002, k); e.synchronize()
            best = min(best, (time.perf_counter() - t) / k)
    return best, cfg


for n in sizes:
    posm, vel = nb.ic_plummer(n, seed=1)
    distinct = posm.copy()
    distinct[:, 3] *= np.random.default_rng(1).uniform(0.5, 1.5, n).astype(np.float32)
    for name, pm in (("distinct", distinct), ("equal", posm)):
        before = ""
        if BEFORE_AFTER:
            os.environ["NBODY_SYM_DUP_FACTOR"] = "2"
            tb, cb = run(n, pm, vel, False, 0)
            os.environ.pop("NBODY_SYM_DUP_FACTOR")
            before = f"guided strips, table of >= 2 N slots: ipt {cb['i_per_thread']:2d} {tb * 1e6:8.1f} us {n * n * 20 / tb / PEAK * 100:5.1f} %  ->  "
        t0, cfg = run(n, pm, vel, None, 0)                        # the library's default (may be the block kernel)
        line = f"N={n:7d} {name:8s} {before}default {cfg['kernel']:24s} ipt {cfg['i_per_thread']:2d} plan {str(cfg['plan']):6s} {t0 * 1e6:8.1f} us {n * n * 20 / t0 / PEAK * 100:5.1f} %"
        for ipt in (() if DEFAULT_ONLY else (4, 8) if KAHAN else (4, 8, 16)):
            if n < 256 * ipt:
                continue
            tg, cg = run(n, pm, vel, False, ipt)
            te, ce = run(n, pm, vel, True, ipt)
            line += f" | ipt {ipt:2d} guided {tg * 1e6:8.1f} us {n * n * 20 / tg / PEAK * 100:5.1f} % ({cg['blocks']} items) even {te * 1e6:8.1f} us {n * n * 20 / te / PEAK * 100:5.1f} %"
        print(line, flush=True)

#!/usr/bin/env python3
"""Force-kernel sweep on one GPU: tile / i-per-lane / j-split / zero-mode variants, timed with the HIP events
inside the C-ABI (nbody_kernel_time).  Prints one line per variant: ms, pairs/s, % of the fp32 FMA peak.

    python tools/sweep.py --n 1048576 --iters 2 [--quick]
"""
import argparse
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import parallelnbody_amd as nb  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--tiles", default="256")
    ap.add_argument("--ipts", default="1,2,4")
    ap.add_argument("--jsplits", default="0")
    ap.add_argument("--zeros", default="0,1,2")
    ap.add_argument("--precisions", default="f32")
    ap.add_argument("--eps", type=float, default=0.0)
    ap.add_argument("--algos", default="1", help="1 = tiled, 2 = symmetric")
    a = ap.parse_args()
    posm, vel = nb.ic_plummer(a.n, seed=1)
    ints = lambda s: [int(x) for x in s.split(",")]
    print(f"{'algo':>9} {'prec':>9} {'tile':>5} {'ipt':>4} {'jsplit':>6} {'zero':>5} {'blocks':>7} {'ms':>10} {'pairs/s':>12} {'%peak':>7}")
    for algo, prec, tile, ipt, js, zm in itertools.product(ints(a.algos), a.precisions.split(","), ints(a.tiles),
                                                           ints(a.ipts), ints(a.jsplits), ints(a.zeros)):
        try:
            eng = nb.NBodyEngine(a.n, precision=prec, tile=tile, i_per_thread=ipt, j_split=js, zero_mode=zm, eps=a.eps,
                                 time_kernels=True, algorithm=algo)
        except nb.NBodyError:
            continue                     # a combination the library refuses (e.g. 8 bodies per lane, one-sided kernel)
        with eng as e:
            e.set_state(posm, vel)
            # warm-up: ~0.25 s of passes first — the clock needs sustained load to settle (N = 65536: 0.955 ms per
            # pass in a 5-pass run, 0.796 ms in a 500-pass run), then time at least as much again
            e.kernel_time_reset()
            e.compute_forces()
            e.synchronize()
            t1, _ = e.kernel_time(nb.KERNEL_FORCES)
            reps = min(5000, max(1, int(250.0 / max(t1, 1e-3))))
            for _ in range(reps):
                e.compute_forces()
            e.synchronize()
            e.kernel_time_reset()
            for _ in range(max(a.iters, reps)):
                e.compute_forces()
            ms, n = e.kernel_time(nb.KERNEL_FORCES)
            cfg = e.launch_config()
        ms /= n
        pps = float(a.n) ** 2 / (ms * 1e-3)
        peak = 157.3e12 if prec != "f64" else 78.6e12
        print(f"{cfg['algorithm']:>9} {prec:>9} {cfg['tile']:>5} {cfg['i_per_thread']:>4} {cfg['j_split']:>6} {zm:>5} {cfg['blocks']:>7} "
              f"{ms:>10.3f} {pps:>12.4e} {100 * pps * 20 / peak:>7.2f}", flush=True)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — the reference's headline workload on MI355X: O(N^2) all-pairs gravity + kick-drift.

A "step" is one Tick body (OctreeSearch.cpp:25-31) over all N bodies: force pass + update.  The workload is
BASELINE.json's metric configuration, N = 2^20 bodies, fp32, reference-compatible arithmetic (G = 1e4, no
softening, d == 0 pairs skipped); it fits one GPU, so every --gpus value runs the same N (strong scaling:
the bodies are range-partitioned over the ranks with one RCCL all-gather of positions per step).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...                  # no launcher: starts the line above as a CHILD process (before any GPU
                                                  # call of its own) and relays its one JSON line and exit code
    python bench.py --gpus N --host single ...    # ONE process, one thread, N GPUs: nbody_create_multi (the reference's
                                                  # game-thread model, OctreeSearch.cpp:21-34; RCCL grouped send/recv +
                                                  # in-place all-gather between the devices)

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (RCCL / device-tensor sharing fail with hipIpcGetMemHandle: invalid
# argument otherwise); the boxes export it already — kept here so that a bare environment behaves the same
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3      # MI355X vector fp32 peak, MI355X_MICROARCH.md "Chip-level parameters"
FLOP_PER_PAIR = 20            # 3 sub + 6 (dot) + 4 (rsqrt cubed) + 1 (mass) + 6 (3 FMA)
FLOP_PER_EVAL_SYM = 25        # what the symmetric kernel executes per UNORDERED pair: 3 sub + 5 (r^2) + 1 rsq + 2 (cube)
                              # + 2 (the two mass factors) + 12 (six FMAs); matches the PMC count (profiles/r02_pmc_*.txt)
FLOP_PER_EVAL_SYM_EQUAL = 23  # its equal-mass form (all bodies of one mass, as in a Plummer sphere): no mass factors in the loop
# HBM bytes per force launch, measured by PMC passes (tools/profile_kernel.sh: FETCH_SIZE x2 — the gfx950 correction of
# MI355X_MICROARCH.md — + WRITE_SIZE, separate passes), keyed by (algorithm, n, gpus, bodies per lane, precision, equal-mass form) with the
# file the number comes from.  Configurations that were not profiled report null.
TRAFFIC_BYTES_PER_LAUNCH = {
    ("tiled", 1 << 20, 1, 4, "f32", False): (2 * 173606 * 1024 + 262144 * 1024, "profiles/r01_pmc_forces_tile_kernel.txt"),
    ("symmetric", 1 << 20, 1, 16, "f32", True): (5394247488, "profiles/r05_pmc_forces_sym_kernel_equal_mass_n1048576_ipt16.txt"),   # re-taken on the round-5 build (r03: 5408130091)
    ("symmetric", 1 << 20, 1, 16, "f32", False): (5396041643, "profiles/r02_pmc_forces_sym_kernel_n1048576_ipt16.txt"),
    ("symmetric", 1 << 16, 1, 16, "f32", True): (182662522, "profiles/r02_pmc_forces_sym_kernel_equal_mass_n65536_ipt16.txt"),
    ("symmetric", 1 << 16, 1, 16, "f32", False): (301033491, "profiles/r02_pmc_forces_sym_kernel_n65536_ipt16.txt"),
    ("symmetric", 1 << 21, 1, 8, "f32_kahan", True): (33727357803, "profiles/r05_pmc_forces_sym_kernel_kahan_equal_mass_n2097152_ipt8.txt"),   # re-taken on the round-5 build (r02: 31367423019, before the strips' cap of 1024 subtiles)
    ("symmetric", 1 << 21, 1, 8, "f32_kahan", False): (31471647019, "profiles/r02_pmc_forces_sym_kernel_kahan_n2097152_ipt8.txt"),
    ("symmetric", 1 << 18, 1, 4, "f64", True): (2361710743, "profiles/r05_pmc_forces_sym_f64_kernel_equal_mass_n262144_ipt4.txt"),   # re-taken on the round-5 build (r02: 2362121984)
    ("symmetric", 1 << 18, 1, 4, "f64", False): (2361977562, "profiles/r02_pmc_forces_sym_f64_kernel_n262144_ipt4.txt"),
}


# the same under the even-share plan (csrc/sym_plan.h; plain fp32 systems of 16385 <= N < 139264): (n, bodies per lane, precision, equal-mass form)
TRAFFIC_BYTES_PER_LAUNCH_EVEN = {
    (1 << 16, 16, "f32", True): (51966002, "profiles/r05_pmc_forces_sym_kernel_equal_mass_n65536_ipt16_even_shares.txt"),   # guided strips: 182662522
}


def sample_bodies(i_begin, i_count, super_tile, i_per_lane, seed=7):
    """Owned bodies on which the benched force pass is compared with an fp64 direct sum: 24 random ones, the first and
    last body of the first / middle / last owned super tile, and both sides of an i-set boundary inside a super tile
    (where the diagonal workgroup switches from one-sided to symmetric tiles)."""
    import numpy as np
    rng = np.random.default_rng(seed)
    pick = set(int(i) for i in i_begin + rng.choice(i_count, min(24, i_count), replace=False))
    S = super_tile if super_tile else max(1, i_count // 4)
    tiles = max(1, i_count // S)
    for t in sorted({0, tiles // 2, tiles - 1}):
        lo = i_begin + t * S
        hi = min(lo + S, i_begin + i_count) - 1
        pick.update((lo, hi))
        bi = 256 * max(1, i_per_lane)
        if lo + bi <= hi:
            pick.update((lo + bi - 1, lo + bi, lo + bi // 2))
        if lo + 256 <= hi:
            pick.update((lo + 255, lo + 256))                    # a 256-body j-tile boundary
    return sorted(pick)


def sampled_force_error(posm_all, acc_own, i_begin, bodies, G, eps):
    """max over `bodies` of |a_gpu - a_ref| / |a_ref| with a_ref the pair law of OctreeSearch.h:101-104 summed over all j in
    fp64 by numpy (d == 0 pairs skipped, eps^2 added to d^2 when softened).  A checker run after the timed region."""
    import numpy as np
    p = np.asarray(posm_all, np.float64)
    worst = 0.0
    for i in bodies:
        d = p[:, :3] - p[i, :3]
        r2 = (d * d).sum(1) + eps * eps
        with np.errstate(divide="ignore", invalid="ignore"):
            s = np.where(r2 > 0.0, G * p[:, 3] / (r2 * np.sqrt(r2)), 0.0)
        ref = (s[:, None] * d).sum(0)
        got = np.asarray(acc_own[i - i_begin, :3], np.float64)
        worst = max(worst, float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-300)))
    return worst


def cpu_baseline(posm, target_seconds):
    """CPU numbers reported next to the GPU's (never the thing measured or shipped), on the host cores of this box:
      port : the oracle's direct sum in the reference's own arithmetic (fp32 Dist, double pow per pair — OctreeSearch.h:
             101-104), OpenMP over i, on a bounded i-slice of the SAME workload;
      simd : SURVEY 8(d)(A)'s baseline — the build's own plain fp32 direct sum, OpenMP over i and SIMD over j
             (oracle/cpu_baseline.c), at N = 1024 (configs[0]) and N = 65536 (configs[1]), whole passes, best of several.
    `value` is the port's (the reference's arithmetic on the benched workload)."""
    import numpy as np
    from oracle import oracle as O
    import parallelnbody_amd as nb
    O.build()
    n = posm.shape[0]
    pos = np.ascontiguousarray(posm[:, :3]); mass = np.ascontiguousarray(posm[:, 3])
    cores = min(O.max_threads(), os.cpu_count() or 1)
    probe = min(n, 4 * cores)
    t0 = time.perf_counter()
    O.forces_direct_f32(pos, mass, i0=0, i1=probe, nthreads=cores)
    t_probe = time.perf_counter() - t0
    rate = probe * n / max(t_probe, 1e-9)
    ni = int(min(n, max(probe, target_seconds * rate / n)))
    ni = max(cores, ni // cores * cores)
    t0 = time.perf_counter()
    O.forces_direct_f32(pos, mass, i0=0, i1=ni, nthreads=cores)
    dt = time.perf_counter() - t0
    port = {"value": ni * n / dt, "unit": "pair-interactions/s", "cores": cores, "kind": "port",
            "sample": f"oracle direct sum (reference fp32/double-pow arithmetic), bodies 0..{ni - 1} vs all {n} "
                      f"of the same Plummer input, {dt:.1f} s, OpenMP over i"}
    simd = {}
    for m in (1024, 65536):
        pm, _ = nb.ic_plummer(m, total_mass=1000.0, scale_radius=100.0, G=1.0e4, seed=20261003)
        p3 = np.ascontiguousarray(pm[:, :3]); ms = np.ascontiguousarray(pm[:, 3])
        O.forces_simd_f32(p3, ms, nthreads=cores)                       # builds the library, spins the threads up
        best, spent, reps = 1e30, 0.0, 0
        while spent < 2.0 and reps < 200:
            t0 = time.perf_counter()
            O.forces_simd_f32(p3, ms, nthreads=cores)
            d = time.perf_counter() - t0
            best = min(best, d); spent += d; reps += 1
        simd[f"n{m}"] = {"value": m * float(m) / best, "unit": "pair-interactions/s", "cores": cores,
                         "sample": f"whole force pass at N={m}, best of {reps}, {best * 1e3:.3f} ms"}
    out = dict(port)
    out["port"] = port
    out["simd"] = {"kind": "build's own fp32 direct sum, OpenMP over i, omp simd over j, 1/sqrtf cubed (oracle/cpu_baseline.c, "
                           "-O3 -march=native -ffast-math); not the reference's arithmetic", **simd}
    return out


PEAK_FP64_TFLOPS = PEAK_FP32_TFLOPS / 2    # 78.6: MI355X vector fp64 peak (MI355X_MICROARCH.md)


def baseline_config_row(nb, n, precision, eps, steps, warmup, settle_seconds, dt=0.01):
    """One more of BASELINE.json's configs on this GPU, after the headline's timed region: a seeded Plummer sphere of n bodies,
    `steps` whole steps timed from the host (state resident, force pass + kick-drift), the force pass's own time from HIP events
    on the launch stream (nbody_kernel_time), and the sampled parity check of the benched instantiation at the benched size."""
    import numpy as np
    f64 = precision == "f64"
    dtype = np.float64 if f64 else np.float32
    posm, vel = nb.ic_plummer(n, total_mass=1000.0, scale_radius=100.0, G=1.0e4, seed=20261003)
    if f64:
        posm, vel = posm.astype(np.float64), vel.astype(np.float64)
    with nb.NBodyEngine(n, precision=precision, eps=eps, time_kernels=True) as e:
        e.set_state(posm, vel)
        cfg = e.launch_config()
        t0 = time.perf_counter()
        e.compute_forces(); e.synchronize()
        one = max(time.perf_counter() - t0, 1e-5)
        for _ in range(int(min(2000, max(1, settle_seconds / one)))):
            e.compute_forces()
        e.step(dt, warmup); e.synchronize()
        e.kernel_time_reset()
        t0 = time.perf_counter()
        e.step(dt, steps); e.synchronize()
        elapsed = time.perf_counter() - t0
        f_ms, f_n = e.kernel_time(nb.KERNEL_FORCES)
        clk = clock_fields(e, f_ms / max(f_n, 1) * 1e-3, float(n) * float(n))
        equal_mass = e.equal_mass_form()
        p_end = e.state(dtype)[0]
        e.compute_forces()
        acc = e.accelerations(dtype)
    bodies = sample_bodies(0, n, cfg["super_tile"], cfg["i_per_thread"])
    err = sampled_force_error(p_end, acc, 0, bodies, 1.0e4, eps)
    tol = {"f32": 2e-5, "f32_kahan": 2e-6 if eps > 0 else 2e-5, "f64": 1e-12}[precision]
    finite = bool(np.isfinite(p_end).all())
    if not (finite and err < tol):
        raise SystemExit(f"bench.py: configs row N={n} {precision}: the benched force pass disagrees with the fp64 direct sum on "
                         f"sampled bodies: max rel err {err:.3e} >= {tol:.1e} (finite={finite})")
    pairs = float(n) * float(n)
    peak = PEAK_FP64_TFLOPS if f64 else PEAK_FP32_TFLOPS
    avg_launch_s = f_ms / max(f_n, 1) * 1e-3
    achieved = pairs * FLOP_PER_PAIR / avg_launch_s * 1e-12
    traffic = TRAFFIC_BYTES_PER_LAUNCH.get((cfg["algorithm"], n, 1, cfg["i_per_thread"], precision, equal_mass), (None, None))
    if cfg.get("plan") == "even":          # the table's figures for mid sizes are the guided plan's; the even-share plan has its own
        traffic = TRAFFIC_BYTES_PER_LAUNCH_EVEN.get((n, cfg["i_per_thread"], precision, equal_mass), (None, None))
    return {"workload": f"N={n} all-pairs {precision}, seeded Plummer sphere (equal masses), G=1e4, eps={eps}, dt={dt}",
            "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3, "value": pairs * steps / elapsed,
            "unit": "pair-interactions/s", "dtype": "f64" if f64 else "f32", "accumulate": precision,
            "algorithm": cfg["algorithm"], "kernel": cfg["kernel"], "i_per_lane": cfg["i_per_thread"], "lds_tile_bodies": cfg["tile"],
            "plan": cfg.get("plan"), "workgroups": cfg["blocks"], "equal_mass_form": equal_mass,
            "roofline": {"bound": "valu_fp64" if f64 else "valu_fp32", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "avg_launch_ms": avg_launch_s * 1e3, "launches": f_n,
                         "whole_step_frac": pairs * FLOP_PER_PAIR / (elapsed / steps) * 1e-12 / peak,
                         "traffic": traffic[0], "traffic_source": traffic[1], **clk},
            "max_rel_err_sampled": err, "bodies_sampled": len(bodies), "rel_err_tolerance": tol, "finite": finite}


def barnes_hut_row(nb, n, frames, warmup, parity_frames, scene="box", theta=1.0, dt=0.01, size=1000.0):
    """The reference's SHIPPED algorithm (OctreeSearch.cpp:74-89 at Theta = 1.0, .cpp:85) on its shipped kind of scene
    (CreateSpacePoints(N, 1000), .cpp:58-72): whole frames — ComputeCubeSize, the tree rooted at the previous tree's CoM,
    ComputeMass, every body's walk, kick-drift — queued by nbody_step with one host wait per call.  Before the timing, on a
    context of its own: `parity_frames` whole Ticks compared in EVERY BYTE of the FParticle records, Size and the root centre
    with the oracle's restatement of OctreeSearch.h:60-108 / .cpp:25-31 (cube correctly rounded: pow_mode 3) — whose time on this
    box's host cores is the row's cpu_baseline."""
    import ctypes
    import numpy as np
    from oracle import oracle as O
    if scene == "box":
        posm, vel = nb.ic_reference_box(n, size, seed=1)
        what = f"reference box scene (CreateSpacePoints(N, {size:g}), OctreeSearch.cpp:58-72)"
    else:                                                    # BASELINE.json's inputs: a seeded Plummer sphere (the bench's headline scene)
        posm, vel = nb.ic_plummer(n, total_mass=1000.0, scale_radius=100.0, G=1.0e4, seed=20261003)
        what = "seeded Plummer sphere (BASELINE's input kind; equal masses)"
    q = np.zeros(n, nb.PARTICLE_DTYPE)
    q["Mass"] = posm[:, 3]; q["Position"] = posm[:, :3]; q["Velocity"] = vel[:, :3]
    com, sz, cpu_s = None, 0.0, []
    with nb.NBodyEngine(n, theta=theta) as e:
        e.set_state(posm, vel)
        for frame in range(parity_frames):
            size_dev, out = e.tick(dt)
            t0 = time.perf_counter()
            com, sz = O.tick_aos_f32(q, dt, theta=theta, root_com=com, size=sz, pow_mode=3)
            cpu_s.append(time.perf_counter() - t0)
            same = size_dev == sz and out.tobytes() == q.tobytes() and bool(np.all(e.bh_stats()["root_com"] == com))
            if not same:
                raise SystemExit(f"bench.py: Barnes-Hut row N={n}: frame {frame} differs from the oracle's (Size {size_dev} vs {sz}, "
                                 f"{int((out['Acceleration'] != q['Acceleration']).any(axis=1).sum())} bodies' accelerations differ)")
        st = e.bh_stats()
    with nb.NBodyEngine(n, theta=theta) as e:
        e.set_state(posm, vel)
        e.step(dt, warmup); e.synchronize()
        t0 = time.perf_counter()
        e.step(dt, frames); e.synchronize()
        elapsed = time.perf_counter() - t0
        done = e.steps_done()
        finite = bool(np.isfinite(e.state()[0]).all())
        warm, again = ctypes.c_longlong(), ctypes.c_longlong()
        if n <= 4096 or e._L.nbody_debug_bh_sort_counts(e._h, ctypes.byref(warm), ctypes.byref(again)) != 0:
            warm, again = None, None
        else:
            warm, again = warm.value, again.value
    if done != warmup + frames or not finite:
        raise SystemExit(f"bench.py: Barnes-Hut row N={n}: {done} of {warmup + frames} frames built, finite={finite}")
    cores = min(O.max_threads(), os.cpu_count() or 1)
    return {"workload": f"N={n} {what}, theta={theta}, dt={dt}: whole frames (Size, tree, ComputeMass, walks, kick-drift)",
            "frames": frames, "warmup": warmup, "us_per_frame": elapsed / frames * 1e6, "frames_per_s": frames / elapsed,
            "bodies_per_s": n * frames / elapsed, "tree_nodes": st["nodes"], "tree_levels": st["levels"],
            "frames_sorted_from_the_previous_order": warm, "times_frames_were_queued_again": again,
            "parity": f"every byte of {parity_frames} frame(s) (FParticle records, Size, root centre) equal to the oracle's",
            "cpu_baseline": {"value": min(cpu_s) * 1e6, "unit": "us/frame", "kind": "port", "cores": cores,
                             "sample": f"oracle Tick (Octree::Add and ComputeMass on one thread, the walks OpenMP over bodies) of "
                                       f"the same scene, best of {len(cpu_s)} frame(s)"}}


def mid_size_row(nb, n, seconds=0.25, dt=0.002):
    """A mid-size system in the kernel's GENERAL form — the bodies of a seeded Plummer sphere with every mass scaled by its own
    factor in [0.5, 1.5), what the reference's own scene (masses 1 ... 5000, OctreeSearch.cpp:66) runs — stepped as a host steps it:
    the library's defaults, no per-kernel events, `nbody_step` in one call, best of three timed stretches.  `frac` is the WHOLE
    step (force pass + update) against the fp32 vector peak at the 20-flop convention; the benched pass is compared with the fp64
    direct sum on sampled bodies first."""
    import numpy as np
    posm, vel = nb.ic_plummer(n, total_mass=1000.0, scale_radius=100.0, G=1.0e4, seed=20261003)
    posm[:, 3] *= np.random.default_rng(n).uniform(0.5, 1.5, n).astype(np.float32)
    with nb.NBodyEngine(n) as e:
        cfg = e.launch_config()
        e.set_state(posm, vel)
        e.compute_forces()
        bodies = sample_bodies(0, n, cfg["super_tile"], cfg["i_per_thread"])
        err = sampled_force_error(posm, e.accelerations(), 0, bodies, 1.0e4, 0.0)
        if not (err < 2e-5):
            raise SystemExit(f"bench.py: mid_sizes row N={n}: the force pass disagrees with the fp64 direct sum: max rel err {err:.3e}")
        e.step(dt, 200); e.synchronize()
        k = max(100, int(seconds / (n * float(n) / 6e12 + 1e-5)))
        best = float("inf")
        for _ in range(3):
            t0 = time.perf_counter(); e.step(dt, k); e.synchronize()
            best = min(best, (time.perf_counter() - t0) / k)
        general = not e.equal_mass_form()
    return {"workload": f"N={n} all-pairs f32, Plummer positions, distinct masses (the general form), G=1e4, eps=0, dt={dt}",
            "us_per_step": best * 1e6, "value": n * float(n) / best, "unit": "pair-interactions/s", "steps_timed": 3 * k,
            "whole_step_frac": n * float(n) * FLOP_PER_PAIR / best * 1e-12 / PEAK_FP32_TFLOPS, "general_form": general,
            "kernel": cfg["kernel"], "plan": cfg.get("plan"), "i_per_lane": cfg["i_per_thread"], "workgroups": cfg["blocks"],
            "max_rel_err_sampled": err, "bodies_sampled": len(bodies), "rel_err_tolerance": 2e-5}


def extra_rows(args, nb):
    """BASELINE.json's other single-GPU configs and the theta > 0 path, measured in the same driver run AFTER the headline's
    timed region (the headline's fields are untouched): configs[1] N = 65536 fp32, configs[3] N = 262144 fp64, configs[4]
    N = 2097152 softened Kahan — each >= 3 timed whole steps with its own sampled parity check and roofline fraction against its
    own peak — and Barnes-Hut frames at the reference's shipped opening angle."""
    s = args.settle_seconds
    configs = {
        "n65536_f32": baseline_config_row(nb, 1 << 16, "f32", 0.0, steps=100, warmup=5, settle_seconds=s),
        "n262144_f64": baseline_config_row(nb, 1 << 18, "f64", 0.0, steps=5, warmup=1, settle_seconds=s),
        "n2097152_kahan_softened": baseline_config_row(nb, 1 << 21, "f32_kahan", 0.5, steps=3, warmup=1, settle_seconds=0.0),
    }
    bh = {
        "n2000": barnes_hut_row(nb, 2000, frames=200, warmup=20, parity_frames=2),
        "n65536": barnes_hut_row(nb, 1 << 16, frames=100, warmup=10, parity_frames=1, scene="plummer"),
        # the reference's own kind of scene at that size: within these frames runaway bodies own Size (1e9 for a box of 1e3) and every
        # other body shares one cell of level 21 — what round 5's whole-key bucket boundaries are for
        "n65536_reference_box_scene": barnes_hut_row(nb, 1 << 16, frames=300, warmup=10, parity_frames=1, scene="box"),
        "n1048576": barnes_hut_row(nb, 1 << 20, frames=50, warmup=5, parity_frames=1, scene="plummer"),
    }
    # mid sizes in the general form (the round-4 review's item 4): whole steps as a host runs them
    mid = {f"n{n}_distinct_masses": mid_size_row(nb, n) for n in (20480, 32768, 65536)}
    return configs, bh, mid


def clock_fields(engine, avg_launch_s, launch_pairs):
    """roofline.clock_ghz / cycles_per_interaction: the shader clock the timed force kernels ran at (read inside the kernels:
    nbody_kernel_clock) and launch time x clock x SIMD lanes / interactions — SIMD cycles per interaction and lane if every SIMD
    of the device was busy for the whole launch.  The force loops are power-limited and boxes hold different clocks under them;
    the cycles do not depend on the box: ~20 in the equal-mass form of the symmetric kernel, ~22 in its general form, ~37 one-sided
    (DESIGN.md 4.1, 4.1b), ~51-55 in the fp64 symmetric kernel (whose power draw moves the clock most from box to box: 24.4 to 28.7 ms
    a step of configs[3] within one afternoon).  None for kernels without the stamps (the generic scalar kernels)."""
    try:
        mhz, cus = engine.kernel_clock()
    except Exception:  # noqa: BLE001
        return {"clock_ghz": None, "cycles_per_interaction": None}
    if not mhz > 0.0:
        return {"clock_ghz": None, "cycles_per_interaction": None}
    return {"clock_ghz": mhz * 1e-3, "compute_units": cus,
            "cycles_per_interaction": avg_launch_s * mhz * 1e6 * (cus * 4 * 64) / launch_pairs}


_RESULT_FD = None


def keep_stdout_for_the_result():
    """The contract: rank 0 prints ONE JSON line.  Libraries under this process write to stdout as well (RCCL prints a
    version banner there when NCCL_DEBUG says so), so file descriptor 1 is pointed at stderr for the run and the result
    goes out through a duplicate of the original stdout."""
    global _RESULT_FD
    if _RESULT_FD is None:
        sys.stdout.flush()
        _RESULT_FD = os.dup(1)
        os.dup2(2, 1)


def print_result(out):
    line = (json.dumps(out) + "\n").encode()
    sys.stdout.flush()
    os.write(1 if _RESULT_FD is None else _RESULT_FD, line)


def _kill_group(proc, grace=5.0):
    """End a child this process started (and only that: its own session / process group, never a pattern): SIGTERM, then
    SIGKILL after `grace` seconds."""
    import signal
    if proc.poll() is not None:
        return
    for sig, wait in ((signal.SIGTERM, grace), (signal.SIGKILL, 30.0)):
        try:
            os.killpg(proc.pid, sig)
        except (ProcessLookupError, PermissionError):
            pass
        t_end = time.monotonic() + wait
        while time.monotonic() < t_end and proc.poll() is None:
            time.sleep(0.05)
        if proc.poll() is not None:
            return


def _die_with_parent():
    """preexec of every child: if the watchdog itself is killed, the kernel ends the child too (PR_SET_PDEATHSIG)."""
    try:
        import ctypes
        import signal
        ctypes.CDLL("libc.so.6", use_errno=True).prctl(1, signal.SIGKILL)
    except Exception:  # noqa: BLE001
        pass


class _Child:
    """A worker process in its own session, stdout collected, stderr relayed line by line (the last lines kept)."""

    def __init__(self, cmd, env, tag):
        import collections
        import subprocess
        import threading
        self.proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, start_new_session=True,
                                     preexec_fn=_die_with_parent)
        self.out, self.tail, self.tag = [], collections.deque(maxlen=12), tag
        self.threads = [threading.Thread(target=self._pump_out, daemon=True), threading.Thread(target=self._pump_err, daemon=True)]
        for t in self.threads:
            t.start()

    def _pump_out(self):
        for raw in self.proc.stdout:
            self.out.append(raw.decode(errors="replace"))

    def _pump_err(self):
        for raw in self.proc.stderr:
            line = raw.decode(errors="replace")
            self.tail.append(line.rstrip())
            sys.stderr.write(line)
            sys.stderr.flush()

    def finish(self):
        for t in self.threads:
            t.join(timeout=5.0)

    def json_line(self):
        for ln in reversed("".join(self.out).splitlines()):
            if ln.startswith("{"):
                try:
                    return json.loads(ln)
                except ValueError:
                    pass
        return None


def _write_atomically(path, text):
    tmp = f"{path}.tmp{os.getpid()}"
    with open(tmp, "w") as f:
        f.write(text)
    os.replace(tmp, path)


def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


FALLBACK_ARGS = ["--algorithm", "tiled", "--no-overlap"]


def rank_watchdog(args, argv):
    """What every rank of `bench.py --gpus N` (N > 1) is before any GPU call: a watchdog that never touches the GPU itself.

    The default multi-rank step (symmetric pass + all-to-all + all-gather under the next pass's own-slice strips) has more
    moving parts than north_star's literal one (one-sided kernel, per-step all-gather, nothing else).  A failure or a hang in
    the first must not cost the whole measurement, and ranks inside a job cannot agree on a failure without risking
    mismatched collectives — so each attempt is a FRESH set of worker processes:

      attempt 1: this command line as given, as a child process with a time limit (--child-timeout);
      attempt 2: only if attempt 1 failed on any rank or ran out of time (and --algorithm was left on auto): the children's
                 process groups are killed and a new child is started with `--algorithm tiled --no-overlap`, on a rendezvous
                 port of its own.  Its line carries config.fallback: what failed, where, and the last lines of stderr.

    The watchdogs of one job agree through small files in a directory only they know (rank 0 decides: `done`, or `retry` with
    the next attempt's port); nothing here is a collective and nothing re-execs a process that has used the GPU.  Exit code 0
    iff a contract-complete line went out."""
    import shutil
    import signal
    import tempfile
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ["WORLD_SIZE"])
    run_dir = os.environ.get("NBODY_BENCH_RUN_DIR") or os.path.join(
        tempfile.gettempdir(), "nbody_bench_%s_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none"),
                                                          os.getppid() if "TORCHELASTIC_RUN_ID" in os.environ else "x"))
    os.makedirs(run_dir, exist_ok=True)
    for name in os.listdir(run_dir):                      # leftovers of an earlier job under the same name: this rank's own files only
        if name.endswith(f".rank{rank}") or (rank == 0 and (".verdict" in name or name == "job")):
            try:
                os.unlink(os.path.join(run_dir, name))
            except OSError:
                pass
    # Files of ANOTHER job that used the same directory name (same port, no launcher id) must never be taken for this job's: rank 0
    # names the job (start time + pid), every status and verdict file carries that name, and files without it are ignored.
    # (The directory is node-local: a job over several nodes sets NBODY_BENCH_RUN_DIR to a directory on a shared file system.)
    job_path = os.path.join(run_dir, "job")
    t_start = time.time_ns()
    if rank == 0:
        job = "%d-%d" % (t_start, os.getpid())
        _write_atomically(job_path, job)
    else:
        # a name left by an earlier job is older than this process by more than the ranks of one launch are apart (60 s allowed):
        # not ours — rank 0 replaces it as it starts
        def fresh(name):
            try:
                return name is not None and int(name.split("-")[0]) >= t_start - 60 * 10 ** 9
            except ValueError:
                return False
        t_end, job = time.monotonic() + 120.0, None
        while time.monotonic() < t_end:
            job = _read(job_path)
            if fresh(job):
                break
            job = None
            time.sleep(0.05)
        if job is None:
            print(f"[bench watchdog rank {rank}] rank 0's watchdog never named the job in {run_dir}", file=sys.stderr, flush=True)
            return 1

    def read_job_file(path):
        """the JSON / text of a status or verdict file of THIS job, else None"""
        text = _read(path)
        if text is None or not text.startswith(job + "\n"):
            return None
        return text[len(job) + 1:]
    attempts = [("as given", [])]
    if args.algorithm == "auto" and not args.no_fallback:
        attempts.append(("all-gather-only step (one-sided kernel, every collective in stream order)", FALLBACK_ARGS))
    worker_cmd = os.environ.get("NBODY_BENCH_WORKER_CMD")     # tests/test_bench_watchdog.py: a stand-in worker, no GPU
    base = (worker_cmd.split() if worker_cmd else [sys.executable, os.path.abspath(__file__)]) + list(argv) + ["--worker"]
    live = {"child": None}

    def on_signal(signum, _frame):
        if live["child"] is not None:
            _kill_group(live["child"].proc, grace=2.0)
        sys.exit(128 + signum)
    for sg in (signal.SIGTERM, signal.SIGINT):
        signal.signal(sg, on_signal)

    say = lambda msg: print(f"[bench watchdog rank {rank}] {msg}", file=sys.stderr, flush=True)
    failures, port, line = [], None, None
    for k, (what, extra) in enumerate(attempts, start=1):
        env = dict(os.environ, NBODY_BENCH_ATTEMPT=str(k))
        if k > 1:                                         # a rendezvous of its own: rank 0's worker hosts the store on the new port
            env["MASTER_PORT"] = str(port)
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        child = _Child(base + extra, env, f"attempt {k}")
        live["child"] = child
        t_limit = time.monotonic() + args.child_timeout
        status_path = os.path.join(run_dir, f"a{k}.rank{rank}")
        verdict_path = os.path.join(run_dir, f"a{k}.verdict")
        status, verdict, t_status = None, None, 0.0
        while verdict is None:
            rc = child.proc.poll()
            if status is None and rc is not None:
                child.finish()
                status = "ok" if rc == 0 else f"fail worker exit code {rc}"
            elif status is None and time.monotonic() > t_limit:
                say(f"attempt {k} ({what}): no result after {args.child_timeout:.0f} s: killing the worker's process group")
                _kill_group(child.proc)
                child.finish()
                status = f"fail time limit of {args.child_timeout:.0f} s"
            if status is not None and not os.path.exists(status_path):
                t_status = time.time()
                _write_atomically(status_path, job + "\n" + json.dumps({"status": status, "t": t_status, "stderr_tail": list(child.tail)}))
            if rank == 0:
                peers = {q: json.loads(read_job_file(os.path.join(run_dir, f"a{k}.rank{q}")) or "null") for q in range(world)}
                bad = {q: s for q, s in peers.items() if s and s["status"] != "ok"}
                # a line that is already out counts even if some rank then stumbles in its teardown
                got = child.json_line() if status == "ok" or (status is None and bad) else None
                if got is not None:
                    line, verdict = got, "done"
                elif status is not None and status != "ok" and len(peers) - list(peers.values()).count(None) < world and time.time() < t_status + 3.0:
                    pass          # the own worker failed: a moment for the others to say when theirs did (the first failure is the cause)
                elif status is not None or bad:
                    # the own worker failed, or another rank's did (this one may sit in a collective the other never enters)
                    if status is None:
                        say(f"attempt {k}: rank(s) {sorted(bad)} failed: killing the own worker")
                        _kill_group(child.proc)
                        child.finish()
                        bad[0] = {"status": "fail killed: another rank failed first", "t": time.time(), "stderr_tail": list(child.tail)}
                    elif status == "ok":
                        bad[0] = {"status": "fail no JSON line on stdout", "t": time.time(), "stderr_tail": list(child.tail)}
                    elif 0 not in bad:
                        bad[0] = {"status": status, "t": t_status, "stderr_tail": list(child.tail)}
                    first = min(bad, key=lambda q: (bad[q]["status"].startswith("fail killed"), bad[q].get("t", 0.0), q))
                    order = sorted(bad, key=lambda q: (bad[q]["status"].startswith("fail killed"), bad[q].get("t", 0.0), q))
                    failures.append({"attempt": k, "ran": what, "rank": first, "reason": bad[first]["status"][5:],
                                     "ranks_failed": sorted(bad), "stderr_tail": bad[first]["stderr_tail"],
                                     "others": [{"rank": q, "reason": bad[q]["status"][5:], "stderr_tail": bad[q]["stderr_tail"][-4:]}
                                                for q in order[1:4]]})
                    port = _free_port()
                    verdict = f"retry {port}" if k < len(attempts) else "give up"
                if verdict is not None:
                    _write_atomically(verdict_path, job + "\n" + verdict)
            else:
                verdict = read_job_file(verdict_path)
                if verdict is None and time.monotonic() > t_limit + 90.0:
                    verdict = "give up"               # rank 0's watchdog is gone
            if verdict is None:
                time.sleep(0.1)
        _kill_group(child.proc)                           # no-op when the worker has left by itself
        child.finish()
        live["child"] = None
        if verdict == "done":
            if rank == 0:
                if failures:
                    line.setdefault("config", {})["fallback"] = {
                        "ran": what, "instead_of": failures[0]["ran"] + " (the default multi-rank step: symmetric pass + all-to-all)",
                        "because": f"rank {failures[0]['rank']}: {failures[0]['reason']}", "ranks_failed": failures[0]["ranks_failed"],
                        "stderr_tail": failures[0]["stderr_tail"], "other_failures": failures[0]["others"]}
                print_result(line)
                t_end = time.monotonic() + 10.0           # the other watchdogs say when they have read the verdict
                while time.monotonic() < t_end and not all(os.path.exists(os.path.join(run_dir, f"a{k}.seen.rank{q}")) for q in range(1, world)):
                    time.sleep(0.05)
                shutil.rmtree(run_dir, ignore_errors=True)
            else:
                _write_atomically(os.path.join(run_dir, f"a{k}.seen.rank{rank}"), "")
            return 0
        if verdict.startswith("retry"):
            port = int(verdict.split()[1])
            say(f"attempt {k} ({what}) failed; starting a fresh worker: {attempts[k][0]}")
            continue
        say(f"attempt {k} ({what}) failed and there is nothing left to try")
        break
    if rank == 0:                                         # every way out takes the directory along (the others have read "give up" or timed out)
        time.sleep(1.0)
        shutil.rmtree(run_dir, ignore_errors=True)
    return 1


def relaunch_under_torchrun(n_gpus, limit_s):
    """`python bench.py --gpus N` with no launcher around it: run the very same command line under
    `python -m torch.distributed.run` (one rank per GPU) as a CHILD process — started before this process has made any GPU
    call, never an exec after one — relay what it prints and leave with its exit code.  Every rank it starts is a
    rank_watchdog; `limit_s` bounds the whole job from out here as well (a launcher that never returns)."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] no launcher around --gpus %d: starting %s" % (n_gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, start_new_session=True, preexec_fn=_die_with_parent)
    try:
        return proc.wait(timeout=limit_s)
    except subprocess.TimeoutExpired:
        print(f"[bench] the launcher has not returned after {limit_s:.0f} s: killing its process group", file=sys.stderr, flush=True)
        _kill_group(proc)
        return 1
    except KeyboardInterrupt:
        _kill_group(proc)
        raise


def run_single_host(args):
    """--host single: ONE process, one caller thread, args.gpus devices behind one context (nbody_create_multi, csrc/multi.hip)
    — how a UE4 host, whose only caller is the game thread (OctreeSearch.cpp:21-34), would use several GPUs.  Per step the
    library queues, on every device's stream and without waiting for the host: force pass, the symmetric algorithm's
    exchange as grouped ncclSend/ncclRecv, kick-drift, one grouped in-place ncclAllGather of the positions.  Same metric,
    same workload, same parity check as the one-process-per-GPU line."""
    import numpy as np
    import torch
    import parallelnbody_amd as nb
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: parallelnbody_amd has no CPU path")
    if nb.device_count() < args.gpus:
        raise SystemExit(f"--host single --gpus {args.gpus}: only {nb.device_count()} device(s) visible")
    n, g = args.n, args.gpus
    posm, vel = nb.ic_plummer(n, total_mass=1000.0, scale_radius=100.0, G=1.0e4, seed=20261003)
    if args.precision == "f64":
        posm, vel = posm.astype(np.float64), vel.astype(np.float64)

    def fence(e):
        e.synchronize()                              # every device's stream
        for d in range(g):
            torch.cuda.synchronize(d)

    with nb.NBodyEngine(n, devices=list(range(g)), precision=args.precision, eps=args.eps, tile=args.tile,
                        i_per_thread=args.ipt, j_split=args.jsplit, time_kernels=True,
                        algorithm={"auto": 0, "tiled": 1, "symmetric": 2}[args.algorithm],
                        zero_mode={"exact": 0, "floor": 2}[args.zero_mode]) as e:
        e.set_state(posm, vel)
        cfg = e.launch_config()                      # per device
        t0 = time.perf_counter()                     # untimed force passes: the clock needs sustained load to settle
        e.compute_forces(); fence(e)
        one = max(time.perf_counter() - t0, 1e-5)
        for _ in range(int(min(2000, max(1, args.settle_seconds / one)))):
            e.compute_forces()
        e.step(args.dt, args.warmup)
        fence(e)
        e.kernel_time_reset()
        t0 = time.perf_counter()
        e.step(args.dt, args.steps)
        fence(e)
        elapsed = time.perf_counter() - t0
        f_ms, f_n = e.kernel_time(nb.KERNEL_FORCES)   # the slowest device's total
        u_ms, u_n = e.kernel_time(nb.KERNEL_UPDATE)
        clk = clock_fields(e, f_ms / max(f_n, 1) * 1e-3, float(n // g) * float(n))
        equal_mass = e.equal_mass_form()
        p_end = e.state(np.float64 if args.precision == "f64" else np.float32)[0]
        finite = bool(np.isfinite(p_end).all())
        e.compute_forces()                           # parity of the benched instantiation, after the timed region
        acc = e.accelerations(np.float64 if args.precision == "f64" else np.float32)
    slice_n = n // g
    bodies = sorted({b for k in range(g) for b in sample_bodies(k * slice_n, slice_n, cfg["super_tile"], cfg["i_per_thread"], seed=7 + k)})
    err = sampled_force_error(p_end, acc, 0, bodies, 1.0e4, args.eps)
    tol = {"f32": 2e-5, "f32_kahan": 2e-6 if args.eps > 0 else 2e-5, "f64": 1e-12}[args.precision]
    if not (finite and err < tol):
        raise SystemExit(f"bench.py: the benched force pass disagrees with the fp64 direct sum on sampled bodies: "
                         f"max rel err {err:.3e} >= {tol:.1e} (finite={finite})")
    pairs_per_step = float(n) * float(n)
    launch_pairs = float(slice_n) * float(n)
    avg_launch_s = f_ms / max(f_n, 1) * 1e-3
    peak = PEAK_FP32_TFLOPS if args.precision != "f64" else PEAK_FP32_TFLOPS / 2
    achieved = launch_pairs * FLOP_PER_PAIR / avg_launch_s * 1e-12
    flop_eval = FLOP_PER_EVAL_SYM_EQUAL if equal_mass else FLOP_PER_EVAL_SYM
    executed = (launch_pairs / 2 * flop_eval if cfg["algorithm"] == "symmetric" else launch_pairs * FLOP_PER_PAIR) / avg_launch_s * 1e-12
    traffic = TRAFFIC_BYTES_PER_LAUNCH.get((cfg["algorithm"], n, g, cfg["i_per_thread"], args.precision, equal_mass), (None, None))
    exchange = cfg["algorithm"] == "symmetric" and g > 1
    out = {
        "metric": "body-pair interactions/s at N=2^20" if n == (1 << 20) else f"body-pair interactions/s at N={n}",
        "value": pairs_per_step * args.steps / elapsed, "unit": "pair-interactions/s", "n_gpus": g, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None,
        "dtype": {"f32": "f32", "f32_kahan": "f32", "f64": "f64"}[args.precision], "data": "synthetic",
        "config": {"workload": f"N={n} all-pairs {args.precision}, seeded Plummer sphere (equal masses), G=1e4, eps={args.eps}, "
                               f"dt={args.dt}, one force pass + kick-drift per step",
                   "host": "single process, one thread (nbody_create_multi)",
                   "parallelism": (f"range-partition x{g}, per step 1 grouped in-place ncclAllGather(posm)"
                                   + (" + grouped ncclSend/ncclRecv(j-side sums)" if exchange else "")
                                   + (" (a one-rank communicator: every collective still runs)" if g == 1 else "")),
                   "algorithm": cfg["algorithm"], "zero_distance": args.zero_mode, "lds_tile_bodies": cfg["tile"],
                   "i_per_lane": cfg["i_per_thread"], "j_split": cfg["j_split"] if cfg["algorithm"] == "tiled" else None,
                   "super_tile_bodies": cfg["super_tile"] or None, "workgroups_per_device": cfg["blocks"],
                   "accumulate": args.precision, "finite": finite, "equal_mass_form": equal_mass,
                   "max_rel_err_sampled": err, "bodies_sampled": len(bodies), "rel_err_tolerance": tol},
        "roofline": {"bound": "valu_fp32" if args.precision != "f64" else "valu_fp64", "achieved": achieved, "peak": peak,
                     "unit": "TFLOP/s", "frac": achieved / peak,
                     "achieved_is": "algorithmic: N_i x N ordered interactions x 20 flop (SURVEY 8d) / the slowest device's launch time",
                     "executed": executed, "executed_frac": executed / peak, **clk,
                     "traffic": traffic[0], "traffic_source": traffic[1], "kernel": cfg["kernel"],
                     "avg_launch_ms": avg_launch_s * 1e3, "launches": f_n, "flop_per_pair": FLOP_PER_PAIR,
                     "pairs_per_launch": launch_pairs, "update_kernel_avg_ms": u_ms / max(u_n, 1)},
    }
    if args.cpu_seconds > 0 and g == 1:
        out["cpu_baseline"] = cpu_baseline(posm.astype(np.float32), args.cpu_seconds)
    print_result(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n", "--bodies", dest="n", type=int, default=1 << 20,
                    help="bodies (default 2^20 = BASELINE metric); use --bodies under torch.distributed.run, whose own parser claims --n")
    ap.add_argument("--precision", default="f32", choices=["f32", "f32_kahan", "f64"])
    ap.add_argument("--eps", type=float, default=0.0)
    ap.add_argument("--dt", type=float, default=0.01)
    ap.add_argument("--tile", type=int, default=0)
    ap.add_argument("--ipt", type=int, default=0)
    ap.add_argument("--jsplit", type=int, default=0)
    ap.add_argument("--algorithm", default="auto", choices=["auto", "tiled", "symmetric"])
    ap.add_argument("--zero-mode", default="exact", choices=["exact", "floor"],
                    help="exact = the reference's d == 0 skip for every distance; floor = ~1e-20 eps^2 floor")
    ap.add_argument("--no-distinct-row", action="store_true",
                    help="skip the second measurement of the same bodies with distinct masses (one GPU, equal-mass workloads)")
    ap.add_argument("--no-tiled-row", action="store_true",
                    help="multi-GPU: skip the extra timing of the one-sided kernel + all-gather-only step (config.all_gather_only_row)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU baseline sample length; 0 = skip")
    ap.add_argument("--no-extra-rows", action="store_true",
                    help="one GPU, default workload: skip BASELINE's other configs (N=65536 fp32, N=262144 fp64, N=2097152 Kahan) "
                         "and the Barnes-Hut frames that follow the headline's timed region (`configs`, `bh` in the line)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="multi-GPU: every collective in the simulation's one stream order (default: the all-gather of a step's "
                         "positions runs on a stream of its own, under the next force pass's own-slice strips; same bits)")
    ap.add_argument("--host", default="ranks", choices=["ranks", "single"],
                    help="ranks = one process per GPU under torch.distributed (default; started as a child process when no "
                         "launcher did); single = one process drives all GPUs through nbody_create_multi")
    ap.add_argument("--settle-seconds", type=float, default=0.3,
                    help="untimed force passes before the warm-up steps (state unchanged): lets the GPU clock settle")
    ap.add_argument("--child-timeout", type=float, default=300.0,
                    help="multi-GPU: seconds a rank's worker process may take before its watchdog kills it and the job falls "
                         "back to the all-gather-only step in fresh workers (rank_watchdog)")
    ap.add_argument("--no-fallback", action="store_true", help="multi-GPU: one attempt only, exit non-zero if it fails")
    ap.add_argument("--worker", action="store_true", help=argparse.SUPPRESS)      # set by rank_watchdog on the processes that do the work
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and args.host != "single" and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args.gpus, 2 * (args.child_timeout + 120.0)))
    if args.gpus > 1 and args.host != "single" and not args.worker:
        sys.exit(rank_watchdog(args, sys.argv[1:]))
    keep_stdout_for_the_result()
    if args.host == "single":
        return run_single_host(args)

    import numpy as np
    import torch
    import parallelnbody_amd as nb
    from parallelnbody_amd.sharded import hip_engine_factory

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: parallelnbody_amd has no CPU path")
    # NBODY_DIST_BACKEND=gloo: rehearsal of this very code path on a box with fewer GPUs than ranks (ranks then share
    # devices and the collectives go over gloo on device tensors; RCCL refuses two ranks on one device)
    backend = os.environ.get("NBODY_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = torch.distributed
    # rehearsals of rank_watchdog (tests/test_two_rank_gpu.py): on the named rank the all-to-all of the symmetric step raises,
    # or never returns.  The all-gather-only step has no all-to-all, so the fallback's workers are unaffected.
    a2a_fail, a2a_hang = os.environ.get("NBODY_REHEARSE_A2A_FAILURE"), os.environ.get("NBODY_REHEARSE_A2A_HANG")
    if str(rank) in (a2a_fail, a2a_hang):
        def rehearsed_all_to_all(*_a, **_kw):
            if a2a_hang == str(rank):
                print(f"[bench rank {rank}] NBODY_REHEARSE_A2A_HANG: this rank never returns from the all-to-all", file=sys.stderr, flush=True)
                time.sleep(1e6)
            raise RuntimeError("NBODY_REHEARSE_A2A_FAILURE: injected failure inside the all-to-all")
        dist.all_to_all_single = rehearsed_all_to_all
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    n = args.n
    posm, vel = nb.ic_plummer(n, total_mass=1000.0, scale_radius=100.0, G=1.0e4, seed=20261003)
    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def factory(*a, **kw):
        # NBODY_REHEARSE_CREATE_FAILURE=<rank>: rehearsal (tests/test_two_rank_gpu.py) of one rank failing to create its
        # symmetric engine; everything else is the product's own factory
        if os.environ.get("NBODY_REHEARSE_CREATE_FAILURE") == str(rank) and kw.get("algorithm", 0) == 0:
            raise RuntimeError("NBODY_REHEARSE_CREATE_FAILURE")
        return hip_engine_factory(*a, **kw)

    def build(algorithm):
        sim = nb.ShardedSimulation(posm, vel, rank=rank, world_size=world, device=f"cuda:{local_rank}", engine_factory=factory,
                                   precision=args.precision, eps=args.eps, tile=args.tile, i_per_thread=args.ipt,
                                   j_split=args.jsplit, time_kernels=True, overlap=not args.no_overlap,
                                   algorithm={"auto": 0, "tiled": 1, "symmetric": 2}[algorithm],
                                   zero_mode={"exact": 0, "floor": 2}[args.zero_mode])
        sim.warm_collectives()
        sim.settle(args.settle_seconds)     # untimed force passes: the clock needs sustained load to settle
        sim.step(args.dt, args.warmup)
        return sim

    # Multi-GPU only: the symmetric algorithm needs equal slices of whole i-sets and room for its partial sums.  Engine
    # creation is the one stage whose outcome the ranks agree on with a matched collective (ShardedSimulation raises
    # EngineCreationFailed on EVERY rank), so that is the one failure answered by a rebuild — loudly, and named in
    # config.fallback — on the one-sided HIP kernel, whose step needs the all-gather alone.  Any later failure ends the
    # job with a non-zero exit code: ranks cannot agree on it without risking mismatched collectives.
    fallback = None
    try:
        sim = build(args.algorithm)
    except nb.EngineCreationFailed as e:
        if args.algorithm != "auto":
            raise
        print(f"[bench rank {rank}] symmetric multi-GPU engines could not be created ({e}); "
              "every rank rebuilds on the one-sided kernel", file=sys.stderr, flush=True)
        sim = build("tiled")
        fallback = f"tiled: symmetric engines could not be created ({e})"
    cfg = sim.engine.launch_config()
    fence()
    sim.engine.kernel_time_reset()
    t0 = time.perf_counter()
    sim.step(args.dt, args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    f_ms, f_n = sim.engine.kernel_time(nb.KERNEL_FORCES)
    u_ms, u_n = sim.engine.kernel_time(nb.KERNEL_UPDATE)
    clk = clock_fields(sim.engine, f_ms / max(f_n, 1) * 1e-3, float(sim.i_count) * float(n))
    pairs_per_step = float(n) * float(n)
    value = pairs_per_step * args.steps / elapsed
    # dominant kernel: the force pass of this rank = i_count x n_total pair interactions per launch (the symmetric
    # kernel evaluates each unordered pair once and credits both bodies: same interaction count, fewer instructions)
    launch_pairs = float(sim.i_count) * float(n)
    avg_launch_s = (f_ms / max(f_n, 1)) * 1e-3
    # slowest and fastest rank's force pass (the step waits for the slowest): where scaling is lost, if it is
    launch_ms_minmax = [avg_launch_s * 1e3, avg_launch_s * 1e3]
    if world > 1:
        t = torch.tensor([-avg_launch_s * 1e3, avg_launch_s * 1e3], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        launch_ms_minmax = [-float(t[0]), float(t[1])]
        g_ = clk["clock_ghz"] or 0.0
        t = torch.tensor([-g_, g_], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        clk["clock_ghz_min_max_over_ranks"] = [-float(t[0]), float(t[1])]
    achieved_tflops = launch_pairs * FLOP_PER_PAIR / avg_launch_s * 1e-12
    # what the hardware executed: the symmetric kernel evaluates each unordered pair once (25 flop) and credits both
    # bodies; the one-sided kernel evaluates every ordered pair (20 flop)
    equal_mass = sim.engine.equal_mass_form()      # which form of the symmetric kernel the timed passes ran
    flop_eval = FLOP_PER_EVAL_SYM_EQUAL if equal_mass else FLOP_PER_EVAL_SYM
    traffic = TRAFFIC_BYTES_PER_LAUNCH.get((cfg["algorithm"], n, world, cfg["i_per_thread"], args.precision, equal_mass), (None, None))
    executed_tflops = (launch_pairs / 2 * flop_eval if cfg["algorithm"] == "symmetric"
                       else launch_pairs * FLOP_PER_PAIR) / avg_launch_s * 1e-12
    peak = PEAK_FP32_TFLOPS if args.precision != "f64" else PEAK_FP32_TFLOPS / 2
    p_end, _ = sim.gather_state()
    finite = bool(np.isfinite(p_end).all())
    # parity of the benched instantiation at the benched size, after the timed region: one more force pass of the
    # final state, sampled bodies against an fp64 direct sum (asserted: 2e-5 fp32, 2e-6 Kahan, 1e-12 fp64)
    sim.compute_forces()
    acc_own = sim.engine.accelerations(np.float64 if args.precision == "f64" else np.float32)
    bodies = sample_bodies(sim.i_begin, sim.i_count, cfg["super_tile"], cfg["i_per_thread"])
    err = sampled_force_error(p_end, acc_own, sim.i_begin, bodies, 1.0e4, args.eps)
    n_sampled = len(bodies)
    if world > 1:
        t = torch.tensor([err, float(n_sampled)], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        err = float(t[0])
        n_sampled = int(t[1]) * world
    tol = {"f32": 2e-5, "f32_kahan": 2e-6 if args.eps > 0 else 2e-5, "f64": 1e-12}[args.precision]
    if not (finite and err < tol):
        raise SystemExit(f"bench.py: the benched force pass disagrees with the fp64 direct sum on sampled bodies: "
                         f"max rel err {err:.3e} >= {tol:.1e} (finite={finite})")

    # The Plummer sphere's bodies all have one mass, and the symmetric kernel has a form for exactly that (no mass factor
    # inside the pair loop: nbody_equal_mass_form).  One GPU only: the same bodies with DISTINCT masses — the kernel's
    # general form — timed right after, same K steps, same parity check, so that both numbers stand in the one line.
    distinct = None
    if world == 1 and equal_mass and not args.no_distinct_row:
        posm2 = posm.copy()
        posm2[:, 3] *= np.random.default_rng(20261004).uniform(0.5, 1.5, n).astype(np.float32)
        sim2 = nb.ShardedSimulation(posm2, vel, rank=0, world_size=1, device=f"cuda:{local_rank}", engine_factory=factory,
                                    precision=args.precision, eps=args.eps, tile=args.tile, i_per_thread=args.ipt,
                                    j_split=args.jsplit, time_kernels=True,
                                    algorithm={"auto": 0, "tiled": 1, "symmetric": 2}[args.algorithm],
                                    zero_mode={"exact": 0, "floor": 2}[args.zero_mode])
        sim2.step(args.dt, max(args.warmup, 1))
        fence()
        sim2.engine.kernel_time_reset()
        t0 = time.perf_counter()
        sim2.step(args.dt, args.steps)
        fence()
        el2 = time.perf_counter() - t0
        f2_ms, f2_n = sim2.engine.kernel_time(nb.KERNEL_FORCES)
        clk2 = clock_fields(sim2.engine, f2_ms / max(f2_n, 1) * 1e-3, launch_pairs)
        p2, _ = sim2.gather_state()
        sim2.compute_forces()
        err2 = sampled_force_error(p2, sim2.engine.accelerations(np.float64 if args.precision == "f64" else np.float32), 0, bodies, 1.0e4, args.eps)
        still_equal = sim2.engine.equal_mass_form()
        sim2.close()
        if still_equal or not err2 < tol:
            raise SystemExit(f"bench.py: distinct-mass row: equal-mass form in use = {still_equal}, max rel err {err2:.3e} (tolerance {tol:.1e})")
        distinct = {"masses": "the same bodies, each mass scaled by its own factor in [0.5, 1.5): the kernel's general form",
                    "value": pairs_per_step * args.steps / el2, "ms_per_step": el2 / args.steps * 1e3,
                    "force_pass_avg_ms": f2_ms / max(f2_n, 1),
                    "roofline_frac": launch_pairs * FLOP_PER_PAIR / (f2_ms / max(f2_n, 1) * 1e-3) * 1e-12 / peak,
                    "clock_ghz": clk2["clock_ghz"], "cycles_per_interaction": clk2["cycles_per_interaction"],
                    "max_rel_err_sampled": err2}

    # Multi-GPU only: north_star's literal step — one-sided kernel, per-step all-gather of positions, no other collective —
    # timed next to the default (symmetric + all-to-all) in the same job, same K steps, after the main measurement.
    tiled_row = None
    if world > 1 and args.algorithm == "auto" and cfg["algorithm"] == "symmetric" and not args.no_tiled_row:
        sim.close()
        sim = build("tiled")
        fence()
        t0 = time.perf_counter()
        sim.step(args.dt, args.steps)
        fence()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        tiled_row = {"algorithm": "tiled", "parallelism": f"range-partition x{world}, per step 1 RCCL all-gather(posm)",
                     "value": pairs_per_step * args.steps / float(tt[0]), "ms_per_step": float(tt[0]) / args.steps * 1e3}

    # One GPU, the default workload: BASELINE.json's other configs and the theta > 0 frames, each on a context of its own, after
    # everything of the headline has been measured (its engine released first: N = 2^21 Kahan wants its own pool).
    extra = None
    if world == 1 and n == (1 << 20) and args.precision == "f32" and args.eps == 0.0 and args.algorithm == "auto" \
            and not (args.tile or args.ipt or args.jsplit) and not args.no_extra_rows:
        sim.close()
        extra = extra_rows(args, nb)

    if rank == 0:
        out = {
            "metric": "body-pair interactions/s at N=2^20" if n == (1 << 20) else f"body-pair interactions/s at N={n}",
            "value": value, "unit": "pair-interactions/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None,
            "dtype": {"f32": "f32", "f32_kahan": "f32", "f64": "f64"}[args.precision], "data": "synthetic",
            "config": {"workload": f"N={n} all-pairs {args.precision}, seeded Plummer sphere (equal masses), G=1e4, eps={args.eps}, "
                                   f"dt={args.dt}, one force pass + kick-drift per step",
                       "parallelism": (f"range-partition x{world}, per step 1 RCCL all-gather(posm)"
                                       + (" + 1 all-to-all(j-side sums)" if cfg["algorithm"] == "symmetric" else "")
                                       + ("" if args.no_overlap else "; the all-gather runs on its own stream under the next "
                                          "force pass's own-slice strips"))
                       if world > 1 else "1 GPU",
                       "algorithm": cfg["algorithm"], "zero_distance": args.zero_mode,
                       "lds_tile_bodies": cfg["tile"], "i_per_lane": cfg["i_per_thread"],
                       "j_split": cfg["j_split"] if cfg["algorithm"] == "tiled" else None,
                       "super_tile_bodies": cfg["super_tile"] or None,
                       "workgroups": cfg["blocks"], "plan": cfg.get("plan"), "accumulate": args.precision, "finite": finite,
                       "equal_mass_form": equal_mass,
                       **({"distinct_masses": distinct} if distinct else {}),
                       "max_rel_err_sampled": err, "bodies_sampled": n_sampled, "rel_err_tolerance": tol,
                       **({"fallback": fallback} if fallback else {}),
                       **({"all_gather_only_row": tiled_row} if tiled_row else {})},
            "roofline": {"bound": "valu_fp32" if args.precision != "f64" else "valu_fp64",
                         "achieved": achieved_tflops, "peak": peak, "unit": "TFLOP/s", "frac": achieved_tflops / peak,
                         "achieved_is": "algorithmic: N_i x N ordered interactions x 20 flop (SURVEY 8d) / launch time",
                         "executed": executed_tflops, "executed_frac": executed_tflops / peak, **clk,
                         "executed_is": (f"VALU flops issued: each unordered pair evaluated once, {flop_eval} flop"
                                         + (" (equal-mass form: no mass factors inside the loop)" if equal_mass else "")
                                         if cfg["algorithm"] == "symmetric" else "every ordered pair evaluated, 20 flop"),
                         "traffic": traffic[0], "traffic_source": traffic[1],
                         "kernel": cfg["kernel"] + (" (the events bracket the whole force pass: sym_prep_kernel and reduce_j_kernel too "
                                                    "where they are launches of their own — sharded ranks, fp64; a single fp32 "
                                                    "device folds them into update_sym_fused_kernel)"
                                                    if cfg["algorithm"] == "symmetric" else ""),
                         "avg_launch_ms": avg_launch_s * 1e3, "launches": f_n,
                         "launch_ms_min_max_over_ranks": launch_ms_minmax,
                         "flop_per_pair": FLOP_PER_PAIR, "pairs_per_launch": launch_pairs,
                         "update_kernel_avg_ms": u_ms / max(u_n, 1)},
        }
        if args.cpu_seconds > 0 and world == 1:
            out["cpu_baseline"] = cpu_baseline(posm, args.cpu_seconds)
        if extra is not None:
            out["configs"], out["bh"], out["mid_sizes"] = extra
        print_result(out)
    sim.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Command-line replay of the reference's scene: the parameter surface of BP_ScreenUI
(NParticles, BoxSize, DeltaTime, Pause via --dt 0) driving the AOctreeSearch-shaped engine.

    python -m parallelnbody_amd --n 2000 --size 1000 --dt 0.01 --steps 600          # the shipped scene
    python -m parallelnbody_amd --plummer --n 65536 --eps 1 --steps 100 --energy-every 20
    python -m parallelnbody_amd --n 2000 --steps 300 --checkpoint run.ckpt ; python -m parallelnbody_amd --n 2000 --resume run.ckpt --steps 300
    python -m parallelnbody_amd --n 2000 --theta 1.0 --steps 600 --trajectory run.trj --trajectory-every 10   # as shipped: Barnes-Hut, theta = 1

Trajectory file (nothing in the reference to mirror): header `NBDYTRJ1`, int32 n, int32 reserved, then per
dumped frame int64 frame number + n x 3 float32 positions — `read_trajectory(path)` returns (frames, positions[k, n, 3]).
"""
import argparse
import json
import sys
import time

import numpy as np

from . import NBodyEngine, ic_plummer, ic_reference_box


TRJ_MAGIC = b"NBDYTRJ1"


def read_trajectory(path):
    """(frame numbers [k], positions [k, n, 3] float32) of a file written by --trajectory."""
    with open(path, "rb") as f:
        if f.read(8) != TRJ_MAGIC:
            raise ValueError(f"{path}: not a trajectory file")
        n = int(np.frombuffer(f.read(8), np.int32)[0])
        rec = np.dtype([("frame", "<i8"), ("pos", "<f4", (n, 3))])
        data = np.frombuffer(f.read(), rec)
    return data["frame"].copy(), data["pos"].copy()


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m parallelnbody_amd", description=__doc__.split("\n")[0])
    ap.add_argument("--n", type=int, default=2000, help="NParticles (BP_ScreenUI default 2000)")
    ap.add_argument("--size", type=float, default=1000.0, help="BoxSize (BP_ScreenUI default 1000)")
    ap.add_argument("--dt", type=float, default=0.01, help="DeltaTime = PhDeltaTime (default 0.01; <= 0 pauses)")
    ap.add_argument("--steps", type=int, default=100, help="frames to advance")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--plummer", action="store_true", help="seeded Plummer sphere instead of CreateSpacePoints' box")
    ap.add_argument("--eps", type=float, default=0.0, help="softening length (reference: 0)")
    ap.add_argument("--G", type=float, default=1.0e4, help="gravitational constant (reference: 1e4)")
    ap.add_argument("--precision", default="f32", choices=["f32", "f32_kahan", "f64"])
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--energy-every", type=int, default=0, help="print kinetic/potential energy every K frames")
    ap.add_argument("--checkpoint", help="write the final state here")
    ap.add_argument("--resume", help="start from this checkpoint instead of fresh initial conditions")
    ap.add_argument("--dump-positions", help="write the final positions (n x 3 float32, .npy)")
    ap.add_argument("--theta", type=float, default=None,
                    help="opening angle: 0 = exact all-pairs (default); 1.0 = the reference's shipped Barnes-Hut walk.  With "
                         "--resume: the file's own opening angle unless this option says otherwise (then it is announced)")
    ap.add_argument("--sync-energy", action="store_true",
                    help="energy lines also carry the total with the staggered velocity pulled to the positions' time "
                         "(v_n = v_{n-1/2} + dt/2 a_n: one extra force pass per line.  Positions, velocities and — at theta > 0 — "
                         "the root of the next tree are not touched, so the trajectory is the same with and without; the stored "
                         "accelerations are those of the extra pass)")
    ap.add_argument("--leapfrog-start", action="store_true",
                    help="fresh runs only: store v_{-1/2} = v_0 - dt/2 a_0, so the update (v += dt a; x += dt v) is a proper leapfrog")
    ap.add_argument("--trajectory", help="write positions every --trajectory-every frames to this file (read_trajectory reads it back)")
    ap.add_argument("--trajectory-every", type=int, default=1)
    a = ap.parse_args(argv)

    posm, vel = (ic_plummer(a.n, G=a.G, seed=a.seed) if a.plummer else ic_reference_box(a.n, a.size, seed=a.seed))
    if a.trajectory_every < 1:
        ap.error("--trajectory-every must be >= 1")
    with NBodyEngine(a.n, device=a.device, precision=a.precision, G=a.G, eps=a.eps, theta=a.theta or 0.0) as e:
        e.set_state(posm, vel)
        start = e.load_checkpoint(a.resume) if a.resume else 0
        if a.resume and a.theta is not None and e.theta() != np.float32(a.theta):
            print(f"--resume: {a.resume} was written at theta = {e.theta():g}; continuing at --theta {a.theta:g} as asked "
                  "(not the trajectory the file belongs to)", file=sys.stderr)
            e.set_theta(a.theta)
        f64 = a.precision == "f64"
        if a.leapfrog_start and not a.resume and a.dt > 0:
            e.compute_forces()
            p0, v0, a0 = e.state(np.float64 if f64 else np.float32)
            v0[:, :3] = (v0[:, :3].astype(np.float64) - 0.5 * a.dt * a0[:, :3].astype(np.float64)).astype(v0.dtype)
            e.set_state(p0, v0)

        def energies():
            ke, pe = e.energy()
            out = {"kinetic": ke, "potential": pe, "total": ke + pe}
            if a.sync_energy:
                e.compute_forces()                               # a(x_n); positions and velocities are not touched
                p, v, acc = e.state(np.float64)
                vs = v[:, :3] + 0.5 * a.dt * acc[:, :3]
                out["total_synchronised"] = 0.5 * float((p[:, 3] * (vs ** 2).sum(1)).sum()) + pe
            return out
        trj = None
        if a.trajectory:
            trj = open(a.trajectory, "wb")
            trj.write(TRJ_MAGIC + np.array([a.n, 0], np.int32).tobytes())
            frame_buf = np.empty((a.n, 3), np.float32)
            e.pin(frame_buf)                                     # frames land in it by one DMA
        # advance in chunks that end on every frame somebody wants to see
        marks = [m for m in (a.energy_every, a.trajectory_every if trj else 0) if m > 0]
        if a.energy_every > 0 and a.sync_energy and not a.resume:
            print(json.dumps({"frame": start, **energies()}), flush=True)
        t0 = time.perf_counter()
        done = 0
        while done < a.steps:
            k = min([a.steps - done] + [m - (start + done) % m for m in marks])
            e.step(a.dt, k)
            done += k
            frame = start + done
            if a.energy_every > 0 and frame % a.energy_every == 0:
                print(json.dumps({"frame": frame, **energies()}), flush=True)
            if trj and frame % a.trajectory_every == 0:
                e.positions(out=frame_buf)
                trj.write(np.int64(frame).tobytes() + frame_buf.tobytes())
        e.synchronize()
        if trj:
            trj.close()
        wall = time.perf_counter() - t0
        size = e.bounds()
        if a.checkpoint:
            e.save_checkpoint(a.checkpoint)
        if a.dump_positions:
            np.save(a.dump_positions, e.positions())
        cfg = e.launch_config()
        print(json.dumps({"n": a.n, "frames": a.steps, "first_frame": start, "dt": a.dt, "wall_s": wall,
                          "frames_per_s": a.steps / wall if wall > 0 else None,
                          "pair_interactions_per_s": a.n * a.n * a.steps / wall if wall > 0 and a.dt > 0 else 0.0,
                          "Size": size, "algorithm": cfg["algorithm"], "steps_done": e.steps_done()}))


if __name__ == "__main__":
    main()

"""The sharded HIP path with a REAL process group: two processes share the one GPU of the box, each drives a HIP engine
for half of the bodies, and the per-step collectives (in-place all_gather_into_tensor of positions, all_to_all_single of
the symmetric algorithm's j-side sums) run over gloo on device tensors — RCCL itself refuses two ranks on one device, so
only the transport differs from the 8-GPU job.  The result must match a single-context run."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_processes_one_gpu_end_to_end():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tools", "two_rank_gloo_gpu.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "two ranks on one GPU over gloo: algorithm symmetric, exchange ranks 2" in out.stdout
    assert "at theta = 1.0: four frames, every byte of positions and velocities equal to one context's" in out.stdout


CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                 "dtype", "data", "config", "roofline")


def _bench_two_ranks(extra_env, extra_args=(), launcher=True):
    """bench.py --gpus 2 with both ranks on this box's one GPU (collectives over gloo).  launcher=True: the driver's own command
    line for N > 1 (torch.distributed.run around it); False: the bare `python bench.py --gpus 2`."""
    import json
    args = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--bodies", "65536", "--settle-seconds", "0.02",
            *extra_args]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NBODY_DIST_BACKEND="gloo", **extra_env)
    if launcher:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port)] + args
    else:
        cmd = [sys.executable] + args
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    for key in CONTRACT_KEYS:
        assert key in r, key
    return r, out.stderr


def test_bench_multi_rank_line_on_one_gpu():
    # bench.py's own multi-rank path (the one the driver launches on 8 GPUs), rehearsed with two ranks sharing this GPU over
    # gloo: symmetric kernel + all-to-all as the headline, the all-gather-only step (north_star's literal one) beside it,
    # and the fp64-sampled parity check on every rank's slice
    r, _ = _bench_two_ranks({})
    cf = r["config"]
    assert r["n_gpus"] == 2 and cf["algorithm"] == "symmetric" and "fallback" not in cf
    assert cf["max_rel_err_sampled"] < 2e-5 and cf["bodies_sampled"] >= 48
    # the force pass runs in two goes here (own-slice strips, then the rest): still ONE pass per step in the kernel timers,
    # or roofline.achieved would read twice what the hardware did
    assert r["roofline"]["launches"] == r["steps"] and 0.0 < r["roofline"]["frac"] < 1.0
    row = cf["all_gather_only_row"]
    assert row["algorithm"] == "tiled" and row["value"] > 0 and "all-to-all" not in row["parallelism"]


def test_bench_falls_back_together_when_one_rank_cannot_build_the_symmetric_engine():
    # rank 1 "fails" to create its symmetric engine: both ranks learn it from the same collective and rebuild on the
    # one-sided kernel; the line says so
    r, err = _bench_two_ranks({"NBODY_REHEARSE_CREATE_FAILURE": "1"})
    assert r["config"]["algorithm"] == "tiled" and "symmetric engines could not be created" in r["config"]["fallback"]
    assert "every rank rebuilds" in err


@pytest.mark.parametrize("launcher", [True, False])
def test_bench_survives_a_failure_inside_the_all_to_all(launcher):
    # rank 1 raises inside the symmetric step's all-to-all (its first one: the warm-up of the collectives); rank 0 is left inside
    # that collective.  The watchdogs end both workers and start fresh ones on north_star's literal step — one-sided kernel,
    # per-step all-gather, every collective in stream order — whose line goes out with config.fallback saying what happened
    r, err = _bench_two_ranks({"NBODY_REHEARSE_A2A_FAILURE": "1"}, extra_args=["--child-timeout", "240"], launcher=launcher)
    cf = r["config"]
    assert r["n_gpus"] == 2 and cf["algorithm"] == "tiled" and r["value"] > 0 and cf["max_rel_err_sampled"] < 2e-5
    assert "all-to-all" not in cf["parallelism"] and "own stream" not in cf["parallelism"]
    fb = cf["fallback"]
    # (rank 0's worker fails as well, as soon as gloo sees its peer go: whichever exit was noticed first is named, both are listed)
    assert "worker exit code" in fb["because"] and 1 in fb["ranks_failed"]
    tails = fb["stderr_tail"] + [ln for o in fb["other_failures"] for ln in o["stderr_tail"]]
    assert any("NBODY_REHEARSE_A2A_FAILURE" in ln for ln in tails)
    assert "starting a fresh worker" in err
    assert 0.0 < r["roofline"]["frac"] < 1.0 and r["roofline"]["launches"] == r["steps"]


@pytest.mark.parametrize("launcher", [True, False])
def test_bench_survives_a_rank_that_never_returns_from_the_all_to_all(launcher):
    # the hang: rank 1 sleeps inside the all-to-all; nothing fails, nothing returns.  At the time limit the watchdogs kill the
    # workers' process groups (their own children, by process group id) and the fresh workers deliver the all-gather-only line
    r, err = _bench_two_ranks({"NBODY_REHEARSE_A2A_HANG": "1"}, extra_args=["--child-timeout", "75"], launcher=launcher)
    cf = r["config"]
    assert cf["algorithm"] == "tiled" and r["value"] > 0 and cf["max_rel_err_sampled"] < 2e-5
    assert "time limit of 75 s" in cf["fallback"]["because"]
    assert "killing the worker's process group" in err


def test_bench_without_a_launcher_starts_its_own_ranks():
    # `python bench.py --gpus 2` with no torch.distributed.run around it (the shape of the command the driver uses for one GPU):
    # bench.py starts the launcher as a child process before touching the GPU and relays its line and exit code
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--bodies", "65536",
           "--settle-seconds", "0.02", "--no-tiled-row"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["NBODY_DIST_BACKEND"] = "gloo"
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "no launcher around --gpus 2" in out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["config"]["max_rel_err_sampled"] < 2e-5
    for key in ("metric", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline"):
        assert key in r, key


def test_bench_single_host_line():
    # `--host single`: one process, one thread, nbody_create_multi over devices 0..N-1 (here N = 1: a one-rank RCCL
    # communicator, every collective still runs); the same contract line, cpu_baseline included at N = 1
    import json
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--host", "single", "--steps", "2", "--warmup", "1",
           "--bodies", "65536", "--settle-seconds", "0.02", "--cpu-seconds", "0.5"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["value"] > 0 and "nbody_create_multi" in r["config"]["host"]
    assert r["config"]["algorithm"] == "symmetric" and r["config"]["max_rel_err_sampled"] < 2e-5
    assert 0.0 < r["roofline"]["frac"] < 1.0 and r["cpu_baseline"]["value"] > 0
    # more devices than the box has: refused before anything is measured
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--host", "single"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode != 0 and "device(s) visible" in out.stderr

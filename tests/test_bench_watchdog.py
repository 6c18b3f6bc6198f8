"""bench.py --gpus N (N > 1): every rank is first a watchdog that never touches the GPU; the work is done by child processes
with a time limit, and a failure or a hang of the default multi-rank step is answered by FRESH workers on the
all-gather-only step (bench.py rank_watchdog).  Here on the CPU: the watchdog's own logic with a stand-in worker
(tests/helpers/watchdog_worker.py: a real gloo process group, behaviour by environment); the HIP workers are rehearsed on the
GPU box by tests/test_two_rank_gpu.py."""
import json
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = f"{sys.executable} {os.path.join(ROOT, 'tests', 'helpers', 'watchdog_worker.py')}"


def run_bench(mode, *extra, under_launcher=False, timeout=20):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(NBODY_BENCH_WORKER_CMD=STUB, WATCHDOG_STUB_MODE=mode)
    cmd = [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--child-timeout", str(timeout), *extra]
    if under_launcher:                                     # the driver's own command line for N > 1
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
        cmd = ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port", str(port)] + cmd
    t0 = time.monotonic()
    out = subprocess.run([sys.executable] + cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out, [json.loads(ln) for ln in lines], time.monotonic() - t0


@pytest.mark.parametrize("under_launcher", [False, True])
def test_first_attempt_succeeds(under_launcher):
    out, lines, _ = run_bench("ok", under_launcher=under_launcher)
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1 and lines[0]["attempt"] == 1 and "fallback" not in lines[0]["config"]
    assert "--algorithm" not in lines[0]["argv"] and lines[0]["argv"][-1] == "--worker"


@pytest.mark.parametrize("under_launcher", [False, True])
def test_a_failing_rank_sends_the_job_to_fresh_workers_on_the_all_gather_only_step(under_launcher):
    out, lines, took = run_bench("fail1", under_launcher=under_launcher, timeout=60)
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1, out.stdout
    r = lines[0]
    assert r["attempt"] == 2 and r["argv"][-4:] == ["--worker", "--algorithm", "tiled", "--no-overlap"]
    fb = r["config"]["fallback"]
    assert fb["because"].startswith("rank 1: worker exit code") and 1 in fb["ranks_failed"]
    assert any("injected" in ln for ln in fb["stderr_tail"])
    assert "all-gather-only" in fb["ran"]
    # rank 0 sat in a collective rank 1 never entered: its watchdog ended it as soon as rank 1's failure was known — long
    # before the 60 s limit
    assert took < 45, took


def test_a_hanging_rank_is_ended_at_the_time_limit():
    out, lines, took = run_bench("hang1", timeout=8)
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1 and lines[0]["attempt"] == 2
    fb = lines[0]["config"]["fallback"]
    assert "time limit of 8 s" in fb["because"]
    assert "killing the worker's process group" in out.stderr
    assert took < 60


def test_a_line_that_is_out_survives_a_late_failure_elsewhere():
    out, lines, took = run_bench("late1")
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1 and lines[0]["attempt"] == 1 and "fallback" not in lines[0]["config"]
    assert took < 25                                   # rank 0's worker (asleep for 30 s) was not waited for


def test_both_attempts_failing_is_a_failure():
    out, lines, _ = run_bench("failall")
    assert out.returncode != 0 and not lines
    assert "nothing left to try" in out.stderr


def test_no_fallback_when_the_caller_chose_the_algorithm():
    out, lines, _ = run_bench("fail1", "--algorithm", "symmetric")
    assert out.returncode != 0 and not lines
    out, lines, _ = run_bench("fail1", "--no-fallback")
    assert out.returncode != 0 and not lines


def test_the_fallback_rendezvous_is_on_a_port_of_its_own():
    out, lines, _ = run_bench("fail1", under_launcher=True)
    assert out.returncode == 0 and lines[0]["attempt"] == 2
    launcher_port = out.args[out.args.index("--master-port") + 1]
    assert lines[0]["port"] != launcher_port           # the first attempt's store may hold its keys for ever: not reused


def test_files_of_another_job_in_the_same_directory_are_not_this_jobs(tmp_path):
    # The watchdogs agree through small files in a directory whose name another job may have had (same port, no launcher id).  A
    # leftover `fail` of rank 1 and a leftover `give up` verdict of an earlier job stand in the directory: they carry that job's
    # name, not this one's, and are ignored — the first attempt succeeds and the line goes out.  The directory is gone afterwards.
    run_dir = tmp_path / "shared_name"
    run_dir.mkdir()
    stale = json.dumps({"status": "fail worker exit code 9", "t": 1.0, "stderr_tail": ["from another job"]})
    (run_dir / "a1.rank1").write_text("111-222\n" + stale)       # another job's files
    (run_dir / "a1.verdict").write_text("111-222\ngive up")
    (run_dir / "job").write_text("111-222")
    env_before = os.environ.get("NBODY_BENCH_RUN_DIR")
    os.environ["NBODY_BENCH_RUN_DIR"] = str(run_dir)
    try:
        out, lines, _ = run_bench("ok", under_launcher=True)
    finally:
        if env_before is None:
            del os.environ["NBODY_BENCH_RUN_DIR"]
        else:
            os.environ["NBODY_BENCH_RUN_DIR"] = env_before
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1 and lines[0]["attempt"] == 1 and "fallback" not in lines[0]["config"]
    assert not run_dir.exists()

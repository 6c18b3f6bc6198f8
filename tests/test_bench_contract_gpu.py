"""bench.py's one-line JSON contract (the driver parses it): run it small and check every field the contract names."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                          "--n", "65536", "--cpu-seconds", "0.5", "--settle-seconds", "0.05"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1 and r["higher_is_better"] is True
    assert r["dtype"] == "f32" and r["data"] == "synthetic" and r["vs_baseline"] is None and r["scaling"] == "strong"
    assert "workload" in r["config"] and "model" not in r["config"]
    assert r["value"] == pytest.approx(65536.0 ** 2 * 3 / (r["ms_per_step"] * 3e-3), rel=1e-6)
    rf = r["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in rf, key
    assert rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3 and rf["frac"] == pytest.approx(rf["achieved"] / rf["peak"])
    assert 0.2 < rf["frac"] < 1.0, rf           # a real measurement of the HIP kernel (N = 65536: about two thirds of peak)
    # the clock the timed kernels ran at, read inside the kernels, and what time x clock says about the code alone: SIMD cycles
    # per interaction (equal-mass form at N = 65536: 21.9 by PMC; the launch also holds ramp-up, tail and the fold)
    assert 1.2 < rf["clock_ghz"] <= 2.45 and rf["compute_units"] == 256, rf
    assert 19.0 < rf["cycles_per_interaction"] < 32.0, rf
    assert rf["cycles_per_interaction"] == pytest.approx(rf["avg_launch_ms"] * 1e-3 * rf["clock_ghz"] * 1e9 * 256 * 4 * 64 / rf["pairs_per_launch"])
    cb = r["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in cb, key
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == r["unit"]
    # the two CPU legs VERDICT r1 asked for: the reference-arithmetic port and the SIMD direct sum at N = 1024 and 65536
    assert cb["port"]["value"] == cb["value"]
    assert cb["simd"]["n1024"]["value"] > 0 and cb["simd"]["n65536"]["value"] > cb["port"]["value"]
    # parity of the benched pass inside the bench run itself, and the executed-flop figure next to the credited one
    cf = r["config"]
    assert cf["bodies_sampled"] >= 24 and cf["max_rel_err_sampled"] < cf["rel_err_tolerance"] == 2e-5
    # the Plummer sphere has equal masses: the timed passes ran the equal-mass form (23 flop per unordered pair), and the
    # same bodies with distinct masses — the general form, 25 flop — are timed and checked in the same line
    assert cf["equal_mass_form"] is True
    assert rf["executed"] == pytest.approx(rf["achieved"] * 23 / 40) and "traffic_source" in rf
    dm = cf["distinct_masses"]
    assert dm["max_rel_err_sampled"] < 2e-5 and 0.2 < dm["roofline_frac"] < rf["frac"] * 1.02
    assert dm["value"] == pytest.approx(65536.0 ** 2 * 3 / (dm["ms_per_step"] * 3e-3), rel=1e-6)
    assert 1.2 < dm["clock_ghz"] <= 2.45 and rf["cycles_per_interaction"] < dm["cycles_per_interaction"] < 36.0   # two more packed ops per register pair


def test_stdout_is_the_json_line_alone_when_rccl_prints_its_banner():
    # NCCL_DEBUG=VERSION makes RCCL write a version banner to the process's stdout at communicator creation: the
    # single-process multi-GPU leg creates communicators even for one device.  The driver parses stdout.
    env = dict(os.environ, NCCL_DEBUG="VERSION")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--host", "single", "--steps", "2",
                          "--warmup", "1", "--n", "32768", "--cpu-seconds", "0", "--settle-seconds", "0.05"],
                         capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["config"]["host"].startswith("single process") and r["value"] > 0


def test_the_rows_bench_adds_behind_the_headline_carry_their_own_parity_and_roofline():
    # `configs` (BASELINE.json's other single-GPU configs) and `bh` (theta = 1 frames) of the default one-GPU line, at their smallest
    # members so that the test takes seconds: the same functions, the same checks (a row that disagrees with its checker ends the run)
    sys.path.insert(0, ROOT)
    import bench
    import parallelnbody_amd as nb
    row = bench.baseline_config_row(nb, 1 << 16, "f32", 0.0, steps=5, warmup=1, settle_seconds=0.05)
    assert row["steps"] == 5 and row["dtype"] == "f32" and row["algorithm"] == "symmetric"
    assert row["value"] == pytest.approx(65536.0 ** 2 * 5 / (row["ms_per_step"] * 5e-3), rel=1e-6)
    rf = row["roofline"]
    assert rf["peak"] == 157.3 and 0.2 < rf["whole_step_frac"] <= rf["frac"] < 1.0 and rf["launches"] == 5
    assert row["max_rel_err_sampled"] < row["rel_err_tolerance"] == 2e-5 and row["bodies_sampled"] >= 24
    row64 = bench.baseline_config_row(nb, 1 << 14, "f64", 0.0, steps=3, warmup=1, settle_seconds=0.0)
    assert row64["dtype"] == "f64" and row64["roofline"]["peak"] == pytest.approx(78.65) and row64["max_rel_err_sampled"] < 1e-12
    if row64["kernel"] == "forces_sym_f64_kernel":              # the clock the fp64 pass ran at, from stamps inside the kernel (round 5)
        rf64 = row64["roofline"]
        assert 1.0 < rf64["clock_ghz"] <= 2.45 and rf64["compute_units"] == 256 and 40.0 < rf64["cycles_per_interaction"] < 400.0, rf64
    bh = bench.barnes_hut_row(nb, 2000, frames=50, warmup=5, parity_frames=2)
    assert bh["frames"] == 50 and 5.0 < bh["us_per_frame"] < 2000.0 and "every byte of 2 frame(s)" in bh["parity"]
    assert bh["tree_nodes"] > 2000 and bh["tree_levels"] >= 8
    cb = bh["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "us/frame" and cb["value"] > bh["us_per_frame"] and cb["cores"] >= 1
    # `mid_sizes`: the general form as a host steps it (no events), sampled parity first
    mid = bench.mid_size_row(nb, 20480, seconds=0.05)
    assert mid["general_form"] and mid["plan"] == "even" and mid["kernel"] == "forces_sym_pk_kernel"
    assert mid["max_rel_err_sampled"] < mid["rel_err_tolerance"] == 2e-5 and 0.3 < mid["whole_step_frac"] < 1.0
    assert mid["value"] == pytest.approx(20480.0 ** 2 / (mid["us_per_step"] * 1e-6), rel=1e-9)

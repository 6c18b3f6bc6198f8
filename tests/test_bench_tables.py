"""bench.py's table of measured HBM traffic (roofline.traffic) must say what the committed PMC summaries say: every entry
names a file under profiles/ and carries that file's traffic_bytes_per_launch (or, for the round-1 one-sided kernel, the
FETCH_SIZE / WRITE_SIZE figures it is computed from)."""
import importlib.util
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_traffic_table_matches_the_committed_pmc_summaries():
    table = _bench().TRAFFIC_BYTES_PER_LAUNCH
    assert len(table) >= 8
    for key, (nbytes, source) in table.items():
        path = os.path.join(ROOT, source)
        assert os.path.isfile(path), (key, source)
        text = open(path).read()
        m = re.search(r"^traffic_bytes_per_launch,(\d+)", text, re.M)
        if m:
            assert int(m.group(1)) == nbytes, (key, source, m.group(1), nbytes)
        else:                                   # round-1 summary: raw counters in KB, FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE
            f = re.search(r"FETCH_SIZE\D+([\d.]+)", text); w = re.search(r"WRITE_SIZE\D+([\d.]+)", text)
            assert f and w, (key, source)
            assert abs((2 * float(f.group(1)) + float(w.group(1))) * 1024 - nbytes) <= 0.01 * nbytes, (key, source)
        assert key[0] in ("tiled", "symmetric") and key[4] in ("f32", "f32_kahan", "f64") and isinstance(key[5], bool)


def test_even_share_traffic_table_matches_the_committed_pmc_summaries():
    table = _bench().TRAFFIC_BYTES_PER_LAUNCH_EVEN
    assert len(table) >= 1
    for key, (nbytes, source) in table.items():
        text = open(os.path.join(ROOT, source)).read()
        m = re.search(r"^traffic_bytes_per_launch,(\d+)", text, re.M)
        assert m and int(m.group(1)) == nbytes, (key, source)
        assert 16385 <= key[0] < 139264 and key[2] == "f32" and isinstance(key[3], bool)


def test_flop_conventions_are_what_design_states():
    b = _bench()
    assert (b.FLOP_PER_PAIR, b.FLOP_PER_EVAL_SYM, b.FLOP_PER_EVAL_SYM_EQUAL) == (20, 25, 23)
    assert b.PEAK_FP32_TFLOPS == 157.3


def test_result_line_has_stdout_to_itself():
    """bench.py's contract is ONE JSON line on stdout; libraries under it (RCCL's version banner) write to file descriptor 1
    as well.  keep_stdout_for_the_result() points fd 1 at stderr for the run; print_result() goes out through the original."""
    import subprocess
    code = ("import os, sys; sys.path.insert(0, %r); import bench; bench.keep_stdout_for_the_result(); "
            "os.write(1, b'a library banner on fd 1\\n'); print('a python print'); bench.print_result({'value': 1.5})") % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout == '{"value": 1.5}\n'
    assert "a library banner on fd 1" in out.stderr and "a python print" in out.stderr

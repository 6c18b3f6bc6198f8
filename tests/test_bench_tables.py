"""bench.py's table of measured HBM traffic (roofline.traffic) must say what the committed PMC summaries say: every entry
names a file under profiles/ and carries that file's traffic_bytes_per_launch (or, for the round-1 one-sided kernel, the
FETCH_SIZE / WRITE_SIZE figures it is computed from)."""
import importlib.util
import os
import re

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


def test_flop_conventions_are_what_design_states():
    b = _bench()
    assert (b.FLOP_PER_PAIR, b.FLOP_PER_EVAL_SYM, b.FLOP_PER_EVAL_SYM_EQUAL) == (20, 25, 23)
    assert b.PEAK_FP32_TFLOPS == 157.3

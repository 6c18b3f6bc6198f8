"""Rules the native sources keep that no compiler checks (CPU-only: reads the .hip files as text)."""
import glob
import os
import re

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "parallelnbody_amd", "csrc")


def _functions(text):
    """(start, end) of every top-level brace block: from a line that ends with '{' at nesting depth 0 to the '}' in column 0."""
    out, start = [], None
    for m in re.finditer(r"^(.*)$", text, re.M):
        line = m.group(1)
        if start is None and line.rstrip().endswith("{") and not line.startswith((" ", "\t", "}")):
            start = m.start()
        elif start is not None and line.startswith("}"):
            out.append((start, m.end()))
            start = None
    return out


def test_memsets_on_the_null_stream_are_waited_for_before_the_function_returns():
    # hipMemset returns before the fill has run, and it runs on the NULL stream; every context works on a non-blocking stream,
    # which does not wait for the null stream.  Round 4's frames fuzz found a creation memset landing inside the first frame.
    # So: a function that calls hipMemset( must wait for the null stream after its last one.
    checked = 0
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp"))):
        text = open(path).read()
        for a, b in _functions(text):
            body = text[a:b]
            last = body.rfind("hipMemset(")
            if last < 0:
                continue
            checked += 1
            assert "hipStreamSynchronize(nullptr)" in body[last:] or "hipDeviceSynchronize()" in body[last:], (
                f"{os.path.basename(path)}: a function calls hipMemset( and does not wait for the null stream afterwards:\n"
                + body[:200])
    assert checked >= 2          # nbody_create and bh_create at least


def test_no_stream_is_created_blocking_by_accident():
    # (the rule above rests on it: the streams are non-blocking on purpose — a blocking stream would serialise with torch's
    # null-stream work in the host process)
    for path in glob.glob(os.path.join(CSRC, "*.hip")):
        text = open(path).read()
        assert "hipStreamCreate(" not in text, path

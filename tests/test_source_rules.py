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


WAITS = ("hipStreamSynchronize(nullptr)", "hipDeviceSynchronize()")


def _waited(text):
    return any(w in text for w in WAITS)


def test_memsets_on_the_null_stream_are_waited_for_before_the_function_returns():
    # hipMemset returns before the fill has run, and it runs on the NULL stream; every context works on a non-blocking stream,
    # which does not wait for the null stream.  Round 4's frames fuzz found a creation memset landing inside the first frame, and
    # the fix first missed the small systems' early `return hipSuccess` in front of the wait.  So, for a function that calls
    # hipMemset(:
    #   * every `return` behind its first hipMemset( — other than the error exits that destroy the object (`return bail(`; BH_TRY's
    #     are of that kind and hidden in the macro) — has a wait for the null stream between the last hipMemset( before it and itself;
    #   * or the function waits nowhere, is `static`, and every call of it in its file is followed by the wait before the caller's
    #     next `return` (bh_create_state / bh_create).
    checked = 0
    for path in sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp"))):
        text = re.sub(r"//[^\n]*", "", open(path).read())          # (comments talk about returns and memsets too)
        funcs = _functions(text)
        for a, b in funcs:
            body = text[a:b]
            first = body.find("hipMemset(")
            if first < 0:
                continue
            checked += 1
            name = os.path.basename(path) + ": " + body[:body.index("\n")]
            if not _waited(body):
                head = body[:body.index("(")]
                assert head.startswith("static "), f"{name}\ncalls hipMemset( and neither waits for the null stream nor is static"
                fn = head.split()[-1].lstrip("*")
                calls = 0
                for ca, cb in funcs:
                    caller = text[ca:cb]
                    if (ca, cb) == (a, b):
                        continue
                    for m in re.finditer(re.escape(fn) + r"\(", caller):
                        rest = caller[m.end():]
                        nxt = re.search(r"\breturn\b", rest)
                        assert nxt is not None and _waited(rest[:nxt.start()]), (
                            f"{name}\nis called without a wait for the null stream before the caller returns:\n" + caller[:200])
                        calls += 1
                assert calls >= 1, f"{name}\ncalls hipMemset(, does not wait, and no caller was found"
                continue
            for m in re.finditer(r"\breturn\b(?! bail\()", body):
                if m.start() < first:
                    continue
                last = body.rfind("hipMemset(", 0, m.start())
                assert _waited(body[last:m.start()]), (
                    f"{name}\nreturns behind a hipMemset( without waiting for the null stream:\n" + body[m.start():m.start() + 120])
    assert checked >= 2          # nbody_create and bh_create_state at least


def test_no_stream_is_created_blocking_by_accident():
    # (the rule above rests on it: the streams are non-blocking on purpose — a blocking stream would serialise with torch's
    # null-stream work in the host process)
    for path in glob.glob(os.path.join(CSRC, "*.hip")):
        text = open(path).read()
        assert "hipStreamCreate(" not in text, path

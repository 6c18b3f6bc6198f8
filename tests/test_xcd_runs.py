"""xcd_run_block (csrc/bh_common.h): the renumbering that gives one XCD's workgroups — b, b + 8, b + 16, ... under the round-robin
dispatch — consecutive runs of the key order must be a bijection of [0, grid) for EVERY grid size, or bodies would be walked twice
and others not at all.  Restated here line by line (the GPU tests check the kernels that use it bit by bit at many sizes; this one
checks every grid size up to 5000 and that the text in the header is still the text restated)."""
import os
import re

import numpy as np

HEADER = os.path.join(os.path.dirname(__file__), "..", "parallelnbody_amd", "csrc", "bh_common.h")


def xcd_run_block(b, g):
    x, q, r = b & 7, g >> 3, g & 7          # XCD x holds q + (x < r) workgroups
    return x * q + min(x, r) + (b >> 3)


def test_the_header_still_says_what_is_restated_here():
    text = open(HEADER).read()
    body = text[text.index("__device__ __forceinline__ int xcd_run_block()"):]
    body = body[:body.index("\n}\n")]
    assert re.search(r"x = \(int\)blockIdx\.x & 7, q = g >> 3, r = g & 7;", body)
    assert re.search(r"return x \* q \+ min\(x, r\) \+ \(\(int\)blockIdx\.x >> 3\);", body)


def test_every_grid_size_is_renumbered_one_to_one_and_runs_are_consecutive():
    for g in list(range(1, 1200)) + [2048, 4095, 4096, 4097, 4999, 5000]:
        out = np.array([xcd_run_block(b, g) for b in range(g)])
        assert np.array_equal(np.sort(out), np.arange(g)), g
        for x in range(min(8, g)):              # one XCD's workgroups, in dispatch order, take consecutive numbers
            mine = out[x::8]
            assert np.array_equal(mine, np.arange(mine[0], mine[0] + len(mine))), (g, x)

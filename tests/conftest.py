import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/nbody_oracle.c) — the checker, never the thing under test."""
    from oracle import oracle as O
    O.build()
    O.lib()
    return O


@pytest.fixture(scope="session")
def nb():
    """The product package; its shared library must already be built (graft build())."""
    import parallelnbody_amd as nb
    if not os.path.exists(nb._lib.LIB_PATH):
        nb.build()
    nb.lib()
    return nb


def rel_err(a, ref):
    """Per-body relative error of 3-vectors: |a - ref| / |ref|."""
    a = np.asarray(a, np.float64)
    ref = np.asarray(ref, np.float64)
    den = np.linalg.norm(ref, axis=1)
    den = np.where(den > 0, den, 1.0)
    return np.linalg.norm(a - ref, axis=1) / den


def particles_from(nb, posm, vel):
    p = np.zeros(posm.shape[0], nb.PARTICLE_DTYPE)
    p["Mass"] = posm[:, 3]
    p["Position"] = posm[:, :3]
    p["Velocity"] = vel[:, :3]
    return p

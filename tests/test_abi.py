"""The C-ABI library: loads, exports every symbol the headers declare, validates arguments and refuses
to run without a device.  Host-only logic (initial conditions, launch geometry).  No GPU compute."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    names = []
    for h in ("nbody.h", "nbody_actor.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        names += re.findall(r"NBODY_AMD_API[^;]*?\b(nbody_[a-z0-9_]+)\s*\(", text, flags=re.S)
    return sorted(set(names))


def test_every_declared_symbol_is_exported(nb):
    L = ctypes.CDLL(nb._lib.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 40
    missing = [s for s in syms if not hasattr(L, s)]
    assert not missing, missing


def test_ctypes_binding_covers_the_header(nb):
    L = nb.lib()
    for s in declared_symbols():
        assert getattr(L, s).argtypes is not None, f"{s} has no prototype in parallelnbody_amd/_lib.py"


def test_params_struct_matches_header(nb):
    p = nb._lib.Params()
    assert nb.lib().nbody_default_params(ctypes.byref(p)) == 0
    assert p.struct_size == ctypes.sizeof(nb._lib.Params) == 72
    assert p.G == 1.0e4 and p.eps == 0.0 and p.precision == nb.PREC_F32     # reference constants
    assert nb.PARTICLE_DTYPE.itemsize == 40


def test_create_rejects_bad_arguments(nb):
    L = nb.lib()
    h = ctypes.c_void_p()
    p = nb._lib.Params()
    L.nbody_default_params(ctypes.byref(p))
    p.n_total = 0
    assert L.nbody_create(ctypes.byref(p), ctypes.byref(h)) == nb._lib.ERR_INVALID
    p.n_total = 100; p.i_begin = 50; p.i_count = 51
    assert L.nbody_create(ctypes.byref(p), ctypes.byref(h)) == nb._lib.ERR_INVALID
    p.i_begin = 0; p.i_count = 0; p.tile = 100
    assert L.nbody_create(ctypes.byref(p), ctypes.byref(h)) == nb._lib.ERR_INVALID
    p.tile = 0; p.struct_size = 8
    assert L.nbody_create(ctypes.byref(p), ctypes.byref(h)) == nb._lib.ERR_INVALID
    assert b"struct_size" in L.nbody_last_error(None)
    assert not h.value
    L.nbody_destroy(None)      # like `delete NULL` in CleanParticles (OctreeSearch.cpp:94)


def test_no_device_fails_loudly_not_silently(nb):
    if nb.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(nb.NBodyError) as e:
        nb.NBodyEngine(1024)
    assert e.value.code == nb._lib.ERR_NO_DEVICE and "no CPU path" in str(e.value)
    a = nb.OctreeSearch()
    a.CreateSpacePoints(100, 1000.0)
    assert not a.Initialized and a.LastStatus == nb._lib.ERR_NO_DEVICE
    a.Tick(0.016)              # silent guard, like the reference's `if (!Initialized) return;`
    a.CleanParticles()


def test_product_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "parallelnbody_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp")) or f == "Makefile":
                assert "oracle" not in open(os.path.join(dirpath, f)).read().lower(), os.path.join(dirpath, f)


def test_reference_box_generator(nb):
    # CreateSpacePoints: OctreeSearch.cpp:58-72
    posm, vel = nb.ic_reference_box(2000, 1000.0, seed=5)
    assert posm.shape == (2000, 4) and vel.shape == (2000, 4)
    np.testing.assert_array_equal(posm[0], [0, 0, 0, 5000])
    np.testing.assert_array_equal(vel[0], [0, 0, 0, 0])
    assert np.abs(posm[1:, 0]).max() <= 1000 and np.abs(posm[1:, 1]).max() <= 1000 and np.abs(posm[1:, 2]).max() <= 100
    assert np.abs(posm[1:, 2]).max() > 90
    sp = np.linalg.norm(vel[1:, :3], axis=1)
    assert sp.min() >= 250 - 1e-3 and sp.max() <= 500 + 1e-3
    assert 1 <= posm[1:, 3].min() and posm[1:, 3].max() <= 5000
    assert abs(vel[1:, :3].mean()) < 30          # isotropic directions
    p2, v2 = nb.ic_reference_box(2000, 1000.0, seed=5)
    assert posm.tobytes() == p2.tobytes() and vel.tobytes() == v2.tobytes()
    p3, _ = nb.ic_reference_box(2000, 1000.0, seed=6)
    assert posm.tobytes() != p3.tobytes()
    assert len(np.unique(posm[:, :3], axis=0)) == 2000   # no duplicates: the reference's tree cannot hold them


def test_plummer_generator_is_virialised(nb, oracle):
    n, M, a, G = 4096, 1000.0, 100.0, 1.0e4
    posm, vel = nb.ic_plummer(n, M, a, G, seed=11)
    assert np.allclose(posm[:, 3], M / n)
    assert np.abs(posm[:, :3].mean(0)).max() < 1e-3 and np.abs(vel[:, :3].mean(0)).max() < 1e-3
    r = np.linalg.norm(posm[:, :3], axis=1)
    assert np.median(r) == pytest.approx(1.30 * a, rel=0.08)      # Plummer half-mass radius = 1.305 a
    ke, pe = oracle.energy_f64(posm[:, :3], vel[:, :3], posm[:, 3], g=G, nthreads=4)
    assert -2 * ke / pe == pytest.approx(1.0, abs=0.08)           # virial ratio
    assert pe == pytest.approx(-3 * np.pi / 32 * G * M * M / a, rel=0.08)
    p2, _ = nb.ic_plummer(n, M, a, G, seed=11)
    assert posm.tobytes() == p2.tobytes()


def test_generators_reject_bad_arguments(nb):
    with pytest.raises(nb.NBodyError):
        nb.ic_plummer(0)
    with pytest.raises(nb.NBodyError):
        nb.ic_reference_box(10, -1.0)

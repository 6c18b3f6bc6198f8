"""The equal-mass form of the fp32 symmetric kernel (forces_sym_pk_kernel, UNI; nbody_equal_mass_form): when every body
has the same mass the pair loop carries no mass factor and the common G m is applied once per body.  Same pair law
(OctreeSearch.h:101-104) — so the checker is the same oracle at the same tolerance — and the choice must follow the
masses wherever they come from: the host's upload, a checkpoint, a write through a device pointer handed out."""
import ctypes
import os

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL_ACC = 2e-5          # asserted;  stated contract 1e-4 (tests/test_parity_gpu.py)


def scene(n, seed, equal=True):
    rng = np.random.default_rng(seed)
    posm = np.concatenate([rng.uniform(-500, 500, (n, 3)),
                           np.full((n, 1), 37.5) if equal else rng.uniform(1, 5000, (n, 1))], 1).astype(np.float32)
    vel = np.concatenate([rng.uniform(-5, 5, (n, 3)), np.zeros((n, 1))], 1).astype(np.float32)
    return posm, vel


@pytest.mark.parametrize("eps", [0.0, 0.05])
@pytest.mark.parametrize("precision,ipt", [("f32", 2), ("f32", 4), ("f32", 8), ("f32", 16), ("f32_kahan", 2), ("f32_kahan", 4),
                                           ("f32_kahan", 8), ("f64", 2), ("f64", 4)])
def test_equal_masses_against_the_oracle(nb, oracle, precision, ipt, eps):
    n = 5000                                           # ragged: the last i-set is mostly zero-mass padding
    posm, vel = scene(n, 100 + ipt)
    with nb.NBodyEngine(n, algorithm=2, i_per_thread=ipt, precision=precision, eps=eps) as e:
        e.set_state(posm, vel)
        e.compute_forces()
        assert e.equal_mass_form()
        a = e.accelerations(np.float64 if precision == "f64" else np.float32)
    ref = oracle.forces_direct_f64(posm[:, :3].astype(np.float64), posm[:, 3].astype(np.float64), eps=eps)
    assert rel_err(a[:, :3], ref).max() < (1e-12 if precision == "f64" else TOL_ACC)


@pytest.mark.parametrize("eps", [0.0, 0.05])
@pytest.mark.parametrize("precision,ipt", [("f32", 2), ("f32", 4), ("f32_kahan", 2), ("f32_kahan", 4)])
@pytest.mark.parametrize("n", [9000, 33000])
def test_one_sided_kernel_equal_masses_against_the_oracle(nb, oracle, n, precision, ipt, eps):
    """The packed one-sided kernel has the same form behind a wave-uniform branch (ragged tiles pad far away then);
    N = 33000 also runs the coincident-body detector and with it the unguarded tiles."""
    posm, vel = scene(n, n + ipt)
    with nb.NBodyEngine(n, algorithm=1, i_per_thread=ipt, precision=precision, eps=eps) as e:
        assert e.launch_config()["kernel"] == "forces_tile_pk_kernel"
        e.set_state(posm, vel)
        e.compute_forces()
        assert e.equal_mass_form()
        a = e.accelerations()
        mixed, _ = scene(n, 5, equal=False)
        e.set_state(mixed, vel)
        e.compute_forces()
        assert not e.equal_mass_form()
        b = e.accelerations()
    sample = np.concatenate([np.arange(0, n, n // 16), [n - 1]])
    for acc, pm in ((a, posm), (b, mixed)):
        ref = np.concatenate([oracle.forces_direct_f64(pm[:, :3].astype(np.float64), pm[:, 3].astype(np.float64), eps=eps, i0=int(i), i1=int(i) + 1)
                              for i in sample])
        assert rel_err(acc[sample, :3], ref).max() < TOL_ACC


def test_one_sided_sharded_contexts_agree_bit_for_bit_in_the_equal_mass_form(nb):
    """TILED's promise — every bit independent of how many GPUs share the bodies — holds in the equal-mass form too."""
    n = 12288
    posm, vel = nb.ic_plummer(n, seed=3)
    with nb.NBodyEngine(n, algorithm=1) as one:
        one.set_state(posm, vel)
        one.compute_forces()
        assert one.equal_mass_form()
        whole = one.accelerations()
    parts = []
    for r in range(3):
        with nb.NBodyEngine(n, i_begin=r * 4096, i_count=4096, algorithm=1) as e:
            e.set_state(posm, vel)
            e.compute_forces()
            assert e.equal_mass_form()
            parts.append(e.accelerations())
    np.testing.assert_array_equal(np.concatenate(parts), whole)


def test_one_sided_kernel_sees_a_mass_changed_through_the_device_pointer(nb, oracle):
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    n = 10000
    posm, vel = scene(n, 8)
    with nb.NBodyEngine(n, algorithm=1) as e:
        e.set_state(posm, vel)
        e.step(0.01, 2)
        assert e.equal_mass_form()
        ptr, nbytes = e.device_ptr(nb.BUF_POSM)
        e.step(0.01, 1)
        assert e.equal_mass_form()                        # mass_check_kernel runs before every pass now, and agrees
        cur = e.state()[0].copy()
        cur[123, 3] *= np.float32(2.0)
        assert hip.hipMemcpy(ptr, cur.ctypes.data, nbytes, 1) == 0
        e.compute_forces()
        assert not e.equal_mass_form()
        sample = np.array([0, 123, n - 1])
        ref = np.concatenate([oracle.forces_direct_f64(cur[:, :3].astype(np.float64), cur[:, 3].astype(np.float64), i0=int(i), i1=int(i) + 1)
                              for i in sample])
        assert rel_err(e.accelerations()[sample, :3], ref).max() < TOL_ACC


def test_fp64_equal_and_general_forms_step_alike(nb, oracle):
    """configs[3]'s kernel at a small size: three steps in the equal-mass form against the general form forced on the same
    scene; a body out where the far-away padding sits sends the scene to the general kernels."""
    n = 24576
    posm, vel = nb.ic_plummer(n, seed=11)
    with nb.NBodyEngine(n, precision="f64") as uni:
        os.environ["NBODY_SYM_NO_UNI"] = "1"
        try:
            gen = nb.NBodyEngine(n, precision="f64")
        finally:
            del os.environ["NBODY_SYM_NO_UNI"]
        with gen:
            for e in (uni, gen):
                assert e.launch_config()["kernel"] == "forces_sym_f64_kernel"
                e.set_state(posm, vel)
                e.step(0.01, 3)
            assert uni.equal_mass_form() and not gen.equal_mass_form()
            pu, vu, au = uni.state(np.float64); pg, vg, ag = gen.state(np.float64)
            np.testing.assert_allclose(pu[:, :3], pg[:, :3], rtol=1e-11, atol=1e-11)
            np.testing.assert_allclose(au[:, :3], ag[:, :3], rtol=1e-9, atol=1e-9 * np.abs(ag).max())
            far = pu.copy()
            far[5, :3] = 0.95e120
            uni.set_state(far, vu)
            uni.compute_forces()
            assert not uni.equal_mass_form()
            a = uni.accelerations(np.float64)
            assert np.all(np.isfinite(a))
            np.testing.assert_array_equal(a[5, :3], 0.0)          # too far away to feel or exert anything
    sample = np.array([0, 5, 77, n - 1])
    near = np.delete(far, 5, axis=0)
    for i in (0, 77):
        ref = oracle.forces_direct_f64(near[:, :3], near[:, 3], i0=(i if i < 5 else i - 1), i1=(i if i < 5 else i - 1) + 1)
        assert rel_err(a[i:i + 1, :3], ref).max() < 1e-12


def _general_engine(nb, n):
    """A context kept on the general form of the kernels (NBODY_SYM_NO_UNI is read when the context is created)."""
    os.environ["NBODY_SYM_NO_UNI"] = "1"
    try:
        return nb.NBodyEngine(n)
    finally:
        del os.environ["NBODY_SYM_NO_UNI"]


@pytest.mark.parametrize("n", [24576, 65536])
def test_equal_and_general_forms_agree_on_the_same_positions(nb, oracle, n):
    """ONE force pass each on identical positions, the library's own choice of kernel at these sizes: the equal-mass form
    against the general form forced on the same scene.  Same pairs, same summation order; the two differ by the rounding
    of the mass factor (inside the loop / once per body), so they must agree far inside the oracle tolerance."""
    posm, vel = nb.ic_plummer(n, seed=n)
    with nb.NBodyEngine(n) as uni, _general_engine(nb, n) as gen:
        for e in (uni, gen):
            e.set_state(posm, vel)
            e.compute_forces()
        assert uni.equal_mass_form() and not gen.equal_mass_form()
        assert uni.launch_config()["algorithm"] == "symmetric"
        au, ag = uni.accelerations(), gen.accelerations()
    assert rel_err(au[:, :3], ag[:, :3]).max() < 5e-6
    sample = np.arange(0, n, n // 32)
    p64 = posm.astype(np.float64)
    ref = np.concatenate([oracle.forces_direct_f64(p64[:, :3], p64[:, 3], i0=int(i), i1=int(i) + 1) for i in sample])
    assert rel_err(au[sample, :3], ref).max() < TOL_ACC
    assert rel_err(ag[sample, :3], ref).max() < TOL_ACC


@pytest.mark.parametrize("n", [24576, 65536])
def test_equal_and_general_forms_step_alike(nb, oracle, n):
    """Three steps (fused update) in both forms: a TRAJECTORY bound — after the first step the two runs stand on positions
    that differ in the last bits, so their accelerations are no longer those of one state (the force comparison proper is
    the test above); positions stay together to 1e-3 of a scene 100 units across."""
    posm, vel = nb.ic_plummer(n, seed=n)
    with nb.NBodyEngine(n) as uni, _general_engine(nb, n) as gen:
        for e in (uni, gen):
            e.set_state(posm, vel)
            e.step(0.01, 3)
        assert uni.equal_mass_form() and not gen.equal_mass_form()
        pu, vu, au = uni.state(); pg, vg, ag = gen.state()
    np.testing.assert_allclose(pu[:, :3], pg[:, :3], rtol=0, atol=1e-3)
    np.testing.assert_allclose(vu[:, :3], vg[:, :3], rtol=0, atol=1e-2 * np.abs(vg[:, :3]).max())
    # accelerations in the state belong to the positions BEFORE the last update: recompute on the final positions
    sample = np.arange(0, n, n // 32)
    ref = np.concatenate([oracle.forces_direct_f64(pg[:, :3].astype(np.float64), pg[:, 3].astype(np.float64), i0=int(i), i1=int(i) + 1)
                          for i in sample])
    with nb.NBodyEngine(n) as again:
        again.set_state(pg, vg)
        again.compute_forces()
        assert again.equal_mass_form()
        assert rel_err(again.accelerations()[sample, :3], ref).max() < TOL_ACC


def test_the_choice_follows_every_new_state(nb, oracle, tmp_path):
    n = 6000
    same, vel = scene(n, 1)
    mixed, _ = scene(n, 2, equal=False)
    with nb.NBodyEngine(n, algorithm=2, i_per_thread=4) as e:
        for posm, want in ((same, True), (mixed, False), (same, True)):
            e.set_state(posm, vel)
            e.compute_forces()
            assert e.equal_mass_form() == want
            ref = oracle.forces_direct_f64(posm[:, :3].astype(np.float64), posm[:, 3].astype(np.float64))
            assert rel_err(e.accelerations()[:, :3], ref).max() < TOL_ACC
        # one body heavier than the rest is enough — the last one, in the ragged i-set
        odd = same.copy(); odd[n - 1, 3] = np.nextafter(odd[n - 1, 3], np.float32(np.inf))
        e.set_state(odd, vel)
        e.compute_forces()
        assert not e.equal_mass_form()
        # a checkpoint brings its own masses
        e.set_state(same, vel); e.step(0.01, 2)
        path = str(tmp_path / "equal.ckpt")
        e.save_checkpoint(path)
        p_saved = e.state()[0]
        e.set_state(mixed, vel); e.compute_forces()
        assert not e.equal_mass_form()
        e.load_checkpoint(path)
        e.compute_forces()
        assert e.equal_mass_form()
        ref = oracle.forces_direct_f64(p_saved[:, :3].astype(np.float64), p_saved[:, 3].astype(np.float64))
        assert rel_err(e.accelerations()[:, :3], ref).max() < TOL_ACC


@pytest.mark.parametrize("n", [6000, 40000])
def test_a_mass_changed_through_the_device_pointer_is_seen(nb, oracle, n):
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    posm, vel = scene(n, 3)
    kw = dict(algorithm=2, i_per_thread=4) if n < 20480 else {}
    with nb.NBodyEngine(n, **kw) as e:
        e.set_state(posm, vel)
        e.step(0.01, 2)
        assert e.equal_mass_form()
        ptr, nbytes = e.device_ptr(nb.BUF_POSM)
        e.step(0.01, 1)                                # both forms are launched now; the device's finding picks
        assert e.equal_mass_form()
        cur = e.state()[0].copy()
        cur[n // 2, 3] *= np.float32(3.0)
        assert hip.hipMemcpy(ptr, cur.ctypes.data, nbytes, 1) == 0
        e.compute_forces()
        assert not e.equal_mass_form()
        sample = np.array([0, 7, n // 2, n - 1])
        ref = np.concatenate([oracle.forces_direct_f64(cur[:, :3].astype(np.float64), cur[:, 3].astype(np.float64), i0=int(i), i1=int(i) + 1)
                              for i in sample])
        assert rel_err(e.accelerations()[sample, :3], ref).max() < TOL_ACC
        # put the mass back: the finding is sticky until the host uploads a state (correct either way)
        cur[n // 2, 3] = posm[0, 3]
        assert hip.hipMemcpy(ptr, cur.ctypes.data, nbytes, 1) == 0
        e.compute_forces()
        assert not e.equal_mass_form()
        ref = np.concatenate([oracle.forces_direct_f64(cur[:, :3].astype(np.float64), cur[:, 3].astype(np.float64), i0=int(i), i1=int(i) + 1)
                              for i in sample])
        assert rel_err(e.accelerations()[sample, :3], ref).max() < TOL_ACC
        e.set_state(cur, vel)
        e.compute_forces()
        assert e.equal_mass_form()


def test_equal_masses_with_coincident_bodies_and_a_body_at_the_origin(nb, oracle):
    """d == 0 between different bodies selects the guarded kernel — also in the equal-mass form; the reference pins body 0
    at the origin (OctreeSearch.cpp:68-70), where no padding body may sit."""
    n = 4500
    posm, vel = scene(n, 4)
    posm[0, :3] = 0.0
    posm[4000, :3] = posm[17, :3]
    with nb.NBodyEngine(n, algorithm=2, i_per_thread=8) as e:
        e.set_state(posm, vel)
        e.compute_forces()
        assert e.equal_mass_form()
        a = e.accelerations()
    assert np.all(np.isfinite(a))
    ref = oracle.forces_direct_f64(posm[:, :3].astype(np.float64), posm[:, 3].astype(np.float64))
    assert rel_err(a[:, :3], ref).max() < TOL_ACC


def test_equal_mass_form_is_not_used_where_it_does_not_apply(nb):
    n = 4096
    posm, vel = scene(n, 5)
    for kw in (dict(algorithm=2, i_per_thread=4, zero_mode=2), dict(algorithm=1, zero_mode=1), dict(algorithm=1, zero_mode=2),
               dict(algorithm=1, precision="f64"), dict(algorithm=1, i_per_thread=1), dict(theta=1.0)):
        with nb.NBodyEngine(n, **kw) as e:
            e.set_state(posm, vel)
            e.compute_forces()
            assert not e.equal_mass_form()


def test_massless_bodies_all_alike(nb):
    n = 3000
    posm, vel = scene(n, 6)
    posm[:, 3] = 0.0
    with nb.NBodyEngine(n, algorithm=2, i_per_thread=4) as e:
        e.set_state(posm, vel)
        e.compute_forces()
        assert e.equal_mass_form()
        np.testing.assert_array_equal(e.accelerations()[:, :3], np.zeros((n, 3), np.float32))

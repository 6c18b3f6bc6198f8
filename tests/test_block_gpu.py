"""forces_block_pk_kernel (csrc/kernels_block.hip): the one-launch step of small and mid-size fp32 systems, through the
C-ABI, against the CPU oracle (OctreeSearch.h:101-104 summed over all j; OctreeSearch.cpp:28-31 for the update).

Tolerance as in test_parity_gpu.py: per-body |a_gpu - a_oracle| / |a_oracle| <= 2e-5 asserted (1e-4 stated)."""
import os

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL_ACC = 2e-5
BLOCK = "forces_block_pk_kernel"


def scene(n, seed, equal=False, box=500.0):
    rng = np.random.default_rng(seed)
    posm = np.concatenate([rng.uniform(-box, box, (n, 3)), rng.uniform(1, 5000, (n, 1))], 1).astype(np.float32)
    if equal:
        posm[:, 3] = np.float32(37.5)
    if n > 3:
        posm[0, :3] = 0.0                                # the shipped scene pins body 0 at the origin
    vel = np.concatenate([rng.uniform(-5, 5, (n, 3)), np.zeros((n, 1))], 1).astype(np.float32)
    return posm, vel


def sampled(n, rng, k=24):
    fixed = [0, 1, 255, 256, 257, n // 2, n - 2, n - 1]
    return sorted(set(int(i) for i in fixed if 0 <= i < n) | set(int(i) for i in rng.choice(n, min(n, k), replace=False)))


@pytest.mark.parametrize("eps", [0.0, 0.7])
@pytest.mark.parametrize("equal", [False, True])
@pytest.mark.parametrize("n", [2560, 3001, 4096, 5000, 8192, 8200, 12289, 16384])
def test_block_kernel_forces_match_the_oracle(nb, oracle, n, equal, eps):
    posm, vel = scene(n, n + 17, equal)
    with nb.NBodyEngine(n, eps=eps) as e:
        assert e.launch_config()["kernel"] == BLOCK
        e.set_state(posm, vel)
        e.compute_forces()
        assert bool(e.equal_mass_form()) == equal
        a = e.accelerations()
    assert np.all(np.isfinite(a))
    p64 = posm.astype(np.float64)
    for i in sampled(n, np.random.default_rng(n)):
        ref = oracle.forces_direct_f64(p64[:, :3], p64[:, 3], eps=eps, i0=i, i1=i + 1)[0]
        assert np.linalg.norm(a[i, :3] - ref) / np.linalg.norm(ref) < TOL_ACC, (n, equal, eps, i)


@pytest.mark.parametrize("n", [2560, 6000, 8192, 10000])
def test_block_kernel_against_the_reference_arithmetic_on_every_body(nb, oracle, n):
    # every body against the oracle's fp32 direct sum (d in fp32, scale factor in double: OctreeSearch.h:101-104)
    posm, vel = scene(n, n + 3)
    with nb.NBodyEngine(n) as e:
        e.set_state(posm, vel)
        e.compute_forces()
        a = e.accelerations()
    ref = oracle.forces_direct_f32(posm[:, :3], posm[:, 3])
    assert rel_err(a, ref).max() < TOL_ACC


@pytest.mark.parametrize("equal", [False, True])
@pytest.mark.parametrize("n", [3000, 8192, 9000])
def test_coincident_bodies_send_a_workgroup_through_the_guarded_walk(nb, oracle, n, equal):
    """OctreeSearch.h:102: `if (d == 0) return`.  The kernel first bets that only self pairs have d == 0; a workgroup whose
    sums come out non-finite walks again with the guard.  The stored bits must be those of a pass that guards every pair."""
    posm, vel = scene(n, n + 29, equal)
    posm[17, :3] = posm[n - 400, :3]                     # different groups of j-bodies
    posm[n - 1, :3] = posm[3, :3]
    posm[600, :3] = posm[601, :3]                        # neighbours: same workgroup, same group
    posm[1200, :3] = posm[0, :3]                         # on the origin, with body 0
    with nb.NBodyEngine(n) as e:
        assert e.launch_config()["kernel"] == BLOCK
        e.set_state(posm, vel)
        e.compute_forces()
        a = e.accelerations()
        os.environ["NBODY_SYM_GUARDED"] = "1"
        try:
            e.compute_forces()
            g = e.accelerations()
        finally:
            del os.environ["NBODY_SYM_GUARDED"]
    assert np.all(np.isfinite(a))
    np.testing.assert_array_equal(a, g)
    ref = oracle.forces_direct_f32(posm[:, :3], posm[:, 3])
    assert rel_err(a, ref).max() < TOL_ACC


def test_underflowing_separations_and_a_body_on_the_padding_point(nb, oracle):
    # two DIFFERENT positions whose squared distance underflows to 0 in fp32 (only possible next to the origin): the
    # reference's d == 0 test skips the pair; and a body exactly where the kernel parks its far-away padding
    n = 5003                                             # ragged: the last group is padded
    posm, vel = scene(n, 91)
    posm[0, :3] = 0.0
    posm[3500, :3] = (1e-30, 0.0, -1e-31)
    posm[n - 1, :3] = np.float32(1.0e30)
    posm[n - 1, 3] = 0.0
    with nb.NBodyEngine(n) as e:
        e.set_state(posm, vel)
        e.compute_forces()
        a = e.accelerations()
    assert np.all(np.isfinite(a))
    ref = oracle.forces_direct_f32(posm[:n - 1, :3], posm[:n - 1, 3])
    assert rel_err(a[:n - 1], ref).max() < TOL_ACC


@pytest.mark.parametrize("equal", [False, True])
@pytest.mark.parametrize("n", [2560, 8192, 12000])
def test_one_launch_step_is_the_reference_update_of_its_own_acceleration(nb, oracle, n, equal):
    # OctreeSearch.cpp:29-30 with separate multiply and add: given the device's own acc, v and x match the oracle's
    # kick-drift bit for bit; and a second step starts from the swapped buffer
    posm, vel = scene(n, n + 5, equal)
    with nb.NBodyEngine(n) as e:
        e.set_state(posm, vel)
        e.step(0.01, 1)
        p, v, a = e.state()
        p1, v1 = oracle.kick_drift_f32(posm[:, :3], vel[:, :3], a[:, :3], 0.01)
        np.testing.assert_array_equal(v[:, :3], v1)
        np.testing.assert_array_equal(p[:, :3], p1)
        np.testing.assert_array_equal(p[:, 3], posm[:, 3])
        e.step(0.01, 1)
        p2, v2, a2 = e.state()
        q2, w2 = oracle.kick_drift_f32(p[:, :3], v[:, :3], a2[:, :3], 0.01)
        np.testing.assert_array_equal(v2[:, :3], w2)
        np.testing.assert_array_equal(p2[:, :3], q2)
        e.compute_forces()                               # the same positions again, without the update
        np.testing.assert_array_equal(e.accelerations(), e.accelerations())
    i = n // 3
    p64 = p.astype(np.float64)
    ref = oracle.forces_direct_f64(p64[:, :3], p64[:, 3], i0=i, i1=i + 1)[0]
    assert np.linalg.norm(a2[i, :3] - ref) / np.linalg.norm(ref) < TOL_ACC


@pytest.mark.parametrize("equal", [False, True])
@pytest.mark.parametrize("n", [4096, 8192])
def test_a_position_pointer_handed_out_keeps_the_trajectory(nb, n, equal):
    """Once nbody_device_ptr(NBODY_BUF_POSM) is out the context may not swap its buffers any more (two launches per step,
    and the device looks at the masses itself): the trajectory is the one-launch path's, bit for bit."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    posm, vel = scene(n, n + 7, equal)
    with nb.NBodyEngine(n) as held, nb.NBodyEngine(n) as private:
        for e in (held, private):
            e.set_state(posm, vel)
            e.step(0.01, 3)
        ptr, nbytes = held.device_ptr(nb.BUF_POSM)
        for e in (held, private):
            e.step(0.01, 4)
        held.synchronize()
        assert bool(held.equal_mass_form()) == equal and bool(private.equal_mass_form()) == equal
        seen = np.empty((n, 4), np.float32)
        assert hip.hipMemcpy(seen.ctypes.data, ptr, nbytes, 2) == 0
        np.testing.assert_array_equal(seen, held.state()[0])
        np.testing.assert_array_equal(seen, private.state()[0])
        np.testing.assert_array_equal(held.state()[1], private.state()[1])
        # the caller changes a mass through the pointer: the next pass must notice
        seen[9, 3] *= np.float32(3.0)
        assert hip.hipMemcpy(ptr, seen.ctypes.data, nbytes, 1) == 0
        held.compute_forces()
        assert not held.equal_mass_form()
        private.set_state(seen, private.state()[1])
        private.compute_forces()
        np.testing.assert_array_equal(held.accelerations(), private.accelerations())


@pytest.mark.parametrize("equal", [False, True])
def test_slices_of_a_block_kernel_system_reproduce_the_single_context(nb, equal):
    # range partition: each context owns a slice and sees all positions; a body's sum does not depend on
    # which other bodies share its workgroup, so the bits are the single context's
    n = 6001
    posm, vel = scene(n, 77, equal)
    with nb.NBodyEngine(n) as e:
        assert e.launch_config()["kernel"] == BLOCK
        e.set_state(posm, vel)
        e.step(0.01, 1)
        p_all, v_all, a_all = e.state()
    cuts = [0, 1999, 4001, n]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        with nb.NBodyEngine(n, i_begin=lo, i_count=hi - lo) as e:
            assert e.launch_config()["kernel"] == BLOCK
            e.set_state(posm, vel)
            e.step(0.01, 1)
            p, v, a = e.state()
        np.testing.assert_array_equal(a, a_all[lo:hi])
        np.testing.assert_array_equal(p, p_all[lo:hi])
        np.testing.assert_array_equal(v, v_all[lo:hi])


def test_energy_drift_of_the_one_launch_step(nb):
    # north star: total-energy drift < 1e-4 over 1k steps (softened Plummer sphere), as test_energy_drift_1k_steps
    from test_parity_gpu import _drift
    worst, kernel = _drift(nb, 8192, "f32", 1000, 250)
    assert kernel == BLOCK
    assert worst < 1e-4, worst


def test_bits_do_not_depend_on_how_many_bodies_share_a_workgroup(nb):
    # the register pairs per workgroup follow the CU count (block_pairs, csrc/capi.hip); no sum may depend on them
    n = 7001
    posm, vel = scene(n, 123)
    ref = None
    for np_ in (2, 3, 4, 5, 6, 7, 8):
        os.environ["NBODY_BLOCK_NP"] = str(np_)
        try:
            with nb.NBodyEngine(n) as e:
                assert e.launch_config()["blocks"] == (n + 2 * np_ - 1) // (2 * np_)
                e.set_state(posm, vel)
                e.step(0.01, 2)
                state = e.state()
        finally:
            del os.environ["NBODY_BLOCK_NP"]
        if ref is None:
            ref = state
        else:
            for a, b in zip(state, ref):
                np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("pinned", [False, True])
@pytest.mark.parametrize("n", [2000, 8192, 12001])
def test_tick_in_one_launch_is_bounds_plus_step_plus_mirror(nb, n, pinned):
    """nbody_tick on the one-launch step: Size (of the positions BEFORE the update, OctreeSearch.cpp:26, 47-56) and the
    frame's FParticle records come out of the same launch — the same bytes as bounds, step, mirror as three calls; mixed
    with plain steps, with frames that ask for the mirror only, and through the raw C entry point for Size only."""
    import ctypes
    posm, vel = scene(n, n + 41)
    with nb.NBodyEngine(n) as a, nb.NBodyEngine(n) as b:
        assert b.launch_config()["kernel"] == BLOCK
        a.set_state(posm, vel); b.set_state(posm, vel)
        mine = np.zeros(n, nb.PARTICLE_DTYPE)
        if pinned:
            b.pin(mine)
        for frame in range(5):
            size_a = a.bounds(); a.step(0.01, 1); pa = a.particles()
            size_b, pb = b.tick(0.01, out=mine)
            assert size_b == size_a and pb.tobytes() == pa.tobytes(), frame
            if frame == 2:                                       # plain steps in between leave the two Size words alone
                a.step(0.01, 3); b.step(0.01, 3)
        # Size only (no mirror), through the C entry point
        size_a = a.bounds(); a.step(0.01, 1)
        size = ctypes.c_float(-1.0)
        assert b._L.nbody_tick(b._h, ctypes.c_float(0.01), ctypes.byref(size), None, 0) == 0
        assert size.value == size_a
        np.testing.assert_array_equal(a.state()[0], b.state()[0])
        # mirror only
        a.step(0.01, 1); pa = a.particles()
        assert b._L.nbody_tick(b._h, ctypes.c_float(0.01), None, mine.ctypes.data, nb.PARTICLE_DTYPE.itemsize) == 0
        assert mine.tobytes() == pa.tobytes()
        # and Size again: the word that was cleared two frames ago
        size_a = a.bounds(); a.step(0.01, 1); pa = a.particles()
        size_b, pb = b.tick(0.01, out=mine)
        assert size_b == size_a and pb.tobytes() == pa.tobytes()
        assert a.steps_done() == b.steps_done() == 11


# ---- forces_block_kernel: the same idea in the other two precisions (small systems) ------------------------------------

GENERIC = "forces_block_kernel"


@pytest.mark.parametrize("eps", [0.0, 0.7])
@pytest.mark.parametrize("prec", ["f32_kahan", "f64"])
@pytest.mark.parametrize("n", [1, 2, 255, 1000, 2001, 5000, 6655])
def test_block_kernel_in_the_other_precisions_matches_the_oracle(nb, oracle, n, prec, eps):
    dt = np.float64 if prec == "f64" else np.float32
    posm, vel = scene(n, n + 61)
    posm, vel = posm.astype(dt), vel.astype(dt)
    with nb.NBodyEngine(n, precision=prec, eps=eps) as e:
        assert e.launch_config()["kernel"] == GENERIC
        e.set_state(posm, vel)
        e.compute_forces()
        assert not e.equal_mass_form()
        a = e.accelerations(np.float64)
    p64 = posm.astype(np.float64)
    ref = oracle.forces_direct_f64(p64[:, :3], p64[:, 3], eps=eps)
    if n == 1:
        assert np.all(a == 0)
    else:
        # fp64: the oracle's own order of additions differs; compensated fp32: the terms themselves are fp32 (v_rsq_f32),
        # the sum of the lanes' compensated sums is formed in double and rounded once
        assert rel_err(a[:, :3], ref).max() < (1e-12 if prec == "f64" else 5e-6)


@pytest.mark.parametrize("prec", ["f32_kahan", "f64"])
def test_other_precisions_one_launch_step_update_pointer_slices_and_coincident_bodies(nb, oracle, prec):
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    dt = np.float64 if prec == "f64" else np.float32
    n = 3001
    posm, vel = scene(n, 71)
    posm[17, :3] = posm[n - 400, :3]; posm[1200, :3] = posm[0, :3]          # coincident pairs, one on the origin (d == 0 skips)
    posm, vel = posm.astype(dt), vel.astype(dt)
    kick = oracle.kick_drift_f64 if prec == "f64" else oracle.kick_drift_f32
    with nb.NBodyEngine(n, precision=prec) as e, nb.NBodyEngine(n, precision=prec) as held:
        assert e.launch_config()["kernel"] == GENERIC
        for c in (e, held):
            c.set_state(posm, vel)
        held.device_ptr(nb.BUF_POSM)                                       # two launches per step from here on
        e.step(0.01, 1)
        p, v, a = e.state(dt)
        assert np.all(np.isfinite(a))
        # the reference's update of the device's own acc (PhDeltaTime is a float: the C-ABI takes dt as float)
        p1, v1 = kick(posm[:, :3], vel[:, :3], a[:, :3], float(np.float32(0.01)))
        np.testing.assert_array_equal(v[:, :3], v1)
        np.testing.assert_array_equal(p[:, :3], p1)
        ref = oracle.forces_direct_f64(posm[:, :3].astype(np.float64), posm[:, 3].astype(np.float64))
        assert rel_err(a[:, :3], ref).max() < (1e-12 if prec == "f64" else 5e-6)
        e.step(0.01, 4); held.step(0.01, 5)
        for x, y in zip(e.state(dt), held.state(dt)):
            np.testing.assert_array_equal(x, y)                            # the same trajectory either way
        p_all, v_all, a_all = e.state(dt)
    with nb.NBodyEngine(n, precision=prec) as e:                           # slices reproduce the single context
        e.set_state(posm, vel); e.step(0.01, 1); p_one, v_one, a_one = e.state(dt)
    for lo, hi in ((0, 1000), (1000, 2999), (2999, n)):
        with nb.NBodyEngine(n, precision=prec, i_begin=lo, i_count=hi - lo) as e:
            assert e.launch_config()["kernel"] == GENERIC
            e.set_state(posm, vel); e.step(0.01, 1)
            ps, vs, as_ = e.state(dt)
        np.testing.assert_array_equal(as_, a_one[lo:hi]); np.testing.assert_array_equal(ps, p_one[lo:hi])
        np.testing.assert_array_equal(vs, v_one[lo:hi])


def test_compensated_block_kernel_is_closer_to_fp64_than_the_plain_one(nb, oracle):
    n = 6000
    posm, vel = nb.ic_plummer(n, seed=9)
    ref = oracle.forces_direct_f64(posm[:, :3].astype(np.float64), posm[:, 3].astype(np.float64), eps=0.5)
    err = {}
    for prec in ("f32", "f32_kahan"):
        with nb.NBodyEngine(n, precision=prec, eps=0.5) as e:
            e.set_state(posm, vel); e.compute_forces()
            err[prec] = rel_err(e.accelerations(np.float64)[:, :3], ref)
    assert err["f32_kahan"].max() < 1e-6 and np.median(err["f32_kahan"]) < np.median(err["f32"])


@pytest.mark.parametrize("theta", [0.0, 1.0])
def test_tick_into_strided_and_misaligned_caller_records(nb, theta):
    # FParticle records 48 bytes apart, starting 8 bytes into a pinned buffer: the kernels' direct path (stride 40, 16-byte
    # aligned for the block kernel) does not apply; the records must arrive all the same
    import ctypes
    n = 2000
    posm, vel = scene(n, 5)
    with nb.NBodyEngine(n, theta=theta) as a, nb.NBodyEngine(n, theta=theta) as b:
        a.set_state(posm, vel); b.set_state(posm, vel)
        raw = np.zeros(n * 48 + 64, np.uint8)
        b.pin(raw)
        base = raw.ctypes.data + 8
        for frame in range(3):
            size_a = a.bounds(); a.step(0.01, 1); pa = a.particles()
            size = ctypes.c_float(-1.0)
            assert b._L.nbody_tick(b._h, ctypes.c_float(0.01), ctypes.byref(size), ctypes.c_void_p(base), 48) == 0
            assert size.value == size_a
            got = np.lib.stride_tricks.as_strided(raw[8:8 + 40].view(np.uint8), shape=(n, 40), strides=(48, 1))
            assert got.tobytes() == pa.tobytes(), frame
        # and 40-byte records at a 4-byte-aligned address inside the pinned range
        base2 = raw.ctypes.data + 4
        a.step(0.01, 1); pa = a.particles()
        assert b._L.nbody_tick(b._h, ctypes.c_float(0.01), None, ctypes.c_void_p(base2), 40) == 0
        assert raw[4:4 + n * 40].tobytes() == pa.tobytes()

"""The multi-rank geometries of BASELINE configs[2] and configs[4] at their own sizes, on one GPU.

What each of the 8 ranks of the sharded job launches — its rows of the work plan against all N bodies, the exchange rows,
the fold of what the other ranks contributed — is run here rank after rank on the one device of the test box, the
all-to-all staged through the host (the transport is the only thing that differs from the 8-GPU job), and sampled bodies
of every rank are compared with the oracle's fp64 direct sum of the pair law (OctreeSearch.h:101-104 over all j, the
driver loop OctreeSearch.cpp:83-86 at theta = 0)."""
import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

RANKS = 8


def _rank_samples(i_begin, i_count, bi, seed):
    """>= 8 owned bodies of a rank: both ends of the slice, both sides of the first and the last i-set boundary inside it,
    the middle, and random ones."""
    rng = np.random.default_rng(seed)
    pick = {i_begin, i_begin + i_count - 1, i_begin + i_count // 2}
    if bi < i_count:
        pick.update((i_begin + bi - 1, i_begin + bi, i_begin + i_count - bi - 1, i_begin + i_count - bi))
    pick.update(int(i) for i in i_begin + rng.choice(i_count, 5, replace=False))
    return sorted(pick)


def _run_ranks(nb, n, posm, vel, check, **kw):
    """One force pass of the whole system as RANKS sharded contexts would run it; `check(rank, engine)` is called on every
    context before any exchange.  Returns the accelerations of all bodies (rank slices concatenated) and the launch
    configuration (the same on every rank)."""
    ic = n // RANKS
    engs, cfg = [], None
    try:
        for r in range(RANKS):
            e = nb.NBodyEngine(n, i_begin=r * ic, i_count=ic, **kw)
            engs.append(e)
            c = e.launch_config()
            c.pop("blocks")                                       # the ranks' item counts may differ by a few strips
            cfg = cfg or c
            assert c == cfg                                       # equal GPUs arrive at equal geometries
            check(r, e)
            e.set_state(posm, vel)
            e.step_begin()
        if engs[0].exchange_ranks():
            assert all(e.exchange_ranks() == RANKS for e in engs)
            sends = [e.exchange_read_send() for e in engs]        # [n, 4] each: what rank r's pairs add to every body
            for r, e in enumerate(engs):
                e.exchange_write_recv(np.concatenate([sd[r * ic:(r + 1) * ic] for sd in sends]))
            del sends
        acc = []
        for e in engs:
            e.step_end(0.0)
            acc.append(e.accelerations())
    finally:
        for e in engs:
            e.close()
    return np.concatenate(acc), cfg


def test_config2_eight_rank_geometry_at_full_size(nb, oracle):
    # BASELINE configs[2]: N = 1048576 fp32 over 8 GPUs — each rank owns 131072 bodies = 8 i-sets of sixteen bodies per lane
    n = 1 << 20
    posm, vel = nb.ic_plummer(n, total_mass=1000.0, scale_radius=100.0, G=1.0e4, seed=20261003)

    def check(r, e):
        c = e.launch_config()
        assert c["algorithm"] == "symmetric" and c["kernel"] == "forces_sym_pk_kernel" and c["i_per_thread"] == 16

    a, cfg = _run_ranks(nb, n, posm, vel, check)
    assert np.all(np.isfinite(a))
    p64 = posm.astype(np.float64)
    ic = n // RANKS
    for r in range(RANKS):
        for i in _rank_samples(r * ic, ic, cfg["super_tile"], seed=100 + r):
            ref = oracle.forces_direct_f64(p64[:, :3], p64[:, 3], i0=int(i), i1=int(i) + 1, nthreads=8)
            assert rel_err(a[i:i + 1, :3], ref).max() < 2e-5, (r, i)
    f = (a[:, :3].astype(np.float64) * p64[:, 3:4]).sum(0)       # Newton's third law over the whole job
    assert np.linalg.norm(f) / (np.linalg.norm(a[:, :3], axis=1) * p64[:, 3]).sum() < 1e-6

    # the same bodies with distinct masses: the kernel's general form in the same geometry
    posm2 = posm.copy()
    posm2[:, 3] *= np.random.default_rng(20261004).uniform(0.5, 1.5, n).astype(np.float32)
    a2, _ = _run_ranks(nb, n, posm2, vel, check)
    p64 = posm2.astype(np.float64)
    for r in range(RANKS):
        for i in _rank_samples(r * ic, ic, cfg["super_tile"], seed=200 + r)[:8]:
            ref = oracle.forces_direct_f64(p64[:, :3], p64[:, 3], i0=int(i), i1=int(i) + 1, nthreads=8)
            assert rel_err(a2[i:i + 1, :3], ref).max() < 2e-5, (r, i)


def test_config4_eight_rank_kahan_geometry_at_full_size(nb, oracle):
    # BASELINE configs[4]: N = 2097152 softened Plummer, Kahan accumulation, 8 GPUs — 262144 bodies per rank, eight per lane
    n = 1 << 21
    eps = 0.5
    posm, vel = nb.ic_plummer(n, seed=21)

    def check(r, e):
        c = e.launch_config()
        assert c["algorithm"] == "symmetric" and c["kernel"] == "forces_sym_pk_kernel" and c["i_per_thread"] == 8

    a, cfg = _run_ranks(nb, n, posm, vel, check, precision="f32_kahan", eps=eps)
    assert np.all(np.isfinite(a))
    p64 = posm.astype(np.float64)
    ic = n // RANKS
    for r in range(RANKS):
        for i in _rank_samples(r * ic, ic, cfg["super_tile"], seed=300 + r)[:9]:
            ref = oracle.forces_direct_f64(p64[:, :3], p64[:, 3], eps=eps, i0=int(i), i1=int(i) + 1, nthreads=8)
            assert rel_err(a[i:i + 1, :3], ref).max() < 2e-6, (r, i)
    f = (a[:, :3].astype(np.float64) * p64[:, 3:4]).sum(0)
    assert np.linalg.norm(f) / (np.linalg.norm(a[:, :3], axis=1) * p64[:, 3]).sum() < 1e-7


def test_all_gather_only_step_eight_slices_equal_one_context_bit_for_bit(nb, oracle):
    # north_star's literal step (one-sided kernel, positions all-gathered, no other collective) at N = 1048576: the eight
    # slices' accelerations must be the single context's in every bit (j_split depends on n_total only), and sampled rows
    # must agree with the oracle
    n = 1 << 20
    posm, vel = nb.ic_plummer(n, total_mass=1000.0, scale_radius=100.0, G=1.0e4, seed=20261003)

    def check(r, e):
        assert e.launch_config()["algorithm"] == "tiled" and e.exchange_ranks() == 0

    a8, cfg = _run_ranks(nb, n, posm, vel, check, algorithm=1)
    with nb.NBodyEngine(n, algorithm=1) as one:
        assert one.launch_config()["j_split"] == cfg["j_split"]
        one.set_state(posm, vel)
        one.compute_forces()
        a1 = one.accelerations()
    ic = n // RANKS
    rows = sorted({i for r in range(RANKS) for i in _rank_samples(r * ic, ic, 1024, seed=400 + r)})
    np.testing.assert_array_equal(a8[rows], a1[rows])
    np.testing.assert_array_equal(a8, a1)                         # and every other row as well
    p64 = posm.astype(np.float64)
    for i in rows[::4]:
        ref = oracle.forces_direct_f64(p64[:, :3], p64[:, 3], i0=int(i), i1=int(i) + 1, nthreads=8)
        assert rel_err(a8[i:i + 1, :3], ref).max() < 2e-5, i


@pytest.mark.parametrize("precision,eps,n", [("f32", 0.0, 1 << 17), ("f32_kahan", 0.5, 1 << 17), ("f32", 0.0, 1 << 20)])
def test_force_pass_in_two_goes_equals_the_pass_in_one(nb, oracle, precision, eps, n):
    # nbody_step_begin_local + nbody_step_begin_remote (what lets the all-gather overlap the force pass) against
    # nbody_step_begin on the same sharded contexts: same plan, same segments, same sums — every bit, on every rank,
    # with distinct masses (general form) and with coincident bodies inside one slice and across slices (the own slice's
    # detector verdict guards the first go, the whole system's the second)
    posm, vel = nb.ic_plummer(n, seed=77)
    posm[:, 3] *= np.random.default_rng(3).uniform(0.5, 1.5, n).astype(np.float32)
    ic = n // RANKS
    variants = [posm]
    if n <= 1 << 17:
        twin = posm.copy(); twin[5, :3] = twin[900, :3]                    # two bodies of rank 0 on one point
        cross = posm.copy(); cross[7, :3] = cross[ic + 11, :3]             # one of rank 0, one of rank 1
        variants += [twin, cross]
    for scene in variants:
        results = []
        for goes in (1, 2):
            engs = [nb.NBodyEngine(n, i_begin=r * ic, i_count=ic, precision=precision, eps=eps) for r in range(RANKS)]
            try:
                for e in engs:
                    assert e.launch_config()["algorithm"] == "symmetric"
                    e.set_state(scene, vel)
                    if goes == 1:
                        e.step_begin()
                    else:
                        e.step_begin_local()
                        with pytest.raises(nb.NBodyError):
                            e.step_begin_local()                          # one first go per step
                        e.step_begin_remote()
                sends = [e.exchange_read_send() for e in engs]
                out = []
                for r, e in enumerate(engs):
                    e.exchange_write_recv(np.concatenate([sd[r * ic:(r + 1) * ic] for sd in sends]))
                    e.step_end(0.01)
                    out.append(np.concatenate(e.state(), axis=1))
                results.append((np.concatenate(out), np.stack(sends)))
            finally:
                for e in engs:
                    e.close()
        np.testing.assert_array_equal(results[0][1], results[1][1])        # what every rank sends
        np.testing.assert_array_equal(results[0][0], results[1][0])        # positions, velocities, accelerations
        a = results[0][0][:, 8:11]
        assert np.all(np.isfinite(a))
        if scene is posm:
            p64 = scene.astype(np.float64)
            for i in (0, ic - 1, ic, n // 2 + 3, n - 1):
                ref = oracle.forces_direct_f64(p64[:, :3], p64[:, 3], eps=eps, i0=i, i1=i + 1, nthreads=8)
                assert rel_err(a[i:i + 1], ref).max() < (2e-6 if precision == "f32_kahan" else 2e-5), i
    # contexts without a first go of their own (one device; the one-sided kernel; fp64) accept the two calls as well
    for kw in (dict(), dict(i_begin=0, i_count=n // 2, algorithm=1), dict(i_begin=0, i_count=n // 2, precision="f64")):
        if n > 1 << 17:
            break
        with nb.NBodyEngine(n, **kw) as a1, nb.NBodyEngine(n, **kw) as a2:
            for e in (a1, a2):
                e.set_state(posm, vel)
            a1.step_begin(); a2.step_begin_local(); a2.step_begin_remote()
            if a1.exchange_ranks():
                for e in (a1, a2):
                    e.exchange_write_recv(np.zeros((e.exchange_ranks() * e.i_count, 4), np.float64 if e.f64 else np.float32))
            a1.step_end(0.01); a2.step_end(0.01)
            for x, y in zip(a1.state(), a2.state()):
                np.testing.assert_array_equal(x, y)

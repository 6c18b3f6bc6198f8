"""The even-share plan of the symmetric force pass (csrc/sym_plan.h build_sym_plan_even; forces_sym_pk_kernel<..., EVEN>):
one workgroup per slot of the chip, every one the same cost — rows of the pair matrix cut at four steps of a subtile's 64,
items that run from the own block into the forward blocks and on past the system's last granule.  Same pair law
(OctreeSearch.h:101-104), same `d == 0` rule (.h:102), same update (OctreeSearch.cpp:28-31) as every other all-pairs kernel
of the library, so the checker is the same oracle at the same tolerance; the guided plan of the same context is the second
witness (same pairs, another order of summation)."""
import os

import numpy as np
import pytest

from conftest import rel_err

pytestmark = pytest.mark.gpu

TOL_ACC = 2e-5          # asserted;  stated contract 1e-4 (tests/test_parity_gpu.py)


def scene(n, seed, equal=False):
    rng = np.random.default_rng(seed)
    posm = np.concatenate([rng.uniform(-500, 500, (n, 3)),
                           np.full((n, 1), 37.5) if equal else rng.uniform(1, 5000, (n, 1))], 1).astype(np.float32)
    vel = np.concatenate([rng.uniform(-5, 5, (n, 3)), np.zeros((n, 1))], 1).astype(np.float32)
    return posm, vel


class env:
    def __init__(self, **kw): self.kw = {k: str(v) for k, v in kw.items()}
    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        os.environ.update(self.kw)
    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None: os.environ.pop(k, None)
            else: os.environ[k] = v


def engine(nb, n, even, ipt, **kw):
    with env(NBODY_SYM_EVEN=1 if even else 0):
        e = nb.NBodyEngine(n, algorithm=2, i_per_thread=ipt, **kw)
    cfg = e.launch_config()
    assert cfg["algorithm"] == "symmetric" and cfg["plan"] == ("even" if even else "guided") and cfg["i_per_thread"] == ipt
    return e


def oracle_sample(oracle, posm, sample, eps=0.0):
    p64 = posm.astype(np.float64)
    return np.concatenate([oracle.forces_direct_f64(p64[:, :3], p64[:, 3], i0=int(i), i1=int(i) + 1, eps=eps) for i in sample])


@pytest.mark.parametrize("equal", [False, True])
@pytest.mark.parametrize("n,ipt", [(5000, 4), (16384, 4), (20001, 8), (32768, 8), (32768, 16), (50000, 16), (65536, 16), (12288, 16)])
def test_even_shares_against_the_oracle_and_the_guided_plan(nb, oracle, n, ipt, equal):
    posm, vel = scene(n, n + ipt, equal)
    with engine(nb, n, True, ipt) as ev, engine(nb, n, False, ipt) as gd:
        slots = {4: 1024, 8: 768, 16: 512}[ipt]
        assert ev.launch_config()["blocks"] == max(slots, -(-n // (256 * ipt)))    # one workgroup per slot of a 256-CU chip
        for e in (ev, gd):
            e.set_state(posm, vel)
            e.compute_forces()
        assert ev.equal_mass_form() == equal
        a, b = ev.accelerations(), gd.accelerations()
    assert np.isfinite(a).all()
    # EVERY body against the fp64 pass of the same scene (forces_sym_f64_kernel, itself 1e-12 from the oracle in
    # tests/test_parity_gpu.py) at the oracle tolerance, and against the guided pass — two orders of summation of the same
    # fp32 terms, each within TOL_ACC of the truth — at half of it.  Relative to the body's acceleration, or to a twentieth
    # of the scene's median where the pulls on a body nearly cancel (N = 20001: one body at 0.7 % of the median differs by
    # 2e-5 of its own between the two, the guided pass being 2.9e-5 and this one 1.1e-5 from the fp64 sum there).
    with nb.NBodyEngine(n, precision="f64") as f64:
        f64.set_state(posm.astype(np.float64), vel.astype(np.float64))
        f64.compute_forces()
        ref_all = f64.accelerations(np.float64)[:, :3]
    na = np.linalg.norm(ref_all, axis=1)
    den = np.maximum(na, 0.05 * np.median(na))
    err_even = np.linalg.norm(a[:, :3] - ref_all, axis=1) / den
    err_guided = np.linalg.norm(b[:, :3] - ref_all, axis=1) / den
    diff = np.linalg.norm(a[:, :3].astype(np.float64) - b[:, :3], axis=1) / den
    print(f"N={n} ipt={ipt} equal={equal}: all bodies vs fp64: even {err_even.max():.2e} guided {err_guided.max():.2e}; even vs guided {diff.max():.2e}")
    assert err_even.max() < TOL_ACC and diff.max() < TOL_ACC / 2
    # bodies at the ends of i-sets, of granules, of the system, and a spread
    sample = np.unique(np.concatenate([np.arange(0, n, n // 24), [0, 63, 64, 255, 256, 511, 512, 256 * ipt - 1, 256 * ipt, n - 65, n - 64, n - 1]]))
    sample = sample[sample < n]
    assert rel_err(a[sample, :3], oracle_sample(oracle, posm, sample)).max() < TOL_ACC


@pytest.mark.parametrize("n,ipt", [(20001, 8), (32768, 16)])
def test_even_shares_softened(nb, oracle, n, ipt):
    posm, vel = scene(n, 3 * n)
    with engine(nb, n, True, ipt, eps=0.5) as ev:
        ev.set_state(posm, vel)
        ev.compute_forces()
        a = ev.accelerations()
    sample = np.arange(0, n, n // 32)
    assert rel_err(a[sample, :3], oracle_sample(oracle, posm, sample, eps=0.5)).max() < TOL_ACC


@pytest.mark.parametrize("n,ipt", [(16384, 4), (32768, 8), (40000, 16)])
def test_even_shares_with_coincident_bodies(nb, oracle, n, ipt):
    """Two different bodies on one point (OctreeSearch.h:102: d == 0 is skipped): the detector sends the pass to the guarded
    loops — in the own block and across blocks, inside a partial subtile or not."""
    posm, vel = scene(n, 7 * n)
    pairs = [(1, 2), (5, 256 * ipt + 5), (100, n - 1), (n // 2, n // 2 + 64), (3000, 3001)]
    for i, j in pairs:
        posm[j, :3] = posm[i, :3]
    with engine(nb, n, True, ipt) as ev:
        ev.set_state(posm, vel)
        ev.compute_forces()
        a = ev.accelerations()
    assert np.isfinite(a).all()
    sample = np.unique(np.array([i for p in pairs for i in p] + list(range(0, n, n // 16))))
    assert rel_err(a[sample, :3], oracle_sample(oracle, posm, sample)).max() < TOL_ACC


@pytest.mark.parametrize("n,ipt", [(20480, 8), (65536, 16)])
def test_even_shares_step_the_same_bits_every_time(nb, oracle, n, ipt):
    """Five Ticks (fused update: force kernel + update per step), twice: no atomics, a plan that depends on (n, bodies per
    lane, CU count) only — identical bytes; and the first Tick's update is the oracle's kick-drift of the accelerations."""
    posm, vel = scene(n, 11 * n)
    outs = []
    for _ in range(2):
        with engine(nb, n, True, ipt) as ev:
            ev.set_state(posm, vel)
            ev.step(0.01, 1)
            p1, v1, a1 = ev.state()
            ev.step(0.01, 4)
            outs.append((p1, v1, a1) + tuple(ev.state()))
    for x, y in zip(*outs):
        assert x.tobytes() == y.tobytes()
    p1, v1, a1 = outs[0][:3]
    ref_p, ref_v = oracle.kick_drift_f32(posm[:, :3], vel[:, :3], a1[:, :3], 0.01)
    assert np.array_equal(p1[:, :3], ref_p) and np.array_equal(v1[:, :3], ref_v)


def test_which_systems_take_even_shares_by_default(nb):
    """Plain fp32, one context owning all bodies, 16385 <= N < 139264 — every size the symmetric pass runs below 139264 —
    (csrc/capi.hip sym_even_default): sixteen bodies per lane from 24576, eight from 20480, four below; fp64 and sharded contexts and
    larger systems keep the guided strips."""
    for n, plan, ipt in ((16385, "even", 4), (20479, "even", 4), (20480, "even", 8), (24575, "even", 8), (24576, "even", 16), (65536, "even", 16),
                         (90111, "even", 16), (90112, "even", 16), (139263, "even", 16), (139264, "guided", 16), (1 << 18, "guided", 16)):
        with nb.NBodyEngine(n) as e:
            cfg = e.launch_config()
        assert (cfg["algorithm"], cfg["plan"], cfg["i_per_thread"]) == ("symmetric", plan, ipt), (n, cfg)
        if plan == "even":
            assert cfg["blocks"] == {4: 1024, 8: 768, 16: 512}[ipt] * (2 if n >= 90112 else 1)     # two items per slot from 90112 bodies on
    with nb.NBodyEngine(16384) as e:
        assert e.launch_config()["kernel"] == "forces_block_pk_kernel" and e.launch_config()["plan"] is None
    for kw in (dict(precision="f32_kahan"), dict(precision="f64"), dict(i_begin=0, i_count=32768)):
        with nb.NBodyEngine(65536, **kw) as e:
            cfg = e.launch_config()
        assert cfg["algorithm"] == "symmetric" and cfg["plan"] == "guided", (kw, cfg)
    # compensated sums: even shares between 12288 and 40960 bodies, four bodies per lane below 22528, eight above
    for n, plan, ipt in ((12287, "guided", 2), (12288, "even", 4), (22528, "even", 8), (40959, "even", 8), (40960, "guided", 8)):
        with nb.NBodyEngine(n, precision="f32_kahan") as e:
            cfg = e.launch_config()
        assert (cfg["algorithm"], cfg["plan"], cfg["i_per_thread"]) == ("symmetric", plan, ipt), (n, cfg)


@pytest.mark.parametrize("eps", [0.0, 0.5])
@pytest.mark.parametrize("n,ipt", [(12288, 4), (20001, 4), (24576, 8), (33000, 8)])
def test_even_shares_with_compensated_sums(nb, oracle, n, ipt, eps):
    """The blocked Kahan form (NBODY_PREC_F32_KAHAN) under even shares: a partial subtile's sums are folded, compensated, like a
    whole one's.  Closer to the fp64 sum than the plain pass, and inside the compensated passes' own tolerance (2e-6 on sampled
    bodies, tests/test_parity_gpu.py)."""
    posm, vel = scene(n, 13 * n)
    with engine(nb, n, True, ipt, precision="f32_kahan", eps=eps) as kh, engine(nb, n, True, ipt if ipt > 4 else 8, eps=eps) as pl:
        for e in (kh, pl):
            e.set_state(posm, vel)
            e.compute_forces()
        a, b = kh.accelerations(), pl.accelerations()
    sample = np.unique(np.concatenate([np.arange(0, n, n // 40), [0, 63, 64, 256 * ipt - 1, 256 * ipt, n - 1]]))
    ref = oracle_sample(oracle, posm, sample, eps=eps)
    ek, ep = rel_err(a[sample, :3], ref), rel_err(b[sample, :3], ref)
    assert ek.max() < 2e-6 and np.median(ek) <= np.median(ep)


@pytest.mark.parametrize("n,equal", [(24576, False), (65536, True)])
def test_checkpoint_resume_on_an_even_share_context_continues_bit_for_bit(nb, tmp_path, n, equal):
    """The library's own choice at these sizes (even shares, fused update, two detector tables taking turns): five steps, a
    checkpoint, five more — and a fresh context resumed from the file ends in the same bytes (the plan is a function of the
    parameters and the device, the file carries positions, velocities and accelerations)."""
    posm, vel = scene(n, 17 * n, equal)
    path = str(tmp_path / "even.ckpt")
    with nb.NBodyEngine(n) as e:
        assert e.launch_config()["plan"] == "even"
        e.set_state(posm, vel)
        e.step(0.01, 5)
        e.save_checkpoint(path)
        e.step(0.01, 5)
        want = e.particles()
    with nb.NBodyEngine(n) as e:
        assert e.load_checkpoint(path) == 5
        e.step(0.01, 5)
        assert e.steps_done() == 10 and e.particles().tobytes() == want.tobytes()

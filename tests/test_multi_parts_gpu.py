"""nbody_create_multi with 2, 4 and 8 PARTS on the one GPU of the test box.

csrc/multi.hip drives one sub-context per device from one caller thread — the reference's game-thread model
(OctreeSearch.cpp:21-34) on a node of eight GPUs.  Real RCCL refuses two ranks on one device, so on this box every `k > 0`
index of that file (slice offsets in the all-gather, the send/recv segments of the all-to-all, getters and checkpoints by
slice) would never run.  Here it does: NBODY_RCCL_LIB names tests/cpp/fake_rccl.c — the same eight entry points as
stream-ordered copies between the parts' buffers — and NBODY_MULTI_SHARE_DEVICE=1 lets device 0 be listed several times.
Everything else is the product's own code: the sub-contexts, their streams, the events between update and gather, the
kernels.  What must hold:

  * the all-gather-only step (algorithm = 1) over any number of parts equals ONE context in every bit;
  * the symmetric step over 2/4/8 parts equals, in every bit, the same ranks emulated by hand with the exchange staged
    through the host, and the oracle within 2e-5;
  * overlap on / off, checkpoints, nbody_tick, getters, the actor: identical to their single-context forms."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import rel_err

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", "-Wall", "-Wextra", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "cpp", "fake_rccl.c"), "-o", so, "-L/opt/rocm/lib", "-lamdhip64",
                           "-Wl,-rpath,/opt/rocm/lib"])
    return so


@pytest.fixture
def parts_env(fake_rccl, monkeypatch):
    monkeypatch.setenv("NBODY_RCCL_LIB", fake_rccl)
    monkeypatch.setenv("NBODY_MULTI_SHARE_DEVICE", "1")
    lib = ctypes.CDLL(fake_rccl)
    lib.fake_rccl_counters.argtypes = [ctypes.POINTER(ctypes.c_int64)]

    def counters():
        c = (ctypes.c_int64 * 6)()
        lib.fake_rccl_counters(c)
        return dict(zip(("all_gathers", "sends", "recvs", "copies", "bytes", "groups"), c))
    return counters


def emulate_ranks(nb, n, ranks, posm, vel, steps, dt, **kw):
    """The symmetric multi-rank step by hand: `ranks` ordinary sharded contexts on one device, the all-to-all and the
    all-gather staged through the host (what tests/test_parity_gpu.py's emulation does, for several steps)."""
    ic = n // ranks
    dt_np = np.float64 if kw.get("precision") == "f64" else np.float32
    engs = [nb.NBodyEngine(n, i_begin=r * ic, i_count=ic, **kw) for r in range(ranks)]
    try:
        p, v = posm, vel
        for _ in range(steps):
            for e in engs:
                e.set_state(p, v)
                e.step_begin()
            sends = [e.exchange_read_send() for e in engs]
            for r, e in enumerate(engs):
                e.exchange_write_recv(np.concatenate([sd[r * ic:(r + 1) * ic] for sd in sends]))
                e.step_end(dt)
            st = [e.state(dt_np) for e in engs]
            p, v, a = (np.concatenate([s[k] for s in st]) for k in range(3))
        return p, v, a, engs[0].launch_config()
    finally:
        for e in engs:
            e.close()


@pytest.mark.parametrize("n,parts", [(8192, 8), (24576, 2), (24576, 4), (24576, 8), (65536, 8)])
def test_all_gather_only_step_over_parts_equals_one_context_in_every_bit(nb, parts_env, n, parts):
    posm, vel = nb.ic_reference_box(n, 1000.0, seed=3)             # the reference's own scene: distinct masses
    before = parts_env()
    with nb.NBodyEngine(n, algorithm=1) as one, nb.NBodyEngine(n, algorithm=1, devices=[0] * parts) as many:
        for e in (one, many):
            e.set_state(posm, vel)
            e.compute_forces()
            e.step(0.01, 5)
        for x, y in zip(one.state(), many.state()):
            np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(one.positions(), many.positions())
        np.testing.assert_array_equal(one.positions(first=n // parts - 3, count=7), many.positions(first=n // parts - 3, count=7))
        np.testing.assert_array_equal(one.particles(), many.particles())
        assert one.bounds() == many.bounds() and many.steps_done() == 5
        ke1, pe1 = one.energy(); ke2, pe2 = many.energy()           # the parts' shares are added on the host: to rounding
        assert ke2 == pytest.approx(ke1, rel=1e-12) and pe2 == pytest.approx(pe1, rel=1e-12)
    after = parts_env()
    # five steps = five grouped in-place all-gathers over `parts` communicators, each rank fetching every other rank's slice
    assert after["all_gathers"] - before["all_gathers"] == 5 * parts
    assert after["copies"] - before["copies"] == 5 * parts * (parts - 1)
    assert after["bytes"] - before["bytes"] == 5 * parts * (parts - 1) * (n // parts) * 16
    assert after["sends"] == before["sends"]                       # no all-to-all on this step


@pytest.mark.parametrize("parts", [2, 4, 8])
@pytest.mark.parametrize("precision,n,tol", [("f32", 65536, 2e-5), ("f32_kahan", 32768, 2e-6), ("f64", 16384, 1e-12)])
def test_symmetric_step_over_parts_equals_the_emulated_ranks_and_the_oracle(nb, oracle, parts_env, parts, precision, n, tol):
    if precision == "f64":
        posm, vel = (a.astype(np.float64) for a in nb.ic_plummer(n, seed=5))
    else:
        posm, vel = nb.ic_plummer(n, seed=5)
        if precision == "f32":                                     # distinct masses: the general form of the kernel
            posm[:, 3] *= np.random.default_rng(1).uniform(0.5, 1.5, n).astype(np.float32)
    dt_np = np.float64 if precision == "f64" else np.float32
    eps = 0.5 if precision == "f32_kahan" else 0.0
    kw = dict(precision=precision, eps=eps, algorithm=2)
    before = parts_env()
    with nb.NBodyEngine(n, devices=[0] * parts, **kw) as many:
        cfg = many.launch_config()
        assert cfg["algorithm"] == "symmetric"
        many.set_state(posm, vel)
        many.compute_forces()
        a0 = many.accelerations(dt_np)
        many.step(0.01, 3)
        p, v, a = many.state(dt_np)
    after = parts_env()
    assert after["sends"] - before["sends"] == 4 * parts * parts and after["recvs"] - before["recvs"] == 4 * parts * parts
    # the oracle: the pair law summed over all j (fp64 restatement: at these sizes the oracle's own fp32 index-order sum is
    # ~1e-5 from it), on bodies at the slice ends and at random
    ic = n // parts
    rng = np.random.default_rng(2)
    sample = sorted({0, n - 1, *(k * ic for k in range(parts)), *(k * ic - 1 for k in range(1, parts + 1)), *map(int, rng.choice(n, 16, replace=False))})
    p64 = posm.astype(np.float64)
    for i in sample:
        ref = oracle.forces_direct_f64(p64[:, :3], p64[:, 3], i0=i, i1=i + 1, eps=eps)
        assert rel_err(a0[i:i + 1], ref).max() < tol, (i, parts)
    # the same ranks by hand: every bit
    pe, ve, ae, cfg_e = emulate_ranks(nb, n, parts, posm, vel, 3, 0.01, **kw)
    assert cfg_e == cfg
    np.testing.assert_array_equal(p, pe)
    np.testing.assert_array_equal(v, ve)
    np.testing.assert_array_equal(a, ae)


@pytest.mark.parametrize("algorithm", [1, 2])
def test_overlap_on_and_off_are_the_same_trajectory(nb, parts_env, monkeypatch, algorithm):
    n = 65536
    posm, vel = nb.ic_plummer(n, seed=8)
    out = []
    for no_overlap in ("0", "1"):
        monkeypatch.setenv("NBODY_MULTI_NO_OVERLAP", no_overlap)
        with nb.NBodyEngine(n, algorithm=algorithm, devices=[0] * 4) as e:
            e.set_state(posm, vel)
            e.step(0.01, 4)
            e.compute_forces()                                     # reads the gathered positions outside the stepping loop
            out.append(e.state())
    for x, y in zip(*out):
        np.testing.assert_array_equal(x, y)


def test_checkpoints_by_slice(nb, parts_env, tmp_path):
    n = 24576
    posm, vel = nb.ic_reference_box(n, 1000.0, seed=4)
    path_m, path_s = str(tmp_path / "parts.ckpt"), str(tmp_path / "one.ckpt")
    with nb.NBodyEngine(n, algorithm=1) as s, nb.NBodyEngine(n, algorithm=1, devices=[0] * 4) as m:
        s.set_state(posm, vel); m.set_state(posm, vel)
        s.step(0.01, 2); m.step(0.01, 2)
        s.save_checkpoint(path_s); m.save_checkpoint(path_m)
        assert open(path_s, "rb").read() == open(path_m, "rb").read()          # the file a single context of the whole system writes
        s.step(0.01, 3)
        with nb.NBodyEngine(n, algorithm=1, devices=[0] * 8) as r:              # another partition reads its slices of it
            assert r.load_checkpoint(path_m) == 2
            r.step(0.01, 3)
            assert r.steps_done() == 5
            for x, y in zip(r.state(), s.state()):
                np.testing.assert_array_equal(x, y)
    with nb.NBodyEngine(n, algorithm=2, devices=[0] * 2) as a, nb.NBodyEngine(n, algorithm=2, devices=[0] * 2) as b:
        a.set_state(posm, vel); a.step(0.01, 2); a.save_checkpoint(path_m); a.step(0.01, 2)      # straight ...
        assert b.load_checkpoint(path_m) == 2
        b.step(0.01, 2)                                                         # ... and resumed: every byte
        for x, y in zip(a.state(), b.state()):
            np.testing.assert_array_equal(x, y)


def test_tick_and_the_actor_over_parts(nb, parts_env):
    n = 8192
    posm, vel = nb.ic_reference_box(n, 1000.0, seed=6)
    with nb.NBodyEngine(n, algorithm=1) as one, nb.NBodyEngine(n, algorithm=1, devices=[0] * 4) as many:
        one.set_state(posm, vel); many.set_state(posm, vel)
        for _ in range(3):
            s1, r1 = one.tick(0.01); s2, r2 = many.tick(0.01)
            assert s1 == s2 and r1.tobytes() == r2.tobytes()
        s1, r1 = one.tick(0.0); s2, r2 = many.tick(0.0)             # paused (OctreeSearch.cpp:25): the frame is still delivered
        assert r1.tobytes() == r2.tobytes()
    # the AOctreeSearch mirror over a device list: the shipped scene, theta = 0 (all-pairs), same frames as on one device
    a, b = nb.OctreeSearch(), nb.OctreeSearch()
    b.set_devices([0, 0])
    for act in (a, b):
        act.set_theta(0.0)
        act.set_seed(5)
        act.CreateSpacePoints(2000, 1000.0)
        for _ in range(5):
            act.Tick(1.0 / 60)
        assert act.LastStatus == 0
    np.testing.assert_array_equal(a.Particles, b.Particles)
    assert a.Size == b.Size


def test_argument_errors_with_parts(nb, parts_env, monkeypatch):
    with pytest.raises(nb.NBodyError):
        nb.NBodyEngine(1001, devices=[0, 0])                        # equal slices only
    with pytest.raises(nb.NBodyError):
        nb.NBodyEngine(4096, devices=[0, 0], theta=1.0)             # Barnes-Hut runs on one device
    monkeypatch.delenv("NBODY_MULTI_SHARE_DEVICE")
    with pytest.raises(nb.NBodyError):
        nb.NBodyEngine(4096, devices=[0, 0])                        # without the test switch a device may be listed once
    monkeypatch.setenv("NBODY_RCCL_LIB", "/nonexistent/librccl.so")
    with pytest.raises(nb.NBodyError) as e:
        nb.NBodyEngine(4096, devices=[0])
    assert "cannot load RCCL" in str(e.value)


def test_plain_c_host_over_four_parts(nb, fake_rccl, tmp_path):
    # tests/cpp/multi_parity.c (C11, no Python in the loop) with device 0 listed four times: the `n_dev > 1` branch of that host
    exe = str(tmp_path / "multi_parity")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "multi_parity.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "parallelnbody_amd"), "-lnbody_amd", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "parallelnbody_amd")])
    env = dict(os.environ, NBODY_RCCL_LIB=fake_rccl, NBODY_MULTI_SHARE_DEVICE="1", NBODY_TEST_PARTS="4")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "multi parity: ok" in out.stdout and "on 4 device(s)" in out.stdout and "within tolerance" in out.stdout

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

from conftest import particles_from, rel_err

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
        nb.NBodyEngine(4096, devices=[0, 0], theta=1.0, precision="f64")   # the reference's tree walk is fp32 (OctreeSearch.h:8-18)
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


@pytest.mark.parametrize("n,parts", [(2000, 2), (8192, 2), (8192, 4), (8192, 8), (131072, 2), (131072, 4), (131072, 8)])
def test_barnes_hut_over_parts_equals_one_context_in_every_byte(nb, oracle, parts_env, n, parts):
    # theta > 0 — the reference's SHIPPED algorithm (OctreeSearch.cpp:74-89 at Theta = 1.0, .cpp:85) — over several devices: the tree
    # is one tree (.cpp:79-81) and a body's walk (.cpp:83-86) reads it and writes that body alone, so every device builds the whole
    # tree from its copy of the positions, walks and integrates its own slice [r N/P, (r+1) N/P), and the in-place all-gather brings
    # the moved bodies to everyone.  Five Ticks (three queued in one call, then two nbody_tick frames with the records and Size) on
    # the shipped kind of scene: every byte of the records, Size, the root centre, node and level counts equal the one-device
    # context's — and, at the sizes the oracle walks in a moment, the oracle's.
    posm, vel = nb.ic_reference_box(n, 1000.0, seed=9)
    before = parts_env()
    with nb.NBodyEngine(n, theta=1.0) as one, nb.NBodyEngine(n, theta=1.0, devices=[0] * parts) as many:
        assert many.theta() == 1.0
        for e in (one, many):
            e.set_state(posm, vel)
            e.compute_forces()                                     # a diagnostic pass: accelerations only, the next root stays
        np.testing.assert_array_equal(one.accelerations(), many.accelerations())
        for e in (one, many):
            e.step(0.01, 3)
        assert one.particles().tobytes() == many.particles().tobytes()
        assert many.steps_done() == 3
        for _ in range(2):
            s1, r1 = one.tick(0.01); s2, r2 = many.tick(0.01)
            assert s1 == s2 and r1.tobytes() == r2.tobytes()
        st1, st2 = one.bh_stats(), many.bh_stats()
        assert st1["nodes"] == st2["nodes"] and st1["levels"] == st2["levels"]
        np.testing.assert_array_equal(st1["root_com"], st2["root_com"])
        np.testing.assert_array_equal(one.bh_leaf_order(), many.bh_leaf_order())
        np.testing.assert_array_equal(one.bh_leaf_boxes(), many.bh_leaf_boxes())
        for x, y in zip(one.state(), many.state()):
            np.testing.assert_array_equal(x, y)
        final = many.particles()
        root = st2["root_com"]
    after = parts_env()
    assert after["all_gathers"] - before["all_gathers"] >= 5 * parts     # one grouped in-place all-gather per frame (more if frames were queued again)
    assert after["sends"] == before["sends"]                       # nothing else changes hands
    # the oracle (pow_mode 3: the cube correctly rounded, as the device has it), five Ticks of the same scene
    q = np.zeros(n, nb.PARTICLE_DTYPE)
    q["Mass"] = posm[:, 3]; q["Position"] = posm[:, :3]; q["Velocity"] = vel[:, :3]
    com, size = None, 0.0
    for _ in range(5):
        com, size = oracle.tick_aos_f32(q, 0.01, theta=1.0, root_com=com, size=size, pow_mode=3)
    assert final.tobytes() == q.tobytes()
    np.testing.assert_array_equal(root, com)


def test_barnes_hut_over_parts_when_frames_are_given_up_and_refused(nb, oracle, parts_env):
    # The sort that starts from the previous frame's order gives a frame up when the records are replaced by a scene that has nothing to
    # do with the order on the devices; every device gives the same frame up, the frames queued behind it do nothing on any device
    # (the all-gathers between them move unchanged slices), and the front queues them again — gathers included.  Then two bodies
    # closer than Size / 2^42: the frame is refused on every device alike, the error is reported once and the state stays what it was.
    n, parts = 16384, 4
    rng = np.random.default_rng(3)
    posm, vel = nb.ic_reference_box(n, 1000.0, seed=2)
    q = np.zeros(n, nb.PARTICLE_DTYPE)
    q["Mass"] = posm[:, 3] * np.float32(1e-3); q["Position"] = posm[:, :3]; q["Velocity"] = vel[:, :3]
    com, size = None, 0.0

    def ref(k):
        nonlocal com, size
        for _ in range(k):
            com, size = oracle.tick_aos_f32(q, 0.01, theta=1.0, root_com=com, size=size, pow_mode=3)

    with nb.NBodyEngine(n, theta=1.0, devices=[0] * parts) as e:
        e.set_particles(q)
        e.step(0.01, 2); ref(2)
        assert e.particles().tobytes() == q.tobytes()
        far = np.float32(0.7) * np.abs(q["Position"]).max()
        clump = (rng.uniform(-30, 30, (n - 1, 3)) + far).astype(np.float32)
        assert len(np.unique(clump, axis=0)) == n - 1
        q["Position"][1:] = clump
        e.push_particles(q)
        e.step(0.01, 3); ref(3)                                    # the first of the three is given up everywhere; all three are queued again
        assert e.particles().tobytes() == q.tobytes()
        assert e.steps_done() == 5
        np.testing.assert_array_equal(e.bh_stats()["root_com"], com)
        q["Position"][7] = q["Position"][9]                        # coincident bodies: the reference's Add would never return
        e.push_particles(q)
        with pytest.raises(nb.NBodyError) as err:
            e.step(0.01, 2)
        assert "42" in str(err.value)
        assert e.steps_done() == 5
        got = e.particles()
        np.testing.assert_array_equal(got["Position"], q["Position"])
        np.testing.assert_array_equal(got["Velocity"], q["Velocity"])
        q["Position"][7] += np.float32(3.0)                        # apart again: the simulation goes on
        e.push_particles(q)
        e.step(0.01, 2); ref(2)
        assert e.particles().tobytes() == q.tobytes()


def test_barnes_hut_checkpoints_and_the_opening_angle_over_parts(nb, parts_env, tmp_path):
    # the file a multi-device context writes at theta > 0 is the one-device context's (theta and the next tree's root centre
    # included); any partition resumes from it; nbody_set_theta switches a running multi-device context between the two force passes
    n = 12288
    posm, vel = nb.ic_reference_box(n, 1000.0, seed=4)
    path_m, path_s = str(tmp_path / "parts.ckpt"), str(tmp_path / "one.ckpt")
    with nb.NBodyEngine(n, theta=1.0) as s, nb.NBodyEngine(n, theta=1.0, devices=[0] * 4) as m:
        s.set_state(posm, vel); m.set_state(posm, vel)
        s.step(0.01, 2); m.step(0.01, 2)
        s.save_checkpoint(path_s); m.save_checkpoint(path_m)
        assert open(path_s, "rb").read() == open(path_m, "rb").read()
        s.step(0.01, 3)
        with nb.NBodyEngine(n, devices=[0] * 2) as r:               # created at theta = 0: the file brings its opening angle along
            assert r.load_checkpoint(path_m) == 2 and r.theta() == 1.0
            r.step(0.01, 3)
            for x, y in zip(r.state(), s.state()):
                np.testing.assert_array_equal(x, y)
        m.load_checkpoint(path_s); s.load_checkpoint(path_s)
        for e in (s, m):
            e.set_theta(0.5); e.step(0.01, 2)                      # another opening angle on the running contexts: still every byte
        for x, y in zip(s.state(), m.state()):
            np.testing.assert_array_equal(x, y)
        for e in (s, m):
            e.set_theta(0.0); e.compute_forces()                   # ... and back to the all-pairs pass (its own summation order per geometry)
        assert s.theta() == 0.0 and m.theta() == 0.0
        assert rel_err(m.accelerations(), s.accelerations()).max() < 2e-5


def test_a_slice_context_at_the_shipped_opening_angle_by_hand(nb, parts_env):
    # what a one-process-per-GPU host does (parallelnbody_amd/sharded.py): contexts that own a slice each, nbody_step on every one,
    # the positions all-gathered by the host in between — here by hand through the host, two and three slices of unequal work
    n = 6000
    posm, vel = nb.ic_reference_box(n, 1000.0, seed=12)
    with nb.NBodyEngine(n, theta=1.0) as one:
        one.set_state(posm, vel)
        one.step(0.01, 3)
        want = one.state()
    for ranks in (2, 3):
        ic = n // ranks
        engs = [nb.NBodyEngine(n, i_begin=r * ic, i_count=ic, theta=1.0) for r in range(ranks)]
        try:
            for e in engs:
                e.set_state(posm, vel)
            for _ in range(3):
                for e in engs:
                    e.step(0.01, 1)
                st = [e.state() for e in engs]
                p = np.concatenate([s[0] for s in st]); v = np.concatenate([s[1] for s in st]); a = np.concatenate([s[2] for s in st])
                for e in engs:                                     # the all-gather: every context gets the other slices' new positions
                    e.push_particles(particles_from(nb, p, v))     # (records pushed into a running simulation: history and root stay)
            np.testing.assert_array_equal(p, want[0]); np.testing.assert_array_equal(v, want[1]); np.testing.assert_array_equal(a, want[2])
        finally:
            for e in engs:
                e.close()

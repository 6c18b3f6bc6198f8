"""GPU Barnes-Hut (theta > 0): the reference's own tree, upsweep and walk on the device, against the
oracle's restatement of Octree::Add / ComputeMass / ComputeForces.  The device follows the reference's arithmetic step
by step in the same depth-first order, so the comparison is (almost always) bit for bit; the assertions allow one fp32
ulp on a vanishing fraction of bodies for the one place the two sides differ: glibc's pow(d, 3.0) vs the device's
correctly rounded (d*d)*d in double."""
import os

import numpy as np
import pytest

from conftest import particles_from, rel_err

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
REF_THETA = 1.0     # OctreeSearch.cpp:85
# The device's path keys hold 42 octant digits: a frame is refused when two bodies share all of them — exactly when Octree::Add
# (OctreeSearch.h:60-81) of the same scene splits a cell of depth 42, i.e. reaches depth 43 (oracle.last_max_depth, root = 0).
REFUSED_FROM_DEPTH = 43


def _check_same(a, ref):
    a = np.asarray(a, np.float32); ref = np.asarray(ref, np.float32)
    same = np.all(a == ref, axis=1)
    assert same.mean() > 0.999, same.mean()
    assert rel_err(a, ref).max() < 1e-6


@pytest.mark.parametrize("theta", [1.0, 0.5, 0.25])
@pytest.mark.parametrize("fixture", ["refbox_n2000_seed1", "plummer_n1024_seed1"])
def test_bh_forces_match_the_oracle_tree(nb, oracle, fixture, theta):
    g = np.load(os.path.join(GOLDEN, fixture + ".npz"))
    n = g["posm"].shape[0]
    pos = np.ascontiguousarray(g["posm"][:, :3]); m = np.ascontiguousarray(g["posm"][:, 3])
    ref, com, nodes = oracle.octree_forces_f32(pos, m, theta)           # root: centre 0, half-width = ComputeCubeSize
    with nb.NBodyEngine(n, theta=theta) as e:
        e.set_state(g["posm"], g["vel"])
        e.compute_forces()
        a = e.accelerations()
        st = e.bh_stats()
    _check_same(a, ref)
    assert st["nodes"] == nodes
    np.testing.assert_array_equal(st["root_com"], com)
    # with the cube taken as the correctly rounded (d*d)*d in double (pow_mode 3: what a correctly rounded pow returns,
    # and what the device computes) instead of glibc's pow(d, 3.0), every body agrees in every bit
    ref3, _, _ = oracle.octree_forces_f32(pos, m, theta, pow_mode=3)
    np.testing.assert_array_equal(a, ref3)
    # and it really is an approximation of the all-pairs answer, not the all-pairs kernel
    assert 0.001 < rel_err(a, g["acc_direct"]).mean() < 1.5


@pytest.mark.parametrize("div_mode", [0, 1])
@pytest.mark.parametrize("fixture", ["refbox_n2000_seed1", "plummer_n1024_seed1"])
def test_bh_matches_the_oracle_under_both_readings_of_the_centre_of_mass_division(nb, oracle, fixture, div_mode):
    # `CenterOfMass /= TotalMass` (OctreeSearch.h:95) goes through UE4's FVector::operator/=(float), which is not in the
    # reference tree: reciprocal-multiply (div_mode 0, UE4 4.9 as remembered) or three divisions (1).  Device and oracle
    # carry the same switch, and agree in every bit under either reading.
    g = np.load(os.path.join(GOLDEN, fixture + ".npz"))
    n = g["posm"].shape[0]
    pos = np.ascontiguousarray(g["posm"][:, :3]); m = np.ascontiguousarray(g["posm"][:, 3])
    ref, com, nodes = oracle.octree_forces_f32(pos, m, REF_THETA, pow_mode=3, div_mode=div_mode)
    with nb.NBodyEngine(n, theta=REF_THETA, bh_div_mode=div_mode) as e:
        e.set_state(g["posm"], g["vel"])
        e.compute_forces()
        a = e.accelerations()
        st = e.bh_stats()
    np.testing.assert_array_equal(a, ref)
    np.testing.assert_array_equal(st["root_com"], com)
    assert st["nodes"] == nodes


@pytest.mark.parametrize("n,seed", [(2000, 1), (20000, 2), (70000, 3)])
def test_leaf_boxes_and_draw_order_equal_the_oracle_tree(nb, oracle, n, seed):
    # what DrawOctreeBoxes (OctreeSearch.cpp:36-45) draws: (Origin, Size) of every occupied leaf and the body in it, depth
    # first with children 0..7 — exactly, for the one-workgroup build (n <= 16384) and the level-by-level one
    posm, vel = nb.ic_reference_box(n, 1000.0, seed=seed)
    boxes, order = oracle.octree_leaves_f32(posm[:, :3], posm[:, 3])
    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, vel)
        e.compute_forces()
        got_order = e.bh_leaf_order()
        got_boxes = e.bh_leaf_boxes()              # indexed by body
    np.testing.assert_array_equal(got_order, order)
    np.testing.assert_array_equal(got_boxes[order], boxes)
    inside = np.all(np.abs(posm[order, :3] - boxes[:, :3]) <= boxes[:, 3:4] * (1 + 1e-6), axis=1)
    assert inside.mean() > 0.999                   # a body sits in its leaf's box (the root box need not hold every body)


def test_vanishing_theta_is_all_pairs_in_the_reference_order(nb, oracle):
    # theta -> 0+ never accepts a cell, so the walk visits every leaf in the reference's depth-first order: the exact
    # all-pairs sum, added up in the order the reference adds it.  Bit for bit the oracle's tree at the same theta; and
    # the all-pairs kernels (their own summation order) agree with it within the stated tolerance.
    g = np.load(os.path.join(GOLDEN, "refbox_n2000_seed1.npz"))
    pos = np.ascontiguousarray(g["posm"][:, :3]); m = np.ascontiguousarray(g["posm"][:, 3])
    tiny = 1e-30
    ref3, _, _ = oracle.octree_forces_f32(pos, m, tiny, pow_mode=3)
    with nb.NBodyEngine(2000, theta=tiny) as e:
        e.set_state(g["posm"], g["vel"])
        e.compute_forces()
        a_tree = e.accelerations()
        e.set_theta(0.0)
        e.compute_forces()
        a_pairs = e.accelerations()
    np.testing.assert_array_equal(a_tree, ref3)
    assert rel_err(a_pairs, a_tree).max() < 2e-5
    assert rel_err(a_tree, g["acc_f64"]).max() < 2e-5


def test_bh_ticks_follow_the_reference_frame_loop(nb, oracle):
    # Tick (OctreeSearch.cpp:25-32) with theta = 1.0: bounds -> tree rooted at the PREVIOUS tree's CoM -> walk -> kick-drift
    g = np.load(os.path.join(GOLDEN, "refbox_n2000_seed1.npz"))
    q = particles_from(nb, g["posm"], g["vel"])
    com, size = None, 0.0
    with nb.NBodyEngine(2000, theta=REF_THETA) as e:
        e.set_state(g["posm"], g["vel"])
        for _ in range(4):
            e.step(0.01, 1)
            com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size)
            out = e.particles()
            _check_same(out["Acceleration"], q["Acceleration"])
            np.testing.assert_array_equal(e.bh_stats()["root_com"], com)
        assert np.abs(out["Position"] - q["Position"]).max() / np.abs(q["Position"]).max() < 1e-6
        # switching back to theta = 0 gives the exact all-pairs pass again
        e.set_theta(0.0)
        e.compute_forces()
        a0 = e.accelerations()
    ref = oracle.forces_direct_f32(out["Position"], out["Mass"])
    assert rel_err(a0, ref).max() < 2e-5


def test_bh_frames_equal_the_oracle_in_every_bit(nb, oracle):
    # twenty whole Ticks of the shipped scene at the shipped opening angle: positions, velocities, accelerations, Size and
    # the root centre of every frame equal the oracle's (cube correctly rounded, pow_mode 3) in every bit
    g = np.load(os.path.join(GOLDEN, "refbox_n2000_seed1.npz"))
    q = particles_from(nb, g["posm"], g["vel"])
    com, size = None, 0.0
    with nb.NBodyEngine(2000, theta=REF_THETA) as e:
        e.set_state(g["posm"], g["vel"])
        for frame in range(20):
            size_dev, out = e.tick(0.01)
            com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size, pow_mode=3)
            assert size_dev == size, frame
            np.testing.assert_array_equal(e.bh_stats()["root_com"], com)
            assert out.tobytes() == q.tobytes(), frame


def test_bh_actor_with_the_shipped_opening_angle(nb, oracle):
    g = np.load(os.path.join(GOLDEN, "refbox_n2000_seed1.npz"))
    p = particles_from(nb, g["posm"], g["vel"])
    a = nb.OctreeSearch()
    a.SetParticles(p)
    a.set_theta(REF_THETA)
    q = p.copy()
    com, size = None, 0.0
    for _ in range(3):
        a.Tick(0.0)
        com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size)
    assert a.LastStatus == 0
    out = a.Particles
    _check_same(out["Acceleration"], q["Acceleration"])
    assert a.Size == pytest.approx(size)


def test_bh_large_n(nb, oracle):
    n = 65536
    posm, vel = nb.ic_plummer(n, seed=8)
    ref, com, nodes = oracle.octree_forces_f32(posm[:, :3], posm[:, 3], REF_THETA)
    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, vel)
        e.compute_forces()
        a = e.accelerations()
        st = e.bh_stats()
    _check_same(a, ref)
    assert st["nodes"] == nodes and st["levels"] >= 8


@pytest.mark.parametrize("n", [20480, 49152])
def test_switching_the_opening_angle_on_a_fused_all_pairs_context(nb, oracle, n):
    # N > 16384 single-device fp32: the all-pairs step is the symmetric kernel with the fused update, which prepares the
    # NEXT pass (scaled positions, coincident-body table) while it moves the bodies.  A Barnes-Hut step in between moves
    # them with the plain update: the next all-pairs pass must prepare again, not reuse what the last fused update left.
    # (The actor hands its public Theta field to nbody_set_theta on every Tick, so this sequence is one checkbox away.)
    posm, vel = nb.ic_plummer(n, seed=n + 1)
    posm[:, 3] *= np.random.default_rng(5).uniform(0.5, 1.5, n).astype(np.float32)
    with nb.NBodyEngine(n) as e:
        assert e.launch_config()["algorithm"] == "symmetric"
        e.set_state(posm, vel)
        e.step(0.01, 2)                              # fused all-pairs steps
        e.set_theta(0.5)
        e.step(0.01, 2)                              # Barnes-Hut steps: bodies move without the fused preparation
        e.set_theta(0.0)
        e.compute_forces()
        p, v, a = e.state()
        e.step(0.01, 1)                              # and stepping goes on from there
        e.compute_forces()
        p2, _, a2 = e.state()
    for pp, aa in ((p, a), (p2, a2)):
        p64 = pp.astype(np.float64)
        sample = np.arange(0, n, n // 64)
        ref = np.concatenate([oracle.forces_direct_f64(p64[:, :3], p64[:, 3], i0=int(i), i1=int(i) + 1) for i in sample])
        assert rel_err(aa[sample, :3], ref).max() < 2e-5
    # theta > 0 from the start, then all-pairs
    with nb.NBodyEngine(n, theta=1.0) as e:
        e.set_state(posm, vel)
        e.step(0.01, 1)
        e.set_theta(0.0)
        e.step(0.01, 1)
        e.set_theta(1.0)
        e.step(0.01, 1)
        e.set_theta(0.0)
        e.compute_forces()
        p, v, a = e.state()
    p64 = p.astype(np.float64)
    ref = np.concatenate([oracle.forces_direct_f64(p64[:, :3], p64[:, 3], i0=int(i), i1=int(i) + 1) for i in sample])
    assert rel_err(a[sample, :3], ref).max() < 2e-5


@pytest.mark.parametrize("n", [3000, 6000])
def test_bodies_that_share_twenty_one_levels_and_more(nb, oracle, n):
    # two pairs of bodies closer than Size / 2^21 (but not on one point): their paths agree in the whole first key word, so
    # the order comes from the second one — the in-LDS sort's tie path (n <= 4096) and the two-pass radix sort the larger
    # systems fall back to — and the tree has chains of single-child cells down to level 25 and beyond
    posm, vel = nb.ic_plummer(n, seed=n)
    posm[:, 3] *= np.random.default_rng(n).uniform(0.5, 1.5, n).astype(np.float32)
    size = np.abs(posm[:, :3]).max()
    posm[7, :3] = posm[900, :3] + np.float32(size * 2.0 ** -24) * np.array([1, 0, 0], np.float32)
    posm[2001, :3] = posm[15, :3]                                # one ulp apart in y and z: ~Size / 2^23
    posm[2001, 1] = np.nextafter(posm[15, 1], np.float32(-np.inf)); posm[2001, 2] = np.nextafter(posm[15, 2], np.float32(np.inf))
    assert not np.array_equal(posm[7, :3], posm[900, :3]) and not np.array_equal(posm[2001, :3], posm[15, :3])
    pos = np.ascontiguousarray(posm[:, :3]); m = np.ascontiguousarray(posm[:, 3])
    ref, com, nodes = oracle.octree_forces_f32(pos, m, REF_THETA, pow_mode=3)
    boxes, order = oracle.octree_leaves_f32(pos, m)
    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, vel)
        e.compute_forces()
        a = e.accelerations()
        st = e.bh_stats()
        np.testing.assert_array_equal(e.bh_leaf_order(), order)
        np.testing.assert_array_equal(e.bh_leaf_boxes()[order], boxes)
    np.testing.assert_array_equal(a, ref)
    np.testing.assert_array_equal(st["root_com"], com)
    assert st["nodes"] == nodes and st["levels"] >= 22
    # the same two bodies ON one point: the reference's Add would never return; the frame is refused on either path
    posm[7, :3] = posm[900, :3]
    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, vel)
        with pytest.raises(nb.NBodyError) as err:
            e.step(0.01, 2)
        assert err.value.code == nb._lib.ERR_UNSUPPORTED and "42 levels" in str(err.value) and e.steps_done() == 0
        e.set_theta(0.0)                                   # the all-pairs pass takes such a state (d == 0 pairs are skipped)
        e.step(0.01, 1)
        assert e.steps_done() == 1


def test_bh_limits(nb):
    g = np.load(os.path.join(GOLDEN, "plummer_n1024_seed1.npz"))
    with pytest.raises(nb.NBodyError) as err:
        nb.NBodyEngine(1024, precision="f64", theta=1.0).step(0.01, 1)
    with nb.NBodyEngine(1024, precision="f64") as e:
        with pytest.raises(nb.NBodyError) as err:
            e.set_theta(1.0)
        assert err.value.code == nb._lib.ERR_UNSUPPORTED
    with nb.NBodyEngine(1024, precision="f32_kahan") as e:
        with pytest.raises(nb.NBodyError):
            e.set_theta(0.5)
    with nb.NBodyEngine(1024, i_begin=0, i_count=512) as e:     # a slice may: it builds the whole tree and walks its own bodies
        e.set_theta(0.5)                                        # (round 5; tests/test_multi_parts_gpu.py)
        assert e.theta() == 0.5
    # coincident bodies: the reference's Add recurses without bound; here the frame is refused
    posm = g["posm"].copy(); posm[7, :3] = posm[900, :3]
    with nb.NBodyEngine(1024, theta=1.0) as e:
        e.set_state(posm, g["vel"])
        with pytest.raises(nb.NBodyError) as err:
            e.compute_forces()
        assert err.value.code == nb._lib.ERR_UNSUPPORTED and "42 levels" in str(err.value)
    # a single body: root is a leaf, no force
    with nb.NBodyEngine(1, theta=1.0) as e:
        e.set_state(np.array([[1, 2, 3, 4]], np.float32), np.zeros((1, 4), np.float32))
        e.step(0.01, 1)
        assert np.all(e.accelerations() == 0)


def test_show_octree_leaf_boxes(nb):
    # DrawOctreeBoxes with ShowOctree (OctreeSearch.cpp:39-41): one DrawDebugBox(Origin, Size) per occupied leaf.
    g = np.load(os.path.join(GOLDEN, "refbox_n2000_seed1.npz"))
    p = particles_from(nb, g["posm"], g["vel"])
    a = nb.OctreeSearch()
    a.SetParticles(p)
    a.set_theta(REF_THETA)
    a.ShowOctree = True
    a.PhDeltaTime = 1e-6
    boxes, points = [], []
    a.set_box_callback(lambda o, s: boxes.append((o, s)))
    a.set_draw_callbacks(on_point=lambda pos, sz: points.append(pos))
    a.Tick(0.0)
    assert len(boxes) == 2000 and len(points) == 2000
    o = np.array([b[0] for b in boxes], np.float64); s = np.array([b[1] for b in boxes], np.float64)
    pts = np.array(points, np.float64)
    # every body lies in its leaf's box (the root may not contain everything, deeper cells do), and leaves are
    # smaller than the root
    inside = np.all(np.abs(pts - o) <= s[:, None] * (1 + 1e-6), axis=1)
    assert inside.mean() > 0.95 and s.max() <= a.Size and s.min() > 0
    a.ShowOctree = False
    boxes.clear()
    a.Tick(0.0)
    assert not boxes


def _fuzz_scene(rng, n):
    """A scene with structure at every scale: a few clumps of very different widths (deep, narrow subtrees: chains of
    single-child cells, cells that reach across many 256-body chunks), a uniform background, masses over three decades."""
    kind = rng.integers(0, 3)
    if kind == 0:
        pos = rng.uniform(-1000, 1000, (n, 3))
    else:
        k = int(rng.integers(1, 6))
        centres = rng.uniform(-800, 800, (k, 3))
        widths = 10.0 ** rng.uniform(-3, 2.5, k)
        which = rng.integers(0, k, n)
        pos = centres[which] + rng.normal(0, 1, (n, 3)) * widths[which, None]
        if kind == 2:
            back = rng.random(n) < 0.3
            pos[back] = rng.uniform(-1000, 1000, (int(back.sum()), 3))
    posm = np.concatenate([pos, 10.0 ** rng.uniform(0, 3, (n, 1))], 1).astype(np.float32)
    if n > 3:
        posm[0, :3] = 0.0
    return posm


def test_bh_fuzz_every_bit_of_the_force_pass_on_random_scenes(nb, oracle):
    """Sizes on both sides of every switch of the theta > 0 path (one-workgroup build up to 4096, windows on the global tree
    up to 20480, ComputeMass over chunks of 256 bodies up to 262144 and of 1024 above), random opening angles, clumpy scenes: accelerations, node count
    and root CoM equal the oracle's tree (correctly rounded cube: pow_mode 3) in every bit.  NBODY_FUZZ_SEED /
    NBODY_FUZZ_TRIALS run it longer."""
    rng = np.random.default_rng(int(os.environ.get("NBODY_FUZZ_SEED", "77")))
    trials = int(os.environ.get("NBODY_FUZZ_TRIALS", "24"))
    ran = refused = 0
    for trial in range(trials):
        u = rng.random()
        n = (int(rng.integers(2, 4097)) if u < 0.35 else int(rng.integers(4097, 21000)) if u < 0.7 else
             int(rng.integers(21000, 140000)) if u < 0.92 else int(rng.integers(262145, 400000)))
        theta = float(rng.choice([1.0, 1.0, 0.5, 0.3, 1.7]))
        div_mode = int(rng.integers(0, 2))
        posm = _fuzz_scene(rng, n)
        vel = np.zeros((n, 4), np.float32)
        pos = np.ascontiguousarray(posm[:, :3]); m = np.ascontiguousarray(posm[:, 3])
        with nb.NBodyEngine(n, theta=theta, bh_div_mode=div_mode) as e:
            e.set_state(posm, vel)
            try:
                e.compute_forces()
            except nb.NBodyError as err:                          # deeper than 42 levels: the reference would recurse on
                assert "42" in str(err) or "deep" in str(err).lower(), err
                # ... and the refusal was due only if the reference's own insertion of this scene goes that deep
                depth = oracle.octree_depth_f32(pos)
                assert depth >= REFUSED_FROM_DEPTH, f"trial {trial}: n={n} refused, but Octree::Add of the scene stops at depth {depth}"
                refused += 1
                continue
            a = e.accelerations()
            st = e.bh_stats()
        ref, com, nodes = oracle.octree_forces_f32(pos, m, theta, pow_mode=3, div_mode=div_mode)
        assert oracle.last_max_depth() < REFUSED_FROM_DEPTH, (trial, n)      # (a frame that should have been refused and was not)
        np.testing.assert_array_equal(a, ref, err_msg=f"trial {trial}: n={n} theta={theta} div_mode={div_mode}")
        np.testing.assert_array_equal(st["root_com"], com)
        assert st["nodes"] == nodes, (trial, n)
        ran += 1
    print(f"force-pass fuzz: {ran} of {trials} scenes compared in every bit, {refused} refused and confirmed "
          f"(the oracle's insertion of the same scene passes depth 42)")
    assert ran >= trials * 2 // 3, ran


@pytest.mark.parametrize("n", [5000, 8192, 30000])
def test_bh_frames_of_larger_systems_equal_the_oracle_in_every_bit(nb, oracle, n):
    # the whole-chip build (keys, sort, node words, ComputeMass in two launches), both walks of the larger systems, the
    # update and the cross-frame root (each tree is rooted at the previous tree's CoM, OctreeSearch.cpp:77-79): six whole
    # Ticks, every byte of the FParticle records, Size and the root centre of every frame
    rng = np.random.default_rng(n)
    posm = _fuzz_scene(rng, n)
    vel = np.concatenate([rng.uniform(-20, 20, (n, 3)), np.zeros((n, 1))], 1).astype(np.float32)
    q = particles_from(nb, posm, vel)
    com, size = None, 0.0
    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, vel)
        for frame in range(6):
            size_dev, out = e.tick(0.01)
            com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size, pow_mode=3)
            assert size_dev == size, frame
            np.testing.assert_array_equal(e.bh_stats()["root_com"], com)
            assert out.tobytes() == q.tobytes(), frame


@pytest.mark.parametrize("n,force_radix", [(5000, False), (5000, True), (8192, False), (12289, True), (40000, False), (40000, True),
                                           (131072, False), (131073, False), (150001, False), (300000, False)])
def test_both_sorts_of_the_larger_systems_order_the_bodies_as_the_reference_tree_does(nb, oracle, monkeypatch, n, force_radix):
    # The larger systems' own sorts of the path keys (no library on the path): tiles sorted in LDS + one merge by rank up to
    # 131072 bodies, eight one-launch radix passes above (NBODY_BH_MERGE_MAX_N moves the switch: the radix form at small sizes
    # too), ragged last tiles, and bodies that agree in the whole first key word — inside one tile, across tiles, three in a
    # row — which the merge orders by the second word on the spot and the radix form in bh_ties_kernel.  Draw order
    # (= key order), leaf boxes, node count, root CoM and every acceleration equal the oracle's tree in every bit.
    if force_radix:
        monkeypatch.setenv("NBODY_BH_MERGE_MAX_N", "4096")
    rng = np.random.default_rng(n + int(force_radix))
    posm = _fuzz_scene(rng, n)
    while len(np.unique(posm[:, :3], axis=0)) != n:            # (the narrowest clumps can put two bodies on one fp32 point:
        posm = _fuzz_scene(rng, n)                              #  the reference's Add never returns from that — another scene)
    size = np.abs(posm[:, :3]).max()
    tiny = np.float32(size * 2.0 ** -24)
    far = n - 50
    posm[far, :3] = posm[100, :3] + tiny * np.array([1, 0, 0], np.float32)           # a pair across tiles
    posm[201, :3] = posm[200, :3] + tiny * np.array([0, 1, 0], np.float32)           # three in a row
    posm[n // 2, :3] = posm[200, :3] + tiny * np.array([0, 0, 1], np.float32)
    assert len({tuple(posm[i, :3]) for i in (100, far, 200, 201, n // 2)}) == 5
    vel = np.zeros((n, 4), np.float32)
    pos = np.ascontiguousarray(posm[:, :3]); m = np.ascontiguousarray(posm[:, 3])
    ref, com, nodes = oracle.octree_forces_f32(pos, m, REF_THETA, pow_mode=3)
    boxes, order = oracle.octree_leaves_f32(pos, m)
    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, vel)
        e.compute_forces()
        a = e.accelerations()
        st = e.bh_stats()
        np.testing.assert_array_equal(e.bh_leaf_order(), order)
        np.testing.assert_array_equal(e.bh_leaf_boxes()[order], boxes)
    np.testing.assert_array_equal(a, ref)
    np.testing.assert_array_equal(st["root_com"], com)
    assert st["nodes"] == nodes and st["levels"] >= 22


def _sort_counts(e):
    import ctypes
    warm, retries = ctypes.c_longlong(), ctypes.c_longlong()
    assert e._L.nbody_debug_bh_sort_counts(e._h, ctypes.byref(warm), ctypes.byref(retries)) == 0
    return warm.value, retries.value


@pytest.mark.parametrize("n", [8192, 40000])
def test_the_sort_that_starts_from_the_previous_frames_order(nb, oracle, n):
    # A frame that follows a frame visits the bodies in the previous key order and drops them into buckets bounded by the
    # previous sorted keys (bh_keys_bucket_kernel / bh_bucket_sort_kernel).  The same scene twice: the second pass is such a
    # frame, with bodies that agree in the whole first key word inside it.  Then the host replaces the records of the RUNNING
    # simulation by a scene that has nothing to do with the order on the device — nearly all bodies in one small clump: its
    # buckets run over, the frame is given up on the device and queued again with the cold sorts.  Every acceleration, the draw
    # order, the leaf boxes and the root CoM equal the oracle's tree in every bit each time.
    rng = np.random.default_rng(n)
    posm = _fuzz_scene(rng, n)
    while len(np.unique(posm[:, :3], axis=0)) != n:
        posm = _fuzz_scene(rng, n)
    tiny = np.float32(np.abs(posm[:, :3]).max() * 2.0 ** -24)
    posm[n - 50, :3] = posm[100, :3] + tiny * np.array([1, 0, 0], np.float32)
    posm[201, :3] = posm[200, :3] + tiny * np.array([0, 1, 0], np.float32)
    vel = np.zeros((n, 4), np.float32)

    def check(e, posm):
        pos = np.ascontiguousarray(posm[:, :3]); m = np.ascontiguousarray(posm[:, 3])
        ref, com, nodes = oracle.octree_forces_f32(pos, m, REF_THETA, pow_mode=3)
        boxes, order = oracle.octree_leaves_f32(pos, m)
        e.compute_forces()
        np.testing.assert_array_equal(e.accelerations(), ref)
        st = e.bh_stats()
        np.testing.assert_array_equal(e.bh_leaf_order(), order)
        np.testing.assert_array_equal(e.bh_leaf_boxes()[order], boxes)
        np.testing.assert_array_equal(st["root_com"], com)
        assert st["nodes"] == nodes

    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, vel)
        check(e, posm)
        assert _sort_counts(e) == (0, 0)
        check(e, posm)                                          # from the first pass's order
        assert _sort_counts(e) == (1, 0)
        clump = posm.copy()
        clump[1:, :3] = (posm[1:, :3] * np.float32(2.0 ** -9) + np.float32(0.7) * np.abs(posm[:, :3]).max()).astype(np.float32)
        while len(np.unique(clump[:, :3], axis=0)) != n:
            clump[1:, :3] += rng.normal(0, tiny * 64, (n - 1, 3)).astype(np.float32)
        e.push_particles(particles_from(nb, clump, vel))
        check(e, clump)
        warm, retries = _sort_counts(e)
        assert retries == 1 and warm == 2, (warm, retries)
        check(e, clump)                                         # and on from the retried frame's order
        assert _sort_counts(e) == (3, 1)


def test_frames_queued_behind_a_frame_the_warm_sort_gives_up(nb, oracle):
    # nbody_step queues its frames without waiting.  When the warm sort gives one of them up, that frame and every frame
    # queued behind it leave the state alone; bh_collect queues them again.  Seven Ticks in three calls, the records replaced
    # by the clump in between: every byte of the records equals the oracle's after each call.
    n = 8192
    rng = np.random.default_rng(77)
    posm = _fuzz_scene(rng, n)
    vel = np.concatenate([rng.uniform(-20, 20, (n, 3)), np.zeros((n, 1))], 1).astype(np.float32)
    q = particles_from(nb, posm, vel)
    com, size = None, 0.0
    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, vel)
        e.step(0.01, 2)
        for _ in range(2):
            com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size, pow_mode=3)
        assert e.particles().tobytes() == q.tobytes()
        assert _sort_counts(e) == (1, 0)
        clump = (rng.uniform(-30, 30, (n - 1, 3)) + 500.0).astype(np.float32)    # a scene that has nothing to do with the order on the device
        assert len(np.unique(clump, axis=0)) == n - 1
        q["Position"][1:] = clump
        q["Mass"] *= np.float32(1e-4)                           # (so that the clump does not blow the scene apart within five frames)
        e.push_particles(q)
        e.step(0.01, 3)                                         # the first of the three is given up, all three are queued again
        for _ in range(3):
            com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size, pow_mode=3)
        assert e.particles().tobytes() == q.tobytes()
        warm, retries = _sort_counts(e)
        assert retries >= 1 and warm >= 1 + 3, (warm, retries)     # (the clump flies apart: a later frame may be given up as well)
        e.step(0.01, 2)
        for _ in range(2):
            com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size, pow_mode=3)
        assert e.particles().tobytes() == q.tobytes()
        np.testing.assert_array_equal(e.bh_stats()["root_com"], com)


def test_size_of_the_next_frame_comes_out_of_the_walk_only_while_nothing_else_moves_a_body(nb, oracle):
    # A larger system's walk leaves the NEXT frame's Size (ComputeCubeSize of the positions it has just written) in slot words, so
    # that the next frame needs no pass over the positions.  Whatever else moves a body in between — the two-call step (force pass,
    # then the update kernel), records pushed by the host, a position pointer handed out — must send the next frame back to looking
    # at the positions.  Ten Ticks driven five different ways, with a runaway body that owns Size in some of them: every byte of
    # the records, Size and the root centre equal the oracle's after each.
    n = 6000
    rng = np.random.default_rng(11)
    posm = _fuzz_scene(rng, n)
    vel = np.concatenate([rng.uniform(-20, 20, (n, 3)), np.zeros((n, 1))], 1).astype(np.float32)
    vel[7, :3] = (4.0e5, -1.0e5, 2.0e5)                        # body 7 leaves: it is Size from the second frame on
    q = particles_from(nb, posm, vel)
    com, size = None, 0.0

    def ref_tick():
        nonlocal com, size
        com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size, pow_mode=3)

    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, vel)
        e.step(0.01, 2); ref_tick(); ref_tick()                 # a pass over the positions, then Size out of the first frame's walk
        assert e.particles().tobytes() == q.tobytes()
        e.step_begin(); e.step_end(0.01); ref_tick()            # the two-call step: its update kernel moves the bodies
        assert e.particles().tobytes() == q.tobytes()
        size_dev, out = e.tick(0.01); ref_tick()                # ... so this frame must look at the positions again
        assert size_dev == size and out.tobytes() == q.tobytes()
        q["Position"][7] *= np.float32(0.25); q["Velocity"][7] = 0   # the host pulls the runaway back in: Size shrinks
        e.push_particles(q)
        size_dev, out = e.tick(0.01); ref_tick()
        assert size_dev == size and out.tobytes() == q.tobytes()
        e.step(0.01, 2); ref_tick(); ref_tick()
        assert e.particles().tobytes() == q.tobytes()
        e.device_ptr(nb.BUF_POSM)                              # the caller holds the positions from now on: every frame looks
        e.step(0.01, 3); ref_tick(); ref_tick(); ref_tick()
        assert e.particles().tobytes() == q.tobytes()
        np.testing.assert_array_equal(e.bh_stats()["root_com"], com)


@pytest.mark.parametrize("n", [9000, 30000, 120000])
def test_levels_that_only_the_first_body_in_key_order_opens(nb, oracle, n):
    # ComputeMass over chunks visits only the levels on which the chunk's bodies open cells.  The first body of the key order has no
    # predecessor: its ladder starts at the root.  A shallow scene with a pair of near-twins in every corner of the box — whichever
    # corner comes first in the key order, its pair's common cells go twenty levels deeper than anything else in the first chunk
    # (found by the frames fuzz: those levels were left out and the pair's cells kept stale sums).
    rng = np.random.default_rng(n)
    posm = np.concatenate([rng.uniform(-1000, 1000, (n, 3)), 10.0 ** rng.uniform(0, 3, (n, 1))], 1).astype(np.float32)
    k = 0
    for sx in (-1, 1):
        for sy in (-1, 1):
            for sz in (-1, 1):
                corner = np.array([sx, sy, sz], np.float32) * np.float32(999.5)
                posm[k, :3] = corner
                posm[k + 1, :3] = corner + np.float32(2.0 ** -11) * np.array([1, -1, 1], np.float32)
                k += 2
    assert len(np.unique(posm[:, :3], axis=0)) == n
    pos = np.ascontiguousarray(posm[:, :3]); m = np.ascontiguousarray(posm[:, 3])
    ref, com, nodes = oracle.octree_forces_f32(pos, m, REF_THETA, pow_mode=3)
    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, np.zeros((n, 4), np.float32))
        for _ in range(2):                                      # a cold pass and one from its order
            e.compute_forces()
            st = e.bh_stats()
            np.testing.assert_array_equal(st["root_com"], com)
            np.testing.assert_array_equal(e.accelerations(), ref)
            assert st["nodes"] == nodes and st["levels"] >= 20


@pytest.mark.timeout(300)
def test_a_clump_of_thousands_that_share_the_whole_first_key_word(nb, oracle):
    # A runaway body blows Size up until a whole clump sits in ONE cell of level 21: tens of thousands of bodies agree in the
    # whole first key word and are told apart by the second.  The radix sort of the larger systems orders such a run by every body
    # finding its own rank in it (found by the frames fuzz: the run's first thread sorting it by insertion took a minute).
    # 30000 bodies inside the level-21 cell at the origin, 110000 around: accelerations, draw order, node count and root CoM equal the
    # oracle's tree in every bit, cold and from the previous order.
    n, clump = 140000, 30000
    rng = np.random.default_rng(21)
    posm = np.concatenate([rng.uniform(-1000, 1000, (n, 3)), 10.0 ** rng.uniform(0, 3, (n, 1))], 1).astype(np.float32)
    posm[0, :3] = (1000.0, -1000.0, 1000.0)                     # Size = 1000: a cell of level 21 is 1000 / 2^21 = 4.77e-4 wide
    posm[1:1 + clump, :3] = rng.uniform(1e-5, 4.7e-4, (clump, 3)).astype(np.float32)
    posm[1:1 + clump, 3] *= np.float32(1e-12)                    # (light: the clump's own forces stay finite in fp32)
    assert len(np.unique(posm[:, :3], axis=0)) == n
    pos = np.ascontiguousarray(posm[:, :3]); m = np.ascontiguousarray(posm[:, 3])
    ref, com, nodes = oracle.octree_forces_f32(pos, m, REF_THETA, pow_mode=3)
    _, order = oracle.octree_leaves_f32(pos, m)
    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, np.zeros((n, 4), np.float32))
        for _ in range(2):
            e.compute_forces()
            st = e.bh_stats()
            np.testing.assert_array_equal(e.bh_leaf_order(), order)
            np.testing.assert_array_equal(st["root_com"], com)
            np.testing.assert_array_equal(e.accelerations(), ref)
            assert st["nodes"] == nodes and st["levels"] > 21


def test_bh_fuzz_every_byte_of_the_frames_on_random_scenes(nb, oracle):
    """Whole Ticks (.cpp:25-31) on random scenes, driven the ways a host drives them: sizes on both sides of every switch of the
    theta > 0 path, random opening angles and both readings of the CoM division, velocities from a crawl to speeds that tear the
    previous frame's order apart (the sort that starts from it must then give frames up and queue them again), frames queued
    several at a time (`step`), one by one with the mirror and Size (`tick`) or as force pass + update (`step_begin` / `step_end`),
    and between the calls the host may replace records, change the opening angle or take the position buffer into its own hands.
    After every call every byte of the records —
    Position, Velocity, Acceleration, Mass — equals the oracle's, and so do Size and the root centre at the end.
    NBODY_FUZZ_SEED / NBODY_FUZZ_TRIALS run it longer."""
    rng = np.random.default_rng(int(os.environ.get("NBODY_FUZZ_SEED", "404")))
    trials = int(os.environ.get("NBODY_FUZZ_TRIALS", "30"))
    ran = warm_total = retries_total = refused = 0
    for trial in range(trials):
        u = rng.random()
        n = (int(rng.integers(2, 4097)) if u < 0.3 else int(rng.integers(4097, 21000)) if u < 0.72 else
             int(rng.integers(21000, 70000)) if u < 0.96 else int(rng.integers(70000, 300000)))
        theta = float(rng.choice([1.0, 1.0, 0.5, 1.7]))
        div_mode = int(rng.integers(0, 2))
        posm = _fuzz_scene(rng, n)
        posm[:, 3] *= np.float32(10.0 ** rng.uniform(-7, -2))    # (from scenes that barely move under gravity to ones that collapse)
        speed = 10.0 ** rng.uniform(-1, 4.5)
        vel = np.concatenate([rng.normal(0, speed, (n, 3)), np.zeros((n, 1))], 1).astype(np.float32)
        dt = float(rng.choice([0.01, 0.002, 0.05]))
        q = particles_from(nb, posm, vel)
        com, size = None, 0.0
        what = f"trial {trial}: n={n} theta={theta} div_mode={div_mode} speed={speed:.3g} dt={dt}"
        with nb.NBodyEngine(n, theta=theta, bh_div_mode=div_mode) as e:
            e.set_state(posm, vel)
            try:
                for call in range(int(rng.integers(2, 6))):
                    size_dev = None
                    how = rng.random()
                    if how < 0.4:
                        k = int(rng.integers(1, 4))
                        e.step(dt, k)
                        out = e.particles()
                    elif how < 0.8:
                        k = 1
                        size_dev, out = e.tick(dt)
                    else:                                        # the two-call step: force pass, then the update kernel
                        k = 1
                        e.step_begin(); e.step_end(dt)
                        out = e.particles()
                    for _ in range(k):
                        com, size = oracle.tick_aos_f32(q, dt, theta=theta, root_com=com, size=size, pow_mode=3, div_mode=div_mode)
                        assert oracle.last_max_depth() < REFUSED_FROM_DEPTH, (what, call)   # (a frame that should have been refused)
                    assert out.tobytes() == q.tobytes(), (what, call)
                    if size_dev is not None:
                        assert size_dev == size, (what, call)
                    between = rng.random()                       # what a host may do between frames
                    if between < 0.15:                           # ... replace some records (bodies jump: the order on the device is stale)
                        some = rng.random(n) < rng.choice([0.001, 0.05, 0.9])
                        q["Position"][some] = (q["Position"][some] * np.float32(rng.choice([0.5, 1.0, 3.0])) +
                                               rng.normal(0, 1.0, (int(some.sum()), 3)).astype(np.float32))
                        e.push_particles(q)
                    elif between < 0.25:                         # ... change the opening angle
                        theta = float(rng.choice([1.0, 0.5, 0.7, 1.7]))
                        e.set_theta(theta)
                    elif between < 0.3:                          # ... ask for the position buffer: it is the caller's from now on
                        e.device_ptr(nb.BUF_POSM)
            except nb.NBodyError as err:                          # deeper than 42 levels: the reference would recurse on
                assert "42" in str(err) or "deep" in str(err).lower(), (what, err)
                # The refusal was due only if the reference's own insertion goes that deep in one of this call's k frames: the
                # oracle runs them from the state the call started from; the frames before the refused one must be on the device
                # (a refused frame and everything queued behind it leave the state alone).
                confirmed = False
                for _ in range(k):
                    root = np.zeros(3, np.float32) if com is None else com
                    depth = oracle.octree_depth_f32(q["Position"], root_origin=root)
                    if depth >= REFUSED_FROM_DEPTH:
                        confirmed = True
                        break
                    com, size = oracle.tick_aos_f32(q, dt, theta=theta, root_com=com, size=size, pow_mode=3, div_mode=div_mode)
                assert confirmed, f"{what}, call {call} ({k} frames): refused, but Octree::Add of no frame passes depth 42"
                got = e.particles()
                for f in ("Position", "Velocity", "Mass"):
                    np.testing.assert_array_equal(got[f], q[f], err_msg=f"{what}: {f} after the refused frame")
                refused += 1
                continue
            np.testing.assert_array_equal(e.bh_stats()["root_com"], com, err_msg=what)
            if n > 4096:
                w, r = _sort_counts(e)
                warm_total += w; retries_total += r
        ran += 1
    assert ran >= trials // 2, ran
    print(f"frames fuzz: {ran} of {trials} scenes ran to the end, {refused} refused and confirmed (the oracle's insertion of the "
          f"refused frame passes depth 42); larger systems: {warm_total} frames sorted from the previous order, "
          f"{retries_total} times frames were queued again")


@pytest.mark.parametrize("n", [2559, 2560, 4096, 4097, 12288, 12289, 98304, 98305])
def test_both_sides_of_the_round_four_switches_cold_and_warm(nb, oracle, n):
    # Sizes on both sides of the theta > 0 path's switches that round 4 added or moved — small systems walking the tree in LDS / in
    # global memory (2560), the one-workgroup build / the whole chip (4097), a wave / sixteen lanes per body (12289), ComputeMass over
    # chunks of 256 / 1024 bodies (98305) — each with a first pass (cold sorts, Size from the bounds kernel) and a second one that
    # starts from the first one's order: accelerations, draw order, node count and root CoM equal the oracle's tree in every bit.
    rng = np.random.default_rng(n)
    posm = _fuzz_scene(rng, n)
    while len(np.unique(posm[:, :3], axis=0)) != n:
        posm = _fuzz_scene(rng, n)
    vel = np.zeros((n, 4), np.float32)
    pos = np.ascontiguousarray(posm[:, :3]); m = np.ascontiguousarray(posm[:, 3])
    try:
        ref, com, nodes = oracle.octree_forces_f32(pos, m, REF_THETA, pow_mode=3)
    except Exception:                                          # (a scene deeper than the oracle's own limit)
        pytest.skip("scene too deep for the oracle")
    _, order = oracle.octree_leaves_f32(pos, m)
    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, vel)
        for _ in range(2):
            try:
                e.compute_forces()
            except nb.NBodyError as err:                          # deeper than 42 levels: the reference would recurse on
                assert "42" in str(err) or "deep" in str(err).lower(), err
                assert oracle.octree_depth_f32(pos) >= REFUSED_FROM_DEPTH
                pytest.skip("two bodies closer than Size / 2^42")
            np.testing.assert_array_equal(e.accelerations(), ref)
            np.testing.assert_array_equal(e.bh_leaf_order(), order)
            st = e.bh_stats()
            np.testing.assert_array_equal(st["root_com"], com)
            assert st["nodes"] == nodes


@pytest.mark.timeout(1200)
@pytest.mark.parametrize("n", [1 << 20, (1 << 20) + 1, 1500000, 1 << 21])
def test_a_million_bodies_and_more_equal_the_oracle_in_every_bit(nb, oracle, n):
    # N = 2^20 is the largest system whose ComputeMass runs in two launches (every thread of bh_sweep_top_kernel owns one chunk of
    # 1024 bodies: nchunks == kTopT) and the size of every headline theta > 0 figure; above it the host waits for the deepest level
    # and launches one bh_sweep_level_kernel per level + bh_finish_kernel (OctreeSearch.h:83-97 level by level).  The shipped kind
    # of scene (CreateSpacePoints, .cpp:58-72) at the shipped opening angle: one cold force pass — accelerations, node count, root
    # CoM — and two whole Ticks that start from the previous order (.cpp:25-31: Size, tree rooted at the previous CoM, walk,
    # kick-drift), every bit / byte against the oracle.
    posm, vel = nb.ic_reference_box(n, 1000.0, seed=n % 1000 + 1)
    pos = np.ascontiguousarray(posm[:, :3]); m = np.ascontiguousarray(posm[:, 3])
    ref, com0, nodes = oracle.octree_forces_f32(pos, m, REF_THETA, pow_mode=3)
    assert oracle.last_max_depth() < REFUSED_FROM_DEPTH
    q = particles_from(nb, posm, vel)
    com, size = None, 0.0
    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, vel)
        e.compute_forces()                                     # cold sorts, Size from the bounds kernel; the next root stays at zero
        st = e.bh_stats()
        np.testing.assert_array_equal(e.accelerations(), ref)
        np.testing.assert_array_equal(st["root_com"], com0)
        assert st["nodes"] == nodes
        for frame in range(2):
            size_dev, out = e.tick(0.01)
            com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size, pow_mode=3)
            assert size_dev == size, frame
            np.testing.assert_array_equal(e.bh_stats()["root_com"], com)
            assert out.tobytes() == q.tobytes(), frame
        warm, _ = _sort_counts(e)
        assert warm >= 2                                       # (both Ticks started from the previous frame's order)


@pytest.mark.parametrize("n", [5000, 30000, 140000])
def test_compute_mass_with_a_launch_per_level_at_sizes_that_are_cheap_to_check(nb, oracle, monkeypatch, n):
    # NBODY_BH_LEVEL_SWEEPS=1 (read when the theta > 0 state is created) sends a system of any size down the branch systems above
    # 2^20 bodies take: host wait for the deepest level, one launch per level, bh_finish_kernel.  Clumpy scenes (deep chains), a
    # cold pass and four Ticks, two of them queued in one call (the wait sits INSIDE the queueing of each frame): every bit / byte.
    monkeypatch.setenv("NBODY_BH_LEVEL_SWEEPS", "1")
    rng = np.random.default_rng(n + 5)
    posm = _fuzz_scene(rng, n)
    while len(np.unique(posm[:, :3], axis=0)) != n or oracle.octree_depth_f32(posm[:, :3]) >= REFUSED_FROM_DEPTH:
        posm = _fuzz_scene(rng, n)
    posm[:, 3] *= np.float32(1e-4)                              # (light: the clumps do not collapse within four frames)
    vel = np.concatenate([rng.uniform(-20, 20, (n, 3)), np.zeros((n, 1))], 1).astype(np.float32)
    pos = np.ascontiguousarray(posm[:, :3]); m = np.ascontiguousarray(posm[:, 3])
    ref, com0, nodes = oracle.octree_forces_f32(pos, m, REF_THETA, pow_mode=3)
    q = particles_from(nb, posm, vel)
    com, size = None, 0.0
    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, vel)
        e.compute_forces()
        st = e.bh_stats()
        np.testing.assert_array_equal(e.accelerations(), ref)
        np.testing.assert_array_equal(st["root_com"], com0)
        assert st["nodes"] == nodes
        for k in (1, 2, 1):
            try:
                e.step(0.01, k)
            except nb.NBodyError as err:                        # a narrow clump can put two bodies on one path for 42 levels as it moves:
                assert "42" in str(err), err                    # the refusal must be the oracle's finding too, on the same frame
                for _ in range(k):
                    if oracle.octree_depth_f32(q["Position"], root_origin=np.zeros(3, np.float32) if com is None else com) >= REFUSED_FROM_DEPTH:
                        break
                    com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size, pow_mode=3)
                else:
                    raise AssertionError("refused, but Octree::Add of no frame of the call passes depth 42")
                got = e.particles()
                np.testing.assert_array_equal(got["Position"], q["Position"])
                np.testing.assert_array_equal(got["Velocity"], q["Velocity"])
                return
            for _ in range(k):
                com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size, pow_mode=3)
                assert oracle.last_max_depth() < REFUSED_FROM_DEPTH
            assert e.particles().tobytes() == q.tobytes(), k
        np.testing.assert_array_equal(e.bh_stats()["root_com"], com)


@pytest.mark.parametrize("walk,rows_max,wave_max,sizes", [
    ("a lane per body", "0", "0", (4100, 9000)),
    ("sixteen lanes per body", "1000000", "0", (9000, 15000)),
    ("a wave per body", "1000000", "1000000", (15000, 30000)),
])
def test_every_walk_of_the_larger_systems_at_sizes_it_does_not_get_by_default(nb, oracle, monkeypatch, walk, rows_max, wave_max, sizes):
    # Which walk a system above 4096 bodies gets follows from its size (a wave per body up to 12288, a lane per body above; sixteen
    # lanes per body only on request since round 5); NBODY_BH_ROWS_MAX_N / NBODY_BH_WAVE_MAX_N (read when the theta > 0 state is
    # created) move the boundaries.  Every walk on both kinds of scene at sizes the defaults give to another one: a cold pass and two
    # Ticks, every bit / byte.
    monkeypatch.setenv("NBODY_BH_ROWS_MAX_N", rows_max)
    monkeypatch.setenv("NBODY_BH_WAVE_MAX_N", wave_max)
    for n in sizes:
        for posm, vel in (nb.ic_reference_box(n, 1000.0, seed=n % 97 + 1), nb.ic_plummer(n, seed=n % 89 + 1)):
            pos = np.ascontiguousarray(posm[:, :3]); m = np.ascontiguousarray(posm[:, 3])
            ref, com0, nodes = oracle.octree_forces_f32(pos, m, REF_THETA, pow_mode=3)
            assert oracle.last_max_depth() < REFUSED_FROM_DEPTH
            q = particles_from(nb, posm, vel)
            com, size = None, 0.0
            with nb.NBodyEngine(n, theta=REF_THETA) as e:
                e.set_state(posm, vel)
                e.compute_forces()
                np.testing.assert_array_equal(e.accelerations(), ref)
                assert e.bh_stats()["nodes"] == nodes
                for frame in range(2):
                    size_dev, out = e.tick(0.01)
                    com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size, pow_mode=3)
                    assert oracle.last_max_depth() < REFUSED_FROM_DEPTH
                    assert size_dev == size, (walk, n, frame)
                    assert out.tobytes() == q.tobytes(), (walk, n, frame)


def test_the_first_two_frames_right_after_creation_of_small_systems(nb, oracle):
    # The theta > 0 state is created by the first theta > 0 call, right in front of its first frame, and its creation clears
    # device words with hipMemset on the NULL stream while the frames run on the context's non-blocking stream.  Round 4 waited for
    # the null stream only behind the larger systems' part of the creation (the advisor's finding): for N <= 4096 a late fill could
    # zero the header's frame count or the previous tree's CoM after the first frame had written them — a wrong steps_done, a second
    # tree rooted at zero.  Forty systems, each created and ticked twice at once: every byte, Size, the root centre, steps_done.
    rng = np.random.default_rng(4096)
    ran = 0
    for trial in range(40):
        n = int(rng.integers(2, 4097))
        posm = _fuzz_scene(rng, n)
        vel = np.concatenate([rng.uniform(-50, 50, (n, 3)), np.zeros((n, 1))], 1).astype(np.float32)
        posm[:, 3] *= np.float32(1e-4)                          # (light: no clump collapses within the two frames)
        q = particles_from(nb, posm, vel)
        ref, com, size, deep = [], None, 0.0, False
        for frame in range(2):                                  # the oracle's two frames first: scenes the reference itself could not insert are left out
            com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size, pow_mode=3) if not deep else (com, size)
            deep = deep or oracle.last_max_depth() >= REFUSED_FROM_DEPTH - 1
            ref.append((q.copy(), com.copy(), size))
        if deep:
            continue
        ran += 1
        with nb.NBodyEngine(n, theta=REF_THETA) as e:
            e.set_state(posm, vel)
            for frame in range(2):
                size_dev, out = e.tick(0.01)
                q, com, size = ref[frame]
                assert size_dev == size, (trial, n, frame)
                np.testing.assert_array_equal(e.bh_stats()["root_com"], com, err_msg=f"trial {trial} n={n} frame {frame}")
                assert out.tobytes() == q.tobytes(), (trial, n, frame)
            assert e.steps_done() == 2, (trial, n)
    assert ran >= 30, ran


@pytest.mark.parametrize("n", [20000, 140000])
def test_a_runaway_body_that_owns_size_does_not_cost_the_sort_from_the_previous_order(nb, oracle, n):
    # The shipped kind of scene throws bodies out within a few hundred frames (close encounters at G = 1e4): Size — the largest
    # |coordinate| about the world origin, OctreeSearch.cpp:47-56 — becomes theirs, 1e7 .. 1e9 for a box of 1e3, and every other body
    # then lies in ONE cell of level 21 (Size / 2^21 wide): all first key words agree.  The sort that starts from the previous frame's
    # order used to tell its buckets apart by first words alone: one bucket took every body, EVERY frame was given up and queued again
    # with the cold sorts (round 5's bench line: 267 give-ups in 300 frames at N = 65536).  Its boundaries are whole keys now: six
    # Ticks of such a scene, none given up, every byte of the records, Size and the root centre against the oracle.
    rng = np.random.default_rng(n)
    posm = np.concatenate([rng.uniform(-1.0, 1.0, (n, 3)) + 3.0, 10.0 ** rng.uniform(-12, -9, (n, 1))], 1).astype(np.float32)
    posm[5, :3] = (1.0e7, -2.0e6, 3.0e6)                        # the runaway: Size = 1e7, a cell of level 21 is 9.5 wide
    posm[5, 3] = 1e-24                                          # (light: the trees' root — the previous CoM, .cpp:77-79 — stays with the cloud)
    assert len(np.unique(posm[:, :3], axis=0)) == n
    vel = np.concatenate([rng.normal(0, 0.05, (n, 3)), np.zeros((n, 1))], 1).astype(np.float32)
    vel[5, :3] = (3.0e5, 0.0, 0.0)                              # ... and it keeps going: Size grows by 3e3 a frame
    q = particles_from(nb, posm, vel)
    com, size = None, 0.0
    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, vel)
        for k in (1, 3, 2):
            e.step(0.01, k)
            for _ in range(k):
                com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size, pow_mode=3)
                assert oracle.last_max_depth() < REFUSED_FROM_DEPTH
            assert e.particles().tobytes() == q.tobytes(), k
        st = e.bh_stats()
        np.testing.assert_array_equal(st["root_com"], com)
        assert st["levels"] > 21                                # (the scene really lives below the first key word)
        warm, retries = _sort_counts(e)
        # (frame 1 is rooted at zero, frame 2 at the cloud's own centre: the root box jumps once and that frame is queued again)
        assert retries <= 1 and warm >= 5, (warm, retries)


@pytest.mark.parametrize("n", [9000, 150000])
def test_bucket_counts_that_do_not_add_up_send_the_frame_back_to_the_cold_sorts(nb, oracle, n):
    # what a creation fill that ran late (or never) can do to the first frame sorted from the previous order — the bucket
    # counts are not those of this frame's bodies: ranges past the arrays' ends, places of the order never written, and behind them
    # kernels that chase links through whatever stands there.  Injected once (nbody_debug_bh_poison: every count word := 3 between
    # two frames): the bucket sort sees that the counts do not add up to N and gives the frame up, it comes back with the cold
    # sorts, and every byte of the records equals the oracle's — before, through and after.
    rng = np.random.default_rng(n)
    posm = _fuzz_scene(rng, n)
    posm[:, 3] *= np.float32(1e-5)
    vel = np.concatenate([rng.uniform(-20, 20, (n, 3)), np.zeros((n, 1))], 1).astype(np.float32)
    q = particles_from(nb, posm, vel)
    com, size = None, 0.0

    def ref(k):
        nonlocal com, size
        for _ in range(k):
            com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size, pow_mode=3)
            assert oracle.last_max_depth() < REFUSED_FROM_DEPTH

    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, vel)
        e.step(0.01, 2); ref(2)
        assert e.particles().tobytes() == q.tobytes()
        _, before = _sort_counts(e)
        assert e._L.nbody_debug_bh_poison(e._h, 1) == 0
        e.step(0.01, 3); ref(3)
        assert e.particles().tobytes() == q.tobytes()
        _, after = _sort_counts(e)
        assert after == before + 1, (before, after)
        e.step(0.01, 2); ref(2)
        assert e.particles().tobytes() == q.tobytes()
        np.testing.assert_array_equal(e.bh_stats()["root_com"], com)


def test_kernel_time_counts_every_frame_once_when_frames_are_queued_again(nb, oracle):
    # nbody_kernel_time(FORCES) at theta > 0 brackets every queued frame with an event pair.  A frame the sort from the previous order
    # gives up — and the frames queued behind it — run (almost) empty inside their pairs and are queued again; round 4 left the re-run
    # outside any pair and the empty runs in the count (the advisor's finding).  Now the pairs of frames that did nothing are taken
    # back and the re-run has its own: five frames = five launches, whatever was queued twice.
    n = 8192
    rng = np.random.default_rng(5)
    posm = np.concatenate([rng.uniform(-1000, 1000, (n, 3)), 10.0 ** rng.uniform(-9, -6, (n, 1))], 1).astype(np.float32)   # (a calm scene)
    vel = np.concatenate([rng.uniform(-20, 20, (n, 3)), np.zeros((n, 1))], 1).astype(np.float32)
    q = particles_from(nb, posm, vel)
    com, size = None, 0.0
    with nb.NBodyEngine(n, theta=REF_THETA, time_kernels=True) as e:
        e.set_state(posm, vel)
        e.step(0.01, 2)
        ms2, k2 = e.kernel_time(nb.KERNEL_FORCES)
        assert k2 == 2 and ms2 > 0
        for _ in range(2):
            com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size, pow_mode=3)
        clump = (rng.uniform(-30, 30, (n - 1, 3)) + 500.0).astype(np.float32)    # a scene that has nothing to do with the order on the device
        assert len(np.unique(clump, axis=0)) == n - 1
        q["Position"][1:] = clump
        e.push_particles(q)
        e.step(0.01, 3)                                         # the first of the three is given up; all three are queued again
        _, retries = _sort_counts(e)
        assert retries >= 1
        ms5, k5 = e.kernel_time(nb.KERNEL_FORCES)
        assert k5 == 5 and ms5 > ms2, (k5, ms5, ms2)
        for _ in range(3):
            com, size = oracle.tick_aos_f32(q, 0.01, theta=REF_THETA, root_com=com, size=size, pow_mode=3)
        assert e.particles().tobytes() == q.tobytes()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("sort_both", ["", "0", "1"])
def test_cold_sorts_by_both_key_words_where_first_words_agree_all_over(nb, oracle, monkeypatch, sort_both):
    # Above 131072 bodies a cold frame sorts by radix passes on the FIRST key word and puts runs of equal first words right afterwards,
    # every body of a run finding its rank by counting — quadratic in the run.  In a scene whose Size a runaway body owns nearly all
    # bodies are ONE run.  The frame before has counted the neighbours that agree in the first word (header word 6): where they are
    # all over, the next cold sort goes by both words — eight passes on the second, eight on the first, no ties left to place.
    # So does the first frame of a new scene (nothing is known yet: insurance).  NBODY_BH_WARM_SORT=0 (read when the theta > 0 state is
    # created) makes every frame a cold one; NBODY_BH_SORT_BOTH=0 / 1 pins the one sort or the other (the run placed by counting still
    # has to be right: it is what a cold frame behind an ordinary frame uses).  Accelerations, draw order, node count and root CoM
    # equal the oracle's tree in every bit, all three ways.
    monkeypatch.setenv("NBODY_BH_WARM_SORT", "0")
    monkeypatch.setenv("NBODY_BH_SORT_BOTH", sort_both) if sort_both else monkeypatch.delenv("NBODY_BH_SORT_BOTH", raising=False)
    n = 150000
    rng = np.random.default_rng(150)
    posm = np.concatenate([rng.uniform(-1.0, 1.0, (n, 3)) + 3.0, 10.0 ** rng.uniform(-12, -9, (n, 1))], 1).astype(np.float32)
    posm[5, :3] = (1.0e7, -2.0e6, 3.0e6); posm[5, 3] = 1e-24       # the runaway: a cell of level 21 is 9.5 wide
    posm[100:40100, :3] = rng.uniform(-900.0, 900.0, (40000, 3)).astype(np.float32)   # ... and a part of the bodies elsewhere: runs of every length
    assert len(np.unique(posm[:, :3], axis=0)) == n
    pos = np.ascontiguousarray(posm[:, :3]); m = np.ascontiguousarray(posm[:, 3])
    ref, com, nodes = oracle.octree_forces_f32(pos, m, REF_THETA, pow_mode=3)
    assert oracle.last_max_depth() < REFUSED_FROM_DEPTH
    _, order = oracle.octree_leaves_f32(pos, m)
    with nb.NBodyEngine(n, theta=REF_THETA) as e:
        e.set_state(posm, np.zeros((n, 4), np.float32))
        for _ in range(3):
            e.compute_forces()
            st = e.bh_stats()
            np.testing.assert_array_equal(e.bh_leaf_order(), order)
            np.testing.assert_array_equal(st["root_com"], com)
            np.testing.assert_array_equal(e.accelerations(), ref)
            assert st["nodes"] == nodes and st["levels"] > 21
        assert _sort_counts(e) == (0, 0)                       # (no frame started from the previous order)

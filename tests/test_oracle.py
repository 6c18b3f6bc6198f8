"""The CPU oracle against analytic answers, an independent fp64 sum, its own tree walk and the
committed fixtures.  No GPU.  (The reference ships no tests or vectors for this path and cannot be
built here: the oracle is 'parity unpinned' — these tests pin the restatement to physics and to
itself, which is the most this image allows.)"""
import os

import numpy as np
import pytest

from conftest import rel_err

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_particle_layout(oracle):
    # FParticle: OctreeSearch.h:8-18 -> Mass@0, Position@4, Velocity@16, Acceleration@28, 40 bytes
    assert oracle.lib().oracle_sizeof_particle() == 40
    d = oracle.PARTICLE_DTYPE
    assert d.itemsize == 40
    assert [d.fields[k][1] for k in ("Mass", "Position", "Velocity", "Acceleration")] == [0, 4, 16, 28]


def test_two_body_analytic(oracle):
    # a = G m / d^2 along the separation; G = 1e4 (OctreeSearch.h:104)
    pos = np.array([[0, 0, 0], [3, 4, 0]], np.float32)
    m = np.array([2.0, 5.0], np.float32)
    a = oracle.forces_direct_f32(pos, m)
    d = 5.0
    np.testing.assert_allclose(a[0], 1e4 * 5.0 / d**2 * np.array([3, 4, 0]) / d, rtol=1e-6)
    np.testing.assert_allclose(a[1], -1e4 * 2.0 / d**2 * np.array([3, 4, 0]) / d, rtol=1e-6)


def test_coincident_pair_is_skipped(oracle):
    # OctreeSearch.h:102: d == 0 -> no contribution (direct sum only: the tree cannot hold duplicates)
    pos = np.array([[1, 1, 1], [1, 1, 1], [2, 1, 1]], np.float32)
    m = np.array([1.0, 1.0, 1.0], np.float32)
    a = oracle.forces_direct_f32(pos, m)
    assert np.all(np.isfinite(a))
    np.testing.assert_allclose(a[0], [1e4, 0, 0], rtol=1e-6)
    np.testing.assert_allclose(a[0], a[1])
    with pytest.raises(RuntimeError):
        oracle.octree_forces_f32(pos, m, 0.0)      # the reference's Add recurses without bound here


def test_single_body_and_empty_ranges(oracle):
    a = oracle.forces_direct_f32(np.zeros((1, 3), np.float32), np.ones(1, np.float32))
    assert a.shape == (1, 3) and np.all(a == 0)
    a = oracle.forces_direct_f32(np.zeros((4, 3), np.float32), np.ones(4, np.float32), i0=2, i1=2)
    assert a.shape == (0, 3)


@pytest.mark.parametrize("fixture", ["plummer_n1024_seed1", "refbox_n2000_seed1"])
def test_golden_fixtures_reproduce(oracle, fixture):
    g = np.load(os.path.join(GOLDEN, fixture + ".npz"))
    pos = np.ascontiguousarray(g["posm"][:, :3]); m = np.ascontiguousarray(g["posm"][:, 3])
    a = oracle.forces_direct_f32(pos, m)
    # same compiler flags -> bit-identical; allow 1e-6 in case libm's pow differs by an ulp
    assert rel_err(a, g["acc_direct"]).max() < 1e-6
    t, _, _ = oracle.octree_forces_f32(pos, m, 0.0)
    assert rel_err(t, g["acc_tree0"]).max() < 1e-6
    p1, v1 = oracle.kick_drift_f32(pos, g["vel"][:, :3], g["acc_direct"], float(g["dt"]))
    np.testing.assert_array_equal(p1, g["pos1"])
    np.testing.assert_array_equal(v1, g["vel1"])
    assert oracle.bounds_f32(pos) == float(g["bounds"])


@pytest.mark.parametrize("fixture", ["plummer_n1024_seed1", "refbox_n2000_seed1"])
def test_fp32_restatement_vs_fp64(oracle, fixture):
    g = np.load(os.path.join(GOLDEN, fixture + ".npz"))
    # fp32 pair law in the reference's arithmetic vs an independent double-precision sum
    assert rel_err(g["acc_direct"], g["acc_f64"]).max() < 2e-5
    # the reference's own traversal order at theta = 0 is all-pairs too
    assert rel_err(g["acc_tree0"], g["acc_f64"]).max() < 2e-5


def test_theta0_walk_equals_all_pairs_and_theta1_does_not(oracle):
    rng = np.random.default_rng(7)
    n = 512
    pos = rng.uniform(-100, 100, (n, 3)).astype(np.float32)
    m = rng.uniform(1, 5000, n).astype(np.float32)
    direct = oracle.forces_direct_f32(pos, m)
    t0, com, cnt = oracle.octree_forces_f32(pos, m, 0.0)
    assert rel_err(t0, direct).max() < 2e-5
    assert cnt >= n
    # root CoM = mass-weighted mean
    np.testing.assert_allclose(com, (pos * m[:, None]).sum(0) / m.sum(), rtol=1e-4, atol=1e-2)
    t1, _, _ = oracle.octree_forces_f32(pos, m, 1.0)      # the shipped opening angle is far from all-pairs
    assert rel_err(t1, direct).mean() > 0.05


def test_pow_readings_agree_to_a_few_ulp(oracle):
    g = np.load(os.path.join(GOLDEN, "plummer_n1024_seed1.npz"))
    pos = np.ascontiguousarray(g["posm"][:, :3]); m = np.ascontiguousarray(g["posm"][:, 3])
    for mode in (1, 2):
        a = oracle.forces_direct_f32(pos, m, pow_mode=mode)
        assert rel_err(a, g["acc_direct"]).max() < 3e-6


def test_kick_drift_uses_new_velocity(oracle):
    # OctreeSearch.cpp:29-30: v += dt*a first, then x += dt*v with the NEW v
    pos = np.array([[1, 2, 3]], np.float32); vel = np.array([[10, 0, -10]], np.float32)
    acc = np.array([[100, 200, 300]], np.float32)
    p1, v1 = oracle.kick_drift_f32(pos, vel, acc, 0.5)
    np.testing.assert_array_equal(v1, [[60, 100, 140]])
    np.testing.assert_array_equal(p1, [[31, 52, 73]])


def test_bounds(oracle):
    pos = np.array([[1, -7, 3], [2, 2, -6.5], [0, 0, 0]], np.float32)
    assert oracle.bounds_f32(pos) == 7.0


def test_tick_direct_matches_pieces_and_pause(oracle):
    g = np.load(os.path.join(GOLDEN, "plummer_n1024_seed1.npz"))
    n = 256
    p = np.zeros(n, oracle.PARTICLE_DTYPE)
    p["Mass"] = g["posm"][:n, 3]; p["Position"] = g["posm"][:n, :3]; p["Velocity"] = g["vel"][:n, :3]
    q = p.copy()
    com, size = oracle.tick_aos_f32(q, 0.01, theta=-1.0)
    a = oracle.forces_direct_f32(p["Position"], p["Mass"])
    p1, v1 = oracle.kick_drift_f32(p["Position"], p["Velocity"], a, 0.01)
    np.testing.assert_array_equal(q["Acceleration"], a)
    np.testing.assert_array_equal(q["Position"], p1)
    np.testing.assert_array_equal(q["Velocity"], v1)
    assert size == oracle.bounds_f32(p["Position"])
    # PhDeltaTime <= 0 freezes the physics (OctreeSearch.cpp:25)
    r = p.copy()
    oracle.tick_aos_f32(r, 0.0)
    assert r.tobytes() == p.tobytes()


def test_tick_tree_uses_previous_com_as_root(oracle):
    # OctreeSearch.cpp:77-79: the root of frame k+1 is centred on the CoM of frame k's tree
    rng = np.random.default_rng(3)
    n = 300
    p = np.zeros(n, oracle.PARTICLE_DTYPE)
    p["Mass"] = rng.uniform(1, 10, n); p["Position"] = rng.uniform(-50, 50, (n, 3)) + 5.0
    com0, _ = oracle.tick_aos_f32(p, 0.01, theta=1.0)
    assert np.linalg.norm(com0) > 1.0
    com1, _ = oracle.tick_aos_f32(p, 0.01, theta=1.0, root_com=com0)
    assert np.all(np.isfinite(p["Position"]))
    assert np.linalg.norm(com1 - com0) < 5.0


def test_energy_and_fp64_step_conserve(oracle):
    # circular two-body orbit in fp64: energy stays put under the symplectic update
    G, m, d = 1.0e4, 1.0, 10.0
    v = np.sqrt(G * m / (2 * d))
    pos = np.array([[-d / 2, 0, 0], [d / 2, 0, 0]], np.float64)
    vel = np.array([[0, -v, 0], [0, v, 0]], np.float64)
    mass = np.array([m, m], np.float64)
    ke0, pe0 = oracle.energy_f64(pos, vel, mass, g=G)
    assert pe0 == pytest.approx(-G * m * m / d)
    dt = 1e-4
    for _ in range(2000):
        a = oracle.forces_direct_f64(pos, mass, g=G)
        pos, vel = oracle.kick_drift_f64(pos, vel, a, dt)
    ke, pe = oracle.energy_f64(pos, vel, mass, g=G)
    assert abs((ke + pe) - (ke0 + pe0)) / abs(ke0 + pe0) < 5e-3   # staggered-velocity O(dt) offset only
    assert np.linalg.norm(pos[1] - pos[0]) == pytest.approx(d, rel=1e-2)


def test_openmp_threads_do_not_change_results(oracle):
    g = np.load(os.path.join(GOLDEN, "plummer_n1024_seed1.npz"))
    pos = np.ascontiguousarray(g["posm"][:, :3]); m = np.ascontiguousarray(g["posm"][:, 3])
    a = oracle.forces_direct_f32(pos, m, nthreads=max(2, min(4, oracle.max_threads())))
    np.testing.assert_array_equal(a, g["acc_direct"])


# Chenciner-Montgomery figure-eight: three equal masses (G = m = 1) on one closed curve, period 6.32591398
FIG8_X = np.array([[-0.97000436, 0.24308753, 0.0], [0.97000436, -0.24308753, 0.0], [0.0, 0.0, 0.0]])
FIG8_V = np.array([[0.4662036850, 0.4323657300, 0.0], [0.4662036850, 0.4323657300, 0.0],
                   [-0.93240737, -0.86473146, 0.0]])
FIG8_T = 6.32591398


def test_three_body_figure_eight_known_answer(oracle):
    # known answer independent of the reference: after one period the bodies are back where they started
    n_steps = 20000
    dt = FIG8_T / n_steps
    pos, vel, m = FIG8_X.copy(), FIG8_V.copy(), np.ones(3)
    a = oracle.forces_direct_f64(pos, m, g=1.0)
    vel = vel - 0.5 * dt * a                       # the update keeps v half a step behind x (kick-drift)
    for _ in range(n_steps):
        a = oracle.forces_direct_f64(pos, m, g=1.0)
        pos, vel = oracle.kick_drift_f64(pos, vel, a, dt)
    assert np.abs(pos - FIG8_X).max() < 2e-4
    assert np.abs(pos.sum(0)).max() < 1e-12        # centre of mass stays put


# ---- a second, independent restatement of the reference's octree (pure Python, numpy scalars) --------------------------
# Written from OctreeSearch.h:20-109 on its own, in a different language and a different shape (recursive objects, as the
# reference; the C oracle is iterative over arrays): two readings of the same lines that agree bit for bit are less
# likely to share a transcription error.  Small cases only (pure-Python loops).

_f = np.float32
_d = np.float64


class _PyCell:
    divide = False                                         # reading of `CenterOfMass /= TotalMass` (.h:95)

    def __init__(self, origin, size):                      # .h:31-33
        self.origin, self.size = origin, _f(size)
        self.body = None
        self.mass = _f(0)
        self.com = np.zeros(3, np.float32)
        self.kids = None

    def octant(self, p):                                   # .h:50-56
        return (4 if p[0] >= self.origin[0] else 0) | (2 if p[1] >= self.origin[1] else 0) | (1 if p[2] >= self.origin[2] else 0)

    def add(self, idx, pos):                               # .h:60-81
        if self.kids is None:
            if self.body is None:
                self.body = idx
                return
            old, self.body = self.body, None
            self.kids = []
            for i in range(8):
                c = self.origin.copy()
                # `center.X += Size * (i & 4 ? 0.5 : -0.5)`: float * double in double, added to the float in double, rounded once
                c[0] = _f(_d(c[0]) + _d(self.size) * (0.5 if i & 4 else -0.5))
                c[1] = _f(_d(c[1]) + _d(self.size) * (0.5 if i & 2 else -0.5))
                c[2] = _f(_d(c[2]) + _d(self.size) * (0.5 if i & 1 else -0.5))
                self.kids.append(_PyCell(c, _f(0.5 * _d(self.size))))
            self.kids[self.octant(pos[old])].add(old, pos)
            self.kids[self.octant(pos[idx])].add(idx, pos)
        else:
            self.kids[self.octant(pos[idx])].add(idx, pos)

    def compute_mass(self, pos, mass):                     # .h:83-97
        if self.kids is None:
            if self.body is not None:
                self.com = pos[self.body].copy()
                self.mass = mass[self.body]
            return
        for k in self.kids:
            k.compute_mass(pos, mass)
            self.mass = _f(self.mass + k.mass)
            for a in range(3):
                self.com[a] = _f(self.com[a] + _f(k.mass * k.com[a]))
        if self.mass != 0:
            if _PyCell.divide:                             # the other reading of FVector::operator/=: three divisions
                for a in range(3):
                    self.com[a] = _f(self.com[a] / self.mass)
            else:
                rv = _f(_f(1.0) / self.mass)               # FVector::operator/=(float): scale by the reciprocal [UE4 4.9]
                for a in range(3):
                    self.com[a] = _f(self.com[a] * rv)
        else:
            self.com = self.origin.copy()

    def forces(self, p, theta, acc):                       # .h:99-108
        if self.kids is None and self.body is None:
            return
        dx = _f(self.com[0] - p[0]); dy = _f(self.com[1] - p[1]); dz = _f(self.com[2] - p[2])
        d = np.sqrt(_f(_f(_f(dx * dx) + _f(dy * dy)) + _f(dz * dz)))      # FVector::Dist: fp32 throughout
        if d == 0:
            return
        if _f(self.size / d) < theta or self.body is not None:
            s = _f(1e4 * _d(self.mass) / _d(d) ** 3)        # double expression, then scalar * FVector takes a float
            acc[0] = _f(acc[0] + _f(s * dx)); acc[1] = _f(acc[1] + _f(s * dy)); acc[2] = _f(acc[2] + _f(s * dz))
        elif self.kids is not None:
            for k in self.kids:
                k.forces(p, theta, acc)


def _py_draw(cell, boxes, order):                          # DrawOctreeBoxes, OctreeSearch.cpp:36-45
    if cell is None:
        return
    if cell.kids is None and cell.body is not None:
        boxes.append((cell.origin[0], cell.origin[1], cell.origin[2], cell.size))   # DrawDebugBox(Origin, (Size,Size,Size))
        order.append(cell.body)                                                      # DrawDebugPoint(Particle->Position)
    elif cell.kids is not None:
        for k in cell.kids:
            _py_draw(k, boxes, order)


def _py_create_octree(pos, mass, theta, root_origin, root_size, draw=None):
    root = _PyCell(np.asarray(root_origin, np.float32).copy(), root_size)       # OctreeSearch.cpp:77-79
    for i in range(len(pos)):
        root.add(i, pos)                                                         # .cpp:80
    root.compute_mass(pos, mass)                                                 # .cpp:81
    acc = np.zeros((len(pos), 3), np.float32)
    for i in range(len(pos)):
        root.forces(pos[i], _f(theta), acc[i])                                   # .cpp:83-86
    if draw is not None:
        _py_draw(root, draw[0], draw[1])
    return acc, root.com


@pytest.mark.parametrize("theta", [0.0, 0.5, 1.0])
@pytest.mark.parametrize("n,seed", [(2, 0), (9, 1), (64, 2), (200, 3)])
def test_c_octree_agrees_with_an_independent_python_restatement(oracle, n, seed, theta):
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-500, 500, (n, 3)).astype(np.float32)
    mass = rng.uniform(1, 5000, n).astype(np.float32)
    if n > 2:
        pos[0] = 0.0                                       # the shipped scene pins body 0 at the origin
    origin = np.array([3.5, -1.25, 0.75], np.float32)      # a previous frame's centre of mass, say
    size = oracle.bounds_f32(pos)
    with np.errstate(over="ignore"):
        want, want_com = _py_create_octree(pos, mass, theta, origin, size)
    got, got_com, _ = oracle.octree_forces_f32(pos, mass, theta, root_origin=origin, root_size=size, pow_mode=0)
    # pow_mode 0 = the C library's pow(double, 3), which is also what numpy's float64 `** 3` calls: bit for bit
    np.testing.assert_array_equal(got_com, want_com)
    np.testing.assert_array_equal(got, want)
    # the float-overload reading of pow (pow_mode 2: d*(d*d) in fp32, a pre-C++11 <cmath>) stays within an ulp or two
    got2, _, _ = oracle.octree_forces_f32(pos, mass, theta, root_origin=origin, root_size=size, pow_mode=2)
    assert np.abs(got2 - want).max() <= np.abs(want).max() * 2.0 ** -22


@pytest.mark.parametrize("theta", [0.5, 1.0])
def test_c_octree_agrees_with_the_python_restatement_on_deep_chains(oracle, theta):
    # the scenes round 4's device fuzz tripped over, small enough for the recursive restatement: a pair of near-twins in a corner
    # of the box (a chain of single-child cells twenty levels deep that only the first body in depth-first order opens) and a
    # clump inside ONE cell of level 21 at the origin (its bodies differ only below the 21st octant digit)
    rng = np.random.default_rng(5)
    n = 48
    pos = rng.uniform(-1000, 1000, (n, 3)).astype(np.float32)
    mass = rng.uniform(1, 5000, n).astype(np.float32)
    pos[0] = (1000.0, -1000.0, 1000.0)                      # Size = 1000
    for k, (sx, sy, sz) in enumerate([(-1, -1, -1), (1, 1, 1), (-1, 1, -1)]):
        corner = np.array([sx, sy, sz], np.float32) * np.float32(999.5)
        pos[1 + 2 * k] = corner
        pos[2 + 2 * k] = corner + np.float32(2.0 ** -11) * np.array([1, -1, 1], np.float32)
    pos[10:22] = rng.uniform(1e-5, 4.7e-4, (12, 3)).astype(np.float32)
    mass[10:22] *= np.float32(1e-12)
    assert len(np.unique(pos, axis=0)) == n
    origin = np.zeros(3, np.float32)
    size = oracle.bounds_f32(pos)
    assert size == 1000.0
    with np.errstate(over="ignore"):
        want, want_com = _py_create_octree(pos, mass, theta, origin, size)
    got, got_com, nodes = oracle.octree_forces_f32(pos, mass, theta, root_origin=origin, root_size=size, pow_mode=0)
    np.testing.assert_array_equal(got_com, want_com)
    np.testing.assert_array_equal(got, want)
    assert nodes > 8 * 25                                   # the chains are there


def _py_depth(cell, d=0):
    if cell.kids is None:
        return d
    return max(_py_depth(k, d + 1) for k in cell.kids)


def test_the_depth_the_oracle_reports_is_the_python_restatements(oracle):
    # oracle.last_max_depth / octree_depth_f32 (how deep Octree::Add, OctreeSearch.h:60-81, goes; root = 0) is what the GPU fuzz
    # tests confirm a refused frame with (the device refuses from depth 43 on: its keys hold 42 octant digits).  Against the
    # recursive Python restatement's deepest cell: random scenes, near-twins at chosen separations — on both sides of the
    # device's limit — and an exact duplicate (the reference recurses without bound; the oracle cuts off at 201).
    rng = np.random.default_rng(42)
    for n in (2, 9, 64):
        pos = rng.uniform(-100, 100, (n, 3)).astype(np.float32)
        root = _PyCell(np.zeros(3, np.float32), oracle.bounds_f32(pos))
        for i in range(n):
            root.add(i, pos)
        assert oracle.octree_depth_f32(pos) == _py_depth(root)
    seen = set()
    for e in (-3, -10, -20, -21, -22, -30, -35, -38, -39, -40, -41, -42, -45):
        x0 = 1000.0 * 2.0 ** e                               # a pair about Size * 2^e apart, next to the origin (where fp32 is that fine)
        pos = np.array([[1000.0, -1000.0, 1000.0], [x0, x0, x0], [2 * x0, x0, x0]], np.float32)
        assert pos[2, 0] != pos[1, 0]
        root = _PyCell(np.zeros(3, np.float32), _f(1000.0))
        for i in range(3):
            root.add(i, pos)
        depth = oracle.octree_depth_f32(pos)
        assert depth == _py_depth(root)
        assert abs(depth - (-e)) <= 2                        # leaves about log2(Size / separation) deep
        seen.add(depth >= 43)
        # and the same number through the whole-frame entry points
        oracle.octree_forces_f32(pos, np.ones(3, np.float32), 1.0)
        assert oracle.last_max_depth() == depth
    assert seen == {False, True}
    pos = np.array([[1.0, 2.0, 3.0], [5.0, 5.0, 5.0], [5.0, 5.0, 5.0]], np.float32)
    assert oracle.octree_depth_f32(pos) == 201
    with pytest.raises(RuntimeError):
        oracle.octree_forces_f32(pos, np.ones(3, np.float32), 1.0)


@pytest.mark.parametrize("n,seed", [(9, 1), (200, 3)])
def test_c_octree_draw_order_and_division_reading(oracle, n, seed):
    # what DrawOctreeBoxes draws (occupied leaves depth first, .cpp:36-45) and the second reading of `/=` in ComputeMass
    # (.h:95: divide instead of reciprocal-multiply), C restatement against the Python one
    rng = np.random.default_rng(seed)
    pos = rng.uniform(-500, 500, (n, 3)).astype(np.float32)
    mass = rng.uniform(1, 5000, n).astype(np.float32)
    origin = np.array([3.5, -1.25, 0.75], np.float32)
    size = oracle.bounds_f32(pos)
    boxes, order = [], []
    with np.errstate(over="ignore"):
        _py_create_octree(pos, mass, 1.0, origin, size, draw=(boxes, order))
    got_boxes, got_order = oracle.octree_leaves_f32(pos, mass, root_origin=origin, root_size=size)
    np.testing.assert_array_equal(got_order, np.array(order, np.int32))
    np.testing.assert_array_equal(got_boxes, np.array(boxes, np.float32))
    assert sorted(order) == list(range(n))                  # every body is drawn exactly once
    _PyCell.divide = True
    try:
        with np.errstate(over="ignore"):
            want, want_com = _py_create_octree(pos, mass, 1.0, origin, size)
    finally:
        _PyCell.divide = False
    got, got_com, _ = oracle.octree_forces_f32(pos, mass, 1.0, root_origin=origin, root_size=size, div_mode=1)
    np.testing.assert_array_equal(got_com, want_com)
    np.testing.assert_array_equal(got, want)
    base, _, _ = oracle.octree_forces_f32(pos, mass, 1.0, root_origin=origin, root_size=size, div_mode=0)
    assert np.abs(base - got).max() <= np.abs(got).max() * 1e-5     # the two readings differ by rounding only


def test_c_tick_agrees_with_the_python_restatement_over_frames(nb, oracle):
    # whole frames (OctreeSearch.cpp:24-32, 47-56, 74-89): Size, tree rooted at the previous frame's centre of mass, walk
    # at the shipped theta, kick-drift — the Python restatement above against oracle_tick_aos_f32, every bit, 3 frames
    rng = np.random.default_rng(11)
    n, dt = 48, _f(0.01)
    pos = rng.uniform(-300, 300, (n, 3)).astype(np.float32)
    vel = rng.normal(0, 20, (n, 3)).astype(np.float32)
    mass = rng.uniform(1, 5000, n).astype(np.float32)
    q = np.zeros(n, nb.PARTICLE_DTYPE)
    q["Mass"], q["Position"], q["Velocity"] = mass, pos, vel
    com_c, size_c = None, 0.0
    com_py = np.zeros(3, np.float32)                       # FVector t = ZeroVector before the first tree, .cpp:77
    for frame in range(3):
        size_py = max(float(np.abs(p).max()) for p in pos)                      # ComputeCubeSize, .cpp:47-56
        with np.errstate(over="ignore"):
            acc, com_py = _py_create_octree(pos, mass, 1.0, com_py, size_py)     # CreateOctree, .cpp:74-89
        for i in range(n):                                                       # .cpp:28-31
            for a in range(3):
                vel[i, a] = _f(vel[i, a] + _f(dt * acc[i, a]))
                pos[i, a] = _f(pos[i, a] + _f(dt * vel[i, a]))
        com_c, size_c = oracle.tick_aos_f32(q, float(dt), theta=1.0, root_com=com_c, size=size_c, pow_mode=0)
        assert size_c == size_py, frame
        np.testing.assert_array_equal(com_c, com_py)
        np.testing.assert_array_equal(q["Acceleration"], acc)
        np.testing.assert_array_equal(q["Velocity"], vel)
        np.testing.assert_array_equal(q["Position"], pos)


def test_simd_cpu_baseline_is_the_same_pair_law(oracle):
    # oracle/cpu_baseline.c (bench.py's cpu_baseline.simd leg): plain fp32, SIMD over j — a throughput
    # baseline, not the oracle; it must still be the pair law of OctreeSearch.h:101-104 to fp32 rounding, self pair and
    # coincident bodies included
    rng = np.random.default_rng(12)
    n = 777
    pos = rng.normal(0, 50, (n, 3)).astype(np.float32)
    pos[5] = pos[9]                                          # two bodies on one point: d == 0, skipped (.h:102)
    mass = rng.uniform(1, 100, n).astype(np.float32)
    a = oracle.forces_simd_f32(pos, mass, nthreads=2)
    ref = oracle.forces_direct_f64(pos.astype(np.float64), mass.astype(np.float64))
    assert np.all(np.isfinite(a))
    assert (np.linalg.norm(a - ref, axis=1) / np.linalg.norm(ref, axis=1)).max() < 2e-5
    soft = oracle.forces_simd_f32(pos, mass, eps=2.0, i0=100, i1=200, nthreads=2)
    ref = oracle.forces_direct_f64(pos.astype(np.float64), mass.astype(np.float64), eps=2.0, i0=100, i1=200)
    assert (np.linalg.norm(soft - ref, axis=1) / np.linalg.norm(ref, axis=1)).max() < 2e-5


def test_child_centre_in_fp32_equals_the_reference_double_expression(tmp_path):
    # host-side proof by sampling of an arithmetic identity the device's path keys rely on (tests/cpp/child_centre_equiv.c)
    import subprocess
    exe = str(tmp_path / "child_centre_equiv")
    src = os.path.join(os.path.dirname(__file__), "cpp", "child_centre_equiv.c")
    subprocess.check_call(["gcc", "-O2", "-msse2", "-mfpmath=sse", "-ffp-contract=off", src, "-o", exe])
    out = subprocess.run([exe, "20000000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "bad 0" in out.stdout, out.stdout

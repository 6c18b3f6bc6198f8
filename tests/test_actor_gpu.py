"""The AOctreeSearch-shaped actor (include/nbody_actor.hpp) driven like the reference's Blueprints drive
the real one, checked against the oracle's Tick."""
import os

import numpy as np
import pytest

from conftest import particles_from, rel_err

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def test_defaults_match_the_reference_constructor(nb):
    a = nb.OctreeSearch()
    # OctreeSearch.cpp:8: Size(0), Initialized(false), ShowOctree(false), PhDeltaTime(0.01)
    assert a.Size == 0 and not a.Initialized and not a.ShowOctree and a.PhDeltaTime == pytest.approx(0.01)
    a.Tick(0.016); a.ComputeCubeSize(); a.CreateOctree(); a.CleanParticles()   # all silent before init
    assert a.Particles.shape == (0,)


def test_hud_begin_play_sequence(nb, oracle):
    # BP_NBodyHUD: spawn, CreateSpacePoints(2000, 1000), SetActorTickEnabled(true).  Nothing else is set: the actor runs at
    # the reference's hard-coded opening angle 1.0 (OctreeSearch.cpp:85), like AOctreeSearch itself.
    a = nb.OctreeSearch()
    a.set_seed(42)
    a.CreateSpacePoints(2000, 1000.0)
    assert a.Initialized and a.Size == 1000.0 and a.LastStatus == 0
    p0 = a.Particles
    assert p0.shape == (2000,) and p0["Mass"][0] == 5000 and np.all(p0["Position"][0] == 0)
    points, flushes = [], []
    a.set_draw_callbacks(on_flush=lambda: flushes.append(1), on_point=lambda pos, sz: points.append((pos, sz)))
    a.Tick(1 / 60)
    assert len(flushes) == 1 and len(points) == 2000 and points[0][1] == 10.0   # DrawDebugPoint(..., 10.0, ...)
    p1 = a.Particles
    assert np.any(p1["Position"] != p0["Position"]) and np.any(p1["Acceleration"] != 0)
    # DrawOctreeBoxes (.cpp:36-45) walks the frame's tree depth first: the points arrive in that order, each body once,
    # at its position AFTER the update (the tree was built before it, around the zero "previous CoM" of a first frame)
    _, order = oracle.octree_leaves_f32(p0["Position"], p0["Mass"])
    assert sorted(order) == list(range(2000))
    np.testing.assert_array_equal(np.array([q[0] for q in points], np.float32), p1["Position"][order])
    assert a.Size == pytest.approx(np.abs(p0["Position"]).max())                 # ComputeCubeSize ran before the step
    # the frame itself is the oracle's Tick at theta = 1.0
    q = p0.copy()
    oracle.tick_aos_f32(q, 0.01, theta=1.0, pow_mode=3)
    assert p1.tobytes() == q.tobytes()


def test_show_octree_draws_box_then_point_per_leaf_in_tree_order(nb, oracle):
    # ShowOctree: DrawDebugBox(Origin, (Size, Size, Size)) then DrawDebugPoint for every occupied leaf, depth first
    # (OctreeSearch.cpp:39-41)
    a = nb.OctreeSearch()
    a.set_seed(7)
    a.CreateSpacePoints(600, 500.0)
    p0 = a.Particles
    calls = []
    a.set_draw_callbacks(on_point=lambda pos, sz: calls.append(("point", pos)))
    a.set_box_callback(lambda o, sz: calls.append(("box", (o[0], o[1], o[2], sz))))
    a.ShowOctree = True
    a.Tick(1 / 60)
    boxes, order = oracle.octree_leaves_f32(p0["Position"], p0["Mass"])
    p1 = a.Particles
    assert [c[0] for c in calls] == ["box", "point"] * 600
    np.testing.assert_array_equal(np.array([c[1] for c in calls[0::2]], np.float32), boxes)
    np.testing.assert_array_equal(np.array([c[1] for c in calls[1::2]], np.float32), p1["Position"][order])


def test_tick_matches_oracle_tick(nb, oracle):
    g = np.load(os.path.join(GOLDEN, "refbox_n2000_seed1.npz"))
    p = particles_from(nb, g["posm"], g["vel"])
    a = nb.OctreeSearch()
    a.set_theta(0.0)                         # the exact all-pairs limit (the reference's own walk at theta -> 0)
    a.SetParticles(p)
    q = p.copy()
    size = 0.0
    a.Tick(0.0)
    _, size = oracle.tick_aos_f32(q, 0.01, theta=-1.0, size=size)
    out = a.Particles
    assert rel_err(out["Acceleration"], q["Acceleration"]).max() < 2e-5
    assert np.abs(out["Position"] - q["Position"]).max() / np.abs(q["Position"]).max() < 1e-6
    assert a.Size == pytest.approx(size)
    # two more frames: fp32 differences of 1e-7 grow through close pairs, so the bound loosens with time
    for _ in range(2):
        a.Tick(0.0)
        _, size = oracle.tick_aos_f32(q, 0.01, theta=-1.0, size=size)
    out = a.Particles
    assert np.median(rel_err(out["Acceleration"], q["Acceleration"])) < 1e-5
    assert np.abs(out["Position"] - q["Position"]).max() / np.abs(q["Position"]).max() < 1e-4
    assert a.Size == pytest.approx(size)


def test_pause_reset_and_theta(nb):
    a = nb.OctreeSearch()
    a.set_seed(1)
    a.set_theta(0.0)
    a.CreateSpacePoints(500, 200.0)
    a.Tick(0.0)
    before = a.Particles
    a.PhDeltaTime = 0.0                      # BP_ScreenUI Pause
    n_drawn = []
    a.set_draw_callbacks(on_point=lambda pos, sz: n_drawn.append(1))
    a.Tick(0.0)
    assert a.Particles.tobytes() == before.tobytes() and len(n_drawn) == 500   # still draws while paused (.cpp:33)
    a.PhDeltaTime = 0.01
    a.set_theta(1.0)                         # the shipped Barnes-Hut opening angle: the reference's own tree walk
    a.Tick(0.0)
    assert a.LastStatus == 0 and a.Particles.tobytes() != before.tobytes()
    a.set_theta(0.0)
    a.CleanParticles()                       # Button_98: CleanParticles -> CreateSpacePoints
    assert not a.Initialized and a.Particles.shape == (0,)
    a.CreateSpacePoints(300, 100.0)
    assert a.Initialized and a.Particles.shape == (300,)
    a.ShowOctree = True
    assert a.ShowOctree

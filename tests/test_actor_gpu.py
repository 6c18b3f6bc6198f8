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


def test_push_particles_keeps_the_history(nb, oracle):
    # the reference's Particles IS the state (OctreeSearch.h:118): an edit between two Ticks changes the simulation and nothing
    # restarts.  Here: the edit reaches the device with PushParticles and the next frames equal the oracle's Ticks of the edited
    # records with the SAME root centre carried over (the previous tree's CoM, OctreeSearch.cpp:77-79) — every byte, theta = 1
    act = nb.OctreeSearch()
    act.set_seed(11)
    act.CreateSpacePoints(2000, 1000.0)
    q = act.Particles.copy()
    com, size = None, 0.0
    for _ in range(3):
        act.Tick(1.0 / 60)
        com, size = oracle.tick_aos_f32(q, 0.01, theta=1.0, root_com=com, size=size, pow_mode=3)
    assert act.Particles.tobytes() == q.tobytes()
    live = act.live_particles()
    for rec in (live, q):
        rec["Velocity"][7, 0] += 1000.0
        rec["Position"][7, 2] -= 25.0
        rec["Mass"][11] = 4321.0
    act.PushParticles()
    assert act.LastStatus == 0
    for _ in range(3):
        act.Tick(1.0 / 60)
        com, size = oracle.tick_aos_f32(q, 0.01, theta=1.0, root_com=com, size=size, pow_mode=3)
        assert act.Size == size and act.Particles.tobytes() == q.tobytes()
    # an edit that is NOT pushed is overwritten by the next frame (the one-way mirror INTEGRATION.md lists)
    live = act.live_particles()
    live["Velocity"][7, 0] += 1000.0
    act.Tick(1.0 / 60)
    com, size = oracle.tick_aos_f32(q, 0.01, theta=1.0, root_com=com, size=size, pow_mode=3)
    assert act.Particles.tobytes() == q.tobytes()
    # records handed over by value work the same
    q["Position"][3, 1] += 5.0
    act.PushParticles(q)
    act.Tick(1.0 / 60)
    com, size = oracle.tick_aos_f32(q, 0.01, theta=1.0, root_com=com, size=size, pow_mode=3)
    assert act.Particles.tobytes() == q.tobytes()


def test_engine_push_particles_at_theta_zero(nb):
    n = 8192
    posm, vel = nb.ic_plummer(n, seed=2)
    from conftest import particles_from
    with nb.NBodyEngine(n) as a, nb.NBodyEngine(n) as b:
        with pytest.raises(nb.NBodyError):
            a.push_particles(particles_from(nb, posm, vel))      # nothing to edit yet
        a.set_state(posm, vel); b.set_state(posm, vel)
        a.step(0.01, 2); b.step(0.01, 2)
        rec = a.particles()
        rec["Velocity"][5] = (1.0, 2.0, 3.0)
        a.push_particles(rec)
        assert a.steps_done() == 2                               # history kept
        b.set_particles(rec)                                     # a new scene with the same records: step count starts again
        assert b.steps_done() == 0
        a.step(0.01, 2); b.step(0.01, 2)
        for x, y in zip(a.state(), b.state()):
            np.testing.assert_array_equal(x, y)

"""Host logic of the multi-GPU path on CPU: world_size 2 over gloo.  The compute engine is replaced by a
stand-in built on the CPU oracle (test-only; the product's default factory is the HIP engine), so this
covers partitioning, the in-place all-gather of positions, gather_state and the energy all-reduce."""
import os
import socket

import numpy as np
import pytest


class OracleEngine:
    """Test stand-in with NBodyEngine's interface; advances its slice with the oracle."""

    def __init__(self, n_total, i_begin, i_count, posm_tensor, device_index, **kw):
        from oracle import oracle as O
        self.O = O
        self.n, self.lo, self.cnt = n_total, i_begin, i_count
        self.posm = posm_tensor.numpy()          # shares memory with the torch tensor that gets gathered into
        self.vel = np.zeros((i_count, 4), np.float32)
        self.acc = np.zeros((i_count, 4), np.float32)

    def set_state(self, posm, vel):
        self.posm[:] = posm
        self.vel[:] = vel[self.lo:self.lo + self.cnt]

    def exchange_ranks(self):
        return 0                                  # one-sided forces: nothing to exchange

    def step_begin(self):
        pos = np.ascontiguousarray(self.posm[:, :3]); m = np.ascontiguousarray(self.posm[:, 3])
        self.acc[:, :3] = self.O.forces_direct_f32(pos, m, i0=self.lo, i1=self.lo + self.cnt)

    def step_end(self, dt):
        if dt > 0:
            pos = np.ascontiguousarray(self.posm[self.lo:self.lo + self.cnt, :3])
            p1, v1 = self.O.kick_drift_f32(pos, self.vel[:, :3], self.acc[:, :3], dt)
            self.vel[:, :3] = v1
            self.posm[self.lo:self.lo + self.cnt, :3] = p1

    def state(self, dtype=np.float32):
        return self.posm[self.lo:self.lo + self.cnt].copy(), self.vel.copy(), self.acc.copy()

    def energy(self):
        # this rank's share: KE of its bodies, half of m_i*phi_i for its bodies
        pos = self.posm[:, :3].astype(np.float64); m = self.posm[:, 3].astype(np.float64)
        ke = 0.5 * (m[self.lo:self.lo + self.cnt] * (self.vel[:, :3].astype(np.float64) ** 2).sum(1)).sum()
        pe = 0.0
        for i in range(self.lo, self.lo + self.cnt):
            d = np.linalg.norm(pos - pos[i], axis=1)
            d[i] = np.inf
            pe += -0.5 * 1e4 * m[i] * (m / d).sum()
        return ke, pe

    def close(self):
        pass


class ExchangingEngine(OracleEngine):
    """Stand-in with the exchange step of the symmetric algorithm: this rank computes, for EVERY body of the system,
    the acceleration due to the sources in its own slice (`send`); after the all-to-all a rank holds one such row per
    source rank for its own bodies (`recv`) and adds them in rank order.  Same transport as the HIP engine's."""

    def __init__(self, n_total, i_begin, i_count, posm_tensor, device_index, **kw):
        super().__init__(n_total, i_begin, i_count, posm_tensor, device_index, **kw)
        self.ranks = n_total // i_count

    def exchange_ranks(self):
        return self.ranks

    def bind_exchange(self, send, recv):
        self.send, self.recv = send.numpy(), recv.numpy()

    def step_begin(self):
        pos = self.posm[:, :3].astype(np.float64); m = self.posm[:, 3].astype(np.float64)
        out = np.zeros((self.n, 3))
        for j in range(self.lo, self.lo + self.cnt):          # sources: my slice; targets: everybody
            d = pos[j] - pos
            r2 = (d * d).sum(1)
            r2[j] = np.inf
            out += (1e4 * m[j] / (r2 * np.sqrt(r2)))[:, None] * d
        self.send[:, :3] = out
        self.send[:, 3] = 0

    def step_end(self, dt):
        a = self.recv.reshape(self.ranks, self.cnt, 4).astype(np.float64).sum(0)[:, :3]
        self.acc[:, :3] = a
        super().step_end(dt)


def _worker_exchange(rank, world, port, n, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import parallelnbody_amd as nb
    posm, vel = nb.ic_plummer(n, seed=9)
    sim = nb.ShardedSimulation(posm, vel, rank=rank, world_size=world, device="cpu", engine_factory=ExchangingEngine)
    assert sim.ex_ranks == world and sim.ex_recv.shape == (world * sim.i_count, 4)
    sim.warm_collectives()
    assert sim.settle(0.0) == 2            # untimed force passes (with their exchange), same count on both ranks, state untouched
    sim.compute_forces()
    _, _, a = sim.engine.state()
    sim.step(0.01, 2)
    p, v = sim.gather_state()
    np.savez(os.path.join(out_dir, f"x{rank}.npz"), p=p, v=v, a=a)
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_with_all_to_all_exchange(nb, oracle, tmp_path):
    import torch.multiprocessing as mp
    n = 128
    mp.spawn(_worker_exchange, args=(2, _free_port(), n, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "x0.npz"); r1 = np.load(tmp_path / "x1.npz")
    np.testing.assert_array_equal(r0["p"], r1["p"])
    posm, vel = nb.ic_plummer(n, seed=9)
    ref = oracle.forces_direct_f64(posm[:, :3].astype(np.float64), posm[:, 3].astype(np.float64))
    a = np.concatenate([r0["a"], r1["a"]])[:, :3]
    assert (np.linalg.norm(a - ref, axis=1) / np.linalg.norm(ref, axis=1)).max() < 1e-5
    pos = posm[:, :3].astype(np.float64); v = vel[:, :3].astype(np.float64)
    for _ in range(2):
        acc = oracle.forces_direct_f64(pos, posm[:, 3].astype(np.float64))
        pos, v = oracle.kick_drift_f64(pos, v, acc, float(np.float32(0.01)))
    assert np.abs(r0["p"][:, :3] - pos).max() / np.abs(pos).max() < 1e-5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, steps, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import parallelnbody_amd as nb
    posm, vel = nb.ic_plummer(n, seed=9)
    sim = nb.ShardedSimulation(posm, vel, rank=rank, world_size=world, device="cpu", engine_factory=OracleEngine)
    assert (sim.i_begin, sim.i_count) == (rank * n // world, n // world)
    sim.step(0.01, steps)
    sim.step(0.0, 3)                       # paused: no exchange, no change
    p, v = sim.gather_state()
    ke, pe = sim.energy()
    # every rank must hold the same replicated positions after the last all-gather
    np.testing.assert_array_equal(sim.posm.numpy(), p)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), p=p, v=v, ke=ke, pe=pe)
    dist.barrier()
    dist.destroy_process_group()


def test_partition_helper(nb):
    assert nb.partition(1 << 20, 8, 0) == (0, 131072) and nb.partition(1 << 20, 8, 7) == (917504, 131072)
    with pytest.raises(ValueError):
        nb.partition(1000, 3, 0)


def test_two_ranks_reproduce_one_rank(nb, oracle, tmp_path):
    import torch.multiprocessing as mp
    n, steps = 256, 3
    mp.spawn(_worker, args=(2, _free_port(), n, steps, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "rank0.npz"); r1 = np.load(tmp_path / "rank1.npz")
    np.testing.assert_array_equal(r0["p"], r1["p"])
    np.testing.assert_array_equal(r0["v"], r1["v"])
    # single-process reference run of the same three ticks
    posm, vel = nb.ic_plummer(n, seed=9)
    pos = posm[:, :3].copy(); v = vel[:, :3].copy()
    for _ in range(steps):
        a = oracle.forces_direct_f32(pos, posm[:, 3])
        pos, v = oracle.kick_drift_f32(pos, v, a, 0.01)
    np.testing.assert_array_equal(r0["p"][:, :3], pos)
    np.testing.assert_array_equal(r0["v"][:, :3], v)
    ke, pe = oracle.energy_f64(pos, v, posm[:, 3])
    assert float(r0["ke"]) == pytest.approx(ke, rel=1e-6) and float(r0["pe"]) == pytest.approx(pe, rel=1e-6)
    assert float(r1["ke"]) == pytest.approx(ke, rel=1e-6)


def _worker_creation_failure(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import parallelnbody_amd as nb
    posm, vel = nb.ic_plummer(128, seed=9)

    def picky(n_total, i_begin, i_count, posm_tensor, device_index, **kw):
        if kw.get("algorithm", 0) == 0 and i_begin > 0:       # only rank 1 cannot build the "auto" engine
            raise MemoryError("no room for the partial sums on this rank")
        return OracleEngine(n_total, i_begin, i_count, posm_tensor, device_index, **kw)

    outcome = "created"
    try:
        nb.ShardedSimulation(posm, vel, rank=rank, world_size=world, device="cpu", engine_factory=picky)
    except nb.EngineCreationFailed as e:
        outcome = f"failed together: {e}"
    # both ranks are at the same point: the rebuild's collectives match
    sim = nb.ShardedSimulation(posm, vel, rank=rank, world_size=world, device="cpu", engine_factory=picky, algorithm=1)
    sim.step(0.01, 1)
    p, _ = sim.gather_state()
    with open(os.path.join(out_dir, f"o{rank}.txt"), "w") as f:
        f.write(outcome + "\n" + repr(float(p[:, :3].sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_engine_creation_failure_on_one_rank_is_seen_by_all(nb, tmp_path):
    # one rank failing to create its engine must not leave the other inside collectives nobody else enters: both raise
    # EngineCreationFailed after the same all-reduce, and a rebuild (bench.py falls back to the one-sided kernel) works
    import torch.multiprocessing as mp
    mp.spawn(_worker_creation_failure, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    o0 = (tmp_path / "o0.txt").read_text().splitlines(); o1 = (tmp_path / "o1.txt").read_text().splitlines()
    assert o0[0].startswith("failed together") and o1[0].startswith("failed together")
    assert "another rank" in o0[0] and "MemoryError" in o1[0]
    assert o0[1] == o1[1]


class TwoGoEngine(OracleEngine):
    """Stand-in with the force pass in two goes (nbody_step_begin_local / _remote): the first go may only read the OWN slice
    of the positions — it poisons everything else it could see while it runs, so a driver that let it depend on the gathered
    positions (or started the second go before the gather had landed) would produce NaNs."""

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.calls = []

    def step_begin_local(self):
        self.calls.append("local")
        own = self.posm[self.lo:self.lo + self.cnt]
        pos = np.ascontiguousarray(own[:, :3]); m = np.ascontiguousarray(own[:, 3])
        self.acc_local = self.O.forces_direct_f32(pos, m)            # own bodies against own bodies

    def step_begin_remote(self):
        assert self.calls[-1] == "local"
        self.calls.append("remote")
        pos = np.ascontiguousarray(self.posm[:, :3]); m = self.posm[:, 3].copy()
        m[self.lo:self.lo + self.cnt] = 0                              # the others against the own bodies
        self.acc[:, :3] = self.acc_local + self.O.forces_direct_f32(pos, m, i0=self.lo, i1=self.lo + self.cnt)

    def step_begin(self):
        self.calls.append("whole")
        super().step_begin()


def _worker_two_goes(rank, world, port, n, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import parallelnbody_amd as nb
    posm, vel = nb.ic_plummer(n, seed=4)
    out = {}
    for overlap in (True, False):
        sim = nb.ShardedSimulation(posm, vel, rank=rank, world_size=world, device="cpu", engine_factory=TwoGoEngine, overlap=overlap)
        sim.step(0.01, 3)
        sim.compute_forces()
        p, v = sim.gather_state()
        assert sim.engine.calls == (["local", "remote"] * 4 if overlap else ["whole"] * 4)
        out["p%d" % overlap], out["v%d" % overlap] = p, v
        sim.close()
    np.savez(os.path.join(out_dir, f"g{rank}.npz"), **out)
    dist.barrier()
    dist.destroy_process_group()


def test_the_driver_runs_the_force_pass_in_two_goes_when_the_engine_has_them(nb, oracle, tmp_path):
    # ShardedSimulation(overlap=True) drives step_begin_local -> (gather of the previous step) -> step_begin_remote; with
    # overlap=False the engine's one-go step_begin.  Same physics either way (the stand-in splits the sum differently, so
    # positions agree to rounding, not in every bit — the HIP engine's two goes are bit-identical: tests/test_rank_geometries_gpu.py)
    import torch.multiprocessing as mp
    n = 96
    mp.spawn(_worker_two_goes, args=(2, _free_port(), n, str(tmp_path)), nprocs=2, join=True)
    r0 = np.load(tmp_path / "g0.npz"); r1 = np.load(tmp_path / "g1.npz")
    for k in ("p1", "v1", "p0", "v0"):
        np.testing.assert_array_equal(r0[k], r1[k])
    assert np.abs(r0["p1"][:, :3] - r0["p0"][:, :3]).max() / np.abs(r0["p0"][:, :3]).max() < 1e-6

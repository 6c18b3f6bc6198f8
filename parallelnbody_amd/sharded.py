"""Range-partitioned multi-GPU stepping: one process per GPU, one all-gather of positions per step.

Body i of an n_total-body system belongs to rank r = i // (n_total / world).  Every rank keeps the full
position+mass array (16 B/body; 16 MiB at N = 2^20) and the velocities/accelerations of its own slice only.
Per step (the reference's Tick body, OctreeSearch.cpp:25-31, per slice):

    step_begin: forces of the owned bodies
    [symmetric algorithm only] all_to_all_single(recv, send)   # 16 B/body: what my pairs add to your bodies
    step_end:   kick-drift(own slice, in place in the full array)
    all_gather_into_tensor(full array, own slice)               # 16 B/body over RCCL/xGMI

The all-gather is issued on a second stream and the next step does not wait for it at once: step_begin_local runs the
strips of the force pass that lie inside the own slice (about 1/world_size of the rank's work — milliseconds at
N = 2^20 — against a gather of ~0.1 ms), and only step_begin_remote is ordered behind the gather's event.
The results are the non-overlapped order's in every bit (`overlap=False` runs that order: same plan, same sums).

No all-reduce anywhere: the force on a body needs every position but no other body's velocity.  With the TILED
algorithm every rank evaluates its bodies against all others (no all-to-all) and the summation order per body
does not depend on the partition, so the trajectory is bit-identical for any world size.  With the SYMMETRIC
algorithm (the fp32 default for large N) each body PAIR is evaluated once in the whole job, so half of every
interaction is computed on another rank and travels through one all-to-all per step; results are
deterministic for a given world size.

torch is plumbing here: device memory, the current stream and torch.distributed.  The compute engine is
injected (`engine_factory`) so that the world_size-2 gloo test can run the host logic on CPU with a stand-in
defined in tests/; the default factory is the HIP engine and nothing else.
"""
import os

import numpy as np

from .engine import REF_DT, NBodyEngine


def partition(n_total, world_size, rank):
    """Contiguous equal slices (what all_gather_into_tensor needs).  Returns (i_begin, i_count)."""
    if n_total % world_size != 0:
        raise ValueError(f"n_total={n_total} must be divisible by world_size={world_size}")
    cnt = n_total // world_size
    return rank * cnt, cnt


def hip_engine_factory(n_total, i_begin, i_count, posm_tensor, device_index, stream=None, **kw):
    eng = NBodyEngine(n_total, i_begin=i_begin, i_count=i_count, device=device_index, **kw)
    if i_count != n_total:
        # the replicated array is a torch tensor so that torch.distributed can gather into it.  A single rank gathers
        # nothing: the engine keeps its own buffer, which also lets it fuse the update with the next pass's preparation
        # (nobody else can move a body then)
        eng.bind_device_state(posm=posm_tensor)
    if stream is not None:
        # Kernels and collectives must be ordered on ONE stream.  torch's default stream has handle 0, which the
        # C-ABI reads as "use the context's own stream", so the simulation owns a real side stream and issues both
        # its kernels (through this handle) and its collectives (under torch.cuda.stream) on it.
        assert stream.cuda_stream != 0
        eng.set_stream(stream.cuda_stream)
    return eng


class EngineCreationFailed(RuntimeError):
    """Some rank could not create its engine.  Raised on EVERY rank at the same point — right after the one collective
    that follows engine creation — so a caller may tear down and rebuild (e.g. with another algorithm) without any rank
    being left inside a collective the others never enter."""


class ShardedSimulation:
    def __init__(self, posm, vel, *, rank=0, world_size=1, device=None, group=None, engine_factory=None, overlap=True,
                 timeline=False, **engine_kw):
        import torch
        self.torch = torch
        self.overlap = overlap and os.environ.get("NBODY_NO_OVERLAP") != "1"    # the environment can put every collective back in one stream order
        self.marks = [] if timeline else None    # timed events around the gather and the force pass's goes (timeline_ms)
        posm = np.ascontiguousarray(posm)
        self.n_total = posm.shape[0]
        self.rank, self.world_size, self.group = rank, world_size, group
        self.i_begin, self.i_count = partition(self.n_total, world_size, rank)
        self.f64 = engine_kw.get("precision") in ("f64", 2)
        tdt = torch.float64 if self.f64 else torch.float32
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        # the replicated position+mass array lives in a torch tensor so that torch.distributed can gather into it
        self.posm = torch.empty((self.n_total, 4), dtype=tdt, device=self.device)
        factory = engine_factory or hip_engine_factory
        dev_index = self.device.index if self.device.type == "cuda" else -1
        self.stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" else None
        # the all-gather's own stream and the event that says "all positions of the last step are in"
        self.gather_stream = torch.cuda.Stream(device=self.device) if self.device.type == "cuda" and world_size > 1 else None
        self.gathered = None
        self.gather_work = None
        if self.stream is not None:          # GPU engines launch on the simulation's stream (hip_engine_factory's `stream`);
            import inspect                   # a caller's own factory gets the keyword only if it takes it
            params = inspect.signature(factory).parameters
            if "stream" in params or any(q.kind is inspect.Parameter.VAR_KEYWORD for q in params.values()):
                engine_kw = dict(engine_kw, stream=self.stream)
        # Engine creation can fail on one rank only (memory, an unsupported geometry).  The ranks agree on the outcome
        # with ONE matched collective, the first this object issues, before anything else can diverge.
        self.engine, failure = None, None
        try:
            self.engine = factory(self.n_total, self.i_begin, self.i_count, self.posm, dev_index, **engine_kw)
        except Exception as e:  # noqa: BLE001
            if world_size == 1:
                raise
            failure = f"{type(e).__name__}: {e}"
        if world_size > 1:
            ok = torch.tensor([0 if failure else 1], dtype=torch.int32, device=self.device)
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=group)
            if int(ok[0]) == 0:
                if self.engine is not None:
                    self.engine.close()
                raise EngineCreationFailed(failure or "engine creation failed on another rank")
        if world_size == 1 and factory is hip_engine_factory:
            self.posm = None                 # a single HIP rank keeps the positions in the engine's own buffer (fused stepping)
        self.engine.set_state(posm.astype(np.float64 if self.f64 else np.float32, copy=False),
                              np.ascontiguousarray(vel).astype(np.float64 if self.f64 else np.float32, copy=False))
        self.steps_done = 0
        # exchange buffers of the symmetric algorithm (none for the tiled one)
        self.ex_ranks = self.engine.exchange_ranks() if hasattr(self.engine, "exchange_ranks") else 0
        if self.ex_ranks:
            assert self.ex_ranks == world_size
            self.ex_send = torch.zeros((self.n_total, 4), dtype=tdt, device=self.device)
            self.ex_recv = torch.zeros((world_size * self.i_count, 4), dtype=tdt, device=self.device)
            self.engine.bind_exchange(self.ex_send, self.ex_recv)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)   # the zero-fills above ran on torch's default stream, the engine will not
        if world_size > 1:
            with self._on_stream():
                self._check_same_geometry()

    def _on_stream(self):
        """Context manager that makes the simulation's stream torch's current stream (no-op on CPU)."""
        import contextlib
        return self.torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def warm_collectives(self):
        """Run each collective of the step once on scratch tensors, so that RCCL's lazy communicator/channel setup
        (seconds, first call only) never lands in a timed region."""
        if self.world_size == 1:
            return
        with self._on_stream():
            self._warm_collectives()

    def _warm_collectives(self):
        torch, dist = self.torch, self.torch.distributed
        scratch = torch.zeros((self.world_size * 8, 4), dtype=self.posm.dtype, device=self.device)
        dist.all_gather_into_tensor(scratch, scratch[self.rank * 8:(self.rank + 1) * 8], group=self.group)
        if self.ex_ranks:
            a = torch.zeros((self.world_size * 8, 4), dtype=self.posm.dtype, device=self.device)
            dist.all_to_all_single(torch.empty_like(a), a, group=self.group)
        t = torch.zeros(2, dtype=torch.float64, device=self.device)
        dist.all_reduce(t, group=self.group)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def _check_same_geometry(self):
        """Every rank must have arrived at the same algorithm and i-set size: which rank evaluates which body pair
        depends on them, and a disagreement would double-count or drop pairs silently."""
        if not hasattr(self.engine, "launch_config"):
            return
        torch, dist = self.torch, self.torch.distributed
        cfg = self.engine.launch_config()
        mine = [1 if cfg.get("algorithm") == "symmetric" else 0, int(cfg.get("super_tile") or 0), int(self.ex_ranks)]
        lo = torch.tensor(mine, dtype=torch.int64, device=self.device)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
        if lo.tolist() != hi.tolist():
            raise RuntimeError(f"ranks disagree on the force-pass geometry (algorithm, bodies per i-set, exchange ranks): "
                               f"this rank {mine}, minimum {lo.tolist()}, maximum {hi.tolist()}")

    def _wait_gather(self):
        """Order the simulation's stream behind the last all-gather.  RCCL: an event wait, the host goes on.  Backends
        whose collectives complete on the host (gloo, the one-GPU rehearsal) are waited for here, as late as possible."""
        if self.gather_work is not None:
            self.gather_work.wait()
            self.gather_work = None
            self._mark("all-gather end (seen by the host)", self.gather_stream, step=self.steps_done - 1)
        if self.gathered is not None:
            self.stream.wait_event(self.gathered)
            self.gathered = None

    def _mark(self, name, stream=None, step=None):
        """timeline=True: a timed event on `stream` (default: the simulation's), read back by `timeline_ms()`."""
        if self.marks is not None:
            ev = self.torch.cuda.Event(enable_timing=True)
            ev.record(stream if stream is not None else self.stream)
            self.marks.append((self.steps_done if step is None else step, name, ev))

    def timeline_ms(self):
        """[(step, name, milliseconds since the first mark)] — where the all-gather ran relative to the force pass's two goes
        (tools/overlap_timeline.py).  Synchronises."""
        self.torch.cuda.synchronize(self.device)
        t0 = self.marks[0][2]
        return [(st, name, t0.elapsed_time(ev)) for st, name, ev in self.marks]

    def _forces(self):
        two_goes = self.overlap and hasattr(self.engine, "step_begin_local")
        if two_goes:
            self._mark("local strips begin")
            self.engine.step_begin_local()        # strips inside the own slice: the gather may still be in flight
            self._mark("local strips end")
            self._wait_gather()
            self.engine.step_begin_remote()
            self._mark("remote strips end")
        else:
            self._wait_gather()
            self.engine.step_begin()
        if self.ex_ranks:
            self.torch.distributed.all_to_all_single(self.ex_recv, self.ex_send, group=self.group)

    def _gather_positions(self):
        """In-place all-gather of the owned slices.  GPU: on the gather stream, ordered behind the update by an event; the
        simulation's stream only waits for it where the next force pass needs the other ranks' positions."""
        dist = self.torch.distributed
        own = self.posm[self.i_begin:self.i_begin + self.i_count]
        if self.gather_stream is None or not self.overlap:
            dist.all_gather_into_tensor(self.posm, own, group=self.group)
            return
        updated = self.torch.cuda.Event()
        updated.record(self.stream)
        with self.torch.cuda.stream(self.gather_stream):
            self.gather_stream.wait_event(updated)
            self._mark("all-gather begin", self.gather_stream)
            if dist.get_backend(self.group) == "nccl":      # RCCL: queued; the gather stream waits for it, the host does not
                dist.all_gather_into_tensor(self.posm, own, group=self.group)
                self._mark("all-gather end", self.gather_stream)
                self.gathered = self.torch.cuda.Event()
                self.gathered.record(self.gather_stream)
            else:                                           # completes on the host: started now, waited for in _wait_gather
                self.gather_work = dist.all_gather_into_tensor(self.posm, own, group=self.group, async_op=True)

    def compute_forces(self):
        """Accelerations of the current positions (the reference's CreateOctree force loop), no update."""
        with self._on_stream():
            self._forces()
            self.engine.step_end(0.0)

    def step(self, dt=REF_DT, nsteps=1):
        with self._on_stream():            # kernels and collectives ordered by streams and events, never by the host
            for _ in range(nsteps):
                if dt > 0:                 # OctreeSearch.cpp:25: PhDeltaTime <= 0 freezes the physics
                    self._forces()
                    self.engine.step_end(dt)
                    if self.world_size > 1:
                        self._gather_positions()
                self.steps_done += 1

    def wait_for_positions(self):
        """Everything that reads the replicated positions outside the stepping loop comes through here first."""
        if self.stream is not None:
            with self._on_stream():
                self._wait_gather()

    def settle(self, seconds=0.3):
        """Untimed force passes (state unchanged: nothing is integrated) for about `seconds`, the same number on every
        rank: the GPU clock needs sustained load to settle (a run of a few ms-long passes measures the ramp, not the
        kernel).  Returns the number of passes."""
        import time
        torch, dist = self.torch, self.torch.distributed
        with self._on_stream():
            sync = self.stream.synchronize if self.stream is not None else (lambda: None)
            t0 = time.perf_counter()
            self._forces(); self.engine.step_end(0.0)      # dt = 0: accelerations only, nothing is integrated
            sync()
            t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=self.device)
            if self.world_size > 1:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            reps = int(min(2000, max(1, seconds / max(float(t[0]), 1e-5))))
            for _ in range(reps):
                self._forces(); self.engine.step_end(0.0)
            sync()
        return reps + 1

    def gather_state(self):
        """(posm[n_total,4], vel[n_total,4]) on every rank, as numpy (for tests and checkpoints)."""
        torch, dist = self.torch, self.torch.distributed
        self.wait_for_positions()
        p, v, _ = self.engine.state(np.float64 if self.f64 else np.float32)
        if self.world_size == 1:
            return p, v
        out = []
        with self._on_stream():
            for a in (p, v):
                t = torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
                full = torch.empty((self.n_total, 4), dtype=t.dtype, device=self.device)
                dist.all_gather_into_tensor(full, t, group=self.group)
                out.append(full.cpu().numpy())
        return out[0], out[1]

    def energy(self):
        """System kinetic and potential energy (sum of the ranks' shares)."""
        self.wait_for_positions()
        ke, pe = self.engine.energy()
        if self.world_size > 1:
            with self._on_stream():
                t = self.torch.tensor([ke, pe], dtype=self.torch.float64, device=self.device)
                self.torch.distributed.all_reduce(t, group=self.group)
                ke, pe = float(t[0]), float(t[1])
        return ke, pe

    def close(self):
        if self.gather_stream is not None:
            if self.gather_work is not None:
                self.gather_work.wait()
            self.gather_stream.synchronize()      # no collective may still be writing the tensor the engine is bound to
        self.engine.close()

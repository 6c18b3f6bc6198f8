"""NBodyEngine — Python handle on one nbody_ctx (include/nbody.h).  All arithmetic happens in the HIP
kernels of libnbody_amd.so; this file only moves numpy buffers across the C-ABI."""
import ctypes
import os

import numpy as np

from . import _lib
from ._lib import NBodyError, Params

# FParticle, /root/reference/Source/NBody/OctreeSearch.h:8-18 (40 bytes)
PARTICLE_DTYPE = np.dtype(
    [("Mass", "<f4"), ("Position", "<f4", (3,)), ("Velocity", "<f4", (3,)), ("Acceleration", "<f4", (3,))])

REF_G = 1.0e4       # OctreeSearch.h:104
REF_DT = 0.01       # OctreeSearch.cpp:8

_PREC = {"f32": _lib.PREC_F32, "f32_kahan": _lib.PREC_F32_KAHAN, "f64": _lib.PREC_F64}


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def ic_reference_box(n, size=200.0, center=(0.0, 0.0, 0.0), seed=1):
    """Seeded CreateSpacePoints distribution (OctreeSearch.cpp:58-72).  Returns (posm[n,4], vel[n,4]) fp32."""
    posm = np.empty((n, 4), np.float32)
    vel = np.empty((n, 4), np.float32)
    c = np.asarray(center, np.float32)
    rc = _lib.lib().nbody_ic_reference_box(n, size, _fp(c), seed, _fp(posm), _fp(vel))
    if rc:
        raise NBodyError(rc, "nbody_ic_reference_box: invalid argument")
    return posm, vel


def ic_plummer(n, total_mass=1000.0, scale_radius=100.0, G=REF_G, seed=1):
    """Seeded equal-mass Plummer sphere in virial equilibrium.  Returns (posm[n,4], vel[n,4]) fp32."""
    posm = np.empty((n, 4), np.float32)
    vel = np.empty((n, 4), np.float32)
    rc = _lib.lib().nbody_ic_plummer(n, total_mass, scale_radius, G, seed, _fp(posm), _fp(vel))
    if rc:
        raise NBodyError(rc, "nbody_ic_plummer: invalid argument")
    return posm, vel


def sym_plan(n_total, i_begin=0, i_count=0, bodies_per_iset=4096, slots=512, k_guided=3, min_sub=4, own_mode=1):
    """The work plan of the symmetric force pass (csrc/sym_plan.h), host only.  Returns (items[n,8] int32 —
    i0, j0, n_sub, flags (1 = own-block strip, 2 = no j-side sums), slot_i, slot_j, 0, 0 —, pool_elems).
    k_guided may be fractional in tenths (the library itself uses 1, 1.5, 3 and 6)."""
    L = _lib.lib()
    n, pe = ctypes.c_int32(), ctypes.c_uint64()
    k10 = int(round(float(k_guided) * 10))
    if abs(k10 - float(k_guided) * 10) > 1e-9:
        raise ValueError("k_guided must be a multiple of 0.1")
    rc = L.nbody_sym_plan_describe_tenths(n_total, i_begin, i_count, bodies_per_iset, slots, k10, min_sub, own_mode, ctypes.byref(n),
                                          ctypes.byref(pe), None, 0)
    if rc:
        raise NBodyError(rc, "nbody_sym_plan_describe: this range cannot be planned")
    items = np.zeros((n.value, 8), np.int32)
    rc = L.nbody_sym_plan_describe_tenths(n_total, i_begin, i_count, bodies_per_iset, slots, k10, min_sub, own_mode, ctypes.byref(n),
                                          ctypes.byref(pe), items.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n.value)
    if rc:
        raise NBodyError(rc, "nbody_sym_plan_describe: this range cannot be planned")
    return items, pe.value


def sym_plan_phased(n_total, j_budget_elems, i_begin=0, i_count=0, bodies_per_iset=4096, slots=512, k_guided=3, min_sub=4):
    """A plan whose j-side segments share a pool area of at most j_budget_elems elements (csrc/sym_plan.h, pool phases).
    Returns (items[n,8], pool_elems, phase_item0[n_phases + 1])."""
    L = _lib.lib()
    n, pe, nph = ctypes.c_int32(), ctypes.c_uint64(), ctypes.c_int32()
    k10 = int(round(float(k_guided) * 10))
    args = (n_total, i_begin, i_count, bodies_per_iset, slots, k10, min_sub, int(j_budget_elems))
    rc = L.nbody_sym_plan_describe_phased(*args, ctypes.byref(n), ctypes.byref(pe), None, 0, ctypes.byref(nph), None, 0)
    if rc:
        raise NBodyError(rc, "nbody_sym_plan_describe_phased: this range cannot be planned")
    items = np.zeros((n.value, 8), np.int32)
    ph = np.zeros(nph.value + 1, np.int32)
    rc = L.nbody_sym_plan_describe_phased(*args, ctypes.byref(n), ctypes.byref(pe), items.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                                          n.value, ctypes.byref(nph), ph.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), nph.value + 1)
    if rc:
        raise NBodyError(rc, "nbody_sym_plan_describe_phased: this range cannot be planned")
    return items, pe.value, ph


def sym_plan_even(n_total, bodies_per_iset=2048, n_items=768):
    """The even-share plan of the symmetric pass (csrc/sym_plan.h), host only.  Returns (items[n,8] int32 — i0, j0, n_sub,
    flags, slot_i, slot_j, k0, k_skip —, pool_elems)."""
    L = _lib.lib()
    n, pe = ctypes.c_int32(), ctypes.c_uint64()
    rc = L.nbody_sym_plan_describe_even(n_total, bodies_per_iset, n_items, ctypes.byref(n), ctypes.byref(pe), None, 0)
    if rc:
        raise NBodyError(rc, "nbody_sym_plan_describe_even: this system cannot be planned")
    items = np.zeros((n.value, 8), np.int32)
    rc = L.nbody_sym_plan_describe_even(n_total, bodies_per_iset, n_items, ctypes.byref(n), ctypes.byref(pe),
                                        items.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)), n.value)
    if rc:
        raise NBodyError(rc, "nbody_sym_plan_describe_even: this system cannot be planned")
    return items, pe.value


def device_count():
    return int(_lib.lib().nbody_device_count())


class NBodyEngine:
    """One context = one GPU's share [i_begin, i_begin+i_count) of an n_total-body system."""

    def __init__(self, n_total, *, i_begin=0, i_count=0, device=0, precision="f32", G=REF_G, eps=0.0, tile=0,
                 i_per_thread=0, j_split=0, time_kernels=False, zero_mode=0, algorithm=0, theta=0.0, devices=None, bh_div_mode=0):
        L = _lib.lib()
        p = Params()
        L.nbody_default_params(ctypes.byref(p))
        p.n_total, p.i_begin, p.i_count, p.device = n_total, i_begin, i_count, device
        p.precision = _PREC[precision] if isinstance(precision, str) else int(precision)
        p.G, p.eps = G, eps
        p.tile, p.i_per_thread, p.j_split = tile, i_per_thread, j_split
        p.time_kernels = 1 if time_kernels else 0
        p.zero_mode = zero_mode
        p.algorithm = algorithm
        p.theta = theta
        p.bh_div_mode = bh_div_mode
        h = ctypes.c_void_p()
        if devices is not None:
            # one context over several GPUs, driven from this thread (nbody_create_multi: RCCL between the devices)
            devs = (ctypes.c_int32 * len(devices))(*devices)
            rc = L.nbody_create_multi(ctypes.byref(p), devs, len(devices), ctypes.byref(h))
        else:
            rc = L.nbody_create(ctypes.byref(p), ctypes.byref(h))
        if rc:
            raise NBodyError(rc, L.nbody_last_error(None).decode())
        self._L, self._h = L, h
        self.n_total = n_total
        self.i_begin = i_begin
        self.i_count = i_count if i_count else n_total - i_begin
        self.f64 = p.precision == _lib.PREC_F64
        self._keep = []   # externally bound tensors kept alive

    # -- plumbing --
    def _check(self, rc):
        if rc:
            raise NBodyError(rc, self._L.nbody_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._L.nbody_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- state in --
    def set_particles(self, particles):
        a = np.ascontiguousarray(particles)
        assert a.dtype.itemsize >= 40
        self._check(self._L.nbody_set_particles(self._h, a.ctypes.data, a.dtype.itemsize, a.shape[0]))

    def set_state(self, posm, vel):
        if np.asarray(posm).dtype == np.float64:
            p = np.ascontiguousarray(posm, np.float64); v = np.ascontiguousarray(vel, np.float64)
            if p.ndim != 2 or p.shape[1] != 4 or v.shape != p.shape:
                raise ValueError("posm and vel must both be [n, 4]")
            self._check(self._L.nbody_set_state_soa_f64(self._h, _dp(p), _dp(v), p.shape[0]))
        else:
            p = np.ascontiguousarray(posm, np.float32); v = np.ascontiguousarray(vel, np.float32)
            if p.ndim != 2 or p.shape[1] != 4 or v.shape != p.shape:
                raise ValueError("posm and vel must both be [n, 4]")
            self._check(self._L.nbody_set_state_soa(self._h, _fp(p), _fp(v), p.shape[0]))

    # -- hot path --
    def compute_forces(self):
        self._check(self._L.nbody_compute_forces(self._h))

    def step(self, dt=REF_DT, nsteps=1):
        self._check(self._L.nbody_step(self._h, dt, nsteps))

    def step_begin(self):
        """Force pass of the owned bodies (first phase of a step driven by a multi-GPU host)."""
        self._check(self._L.nbody_step_begin(self._h))

    def step_begin_local(self):
        """First go of step_begin: what needs the OWNED slice of the positions only (nbody_step_begin_local)."""
        self._check(self._L.nbody_step_begin_local(self._h))

    def step_begin_remote(self):
        """Second go: the rest of the force pass, once all positions are in (nbody_step_begin_remote)."""
        self._check(self._L.nbody_step_begin_remote(self._h))

    def step_end(self, dt=REF_DT):
        """Kick-drift of the owned bodies (dt <= 0: only store the accelerations)."""
        self._check(self._L.nbody_step_end(self._h, dt))

    def exchange_ranks(self):
        """Number of ranks in the all-to-all between step_begin and step_end (0: no exchange)."""
        n = ctypes.c_int32()
        self._check(self._L.nbody_exchange_info(self._h, None, None, None, ctypes.byref(n)))
        return n.value

    def bind_exchange(self, send, recv):
        """Caller-owned device tensors for the exchange: send [n_total,4], recv [ranks*i_count,4] fp32."""
        self._keep += [send, recv]
        self._check(self._L.nbody_bind_exchange(self._h, ctypes.c_void_p(send.data_ptr()), ctypes.c_void_p(recv.data_ptr())))

    def exchange_read_send(self):
        out = np.empty((self.n_total, 4), np.float64 if self.f64 else np.float32)
        self._check(self._L.nbody_exchange_read_send(self._h, out.ctypes.data))
        return out

    def exchange_write_recv(self, recv):
        r = np.ascontiguousarray(recv, np.float64 if self.f64 else np.float32)
        assert r.shape == (self.exchange_ranks() * self.i_count, 4)
        self._check(self._L.nbody_exchange_write_recv(self._h, r.ctypes.data))

    def set_theta(self, theta):
        """Barnes-Hut opening angle (0 = exact all-pairs; the reference ships 1.0, OctreeSearch.cpp:85)."""
        self._check(self._L.nbody_set_theta(self._h, theta))

    def theta(self):
        """The opening angle in force (a checkpoint brings its own: nbody_load_checkpoint)."""
        v = ctypes.c_float()
        self._check(self._L.nbody_get_theta(self._h, ctypes.byref(v)))
        return v.value

    def bh_stats(self):
        n, l = ctypes.c_int32(), ctypes.c_int32()
        com = np.zeros(3, np.float32)
        self._check(self._L.nbody_bh_stats(self._h, ctypes.byref(n), ctypes.byref(l), _fp(com)))
        return {"nodes": n.value, "levels": l.value, "root_com": com}

    def bh_leaf_boxes(self):
        """[n,4]: (Origin, Size) of the leaf holding each body in the last Barnes-Hut tree."""
        out = np.empty((self.n_total, 4), np.float32)
        self._check(self._L.nbody_bh_leaf_boxes(self._h, _fp(out), 16))
        return out

    def bh_leaf_order(self):
        """order[k] = the body in the k-th occupied leaf of a depth-first walk (children 0..7) of the last tree: the order in
        which the reference's DrawOctreeBoxes draws (OctreeSearch.cpp:36-45)."""
        out = np.empty(self.n_total, np.int32)
        self._check(self._L.nbody_bh_leaf_order(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
        return out

    def synchronize(self):
        self._check(self._L.nbody_synchronize(self._h))

    def bounds(self):
        s = ctypes.c_float()
        self._check(self._L.nbody_get_bounds(self._h, ctypes.byref(s)))
        return s.value

    def energy(self):
        ke, pe = ctypes.c_double(), ctypes.c_double()
        self._check(self._L.nbody_energy(self._h, ctypes.byref(ke), ctypes.byref(pe)))
        return ke.value, pe.value

    # -- state out --
    def positions(self, first=0, count=None, out=None):
        """Positions [count,3] (what the renderer reads each frame).  `out`: a C-contiguous float32 array to fill — if it
        was handed to `pin()` the copy is one DMA from the device into it."""
        count = self.n_total - first if count is None else count
        if out is None:
            out = np.empty((count, 3), np.float32)
        elif out.dtype != np.float32 or out.shape != (count, 3) or not out.flags.c_contiguous:
            raise ValueError("positions: out must be a C-contiguous float32 [count,3] array")
        self._check(self._L.nbody_get_positions(self._h, _fp(out), 12, first, count))
        return out

    def particles(self, out=None):
        if out is None:
            out = np.zeros(self.i_count, PARTICLE_DTYPE)
        elif out.dtype != PARTICLE_DTYPE or out.shape != (self.i_count,) or not out.flags.c_contiguous:
            raise ValueError("particles: out must be a C-contiguous PARTICLE_DTYPE [i_count] array")
        self._check(self._L.nbody_get_particles(self._h, out.ctypes.data, PARTICLE_DTYPE.itemsize))
        return out

    def tick(self, dt=REF_DT, out=None):
        """One frame of AOctreeSearch::Tick with a single host synchronisation (nbody_tick): returns (Size of the
        positions before the step — None when dt <= 0 —, the owned FParticle records after it)."""
        if out is None:
            out = np.zeros(self.i_count, PARTICLE_DTYPE)
        elif out.dtype != PARTICLE_DTYPE or out.shape != (self.i_count,) or not out.flags.c_contiguous:
            raise ValueError("tick: out must be a C-contiguous PARTICLE_DTYPE [i_count] array")
        size = ctypes.c_float(0.0)
        self._check(self._L.nbody_tick(self._h, dt, ctypes.byref(size), out.ctypes.data, PARTICLE_DTYPE.itemsize))
        return (size.value if dt > 0 else None), out

    def pin(self, array):
        """Page-lock a caller-owned numpy array for this context (nbody_pin_host_buffer): `positions(out=array)` /
        `particles(out=array)` then land in it with a single device-to-destination copy.  Keep the array alive until
        `unpin(array)` or `close()`."""
        self._check(self._L.nbody_pin_host_buffer(self._h, array.ctypes.data, array.nbytes))
        self._pinned = getattr(self, "_pinned", [])
        self._pinned.append(array)

    def unpin(self, array):
        self._check(self._L.nbody_unpin_host_buffer(self._h, array.ctypes.data))
        self._pinned = [a for a in getattr(self, "_pinned", []) if a is not array]

    def state(self, dtype=np.float32):
        """(posm, vel, acc) of the owned bodies, [i_count,4] each."""
        n = self.i_count
        if dtype == np.float64:
            p, v, a = (np.empty((n, 4), np.float64) for _ in range(3))
            self._check(self._L.nbody_get_state_soa_f64(self._h, _dp(p), _dp(v), _dp(a)))
        else:
            p, v, a = (np.empty((n, 4), np.float32) for _ in range(3))
            self._check(self._L.nbody_get_state_soa(self._h, _fp(p), _fp(v), _fp(a)))
        return p, v, a

    def accelerations(self, dtype=np.float32):
        return self.state(dtype)[2][:, :3]

    # -- checkpoint / resume --
    def save_checkpoint(self, path):
        self._check(self._L.nbody_save_checkpoint(self._h, os.fsencode(path)))

    def load_checkpoint(self, path):
        n = ctypes.c_int64()
        self._check(self._L.nbody_load_checkpoint(self._h, os.fsencode(path), ctypes.byref(n)))
        return n.value

    def steps_done(self):
        n = ctypes.c_int64()
        self._check(self._L.nbody_steps_done(self._h, ctypes.byref(n)))
        return n.value

    # -- device plumbing --
    def set_stream(self, hip_stream_handle):
        self._check(self._L.nbody_set_stream(self._h, ctypes.c_void_p(hip_stream_handle)))

    def device_ptr(self, which):
        ptr, nbytes = ctypes.c_void_p(), ctypes.c_size_t()
        self._check(self._L.nbody_device_ptr(self._h, which, ctypes.byref(ptr), ctypes.byref(nbytes)))
        return ptr.value, nbytes.value

    def bind_device_state(self, posm=None, vel=None, acc=None):
        """Use caller-owned device buffers (objects with .data_ptr(), e.g. torch tensors)."""
        ptrs = []
        for t in (posm, vel, acc):
            ptrs.append(ctypes.c_void_p(t.data_ptr()) if t is not None else None)
            if t is not None:
                self._keep.append(t)
        self._check(self._L.nbody_bind_device_state(self._h, *ptrs))

    def kernel_time(self, which=_lib.KERNEL_FORCES):
        ms, n = ctypes.c_double(), ctypes.c_int64()
        self._check(self._L.nbody_kernel_time(self._h, which, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def kernel_time_reset(self):
        self._check(self._L.nbody_kernel_time_reset(self._h))

    def kernel_clock(self):
        """(shader clock in MHz the timed force kernels ran at since the last reset — 0.0 if none of them is instrumented —,
        compute units of the device): nbody_kernel_clock.  Needs time_kernels."""
        mhz, cus = ctypes.c_double(), ctypes.c_int32()
        self._check(self._L.nbody_kernel_clock(self._h, ctypes.byref(mhz), ctypes.byref(cus)))
        return mhz.value, cus.value

    def push_particles(self, particles):
        """Records of the RUNNING simulation edited by the host (nbody_push_particles): like set_particles, but the step count
        and the Barnes-Hut root centre stay."""
        a = np.ascontiguousarray(particles)
        assert a.dtype.itemsize >= 40
        self._check(self._L.nbody_push_particles(self._h, a.ctypes.data, a.dtype.itemsize, a.shape[0]))

    def equal_mass_form(self):
        """Did the last force pass run the equal-mass form of the fp32 symmetric kernel (nbody.h)?"""
        v = ctypes.c_int32()
        self._check(self._L.nbody_equal_mass_form(self._h, ctypes.byref(v)))
        return bool(v.value)

    def sym_pool(self):
        """(bytes of the symmetric pass's partial-sum pool, phases sharing its j-side area) — (0, 0) on the one-sided kernels."""
        b, ph = ctypes.c_uint64(), ctypes.c_int32()
        self._check(self._L.nbody_sym_pool_info(self._h, ctypes.byref(b), ctypes.byref(ph)))
        return b.value, ph.value

    def launch_config(self):
        v = [ctypes.c_int32() for _ in range(5)]
        self._check(self._L.nbody_get_launch_config(self._h, *[ctypes.byref(x) for x in v]))
        cfg = dict(zip(("tile", "i_per_thread", "j_split", "blocks", "threads"), (x.value for x in v)))
        algo, st = ctypes.c_int32(), ctypes.c_int32()
        self._check(self._L.nbody_get_algorithm(self._h, ctypes.byref(algo), ctypes.byref(st)))
        cfg["algorithm"] = {_lib.ALGO_TILED: "tiled", _lib.ALGO_SYMMETRIC: "symmetric"}[algo.value]
        cfg["super_tile"] = st.value
        cfg["kernel"] = self._L.nbody_force_kernel_name(self._h).decode()
        cfg["plan"] = ("even" if self._L.nbody_sym_plan_is_even(self._h) else "guided") if cfg["algorithm"] == "symmetric" else None
        return cfg

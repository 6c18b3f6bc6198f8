"""ctypes/numpy front end of the CPU oracle (oracle/nbody_oracle.c).  TEST INFRASTRUCTURE ONLY.

Importers allowed: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg.  The product
package (parallelnbody_amd/) never imports this module.  PARITY UNPINNED — see nbody_oracle.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libnbody_oracle.so")
_lib = None

REF_G = 1.0e4          # OctreeSearch.h:104
REF_THETA = 1.0        # OctreeSearch.cpp:85
REF_DT = 0.01          # OctreeSearch.cpp:8

PARTICLE_DTYPE = np.dtype(
    [("Mass", "<f4"), ("Position", "<f4", (3,)), ("Velocity", "<f4", (3,)), ("Acceleration", "<f4", (3,))]
)  # OctreeSearch.h:8-18, 40 bytes


def build(force=False):
    """Compile the oracle with gcc (see oracle/Makefile)."""
    src = os.path.join(_HERE, "nbody_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnbody_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_BASE_PATH = os.path.join(_HERE, "libnbody_cpubase.so")
_base = None


def cpubase():
    """The SIMD fp32 direct sum of oracle/cpu_baseline.c (a throughput baseline, not the oracle).  Built here with
    -march=native, so always for the host that runs it."""
    global _base
    if _base is None:
        src = os.path.join(_HERE, "cpu_baseline.c")
        subprocess.check_call(["make", "-C", _HERE, "-B", "libnbody_cpubase.so"], stdout=subprocess.DEVNULL)
        assert os.path.getmtime(_BASE_PATH) >= os.path.getmtime(src)
        B = ctypes.CDLL(_BASE_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        B.cpubase_max_threads.restype = ctypes.c_int
        B.cpubase_forces_f32.argtypes = [ctypes.c_int, fp, fp, fp, fp, ctypes.c_float, ctypes.c_float, ctypes.c_int,
                                         ctypes.c_int, fp, ctypes.c_int]
        B.cpubase_forces_f32.restype = ctypes.c_int
        _base = B
    return _base


def forces_simd_f32(pos, mass, g=REF_G, eps=0.0, i0=0, i1=None, nthreads=1):
    """The build's own fp32 direct sum (OpenMP over i, SIMD over j).  Returns acc[i0:i1]."""
    pos = _f32(pos); mass = _f32(mass)
    n = pos.shape[0]
    i1 = n if i1 is None else i1
    x, y, z = (np.ascontiguousarray(pos[:, k]) for k in range(3))
    acc = np.zeros((n, 3), np.float32)
    rc = cpubase().cpubase_forces_f32(n, _fp(x), _fp(y), _fp(z), _fp(mass), np.float32(g), np.float32(eps * eps), i0, i1,
                                      _fp(acc), nthreads)
    if rc:
        raise RuntimeError(f"cpubase_forces_f32 rc={rc}")
    return acc[i0:i1]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = ctypes.CDLL(_LIB_PATH)
        fp = ctypes.POINTER(ctypes.c_float)
        dp = ctypes.POINTER(ctypes.c_double)
        ip = ctypes.POINTER(ctypes.c_int)
        L.oracle_sizeof_particle.restype = ctypes.c_int
        L.oracle_forces_direct_f32.argtypes = [ctypes.c_int, fp, fp, ctypes.c_double, ctypes.c_float, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, fp, ctypes.c_int]
        L.oracle_forces_direct_f32.restype = ctypes.c_int
        L.oracle_octree_forces_f32.argtypes = [ctypes.c_int, fp, fp, fp, ctypes.c_float, ctypes.c_float,
                                               ctypes.c_double, ctypes.c_int, fp, fp, ip]
        L.oracle_octree_forces_f32.restype = ctypes.c_int
        L.oracle_octree_f32.argtypes = [ctypes.c_int, fp, fp, fp, ctypes.c_float, ctypes.c_float, ctypes.c_double, ctypes.c_int,
                                        ctypes.c_int, fp, fp, ip, fp, ip]
        L.oracle_octree_f32.restype = ctypes.c_int
        L.oracle_tick_aos2_f32.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_double,
                                           ctypes.c_int, ctypes.c_int, fp, fp]
        L.oracle_tick_aos2_f32.restype = ctypes.c_int
        L.oracle_kick_drift_f32.argtypes = [ctypes.c_int, fp, fp, fp, ctypes.c_float]
        L.oracle_kick_drift_f32.restype = None
        L.oracle_bounds_f32.argtypes = [ctypes.c_int, fp]
        L.oracle_bounds_f32.restype = ctypes.c_float
        L.oracle_tick_aos_f32.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_float, ctypes.c_float,
                                          ctypes.c_double, ctypes.c_int, fp, fp]
        L.oracle_tick_aos_f32.restype = ctypes.c_int
        L.oracle_forces_direct_f64.argtypes = [ctypes.c_int, dp, dp, ctypes.c_double, ctypes.c_double, ctypes.c_int,
                                               ctypes.c_int, dp, ctypes.c_int]
        L.oracle_forces_direct_f64.restype = ctypes.c_int
        L.oracle_kick_drift_f64.argtypes = [ctypes.c_int, dp, dp, dp, ctypes.c_double]
        L.oracle_kick_drift_f64.restype = None
        L.oracle_energy_f64.argtypes = [ctypes.c_int, dp, dp, dp, ctypes.c_double, ctypes.c_double, dp, dp, ctypes.c_int]
        L.oracle_energy_f64.restype = None
        L.oracle_max_threads.restype = ctypes.c_int
        L.oracle_last_max_depth.restype = ctypes.c_int
        _lib = L
    return _lib


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        assert a.shape == shape, (a.shape, shape)
    return a


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _dp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def max_threads():
    return int(lib().oracle_max_threads())


def forces_direct_f32(pos, mass, g=REF_G, eps=0.0, pow_mode=0, i0=0, i1=None, nthreads=1):
    """All-pairs pass in index order (OctreeSearch.h:101-104 summed over j; .cpp:83-86)."""
    pos = _f32(pos); mass = _f32(mass)
    n = pos.shape[0]
    i1 = n if i1 is None else i1
    acc = np.zeros((n, 3), np.float32)
    rc = lib().oracle_forces_direct_f32(n, _fp(pos), _fp(mass), float(g), np.float32(eps * eps), pow_mode, i0, i1,
                                        _fp(acc), nthreads)
    if rc:
        raise RuntimeError(f"oracle_forces_direct_f32 rc={rc}")
    return acc[i0:i1]


def octree_forces_f32(pos, mass, theta, root_origin=(0.0, 0.0, 0.0), root_size=None, g=REF_G, pow_mode=0, div_mode=0):
    """The reference's CreateOctree (OctreeSearch.cpp:74-89): tree build, upsweep, walk per body.

    Returns (acc, root_com, node_count).  root_size defaults to ComputeCubeSize's value.  div_mode: the reading of
    FVector::operator/= in ComputeMass (0 = reciprocal multiply, 1 = divide)."""
    pos = _f32(pos); mass = _f32(mass)
    n = pos.shape[0]
    if root_size is None:
        root_size = bounds_f32(pos)
    origin = _f32(np.asarray(root_origin, np.float32))
    acc = np.zeros((n, 3), np.float32)
    com = np.zeros(3, np.float32)
    cnt = ctypes.c_int(0)
    rc = lib().oracle_octree_f32(n, _fp(pos), _fp(mass), _fp(origin), np.float32(root_size), np.float32(theta), float(g),
                                 pow_mode, div_mode, _fp(acc), _fp(com), ctypes.byref(cnt), None, None)
    if rc:
        raise RuntimeError(f"oracle_octree_f32 rc={rc} (1 = duplicate positions)")
    return acc, com, cnt.value


def octree_leaves_f32(pos, mass, root_origin=(0.0, 0.0, 0.0), root_size=None):
    """What DrawOctreeBoxes (OctreeSearch.cpp:36-45) draws on the tree of these bodies: (boxes[n,4] = Origin, Size of
    every occupied leaf; order[n] = the particle in it), both in the reference's depth-first order."""
    pos = _f32(pos); mass = _f32(mass)
    n = pos.shape[0]
    if root_size is None:
        root_size = bounds_f32(pos)
    origin = _f32(np.asarray(root_origin, np.float32))
    boxes = np.zeros((n, 4), np.float32)
    order = np.zeros(n, np.int32)
    rc = lib().oracle_octree_f32(n, _fp(pos), _fp(mass), _fp(origin), np.float32(root_size), np.float32(1.0), REF_G, 0, 0,
                                 None, None, None, _fp(boxes), order.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if rc:
        raise RuntimeError(f"oracle_octree_f32 rc={rc} (1 = duplicate positions)")
    return boxes, order


def last_max_depth():
    """Deepest node (root = 0) the last octree_* / tick_aos_f32 call reached while inserting the bodies."""
    return int(lib().oracle_last_max_depth())


def octree_depth_f32(pos, root_origin=(0.0, 0.0, 0.0), root_size=None):
    """Octree::Add (OctreeSearch.h:60-81) of these bodies only: how deep the insertion goes (root = 0; 201 = cut off, the
    reference would recurse without bound).  Bodies that share L octant digits end in leaves of depth L + 1."""
    pos = _f32(pos)
    n = pos.shape[0]
    if root_size is None:
        root_size = bounds_f32(pos)
    origin = _f32(np.asarray(root_origin, np.float32))
    mass = np.ones(n, np.float32)
    lib().oracle_octree_f32(n, _fp(pos), _fp(mass), _fp(origin), np.float32(root_size), np.float32(1.0), REF_G, 0, 0,
                            None, None, None, None, None)
    return last_max_depth()


def kick_drift_f32(pos, vel, acc, dt):
    """OctreeSearch.cpp:28-31.  Returns new (pos, vel)."""
    pos = _f32(pos).copy(); vel = _f32(vel).copy(); acc = _f32(acc)
    lib().oracle_kick_drift_f32(pos.shape[0], _fp(pos), _fp(vel), _fp(acc), np.float32(dt))
    return pos, vel


def bounds_f32(pos):
    """OctreeSearch.cpp:47-56."""
    pos = _f32(pos)
    return float(lib().oracle_bounds_f32(pos.shape[0], _fp(pos)))


def tick_aos_f32(particles, dt, theta=REF_THETA, g=REF_G, pow_mode=0, root_com=None, size=0.0, div_mode=0):
    """AOctreeSearch::Tick physics (OctreeSearch.cpp:25-32) on FParticle records, in place.

    theta < 0 → index-order direct sum.  Returns (root_com, size)."""
    assert particles.dtype == PARTICLE_DTYPE and particles.flags.c_contiguous
    com = np.zeros(3, np.float32) if root_com is None else _f32(root_com).copy()
    sz = ctypes.c_float(size)
    rc = lib().oracle_tick_aos2_f32(particles.shape[0], particles.ctypes.data, np.float32(dt), np.float32(theta),
                                    float(g), pow_mode, div_mode, _fp(com), ctypes.byref(sz))
    if rc:
        raise RuntimeError(f"oracle_tick_aos_f32 rc={rc}")
    return com, sz.value


def forces_direct_f64(pos, mass, g=REF_G, eps=0.0, i0=0, i1=None, nthreads=1):
    pos = _f64(pos); mass = _f64(mass)
    n = pos.shape[0]
    i1 = n if i1 is None else i1
    acc = np.zeros((n, 3), np.float64)
    rc = lib().oracle_forces_direct_f64(n, _dp(pos), _dp(mass), float(g), float(eps * eps), i0, i1, _dp(acc), nthreads)
    if rc:
        raise RuntimeError(f"oracle_forces_direct_f64 rc={rc}")
    return acc[i0:i1]


def kick_drift_f64(pos, vel, acc, dt):
    pos = _f64(pos).copy(); vel = _f64(vel).copy(); acc = _f64(acc)
    lib().oracle_kick_drift_f64(pos.shape[0], _dp(pos), _dp(vel), _dp(acc), float(dt))
    return pos, vel


def energy_f64(pos, vel, mass, g=REF_G, eps=0.0, nthreads=1):
    pos = _f64(pos); vel = _f64(vel); mass = _f64(mass)
    ke = ctypes.c_double(0); pe = ctypes.c_double(0)
    lib().oracle_energy_f64(pos.shape[0], _dp(pos), _dp(vel), _dp(mass), float(g), float(eps * eps),
                            ctypes.byref(ke), ctypes.byref(pe), nthreads)
    return ke.value, pe.value

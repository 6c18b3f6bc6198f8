"""OctreeSearch — Python view of the AOctreeSearch mirror (include/nbody_actor.hpp; reference
/root/reference/Source/NBody/OctreeSearch.h:111-149, OctreeSearch.cpp).  Same method names, defaults and
silent-guard behaviour as the UE4 actor, so parity tests read like calls on the reference class."""
import ctypes

import numpy as np

from . import _lib
from .engine import PARTICLE_DTYPE


class OctreeSearch:
    def __init__(self, device=0, precision=_lib.PREC_F32, G=1.0e4, eps=0.0):
        self._L = _lib.lib()
        self._h = ctypes.c_void_p(self._L.nbody_actor_create())
        if not self._h:
            raise MemoryError("nbody_actor_create failed")
        self._L.nbody_actor_set_engine(self._h, device, precision, G, eps)
        self._cb = None

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.nbody_actor_destroy(self._h)
            self._h = None

    # fields (OctreeSearch.h:117-127)
    @property
    def Size(self):
        return self._L.nbody_actor_get_size(self._h)

    @property
    def Initialized(self):
        return bool(self._L.nbody_actor_get_initialized(self._h))

    @property
    def PhDeltaTime(self):
        return self._L.nbody_actor_get_ph_delta_time(self._h)

    @PhDeltaTime.setter
    def PhDeltaTime(self, v):
        self._L.nbody_actor_set_ph_delta_time(self._h, v)

    @property
    def ShowOctree(self):
        return bool(self._L.nbody_actor_get_show_octree(self._h))

    @ShowOctree.setter
    def ShowOctree(self, v):
        self._L.nbody_actor_set_show_octree(self._h, int(bool(v)))

    @property
    def Particles(self):
        n = self._L.nbody_actor_num_particles(self._h)
        out = np.zeros(n, PARTICLE_DTYPE)
        if n:
            self._L.nbody_actor_get_particles(self._h, out.ctypes.data, n)
        return out

    @property
    def LastStatus(self):
        return self._L.nbody_actor_last_status(self._h)

    def set_theta(self, theta):
        self._L.nbody_actor_set_theta(self._h, theta)

    def set_devices(self, devices):
        """Share the bodies over several GPUs (nbody_create_multi) from the next CreateSpacePoints / SetParticles on."""
        devs = (ctypes.c_int32 * len(devices))(*devices) if devices else None
        self._L.nbody_actor_set_devices(self._h, devs, len(devices) if devices else 0)

    def set_seed(self, seed):
        self._L.nbody_actor_set_seed(self._h, seed)

    # methods (OctreeSearch.h:130-148)
    def BeginPlay(self):
        pass

    def Tick(self, DeltaSeconds=0.0):
        self._L.nbody_actor_tick(self._h, DeltaSeconds)

    def ComputeCubeSize(self):
        self._L.nbody_actor_compute_cube_size(self._h)

    def CreateSpacePoints(self, N, Size=200.0):
        self._L.nbody_actor_create_space_points(self._h, N, Size)

    def CreateOctree(self):
        self._L.nbody_actor_create_octree(self._h)

    def CleanParticles(self):
        self._L.nbody_actor_clean_particles(self._h)

    def SetParticles(self, particles):
        a = np.ascontiguousarray(particles, PARTICLE_DTYPE)
        self._L.nbody_actor_set_particles(self._h, a.ctypes.data, a.shape[0])

    def PushParticles(self, particles=None):
        """The host has edited records between two Ticks (in the reference `Particles` IS the state: OctreeSearch.h:118,
        OctreeSearch.cpp:28-31).  particles=None pushes the actor's live records as they stand (edit them through
        `live_particles()`); otherwise the given records are copied in first.  History is kept (nbody_push_particles)."""
        if particles is None:
            self._L.nbody_actor_push_particles(self._h, None, 0)
        else:
            a = np.ascontiguousarray(particles, PARTICLE_DTYPE)
            self._L.nbody_actor_push_particles(self._h, a.ctypes.data, a.shape[0])

    def live_particles(self):
        """The actor's own records as a numpy VIEW (no copy): what the device writes every frame, and what PushParticles() uploads."""
        n = self._L.nbody_actor_num_particles(self._h)
        ptr = self._L.nbody_actor_particle_data(self._h)
        if not n or not ptr:
            return np.zeros(0, PARTICLE_DTYPE)
        buf = (ctypes.c_char * (n * PARTICLE_DTYPE.itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=PARTICLE_DTYPE, count=n)

    def set_box_callback(self, on_box=None):
        """on_box((ox,oy,oz), size) <- DrawDebugBox of an occupied leaf (ShowOctree and theta > 0)."""
        b = _lib.DRAW_BOX_FN((lambda user, o, sz: on_box((o[0], o[1], o[2]), sz)) if on_box else 0)
        self._cb_box = b
        self._L.nbody_actor_set_box_callback(self._h, b, None)

    def set_draw_callbacks(self, on_flush=None, on_point=None):
        """on_flush() <- FlushPersistentDebugLines; on_point((x,y,z), size) <- DrawDebugPoint."""
        f = _lib.FLUSH_FN((lambda user: on_flush()) if on_flush else 0)
        p = _lib.DRAW_POINT_FN((lambda user, pos, sz: on_point((pos[0], pos[1], pos[2]), sz)) if on_point else 0)
        self._cb = (f, p)
        self._L.nbody_actor_set_draw_callbacks(self._h, f, p, None)

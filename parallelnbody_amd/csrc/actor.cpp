// C shim over nbody::OctreeSearchActor (include/nbody_actor.hpp) — see include/nbody_actor.h.
#include <algorithm>
#include <new>

#include "../../include/nbody_actor.h"
#include "../../include/nbody_actor.hpp"

struct nbody_actor {
  nbody::OctreeSearchActor impl;
};

extern "C" {

nbody_actor *nbody_actor_create(void) { return new (std::nothrow) nbody_actor(); }
void nbody_actor_destroy(nbody_actor *a) { delete a; }

void nbody_actor_create_space_points(nbody_actor *a, int32_t n, float size) { if (a) a->impl.CreateSpacePoints(n, size); }
void nbody_actor_set_particles(nbody_actor *a, const nbody_particle *p, int32_t n) { if (a) a->impl.SetParticles(p, n); }
void nbody_actor_compute_cube_size(nbody_actor *a) { if (a) a->impl.ComputeCubeSize(); }
void nbody_actor_create_octree(nbody_actor *a) { if (a) a->impl.CreateOctree(); }
void nbody_actor_tick(nbody_actor *a, float delta_seconds) { if (a) a->impl.Tick(delta_seconds); }
void nbody_actor_clean_particles(nbody_actor *a) { if (a) a->impl.CleanParticles(); }

void nbody_actor_set_draw_callbacks(nbody_actor *a, nbody_flush_fn flush, nbody_draw_point_fn point, void *user) {
  if (!a) return;
  if (flush) a->impl.OnFlushPersistentDebugLines = [flush, user]() { flush(user); };
  else a->impl.OnFlushPersistentDebugLines = nullptr;
  if (point) a->impl.OnDrawDebugPoint = [point, user](const float *pos, float sz) { point(user, pos, sz); };
  else a->impl.OnDrawDebugPoint = nullptr;
}

void nbody_actor_set_box_callback(nbody_actor *a, nbody_draw_box_fn box, void *user) {
  if (!a) return;
  if (box) a->impl.OnDrawDebugBox = [box, user](const float *o, float sz) { box(user, o, sz); };
  else a->impl.OnDrawDebugBox = nullptr;
}

float nbody_actor_get_size(const nbody_actor *a) { return a ? a->impl.Size : 0.0f; }
int32_t nbody_actor_get_initialized(const nbody_actor *a) { return a && a->impl.Initialized ? 1 : 0; }
int32_t nbody_actor_num_particles(const nbody_actor *a) { return a ? (int32_t)a->impl.NumParticles() : 0; }
float nbody_actor_get_ph_delta_time(const nbody_actor *a) { return a ? a->impl.PhDeltaTime : 0.0f; }
void nbody_actor_set_ph_delta_time(nbody_actor *a, float dt) { if (a) a->impl.PhDeltaTime = dt; }
int32_t nbody_actor_get_show_octree(const nbody_actor *a) { return a && a->impl.ShowOctree ? 1 : 0; }
void nbody_actor_set_show_octree(nbody_actor *a, int32_t show) { if (a) a->impl.ShowOctree = show != 0; }
void nbody_actor_set_theta(nbody_actor *a, float theta) { if (a) a->impl.Theta = theta; }
void nbody_actor_set_seed(nbody_actor *a, uint64_t seed) { if (a) a->impl.Seed = seed; }
void nbody_actor_set_engine(nbody_actor *a, int32_t device, int32_t precision, double G, double eps) {
  if (!a) return;
  a->impl.Device = device; a->impl.Precision = precision; a->impl.G = G; a->impl.Eps = eps;
}
void nbody_actor_set_devices(nbody_actor *a, const int32_t *devices, int32_t n) {
  if (!a) return;
  if (devices && n > 0) a->impl.Devices.assign(devices, devices + n);
  else a->impl.Devices.clear();
}
int32_t nbody_actor_last_status(const nbody_actor *a) { return a ? a->impl.LastStatus : NBODY_ERR_INVALID; }

int32_t nbody_actor_get_particles(nbody_actor *a, nbody_particle *out, int32_t capacity) {
  if (!a || !out || capacity < 0) return 0;
  a->impl.SyncParticles();
  const int32_t n = std::min<int32_t>(capacity, (int32_t)a->impl.NumParticles());
  std::copy(a->impl.ParticleData(), a->impl.ParticleData() + n, out);
  return n;
}

nbody_particle *nbody_actor_particle_data(nbody_actor *a) {
  if (!a) return nullptr;
  a->impl.SyncParticles();
  return a->impl.ParticleData();
}

void nbody_actor_push_particles(nbody_actor *a, const nbody_particle *p, int32_t n) {
  if (!a) return;
  if (p) {
    if (n != (int32_t)a->impl.NumParticles()) { a->impl.LastStatus = NBODY_ERR_INVALID; return; }
    if (p != a->impl.ParticleData()) std::copy(p, p + n, a->impl.ParticleData());
  }
  a->impl.PushParticles();
}

void nbody_actor_release_storage(nbody_actor *a) { if (a) a->impl.ReleaseStorage(); }

}  // extern "C"

/*
 * nbody_actor.h — C view of nbody::OctreeSearchActor (nbody_actor.hpp), the mirror of the reference's
 * AOctreeSearch (Source/NBody/OctreeSearch.h:111-149).  For hosts that cannot include C++ (ctypes,
 * cgo-style bindings) and for the parity tests.  Same conventions as nbody.h; methods that are
 * `void` in the reference return nothing here either — nbody_actor_last_status reports the last
 * C-ABI code.
 */
#ifndef NBODY_AMD_ACTOR_H
#define NBODY_AMD_ACTOR_H

#include "nbody.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nbody_actor nbody_actor;

/* Draw callbacks: FlushPersistentDebugLines (OctreeSearch.cpp:24) and DrawDebugPoint (.cpp:41). */
typedef void (*nbody_flush_fn)(void *user);
typedef void (*nbody_draw_point_fn)(void *user, const float position[3], float point_size);
typedef void (*nbody_draw_box_fn)(void *user, const float origin[3], float size);   /* DrawDebugBox, .cpp:40 */

/* AOctreeSearch(), .cpp:8.  The opening angle starts at 1.0 — what the reference hard-codes (OctreeSearch.cpp:85) — so
 * the actor reproduces the shipped Barnes-Hut trajectories and ShowOctree boxes as is; nbody_actor_set_theta(a, 0)
 * selects the exact O(N^2) all-pairs limit (the hot path of this engine; needed for fp64 / Kahan and device lists). */
NBODY_AMD_API nbody_actor *nbody_actor_create(void);
NBODY_AMD_API void nbody_actor_destroy(nbody_actor *a);
NBODY_AMD_API void nbody_actor_create_space_points(nbody_actor *a, int32_t n, float size);   /* .cpp:58-72 */
NBODY_AMD_API void nbody_actor_set_particles(nbody_actor *a, const nbody_particle *p, int32_t n);
NBODY_AMD_API void nbody_actor_compute_cube_size(nbody_actor *a);                      /* .cpp:47-56 */
NBODY_AMD_API void nbody_actor_create_octree(nbody_actor *a);                          /* .cpp:74-89 */
NBODY_AMD_API void nbody_actor_tick(nbody_actor *a, float delta_seconds);              /* .cpp:21-34 */
NBODY_AMD_API void nbody_actor_clean_particles(nbody_actor *a);                        /* .cpp:91-97 */
NBODY_AMD_API void nbody_actor_set_draw_callbacks(nbody_actor *a, nbody_flush_fn flush, nbody_draw_point_fn point, void *user);
NBODY_AMD_API void nbody_actor_set_box_callback(nbody_actor *a, nbody_draw_box_fn box, void *user);

/* Fields (OctreeSearch.h:117-127 + the build-defined knobs of nbody_actor.hpp). */
NBODY_AMD_API float nbody_actor_get_size(const nbody_actor *a);
NBODY_AMD_API int32_t nbody_actor_get_initialized(const nbody_actor *a);
NBODY_AMD_API int32_t nbody_actor_num_particles(const nbody_actor *a);
NBODY_AMD_API float nbody_actor_get_ph_delta_time(const nbody_actor *a);
NBODY_AMD_API void nbody_actor_set_ph_delta_time(nbody_actor *a, float dt);
NBODY_AMD_API int32_t nbody_actor_get_show_octree(const nbody_actor *a);
NBODY_AMD_API void nbody_actor_set_show_octree(nbody_actor *a, int32_t show);
NBODY_AMD_API void nbody_actor_set_theta(nbody_actor *a, float theta);
NBODY_AMD_API void nbody_actor_set_seed(nbody_actor *a, uint64_t seed);
NBODY_AMD_API void nbody_actor_set_engine(nbody_actor *a, int32_t device, int32_t precision, double G, double eps);
/* Share the bodies over several GPUs from the next CreateSpacePoints / SetParticles on (nbody_create_multi); n = 0: one GPU again. */
NBODY_AMD_API void nbody_actor_set_devices(nbody_actor *a, const int32_t *devices, int32_t n);
NBODY_AMD_API int32_t nbody_actor_last_status(const nbody_actor *a);
/* Copy the (synchronised) Particles array out; returns the number of records written. */
NBODY_AMD_API int32_t nbody_actor_get_particles(nbody_actor *a, nbody_particle *out, int32_t capacity);
/* The live (synchronised) records themselves — the counterpart of the reference's `Particles` member (OctreeSearch.h:118).  The
 * device writes every frame's records here; what the host edits here reaches the simulation with nbody_actor_push_particles. */
NBODY_AMD_API nbody_particle *nbody_actor_particle_data(nbody_actor *a);
/* The host has edited records between two Ticks (in the reference that alone changes the simulation: `Particles` is the state,
 * OctreeSearch.cpp:28-31).  p == NULL: push the live records as they stand; otherwise n records (n = the actor's count) are
 * copied in first.  History is kept (step count, the next tree's root centre): nbody_push_particles. */
NBODY_AMD_API void nbody_actor_push_particles(nbody_actor *a, const nbody_particle *p, int32_t n);
/* A host that gave the actor its own array (C++: OctreeSearchActor::AllocateParticles) calls this BEFORE it resizes or frees that
 * array: the context unpins the storage while it is still allocated.  No-op for records the actor allocated itself. */
NBODY_AMD_API void nbody_actor_release_storage(nbody_actor *a);

#ifdef __cplusplus
}
#endif
#endif

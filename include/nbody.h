/*
 * nbody.h — C-ABI of the MI355X-native N-body engine (libnbody_amd.so).
 *
 * The reference (Milias/ParallelNbody) has no FFI; its boundary for this path is the public
 * surface of the UE4 actor AOctreeSearch (Source/NBody/OctreeSearch.h:111-149).  Each entry point
 * below replaces one responsibility of that class and cites it.  Paths are relative to
 * /root/reference/Source/NBody/.
 *
 * Conventions: plain C types only; every call returns 0 (NBODY_OK) or a negative NBODY_ERR_*;
 * nothing throws across the boundary; the context is an opaque caller-owned pointer; all host
 * buffers are caller-allocated; calls on one context are not thread-safe (the reference runs on
 * the UE4 game thread only).  There is NO CPU fallback: without a HIP device nbody_create fails
 * with NBODY_ERR_NO_DEVICE.
 */
#ifndef NBODY_AMD_H
#define NBODY_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* (Not NBODY_API: UnrealBuildTool defines <MODULE>_API for every module, and the reference's module is called NBody — its actor is
 * declared `class NBODY_API AOctreeSearch`, OctreeSearch.h:112.  A header of this library must not redefine the host's macro.) */
#define NBODY_AMD_API __attribute__((visibility("default")))

#define NBODY_VERSION_MAJOR 0
#define NBODY_VERSION_MINOR 1

typedef struct nbody_ctx nbody_ctx;

enum {
  NBODY_OK = 0,
  NBODY_ERR_INVALID = -1,     /* bad argument */
  NBODY_ERR_NO_DEVICE = -2,   /* no HIP device / device ordinal out of range */
  NBODY_ERR_HIP = -3,         /* a HIP runtime call failed; see nbody_last_error */
  NBODY_ERR_STATE = -4,       /* call made in the wrong state (e.g. no particles set) */
  NBODY_ERR_NOMEM = -5,
  NBODY_ERR_UNSUPPORTED = -6
};

/* Arithmetic of the force accumulation. */
enum {
  NBODY_PREC_F32 = 0,         /* reference-compatible: fp32 pair law, fp32 accumulate */
  NBODY_PREC_F32_KAHAN = 1,   /* fp32 pair law, Kahan-compensated fp32 accumulate */
  NBODY_PREC_F64 = 2          /* fp64 state, pair law and accumulate */
};

/*
 * Handling of pairs at distance exactly 0 when eps == 0 (OctreeSearch.h:102, `if (d == 0) return;`).
 * EXACT and SELECT reproduce it for every representable distance; FLOOR adds the smallest eps^2 that
 * keeps G*m_max/d^3 finite in fp32 (about 1e-20: d == 0 pairs still contribute exactly 0, but pairs
 * closer than ~4e-7 length units are softened) and saves two vector instructions per pair.
 */
enum {
  NBODY_ZERO_EXACT = 0,       /* default: r2 += clamp01(1 - r2*2^126), 2 full-rate VALU ops */
  NBODY_ZERO_SELECT = 1,      /* compare + select (half-rate ops on gfx950); for A/B measurements */
  NBODY_ZERO_FLOOR = 2        /* eps^2 floor, not bit-faithful below d ~ 4e-7 */
};

/*
 * Force algorithm.  Both evaluate the same pair law over all pairs; they differ in summation order only.
 * TILED: every ordered pair (i, j), one-sided (kernels.hip).  SYMMETRIC: every unordered pair once, feeding
 * both bodies (kernels_sym.hip, kernels_sym64.hip; work plan csrc/sym_plan.h): fp32, Kahan fp32 and fp64 contexts;
 * when the bodies are sharded the slices must be equal multiples of 256 * i_per_thread bodies and the host drives
 * nbody_step_begin / all-to-all / nbody_step_end.  The per-body summation order then depends on the number of ranks
 * (TILED's does not).  AUTO picks SYMMETRIC where it applies and n_total >= 9216 (where its whole step gets faster
 * than the one-sided kernel's), else TILED.
 */
enum { NBODY_ALGO_AUTO = 0, NBODY_ALGO_TILED = 1, NBODY_ALGO_SYMMETRIC = 2 };

/* Device buffers reachable through nbody_device_ptr / nbody_bind_device_state. */
enum {
  NBODY_BUF_POSM = 0,         /* [n_total] x,y,z,mass  (float4, or double4 for NBODY_PREC_F64) */
  NBODY_BUF_VEL = 1,          /* [i_count] vx,vy,vz,0 */
  NBODY_BUF_ACC = 2           /* [i_count] ax,ay,az,0 */
};

/* Kernels whose device time nbody_kernel_time reports. */
enum { NBODY_KERNEL_FORCES = 0, NBODY_KERNEL_UPDATE = 1 };

/* FParticle — OctreeSearch.h:8-18.  Same field order and offsets (40 bytes). */
typedef struct nbody_particle {
  float Mass;
  float Position[3];
  float Velocity[3];
  float Acceleration[3];
} nbody_particle;

/*
 * Engine parameters.  The reference hard-codes G = 1e4 (OctreeSearch.h:104), no softening
 * (.h:101-104) and fp32; nbody_default_params fills in exactly those.
 */
typedef struct nbody_params {
  uint32_t struct_size;   /* sizeof(nbody_params), for ABI versioning */
  int32_t n_total;        /* bodies in the whole system (Particles.Num(), OctreeSearch.h:118) */
  int32_t i_begin;        /* first body this context owns (range partition over GPUs); default 0 */
  int32_t i_count;        /* bodies this context owns; 0 = n_total - i_begin */
  int32_t device;         /* HIP device ordinal */
  int32_t precision;      /* NBODY_PREC_* */
  double G;               /* gravitational constant; reference 1e4 */
  double eps;             /* Plummer softening length; reference 0 (d == 0 pairs are skipped) */
  int32_t tile;           /* bodies per LDS tile: 64, 128, 256 (default), 512 */
  int32_t i_per_thread;   /* i-bodies per lane: 1, 2, 4 (8: fp32 symmetric kernels, 16: the plain one only); 0 = auto */
  int32_t j_split;        /* j-range chunks summed separately then combined in order; 0 = auto (a function of n_total only) */
  int32_t time_kernels;   /* nonzero: bracket kernels with HIP events for nbody_kernel_time */
  int32_t zero_mode;      /* how d == 0 pairs are dropped when eps == 0 (NBODY_ZERO_*); 0 = default */
  int32_t algorithm;      /* NBODY_ALGO_*; 0 = auto */
  float theta;            /* Barnes-Hut opening angle.  0 (default) = exact all-pairs, the hot path of this engine.  > 0 =
                             the reference's own tree walk (OctreeSearch.h:99-108; it ships 1.0, OctreeSearch.cpp:85) */
  int32_t bh_div_mode;    /* theta > 0 only: reading of `CenterOfMass /= TotalMass` in ComputeMass (OctreeSearch.h:95).  0 (default):
                             FVector::operator/=(float) multiplies by the fp32 reciprocal (UE4 4.9 as remembered — the engine
                             is not vendored); 1: three divisions.  The oracle has the same switch (div_mode). */
} nbody_params;

/* ---- lifecycle ---------------------------------------------------------------------------- */

/* Fill `p` with the reference-compatible defaults (G=1e4, eps=0, fp32, device 0). */
NBODY_AMD_API int nbody_default_params(nbody_params *p);

/* AOctreeSearch ctor + CreateSpacePoints' allocation (OctreeSearch.cpp:8,62).
 *
 * Reproducibility.  Every force pass is deterministic (no atomics in any sum): the same context, state and call sequence
 * give the same bits, run after run.  WHICH sums are formed — the launch geometry — is chosen here, once, from the
 * parameters and from two facts about the device: its compute-unit count (workgroup slots of the symmetric pass's work
 * plan) and, beyond N = 2^22, its total memory (whether the partial-sum pool is shared by phases); the strip-length
 * divisor K of that plan also follows an estimate of the pass's duration from rates measured on MI355X (6.6e12 / 6.0e12 /
 * 2.7e12 interactions/s: fp32 / compensated / fp64); between 16385 and 139264 bodies (compensated: 12288 ... 40960) a context
 * that owns all bodies runs the even-share plan instead — one work item per slot, a function of the body count, the bodies
 * per lane and the CU count (nbody_sym_plan_is_even).  Results are therefore reproducible on every MI355X, and across
 * library versions only where the release notes say so; on a part with another CU count they agree to rounding, not in
 * every bit.  (Plain fp32 systems of up to 16384 bodies run forces_block_pk_kernel: there the CU count only decides how
 * many bodies share a workgroup, which no sum depends on — the same bits on any part.)  nbody_get_launch_config,
 * nbody_get_algorithm and nbody_sym_pool_info say what was chosen;
 * NBODY_ALGO_TILED with explicit tile / i_per_thread / j_split depends on the parameters alone.
 *
 * Masses.  A context that owns all bodies re-reads every mass before every pass: a mass changed through a bound or handed-out
 * device buffer is seen at once.  SHARDED contexts (i_count < n_total) treat masses as immutable between uploads
 * (nbody_set_* / nbody_push_particles / nbody_load_checkpoint): the first go of a pass looks only at the own slice — the other
 * ranks' records may still be arriving — so a mass another rank changes in its buffer selects the general form of the kernels one
 * pass late. */
NBODY_AMD_API int nbody_create(const nbody_params *p, nbody_ctx **out);

/*
 * The same context over several GPUs of one node, driven by ONE caller thread — the reference's only caller is the game
 * thread (OctreeSearch.cpp:21-34).  Bodies are range-partitioned in equal slices over `devices` (n_total a multiple of
 * n_dev); every device keeps all positions.  Per step each device runs the force pass of its slice, [symmetric
 * algorithm] the j-side sums change hands by grouped ncclSend/ncclRecv, the slices are integrated, and ONE in-place
 * ncclAllGather of the positions (RCCL over xGMI; communicators from ncclCommInitAll, librccl loaded at this call)
 * brings every device up to date.  p->i_begin / i_count / device are ignored (the context owns all bodies).
 * theta > 0 (fp32): the reference's tree is ONE tree (OctreeSearch.cpp:79-81) and a body's walk (.cpp:83-86) reads it and writes
 * that body alone — every device builds the whole tree from its copy of the positions (the build is the reference's arithmetic in
 * a fixed order: the same bits everywhere), walks and integrates its own slice, and the same all-gather follows: frames equal the
 * one-device context's in EVERY byte, whatever n_dev is.
 * Every entry point of this header works on the result except the device-plumbing ones (nbody_set_stream, nbody_device_ptr,
 * nbody_bind_*, nbody_step_begin/_end, nbody_exchange_*), which report NBODY_ERR_UNSUPPORTED.  With n_dev = 1 results equal
 * nbody_create's bit for bit.
 */
NBODY_AMD_API int nbody_create_multi(const nbody_params *p, const int32_t *devices, int32_t n_dev, nbody_ctx **out);

/* CleanParticles (OctreeSearch.cpp:91-97).  NULL is allowed, like `delete NULL` there. */
NBODY_AMD_API void nbody_destroy(nbody_ctx *ctx);

/* Message of the last error on `ctx` (or of the last failed nbody_create when ctx is NULL). */
NBODY_AMD_API const char *nbody_last_error(const nbody_ctx *ctx);

NBODY_AMD_API int nbody_version(void);

/* Number of HIP devices visible (0 when there is none; never fails). */
NBODY_AMD_API int nbody_device_count(void);

/* ---- state in ----------------------------------------------------------------------------- */

/* TArray<FParticle> contents (OctreeSearch.h:8-18,118): all n_total records, `stride` bytes apart (>= 40). */
NBODY_AMD_API int nbody_set_particles(nbody_ctx *ctx, const void *aos, size_t stride, int32_t n);

/* The host has EDITED records of a running simulation.  In the reference `Particles` is the state itself (OctreeSearch.h:118;
 * the Tick reads and writes it in place, OctreeSearch.cpp:28-31): code that changes Particles[i] between two Ticks changes the
 * simulation, and nothing else restarts — the next tree is still rooted at the previous tree's CoM (OctreeSearch.cpp:77-79).
 * Same upload as nbody_set_particles (Mass, Position, Velocity, Acceleration of all n_total records), but the history stays:
 * nbody_steps_done goes on counting and the Barnes-Hut root centre is kept. */
NBODY_AMD_API int nbody_push_particles(nbody_ctx *ctx, const void *aos, size_t stride, int32_t n);

/* Native layout: posm4 = n_total x {x,y,z,m}, vel4 = n_total x {vx,vy,vz,unused}, fp32. */
NBODY_AMD_API int nbody_set_state_soa(nbody_ctx *ctx, const float *posm4, const float *vel4, int32_t n);

/* Same in fp64 (converted down for fp32 contexts). */
NBODY_AMD_API int nbody_set_state_soa_f64(nbody_ctx *ctx, const double *posm4, const double *vel4, int32_t n);

/* ---- the hot path ------------------------------------------------------------------------- */

/*
 * The force loop of CreateOctree (OctreeSearch.cpp:83-86) at theta = 0: Acceleration_i =
 * sum over j of the pair law (OctreeSearch.h:101-104) for the owned bodies against all n_total.
 */
NBODY_AMD_API int nbody_compute_forces(nbody_ctx *ctx);

/*
 * The body of Tick (OctreeSearch.cpp:25-32), `nsteps` times: forces(x_n); v += dt*a; x += dt*v.
 * dt <= 0 is a no-op, as PhDeltaTime <= 0 freezes the reference (.cpp:25).  Asynchronous on the
 * context's stream; the getters synchronise.  On a sharded context (i_count < n_total) nsteps
 * must be 1: the caller all-gathers NBODY_BUF_POSM across ranks between steps.
 */
NBODY_AMD_API int nbody_step(nbody_ctx *ctx, float dt, int32_t nsteps);

/*
 * The same step in two phases, for hosts that share the bodies over several GPUs:
 *   nbody_step_begin : force pass (OctreeSearch.cpp:83-86 at theta = 0) for the owned bodies
 *   nbody_step_end   : v += dt*a; x += dt*v for the owned bodies (OctreeSearch.cpp:28-31); dt <= 0 only stores a
 * Between them a sharded context running NBODY_ALGO_SYMMETRIC needs ONE all-to-all: it has evaluated each body pair
 * once and holds, in `send`, what its pairs contribute to every other rank's bodies (n_ranks segments of
 * bytes_per_rank, segment q for rank q); `recv` must receive segment r of every rank's `send`.  nbody_exchange_info
 * reports n_ranks = 0 when no exchange is needed.  After nbody_step_end the owned slice of NBODY_BUF_POSM is
 * all-gathered as with nbody_step.
 */
NBODY_AMD_API int nbody_step_begin(nbody_ctx *ctx);
NBODY_AMD_API int nbody_step_end(nbody_ctx *ctx, float dt);
/*
 * nbody_step_begin in two goes, so that the all-gather of the previous step's positions can still be in flight when the
 * next force pass starts ("compute own-range j-tiles while the gather is in flight"):
 *   nbody_step_begin_local  : the part of the force pass that needs the OWNED slice of NBODY_BUF_POSM only — on a sharded
 *                             fp32 NBODY_ALGO_SYMMETRIC context the strips whose j range lies inside the own slice (about
 *                             1/n_ranks of the rank's work); on any other context nothing
 *   nbody_step_begin_remote : the rest (everything, where the first go did nothing).  Queue it behind the gather:
 *                             an event wait on the context's stream is enough, the host need not block.
 * Same plan, same partial-sum segments, same order of additions as nbody_step_begin: the results are identical in every
 * bit.  The first go reads the masses of ALL bodies (they are what the last upload left; the gather rewrites them with
 * the same bits): a host that changes masses in the bound buffer must let the gather finish first.
 */
NBODY_AMD_API int nbody_step_begin_local(nbody_ctx *ctx);
NBODY_AMD_API int nbody_step_begin_remote(nbody_ctx *ctx);
NBODY_AMD_API int nbody_exchange_info(nbody_ctx *ctx, void **send, void **recv, size_t *bytes_per_rank, int32_t *n_ranks);
/* Use caller-owned device buffers (e.g. torch tensors) for the exchange: send = n_total x float4, recv = n_ranks x i_count x float4
 * (double4 on an fp64 context). */
NBODY_AMD_API int nbody_bind_exchange(nbody_ctx *ctx, void *send, void *recv);
/* Host-staged exchange for callers without a device-side collective: copy `send` out (n_total x 4 floats) /
 * copy `recv` in (n_ranks x i_count x 4 floats); doubles on an fp64 context. */
NBODY_AMD_API int nbody_exchange_read_send(nbody_ctx *ctx, void *host);
NBODY_AMD_API int nbody_exchange_write_recv(nbody_ctx *ctx, const void *host);

/*
 * Barnes-Hut mode: with theta > 0 the force pass is the reference's CreateOctree (OctreeSearch.cpp:
 * 74-89) on the device — same region octree (root centre = previous tree's CoM, half-width = ComputeCubeSize), same
 * mass upsweep, same depth-first walk with `Size/d < Theta`, same arithmetic — instead of the all-pairs kernels.
 * fp32 contexts only.  A context that owns a SLICE of the bodies (i_count < n_total: one rank of a sharded job, one device of
 * nbody_create_multi) builds the whole tree from the replicated positions and walks its own bodies; the caller all-gathers
 * NBODY_BUF_POSM between steps as at theta = 0.  nbody_set_particles / nbody_set_state_* reset the "previous CoM" to zero.
 * Systems of more than 4096 bodies sort a frame's path keys starting from the previous frame's order and take its Size out of the
 * previous frame's walk: a frame whose sort gives up (the records were replaced, the root box jumped) is queued again
 * by the library at the call's one wait; nothing of it shows but the time.
 */
NBODY_AMD_API int nbody_set_theta(nbody_ctx *ctx, float theta);
/* The opening angle in force (nbody_params.theta, nbody_set_theta, or what nbody_load_checkpoint took over from a file). */
NBODY_AMD_API int nbody_get_theta(nbody_ctx *ctx, float *theta);
/* Nodes and levels of the last tree built, and its root CoM (= the next frame's root centre). */
NBODY_AMD_API int nbody_bh_stats(nbody_ctx *ctx, int32_t *nodes, int32_t *levels, float root_com[3]);
/* What DrawOctreeBoxes passes to DrawDebugBox when ShowOctree is set (OctreeSearch.cpp:39-40): for every body the box
 * (Origin.x, Origin.y, Origin.z, Size) of the leaf that held it in the last tree; 4 floats per body, `stride` bytes apart. */
NBODY_AMD_API int nbody_bh_leaf_boxes(nbody_ctx *ctx, float *boxes, size_t stride);
/* The order in which DrawOctreeBoxes (OctreeSearch.cpp:36-45) meets the bodies on the last tree built: depth first,
 * children 0..7; order[k] = index of the body in the k-th occupied leaf.  n_total ints. */
NBODY_AMD_API int nbody_bh_leaf_order(nbody_ctx *ctx, int32_t *order);

/* ComputeCubeSize (OctreeSearch.cpp:47-56): max over owned bodies of max(|x|,|y|,|z|). */
NBODY_AMD_API int nbody_get_bounds(nbody_ctx *ctx, float *size);

/* ---- state out ---------------------------------------------------------------------------- */

/* What DrawDebugPoint reads (OctreeSearch.cpp:41): positions of bodies [first, first+count) of the
 * whole system, 3 floats each, `stride` bytes apart (>= 12). */
NBODY_AMD_API int nbody_get_positions(nbody_ctx *ctx, float *xyz, size_t stride, int32_t first, int32_t count);

/* Owned records [i_begin, i_begin+i_count) into aos[0..i_count): Mass, Position, Velocity, Acceleration. */
NBODY_AMD_API int nbody_get_particles(nbody_ctx *ctx, void *aos, size_t stride);

/* One frame of AOctreeSearch::Tick (OctreeSearch.cpp:21-34) with a single host synchronisation: if dt > 0, *size =
 * ComputeCubeSize of the current positions (.cpp:26) and one Tick body (.cpp:27-31); then the owned FParticle records
 * as nbody_get_particles delivers them (what .cpp:33,41 draws).  size and aos may each be NULL.  Same results as
 * nbody_get_bounds + nbody_step(dt, 1) + nbody_get_particles; not for sharded symmetric contexts (phased step).
 * Systems whose step is a kernel or two (theta = 0 up to 16384 bodies) and every theta > 0 frame (its walk) write the records from that
 * kernel straight into page-locked host memory — into `aos` itself when it lies in a range pinned with
 * nbody_pin_host_buffer (stride 40), so that the frame queues no copy of the mirror at all. */
NBODY_AMD_API int nbody_tick(nbody_ctx *ctx, float dt, float *size, void *aos, size_t stride);

/* Renderer hand-off straight into the caller's buffer (what OctreeSearch.cpp:41 reads every frame):
 * page-lock `bytes` of caller memory at `host` for this context.  nbody_get_positions (stride 12) and
 * nbody_get_particles (stride 40) whose destination lies inside a pinned range then DMA into it directly — one copy,
 * device to destination — instead of going through the context's own staging buffer and a host memcpy.  The memory
 * stays the caller's; unpin it (or destroy the context) before freeing it.  Results are identical either way. */
NBODY_AMD_API int nbody_pin_host_buffer(nbody_ctx *ctx, void *host, size_t bytes);
NBODY_AMD_API int nbody_unpin_host_buffer(nbody_ctx *ctx, void *host);

/* Owned bodies, native layout (fp32; converted down from fp64 contexts).  Either pointer may be NULL. */
NBODY_AMD_API int nbody_get_state_soa(nbody_ctx *ctx, float *posm4, float *vel4, float *acc4);
NBODY_AMD_API int nbody_get_state_soa_f64(nbody_ctx *ctx, double *posm4, double *vel4, double *acc4);

/* Kinetic energy of the owned bodies and their share of the potential energy (1/2 m_i phi_i),
 * evaluated in fp64 on the device.  Sum over contexts for the system total.  Build-defined (no
 * reference counterpart). */
NBODY_AMD_API int nbody_energy(nbody_ctx *ctx, double *ke, double *pe);

/* ---- device plumbing (torch / RCCL interop) -------------------------------------------------- */

/* Launch on the caller's HIP stream (hipStream_t as void*); NULL = the context's own stream. */
NBODY_AMD_API int nbody_set_stream(nbody_ctx *ctx, void *hip_stream);

/* Raw device pointer of one state buffer (e.g. as the send/recv buffer of an all-gather).  It stays valid, and keeps
 * naming the live buffer, until nbody_destroy or a nbody_bind_device_state of that buffer.  Asking for NBODY_BUF_POSM
 * tells the context that positions may change behind its back: from then on it re-reads them before every force pass
 * (no fused update + preparation, no buffer-swapping one-launch step on systems of up to 16384 bodies) — same results, a
 * little slower. */
NBODY_AMD_API int nbody_device_ptr(nbody_ctx *ctx, int32_t which, void **ptr, size_t *bytes);

/* Use caller-owned device memory for the state (any may be NULL = keep the context's own).
 * Sizes as in NBODY_BUF_*.  The caller keeps them alive until nbody_destroy. */
NBODY_AMD_API int nbody_bind_device_state(nbody_ctx *ctx, void *posm, void *vel, void *acc);

/* Block until everything queued on the context's stream has finished. */
NBODY_AMD_API int nbody_synchronize(nbody_ctx *ctx);

/* Sum of device time (ms) and number of launches of one kernel since the last reset, from HIP
 * events recorded on the launch stream (needs params.time_kernels).  Synchronises. */
NBODY_AMD_API int nbody_kernel_time(nbody_ctx *ctx, int32_t which, double *total_ms, int64_t *launches);
NBODY_AMD_API int nbody_kernel_time_reset(nbody_ctx *ctx);

/* The shader clock the timed force kernels actually ran at since the last nbody_kernel_time_reset, in MHz (needs
 * params.time_kernels): every workgroup of forces_sym_pk_kernel / forces_tile_pk_kernel / forces_sym_f64_kernel reads the
 * shader-clock counter and the fixed reference counter at both ends and the ratio of the sums is reported — the power-limited force
 * loops hold a different clock on different boxes (2.13 - 2.33 GHz seen; the fp64 loop moves it most), and time x clock is what tells
 * a slower box from slower code.  0 when no instrumented kernel has run (the generic scalar, block and Barnes-Hut kernels are not
 * instrumented).  compute_units: of the context's device.
 * A multi-device context reports its slowest device.  Synchronises. */
NBODY_AMD_API int nbody_kernel_clock(nbody_ctx *ctx, double *shader_mhz, int32_t *compute_units);

/* Launch geometry actually chosen (for logs and DESIGN.md tables). */
NBODY_AMD_API int nbody_get_launch_config(nbody_ctx *ctx, int32_t *tile, int32_t *i_per_thread, int32_t *j_split,
                                      int32_t *blocks, int32_t *threads);

/* Name of the force kernel this context launches (for logs and profiles). */
NBODY_AMD_API const char *nbody_force_kernel_name(const nbody_ctx *ctx);

/* NBODY_ALGO_* actually in use, and (symmetric only) the bodies per i-set = per block of the ring, 256 x i_per_thread
 * (the parameter keeps its round-1 name). */
NBODY_AMD_API int nbody_get_algorithm(nbody_ctx *ctx, int32_t *algorithm, int32_t *super_tile);

/* The symmetric pass's partial-sum pool: its size in bytes (0 for the one-sided kernels) and how many phases share its
 * j-side area (1 unless the pool of a one-pass plan would exceed a third of the card: N = 2^23 on one 288 GB card). */
NBODY_AMD_API int nbody_sym_pool_info(nbody_ctx *ctx, uint64_t *pool_bytes, int32_t *phases);

/* Symmetric contexts (fp32, Kahan fp32, fp64): did the last force pass run the equal-mass form of the kernel?  When every
 * body has the same mass (the usual Plummer-sphere set-up) the pair loop sums |d|^-3 d with no mass factor — fp32: 14
 * packed instructions per register pair and step instead of 16; fp64: 20 operations per pair instead of 22 — and the
 * common G m is applied once per body afterwards; same pair law, summation
 * order and d == 0 handling, results equal to the general form's to rounding.  Found out on the device before every
 * pass (from the host's copy when only this library writes the positions), so a mass changed through a bound or
 * handed-out buffer is seen.  *in_use = 0 on every other kind of context.  Synchronises the stream. */
NBODY_AMD_API int nbody_equal_mass_form(nbody_ctx *ctx, int32_t *in_use);

/* Host only (no device needed): the work plan of the symmetric force pass for a context owning [i_begin, i_begin +
 * i_count) of n_total bodies — i-sets of `bodies_per_iset` bodies (256 x i_per_thread) against strips of 64-body
 * subtiles, strip lengths by guided self-scheduling over `slots` resident workgroups (csrc/sym_plan.h).  items, when
 * not NULL, receives 8 int32 per work item: i0, j0, n_sub, flags (1 = the strip lies inside the i-set's own block, 2 = it
 * writes no j-side sums), slot_i, slot_j (element offsets of its partial-sum segments), 0, 0.  own_mode: how own-block strips are
 * costed — 1 = the fp32 kernels (symmetric between register pairs of two slots), 2 = the fp64 kernel (between slots).  Build-defined diagnostics; the CPU tests use it
 * to check that every body pair is evaluated exactly once. */
NBODY_AMD_API int nbody_sym_plan_describe(int32_t n_total, int32_t i_begin, int32_t i_count, int32_t bodies_per_iset,
                                      int32_t slots, int32_t k_guided, int32_t min_sub, int32_t own_mode, int32_t *n_items,
                                      uint64_t *pool_elems, int32_t *items, int32_t items_cap);
/* The same with the strip divisor K given in tenths (the library's own choices include K = 1.5). */
NBODY_AMD_API int nbody_sym_plan_describe_tenths(int32_t n_total, int32_t i_begin, int32_t i_count, int32_t bodies_per_iset,
                                             int32_t slots, int32_t k_guided_x10, int32_t min_sub, int32_t own_mode,
                                             int32_t *n_items, uint64_t *pool_elems, int32_t *items, int32_t items_cap);

/* A plan whose j-side partial-sum segments must share an area of at most j_budget_elems pool elements: the items run in
 * phases (consecutive runs of the launch order), each phase's j-side sums are folded before the next reuses the area — how
 * the library keeps the symmetric pass beyond N = 2^22 on one card.  phases[] receives n_phases + 1 item numbers. */
NBODY_AMD_API int nbody_sym_plan_describe_phased(int32_t n_total, int32_t i_begin, int32_t i_count, int32_t bodies_per_iset,
                                             int32_t slots, int32_t k_guided_x10, int32_t min_sub, uint64_t j_budget_elems,
                                             int32_t *n_items, uint64_t *pool_elems, int32_t *items, int32_t items_cap,
                                             int32_t *n_phases, int32_t *phases, int32_t phases_cap);

/* The even-share plan of the symmetric pass (plain fp32 contexts that own all bodies, mid sizes: csrc/sym_plan.h): exactly
 * n_items_wanted work items (at least one per i-set) of equal cost — a row of the pair matrix, i.e. an i-set against its own
 * block and then its forward blocks in ring order, cut at equal cumulative cost to four steps of a subtile's 64.  items
 * receives 8 int32 per work item: i0, j0 (first subtile; the run goes on in ring order, from the system's last granule to
 * granule 0), n_sub (subtiles touched), flags (4 | 1 if the first subtile lies in the own block), slot_i, slot_j, k0 (first
 * step of the first subtile), k_skip (steps of the last subtile left to the next item). */
NBODY_AMD_API int nbody_sym_plan_describe_even(int32_t n_total, int32_t bodies_per_iset, int32_t n_items_wanted, int32_t *n_items,
                                           uint64_t *pool_elems, int32_t *items, int32_t items_cap);

/* 1 when the context's symmetric pass runs an even-share plan (above), 0 otherwise (guided strips, or not symmetric). */
NBODY_AMD_API int32_t nbody_sym_plan_is_even(const nbody_ctx *ctx);

/* Host only: register pairs of bodies (2 ... 8; two bodies each) a workgroup of forces_block_pk_kernel owns for a plain fp32
 * system of n_total bodies on a device with `compute_units` CUs — the rule of csrc/capi.hip (smallest ceil(workgroups /
 * CUs) x pairs, larger workgroups on a tie).  No result depends on it; the CPU tests check the rule. */
NBODY_AMD_API int32_t nbody_block_pairs_describe(int32_t n_total, int32_t compute_units);

/* ---- checkpoint / resume (build-defined: the reference keeps its state in a non-serialised TArray) ---------- */

/* Raw little-endian dump of the context's state: header (format NBDYCKP2: sizes, steps, G, eps, theta and — Barnes-Hut —
 * the previous tree's centre of mass, where the reference roots the next tree, OctreeSearch.cpp:77-79), all
 * positions+masses, owned velocities and accelerations.  Resuming from it continues the trajectory bit for bit, at
 * theta = 0 and at theta > 0.  A sharded job writes one file per rank; a multi-device context writes the single file a
 * one-device context of the whole system would. */
NBODY_AMD_API int nbody_save_checkpoint(nbody_ctx *ctx, const char *path);
/* The context must have the same n_total, precision, G and eps as the one that saved the file, and an owned range inside
 * the file's (so a whole-system file also feeds every slice of a sharded or multi-device job).  An fp32 context that
 * owns all bodies also takes over theta and the tree root. */
NBODY_AMD_API int nbody_load_checkpoint(nbody_ctx *ctx, const char *path, int64_t *steps_done);
/* Updates applied since the state was set or loaded. */
NBODY_AMD_API int nbody_steps_done(nbody_ctx *ctx, int64_t *steps);

/* ---- initial conditions (host only; no device needed) ---------------------------------------- */

/*
 * CreateSpacePoints (OctreeSearch.cpp:58-72) with a seeded generator: uniform box
 * (+-size, +-size, +-size/10) about `center`, isotropic velocities of magnitude 250..500, masses
 * 1..5000, body 0 pinned at the origin at rest with mass 5000.  The reference uses the engine's
 * unseeded RNG, so only the distribution is reproduced.  Out: n x 4 floats each.
 */
NBODY_AMD_API int nbody_ic_reference_box(int32_t n, float size, const float center[3], uint64_t seed,
                                     float *posm4, float *vel4);

/* Seeded equal-mass Plummer sphere in virial equilibrium for constant G (build-defined workload). */
NBODY_AMD_API int nbody_ic_plummer(int32_t n, double total_mass, double scale_radius, double G, uint64_t seed,
                               float *posm4, float *vel4);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_AMD_H */

// Internal launcher interface between the C-ABI (capi.hip) and the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace nbody {

struct ForceLaunch {
  const void *posm;     // [n_total] float4 / double4 : x,y,z,m
  void *accp;           // [j_split][i_count] float4 / double4 partial accelerations
  int n_total;
  int i_begin;
  int i_count;
  int tile;             // LDS tile, bodies
  int ipt;              // i-bodies per lane
  int j_split;          // number of j chunks
  int j_chunk;          // bodies per chunk (multiple of tile)
  double G;
  double eps2;          // > 0: softened (also the "floor" mode); == 0: exact d == 0 skip
  int zero_mode;        // for eps2 == 0: 1 = clamp trick (default), 2 = compare+select (A/B only)
  int precision;        // NBODY_PREC_*
  int wave;             // 0: tile kernels; 2 ... 8: forces_block_pk_kernel (kernels_block.hip), a workgroup per `wave` register
                        // pairs of bodies, its lanes split the j range (j_split must be 1)
  int uni = 0;          // wave >= 2: 1 = the caller knows that every body has body 0's mass (equal-mass form only), 0 = it knows they
                        // do not (general form only), -1 = launch both, each looks at `general`
  int guarded = 0;      // wave >= 2: guard every pair (no optimistic bare pass)
  void *dup_table = nullptr;   // packed fp32 kernel, exact mode: coincident-body detector's table [dup_slots] + {flag, count};
  int dup_slots = 0;           // with it, tiles that hold no self pair run without the d == 0 guard when no two bodies coincide
  // packed fp32 kernel: device int that is 0 when every body has body 0's mass (equal-mass form: no mass factor in the
  // pair loop); nullptr = general form only.  check_masses: run mass_check_kernel on the positions first (somebody else
  // may have written the buffer since the host last looked).
  void *general = nullptr;
  int check_masses = 0;
  void *clk = nullptr;  // packed fp32 tile kernel: two device uint64 the workgroups add their shader-clock / reference-clock intervals to (pk_common.h)
};

// All-pairs force partials.  Returns hipSuccess or the launch error.
hipError_t launch_forces(const ForceLaunch &L, hipStream_t s);
// Small and mid-size single-context fp32 systems (L.wave != 0): the whole Tick body — forces, v += dt*a, x += dt*v — in one launch.
// New positions go to posm_out (a second buffer: the old one is still being read); the caller swaps them afterwards.
// stage / size_bits / size_zero (optional): see BlockLaunch — the frame's FParticle mirror and ComputeCubeSize from the same launch.
hipError_t launch_step_small(const ForceLaunch &L, void *posm_out, void *vel, void *acc, float dt, hipStream_t s,
                             void *stage = nullptr, void *size_bits = nullptr, void *size_zero = nullptr);
// Blocks / threads launch_forces will use for L (for logs).
void forces_geometry(const ForceLaunch &L, int *blocks, int *threads);

// Small and mid-size fp32 systems — kernels_block.hip: a workgroup owns `np` register pairs of bodies and its lanes split
// the j range; a body's whole sum is finished inside its workgroup, so with dt > 0 the Tick's update rides along (new
// positions into posm_out) and a step is one launch with no partial rows.
struct BlockLaunch {
  const void *posm = nullptr;   // [n_total] float4, read only
  void *posm_out = nullptr;     // dt > 0: [n_total] float4, the owned bodies' new (x, y, z, m); must not be posm
  void *vel = nullptr;          // dt > 0: [i_count] float4
  void *acc = nullptr;          // [i_count] float4
  int n_total = 0, i_begin = 0, i_count = 0;
  int np = 2;                   // register pairs per workgroup: 2 ... 8
  double G = 0.0, eps2 = 0.0;   // eps2 > 0 softened (also the floor mode); == 0 exact d == 0 skip
  float dt = 0.f;               // <= 0: accelerations only
  int uni = 0;                  // 1: every body has body 0's mass (the caller knows): no mass factor in the pair loop; 0: general form;
                                // -1: both forms are launched and *general (0 = equal masses) says which one runs
  const void *general = nullptr;
  // dt > 0, optional (the actor's frame in one launch): FParticle records of the owned bodies after the update (10 floats
  // each); a pre-zeroed word for the bit pattern of ComputeCubeSize over the positions before the update, and a second
  // word this launch clears for the next frame
  void *stage = nullptr;
  void *size_bits = nullptr, *size_zero = nullptr;
  int optimistic = 1;           // eps2 == 0: bare pair law outside the own group first, guarded walk only where a sum came out non-finite
};
hipError_t launch_block(const BlockLaunch &L, hipStream_t s);

// acc[i] = sum_c accp[c][i] in chunk order; if dt > 0 also v += dt*a; x += dt*v (owned slice of posm).
hipError_t launch_update(int precision, void *posm, void *vel, void *acc, const void *accp, int i_begin,
                         int i_count, int j_split, float dt, hipStream_t s);

// Symmetric (each unordered pair once) force pass — kernels_sym.hip (fp32), kernels_sym64.hip (fp64).  Who evaluates
// which pairs, where the partial sums go and in which order they are added is the plan of sym_plan.h, uploaded once.
struct SymLaunch {
  const void *posm;     // [n_total] float4 / double4, all bodies
  void *posg;           // fp32 only: [n_pad] float4 (x, y, z, G m) + zero-mass padding, rewritten every pass
  void *pool;           // [pool_elems] float4 / double4: the items' partial-sum segments
  const void *items;    // [n_items] SymItem: one workgroup each
  int n_items;
  const void *i_ptr, *i_off;   // CSR over own 64-body granules: i-side segments
  const void *j_ptr, *j_off;   // CSR over all granules: j-side segments
  void *send;           // [n_total] float4: this rank's j-side contribution to every body (n_src segments of i_count)
  const void *recv;     // [n_src][i_count] float4: what every rank contributed to the own bodies (== send when n_src == 1)
  int n_total;
  int n_pad;            // blocks * bodies per i-set
  int n_src;            // ranks sharing the bodies
  int np;               // fp32: register pairs of i-bodies per lane (1, 2, 4, 8); fp64: bodies per lane / 2 (1, 2)
  int precision;        // NBODY_PREC_F32 (float4 rows) or NBODY_PREC_F64 (double4 rows)
  int kahan;            // fp32 only: Kahan-compensated accumulation
  double G;
  double eps2;          // > 0 softened / floor; == 0 exact d == 0 skip (clamp form)
  void *dup_table;      // eps2 == 0 only: dup_slots x 8-byte hash slots + one flag word; nullptr = always run the guarded kernel
  int dup_slots;        // power of two >= 2 * n_total
  // fused single-device fp32 stepping (update_sym_fused_kernel): the update folds the j-side rows itself and prepares the
  // next pass (posg, the other detector table); a force pass then launches no prep and no reduce_j kernel
  int fused = 0;              // launch_forces_sym: no reduce_j; launch_update_sym: the fused kernel
  int skip_prep = 0;          // posg and dup_table are already those of the current positions
  void *dup_table_next = nullptr;
  // equal-mass form (forces_sym_pk_kernel / forces_sym_f64_kernel, UNI): `general` is a device int the preparation
  // kernel (fp64: mass_check_kernel) raises when a body's mass differs from body 0's; nullptr = general kernels only.  uni_host: 1 = the host knows the masses are equal
  // and nobody else writes the buffer (equal-mass launches only), 0 = it knows they are not (general launches only),
  // -1 = launch both, each looks at `general`.
  void *general = nullptr;
  int uni_host = 0;
  // Sharded fp32 contexts can run the pass in two goes, so that the strips inside the own slice need not wait for the other
  // ranks' positions: phase 1 = prepare the own slice [own_begin, own_begin + own_count) and run items [0, n_local);
  // phase 2 = prepare the rest, run items [n_local, n_items) and fold the j-side rows; phase 0 = everything at once.
  // Same plan, same segments, same summation order: the bits do not depend on how the pass is cut.
  int phase = 0;
  int n_local = 0;
  int own_begin = 0, own_count = 0;
  // What one call of launch_forces_sym does (fp32; the fp64 launcher runs a whole pass): prepare (phase says which bodies),
  // launch the items [item0, item1) of the launch order (item1 < 0: phase's own range), fold the j-side lists j_ptr / j_off
  // into `send` — on top of what is there when fold_accumulate (pool phases after the first, sym_plan.h).
  int item0 = 0, item1 = -1;
  int do_prep = 1, do_fold = -1;     // do_fold < 0: as the phase says (phases 0 and 2 fold)
  int fold_accumulate = 0;
  int clear_detector = 1;            // the fold also clears the coincident-body table for the next pass (the pass's LAST fold only)
  // even-share plan (sym_plan.h, plain fp32 on one context): the items carry first / last steps and follow the ring order,
  // which goes on at body 0 past `wrap` bodies (64 x granules of the system)
  int even = 0, wrap = 0;
  void *clk = nullptr;               // fp32: two device uint64 the force kernel's workgroups add their clock intervals to (pk_common.h)
};
// forces + fold of the j-side rows into L.send
hipError_t launch_forces_sym(const SymLaunch &L, hipStream_t s);
// acc = own i-side rows + L.recv segments; dt > 0: kick-drift of the owned slice
hipError_t launch_update_sym(const SymLaunch &L, void *posm, void *vel, void *acc, int i_begin, int i_count, float dt,
                             hipStream_t s);

// GPU Barnes-Hut with the reference's tree and opening rule — bh_frame.hip (host side), kernels_bh_{small,sort,build,walk}.hip, bh_common.h.  fp32.
struct BhState;
// n bodies; the context owns [i_begin, i_begin + i_count) of them (all: 0, n) — a slice builds the whole tree and walks its own bodies
hipError_t bh_create(BhState **out, int n, int i_begin, int i_count);   // *out is set even on failure: bh_destroy it
void bh_destroy(BhState *b);
void bh_positions_changed(BhState *b);                         // a body was moved by something other than a frame's walk
void bh_positions_external(BhState *b);                        // the caller holds the position buffer from now on
hipError_t bh_reset_root(BhState *b, hipStream_t s);          // previous CoM := 0 (a new scene, OctreeSearch.cpp:77)
// One frame = CreateOctree (OctreeSearch.cpp:74-89: ComputeCubeSize, the tree rooted at the previous CoM, ComputeMass), the walk
// Octree::ComputeForces(body, theta) of every body and — with dt > 0 — the Tick's update of (posm, vel) in place, QUEUED on the
// stream: nothing waits for the host (small systems, bh_is_small: two launches; larger ones up to 2^20 bodies: nine; beyond, one
// wait inside for the deepest level).  bh_collect waits for the stream and reports the frames queued since the last collect:
// *status 0 ok, 1 tree deeper than 42 levels, 2 node pool exhausted; a refused frame and everything queued behind it leave the
// state untouched.  keep_root != 0: the next tree's root centre stays what it was (a diagnostic pass).
// stage (optional): the walk also writes every body's FParticle record (10 floats, body order) there — the frame's mirror.
bool bh_is_small(const BhState *b);
hipError_t bh_frame(BhState *b, void *posm, void *vel, void *acc, float theta, double G, float dt, int keep_root, float *stage,
                    hipStream_t s);
float bh_last_size(const BhState *b);                         // Size of the last frame bh_collect has seen
hipError_t bh_collect(BhState *b, hipStream_t s, int *status, int *frames);   // *status 3: queue the frames that were not built again
hipError_t bh_debug_clocks(BhState *b, long long out[16 + 3 * 512], hipStream_t s);   // tuning builds only (tools/bh_phases.py)
hipError_t bh_debug_poison(BhState *b, int kind, hipStream_t s);   // tests: kind 1 = the warm sort's bucket counts := 3 each (a fill that never ran)
void bh_debug_sort_counts(const BhState *b, long long *warm_frames, long long *retries);   // frames sorted from the previous order; times frames were queued again
const float *bh_root_device(const BhState *b);                // device (ox, oy, oz, Size) of the last tree (small systems)
hipError_t bh_get_tree_com(BhState *b, float out[3], hipStream_t s);   // root CoM of the last tree built
void bh_set_div_mode(BhState *b, int div_mode);            // 0: `/=` in ComputeMass multiplies by the reciprocal; 1: divides
// order[k] = the body whose leaf a depth-first walk (children 0..7) meets k-th in the last tree built (host array, n ints)
hipError_t bh_leaf_order(BhState *b, int *out_host, hipStream_t s);
hipError_t bh_stats(BhState *b, hipStream_t s, int *nodes, int *levels);   // waits for the stream when the last tree's counts are still on their way
// out[body] = (ox, oy, oz, Size) of the leaf holding the body, for the last tree built
hipError_t bh_leaf_boxes(BhState *b, void *out, hipStream_t s);
hipError_t bh_get_root_com(BhState *b, float out[3], hipStream_t s);
hipError_t bh_set_root_com(BhState *b, const float in[3], hipStream_t s);

// out_bits (uint32, pre-zeroed) = bit pattern of max_i max(|x|,|y|,|z|) over the owned slice.
// zero_word (optional): a second word this launch clears — for callers that alternate between two result words.
hipError_t launch_bounds(int precision, const void *posm, int i_begin, int i_count, unsigned int *out_bits,
                         hipStream_t s, unsigned int *zero_word = nullptr);

// out_bits (uint32, pre-zeroed) = bit pattern of max_j |m_j| over all n_total bodies.
hipError_t launch_massmax(int precision, const void *posm, int n_total, unsigned int *out_bits, hipStream_t s);

// Device-side repack for the renderer hand-off: FParticle records (10 floats) of the owned slice / packed xyz.
hipError_t launch_pack_particles(int precision, const void *posm, const void *vel, const void *acc, float *out,
                                 int i_begin, int i_count, hipStream_t s);
hipError_t launch_pack_positions(int precision, const void *posm, float *out, int first, int count, hipStream_t s);

// fp64 energy pieces: out[0] = KE of owned bodies, out[1] = sum_i 1/2 m_i phi_i.  Two launches: per-workgroup parts into
// `partials` (energy_partials(n_total, i_count) doubles), then a fixed-order fold — no atomics, reproducible bits.
size_t energy_partials(int n_total, int i_count);
hipError_t launch_energy(int precision, const void *posm, const void *vel, int n_total, int i_begin, int i_count,
                         double G, double eps2, double *partials, double *out, hipStream_t s);

}  // namespace nbody

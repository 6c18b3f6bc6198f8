// C-ABI of libnbody_amd.so (include/nbody.h): context, device state, launches.  Host C++ over the
// HIP runtime; no torch types, no exceptions across the boundary (the entry points that allocate host memory catch
// std::bad_alloc), no CPU fallback.
#include <hip/hip_runtime.h>
#include <cstdint>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "../../include/nbody.h"
#include "kernels.h"
#include "multi.h"
#include "sym_plan.h"

namespace {

thread_local std::string g_create_error;

struct EventPair { hipEvent_t a, b; bool counts = true; };   // counts: the interval is a whole pass (not the first go of two)

struct KernelTimer {
  std::vector<EventPair> pending;   // recorded, not yet read
  std::vector<EventPair> pool;      // free
  double total_ms = 0.0;
  int64_t launches = 0;
};

}  // namespace

struct nbody_ctx {
  nbody::Multi *multi = nullptr;   // nbody_create_multi: this context is a front for one context per device (multi.h)
  nbody_params p;
  size_t elem;                 // bytes per float4/double4 element
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  void *posm = nullptr, *vel = nullptr, *acc = nullptr, *accp = nullptr;
  void *posm_alt = nullptr;    // small systems: second position buffer of the one-launch step (swapped with posm)
  bool own_posm = false, own_vel = false, own_acc = false;
  void *d_stage = nullptr, *h_stage = nullptr;   // renderer hand-off staging (device repack target, pinned mirror)
  size_t stage_bytes = 0;
  void *scratch = nullptr;     // 64 B device scratch (bounds bits, energy sums)
  void *h_scratch = nullptr;   // pinned mirror
  void *energy_part = nullptr; // nbody_energy: one pair of doubles per workgroup, folded in a fixed order
  int j_split = 1, j_chunk = 0, ipt = 1, tile = 256;
  int sym_np = 1;                        // register pairs per lane of the symmetric kernel
  std::vector<std::pair<char *, size_t>> pinned;   // caller memory page-locked by nbody_pin_host_buffer
  int wave = 0;                // block kernel: register pairs of bodies per workgroup (0 = tile / symmetric kernels)
  int bh_word = 0;             // larger Barnes-Hut systems: which of the two Size words (scratch + 40, + 44) the next frame uses
  int tick_word = 0;           // nbody_tick on the one-launch step: which of the two Size words (scratch + 32, + 36) is cleared and next
  bool have_state = false;
  double floor_eps2 = -1.0;    // NBODY_ZERO_FLOOR: eps^2 floor for the current masses (< 0 = not yet computed)
  // symmetric algorithm (kernels_sym.hip, kernels_sym64.hip; plan: sym_plan.h)
  bool sym = false;
  int sym_bi = 0, sym_pad = 0, sym_items_n = 0, sym_nsrc = 1, sym_slots = 0, sym_min_sub = 0;
  int sym_n_local = 0;                   // items [0, sym_n_local): strips inside the own slice (sym_plan.h)
  std::vector<int> sym_phase_item0;      // pool phases of the plan: phase p = items [p], [p + 1]) (one phase unless the pool had to be shared)
  int sym_n_gran = 0;
  double sym_k = 0.0;
  bool sym_even = false;                 // the plan is an even-share plan (sym_plan.h)
  size_t sym_pool_elems = 0;
  nbody::SymPlan *plan = nullptr;                  // host copy, dropped once uploaded
  void *sym_pool = nullptr, *sym_items = nullptr, *sym_iptr = nullptr, *sym_ioff = nullptr, *sym_jptr = nullptr,
       *sym_joff = nullptr, *sym_posg = nullptr;
  void *sym_send = nullptr, *sym_recv = nullptr;   // exchange buffers (recv == send when the context owns all bodies)
  void *sym_dup_table = nullptr;                   // coincident-body detector (hash slots + flag)
  int sym_dup_slots = 0;
  // fused single-device fp32 stepping: the update prepares the next pass (posg + the OTHER detector table)
  void *sym_dup_table2 = nullptr;
  int sym_dup_cur = 0;                             // which of the two tables holds the verdict on the current positions
  bool sym_posg_valid = false;                     // posg (and that table) describe the current positions
  bool posm_escaped = false;                       // the caller holds / owns the position buffer: it may change behind our back
  // equal-mass kernels: device word the preparation kernel (fp64: mass_check_kernel) raises when two masses differ (sticky; the host resets it
  // with every state it uploads) and what the host itself saw in that state (1 all equal, 0 not, -1 never saw one)
  void *sym_general = nullptr;
  int masses_equal = -1;
  bool own_send = false, own_recv = false;
  bool step_open = false;      // nbody_step_begin done, nbody_step_end pending
  bool step_local = false;     // nbody_step_begin_local done, nbody_step_begin_remote pending
  int64_t steps_done = 0;      // updates applied since the state was set (saved in checkpoints)
  // Barnes-Hut mode (bh_frame.hip, kernels_bh_*.hip)
  float theta = 0.0f;
  nbody::BhState *bh = nullptr;
  void *bh_acc = nullptr;      // [i_count] float4: the walk's output, summed (j_split = 1) by update_kernel
  struct { float dt = 0.f; float *stage = nullptr; int queued = 0; bool timed = false; } bh_batch;   // what bh_enqueue queued since the last bh_finish
  KernelTimer timers[2];
  int clk_items = 0;                   // NBODY_SYM_ITEM_CLOCKS: work items with stamps of their own behind the eight clock words
  unsigned long long *clk = nullptr;   // time_kernels: {shader-clock cycles, reference-clock ticks} summed over the force kernels' workgroups (pk_common.h)
  int wall_khz = 0, cus = 0;           // hipDeviceAttributeWallClockRate, compute units
  std::string err;
};

namespace {

int fail(nbody_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_error = buf;
  return code;
}

int run_update(nbody_ctx *c, float dt);   // (defined below; part_bh_queue_frame comes first)
int queue_forces_bh(nbody_ctx *c, bool diagnostic);

// A call on a multi-device context is answered by its Multi; its message becomes the context's.
int multi_rc(nbody_ctx *c, int rc) {
  if (rc) c->err = nbody::multi_error(c->multi);
  return rc;
}
int multi_unsupported(nbody_ctx *c, const char *who) {
  return fail(c, NBODY_ERR_UNSUPPORTED, "%s: not available on a multi-device context (nbody_create_multi); use one context per device", who);
}

#define HIP_TRY(c, expr)                                                                         \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) return fail((c), NBODY_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// One caller thread may hold contexts on several devices (nbody_create_multi does): every entry point that allocates,
// launches, copies or records makes its context's device the current one first.
int use_device(nbody_ctx *c) {
  if (c->multi) return NBODY_OK;
  HIP_TRY(c, hipSetDevice(c->p.device));
  return NBODY_OK;
}

// Up to N = 16384 a workgroup owns a few bodies and spreads the j range over its lanes (forces_block_pk_kernel,
// kernels_block.hip: one launch per step, no partial rows; 2 ... 8 register pairs of bodies per workgroup).  Above it the
// symmetric pass (plain fp32; Kahan and fp64 go through the tile kernels up to their own threshold).  Whole steps without
// events, same box (profiles/r03_block_kernel_by_n.txt): N = 14336 0.0538 ms against the symmetric pass's 0.0630, 16384 —
// four full workgroups per CU — 0.0674 / 0.0735, 17408 0.0922 / 0.0710; round 2's one-pair-per-workgroup kernel
// (small_pk_kernel, gone): N = 2000 0.0067 ms against 0.0053 now, 4096 0.0143 / 0.0096, 6000 0.0246 / 0.0150.
constexpr int kSmallSystem = 6656;      // the threshold of rounds 1-2 (tile kernel above it); still Kahan's and the forced geometries'
int block_max_n() { static const int v = [] { const char *e = getenv("NBODY_BLOCK_MAX_N"); const int x = e ? atoi(e) : 0; return x > 0 ? x : 16385; }(); return v; }

// Register pairs per workgroup for the block kernel.  A CU works through its workgroups two at a time (2 waves per SIMD at
// ~200 VGPRs) and a workgroup left alone runs about twice as fast, so a CU's time is its number of workgroups times the
// pairs each one carries: the bodies are cut so that ceil(workgroups / CUs) * pairs is smallest, larger workgroups first on
// a tie (fewer prologues).  Measured against all of 2 ... 8 at fifteen sizes (profiles/r03_block_kernel_np_by_n.txt): the
// rule picks the fastest or within 6 % of it.  A function of n_total and the CU count only.
int block_pairs(int n_total, int cus) {
  if (const char *e = getenv("NBODY_BLOCK_NP")) { const int v = atoi(e); if (v >= 1 && v <= 8) return v; }   // tuning only
  if (cus <= 0) cus = 256;
  int best = 8;
  long long best_cost = -1;
  for (int np = 8; np >= 2; --np) {
    const long long wgs = (n_total + 2 * np - 1) / (2 * np);
    const long long cost = (wgs + cus - 1) / cus * np;
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = np; }
  }
  return best;
}

// forces_block_kernel (Kahan, fp64): 4 or 8 bodies per workgroup by the same rule
int block_bodies(int n_total, int cus) {
  if (const char *e = getenv("NBODY_BLOCK_NB")) { const int v = atoi(e); if (v == 4 || v == 8) return v; }   // tuning only
  if (cus <= 0) cus = 256;
  // on a tie eight, unless all workgroups of four are resident at once anyway (measured, whole steps in both precisions:
  // N = 2000 6.8 - 7.5 us with four against 7.7 with eight; N = 4096 16.8 - 18.9 against 15.9 - 18.1)
  const long long r8 = ((n_total + 7) / 8 + cus - 1) / cus, r4 = ((n_total + 3) / 4 + cus - 1) / cus;
  return (r4 * 4 < r8 * 8 || (r4 * 4 == r8 * 8 && r4 <= 2)) ? 4 : 8;
}

int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  if (!e || !*e) return dflt;
  const int v = atoi(e);
  return v > 0 ? v : dflt;
}

int floor_pow2(long long v) { int p = 1; while ((long long)p * 2 <= v) p *= 2; return p; }

// Launch geometry.  j_split is a function of n_total only, so that the per-body summation order
// (and hence every bit of the trajectory) does not depend on how many GPUs share the bodies.
void choose_geometry(nbody_ctx *c) {
  const nbody_params &p = c->p;
  c->tile = p.tile > 0 ? p.tile : 256;
  if (p.i_per_thread > 0) c->ipt = p.i_per_thread > 4 ? 4 : p.i_per_thread;   // 8 and 16 exist for the symmetric kernel only
  // whole steps without events (profiles/r02_small_system_thresholds.txt): N = 8192 0.0277 ms with four bodies per lane, 0.0258 with
  // two; 10240 0.0444 / 0.0427; N = 16384: 0.102 with four, 0.106 with two
  else c->ipt = (p.precision == NBODY_PREC_F64) ? 1 : (p.n_total >= 12288 ? 4 : (p.n_total >= kSmallSystem ? 2 : 1));
  int js;
  if (p.j_split > 0) {
    js = p.j_split;
  } else {
    // aim at >= 2048 workgroups (8 per CU) down to an 8-way body partition; a chunk may be a single tile — small
    // systems are short of workgroups, not of work per workgroup (N = 8192: 52 us with 8 chunks, 29 us with 32)
    const long long per_block = 256LL * c->ipt;
    long long iblocks8 = (p.n_total / 8 + per_block - 1) / per_block;
    if (iblocks8 < 1) iblocks8 = 1;
    js = floor_pow2((2048 + iblocks8 - 1) / iblocks8);
    const int max_js = p.n_total / c->tile;
    if (js > max_js) js = max_js;
    if (js < 1) js = 1;
  }
  int chunk = (p.n_total + js - 1) / js;
  chunk = (chunk + c->tile - 1) / c->tile * c->tile;
  js = (p.n_total + chunk - 1) / chunk;
  c->j_split = js;
  c->j_chunk = chunk;
  // Small and mid-size systems (the reference ships N = 2000): a workgroup owns a few bodies and its lanes split the j range
  // (forces_block_pk_kernel; forces_block_kernel in the other two precisions) — one lane per body cannot fill the chip
  // there.  Only when the caller left the geometry to us.  c->wave: register pairs per workgroup (plain fp32), bodies
  // per workgroup (Kahan, fp64).
  c->wave = 0;
  const bool ours = p.algorithm != NBODY_ALGO_SYMMETRIC && p.tile == 0 && p.i_per_thread == 0 && p.j_split == 0;
  int cus = 256;
  { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, p.device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount; }
  if (ours && p.precision == NBODY_PREC_F32 && p.zero_mode != NBODY_ZERO_SELECT && p.n_total < block_max_n()) {
    c->wave = block_pairs(p.n_total, cus);
  } else if (ours && p.precision == NBODY_PREC_F32_KAHAN && p.zero_mode != NBODY_ZERO_SELECT &&
             p.n_total < env_int("NBODY_BLOCK_MAX_N_KAHAN", kSmallSystem)) {
    // above kSmallSystem the packed Kahan tile kernel is faster (whole steps: profiles/r03_block_kernel_other_precisions.txt)
    c->wave = block_bodies(p.n_total, cus);
  } else if (ours && p.precision == NBODY_PREC_F64 && p.n_total < env_int("NBODY_BLOCK_MAX_N_F64", kSmallSystem)) {
    c->wave = block_bodies(p.n_total, cus);
  }
  if (c->wave != 0) {
    c->j_split = 1;
    c->j_chunk = (p.n_total + c->tile - 1) / c->tile * c->tile;
  }
}

// Which plain fp32 systems take the even-share plan by default: whole steps without events, one box, both forms of the kernel,
// every bodies-per-lane choice under both plans (profiles/r05_even_share_vs_guided_by_n.txt; distinct / equal masses, best
// guided -> best even-share): N = 20480 95.2 / 88.7 us -> 92.3 / 86.4, 24576 128.2 / 119.3 -> 123.7 / 113.4, 32768 206.7 /
// 190.7 -> 199.6 / 182.3, 40960 302.9 / 281.1 -> 289.8 / 260.6, 65536 722.6 / 658.9 -> 703.3 / 630.7, 81920 1103.8 / 1014.1 ->
// 1081.2 / 967.3, 98304 1555.6 / 1421.7 -> 1552.3 / 1392.5; N = 18432 and 131072: nothing in it.
bool sym_even_default(int n_total, bool kahan) {
  // Kahan contexts (eight bodies per lane at most, two waves per SIMD): profiles/r05_even_share_vs_guided_by_n_kahan.txt — N = 12288
  // 51.1 / 49.3 us -> 49.8 / 47.5, 16384 77.8 / 75.0 -> 74.4 / 70.2, 24576 134.0 / 125.2 -> 127.6 / 116.1, 32768 212.8 / 196.1 ->
  // 210.8 / 189.2; from 49152 on the guided strips are ahead (427.8 / 391.8 against 439.0 / 394.6)
  if (kahan) return n_total >= env_int("NBODY_SYM_EVEN_KAHAN_MIN_N", 12288) && n_total < env_int("NBODY_SYM_EVEN_KAHAN_MAX_N", 40960);
  // (with the detector's sparse table, profiles/r05_even_share_vs_block_kernel_13k_to_19k.txt: N = 17408 73.0 / 67.7 -> 70.2 / 67.1
  // with four bodies per lane, 18432 78.4 / 74.6 -> 76.9 / 73.7, 19456 83.8 / 78.3 -> 83.2 / 79.7: everything the symmetric pass
  // runs below 106496 bodies (and, below, up to 139264).  The same table has even shares ahead of the block kernel from N = 15360 — 61.1 / 57.6 -> 57.4 / 54.8,
  // 16384 67.8 / 63.9 -> 64.4 / 61.2 —; the block kernel keeps those sizes for its one-launch step and what nbody_tick gets from it.)
  // Upper end, with TWO items per slot from 90112 bodies on — passes of 1.5 ms and more: slots of unequal speed drift apart —
  // (profiles/r05_even_share_rounds_at_larger_n.txt; guided -> one round -> two, distinct | equal masses): N = 98304 1596 -> 1590 ->
  // 1566 us | 1450 -> 1420 -> 1403, 114688 2073 -> 2063 -> 2037 | 1894 -> 1855 -> 1831, 131072 2650 -> 2664 -> 2645 | 2428 -> 2392 ->
  // 2381; from 147456 on nothing in it either way (3453 -> 3535 -> 3484 | 3160 -> 3185 -> 3118): even shares below 139264 bodies.
  return n_total >= env_int("NBODY_SYM_EVEN_MIN_N", 16385) && n_total < env_int("NBODY_SYM_EVEN_MAX_N", 139264);
}

// Symmetric algorithm: applicability, bodies per lane, and the work plan (sym_plan.h).  Everything here is a function
// of the parameters and of the device's CU count and total memory — never of what happens to be free — so that equal
// GPUs arrive at equal plans (the ranks of a sharded job must) and results are reproducible from box to box.
void choose_algorithm(nbody_ctx *c) {
  const nbody_params &p = c->p;
  c->sym = false;
  if (p.algorithm == NBODY_ALGO_TILED) return;
  if (p.zero_mode == NBODY_ZERO_SELECT) return;                         // compare+select lives in the one-sided kernel only
  // whole steps, one box, final kernels (profiles/r02_threshold_symmetric_vs_one_sided.txt): N = 10240 one-sided 0.0597 ms vs
  // symmetric 0.0597, N = 12288 0.0763 vs 0.0700, 14336 0.0934 vs 0.0841, 16384 (the one-sided geometry's best case) 0.0978
  // vs 0.0948, 18432 0.1287 vs 0.1024, 20480 0.147 vs 0.116; Kahan and fp64 likewise from 12288 (0.0735 vs 0.0639, 0.129 vs 0.119)
  // Round 3 (the fused update folds its two lists side by side; whole steps without events, same box: N = 8192 0.0266 ms
  // one-sided vs 0.0366 symmetric, 9216 0.0351 vs 0.0325, 10240 0.0423 vs 0.0393, 11264 0.0438 vs 0.0413; Kahan 9216 0.0326 vs
  // 0.0289, fp64 0.0774 vs 0.0658; distinct masses 0.0362 vs 0.0340): the symmetric pass from N = 9216
  if (p.algorithm == NBODY_ALGO_AUTO && c->wave != 0) return;            // the block kernels' one-launch step (choose_geometry)
  if (p.algorithm == NBODY_ALGO_AUTO && p.n_total < env_int("NBODY_SYM_MIN_N", 9216)) return;
  const bool f64 = p.precision == NBODY_PREC_F64, kahan = p.precision == NBODY_PREC_F32_KAHAN;
  if (f64 && !(p.eps > 0.0 || p.zero_mode == NBODY_ZERO_EXACT)) return;
  // bodies per lane.  fp32: 2 * register pairs; more of them amortise the travelling sums' dpp moves over more
  // arithmetic (tools/microbench6.hip) but make the i-set — the quantum of work — larger.  fp64: 2, or 4 at 2 waves/SIMD.
  // The even-share plan (sym_plan.h): plain fp32, one context owning all bodies, two register pairs per lane and more —
  // exactly one workgroup per slot, all of equal cost.  NBODY_SYM_EVEN = 0 / 1 forces the choice (A/B measurements, tests).
  int even_env = -1;
  if (const char *e = getenv("NBODY_SYM_EVEN")) { if (e[0] == '0' || e[0] == '1') even_env = e[0] - '0'; }
  // (not where a test forces pool phases on a small system: the phased pass is the guided plan's)
  const bool even_wanted = !f64 && p.i_count == p.n_total && env_int("NBODY_SYM_POOL_BUDGET_MB", 0) == 0 &&
                           (even_env == 1 || (even_env < 0 && sym_even_default(p.n_total, kahan)));
  int ipt = p.i_per_thread;
  if (f64) {
    if (ipt == 0) ipt = p.n_total >= 65536 ? 4 : 2;
    if (ipt != 2 && ipt != 4) return;
  } else {
    if (ipt == 0 && even_wanted) {
      // even shares have no quantum of work to keep small: sixteen bodies per lane (the fewest instructions per interaction)
      // from N = 24576, eight from 20480, four below (same table: N = 20480 93.0 / 86.4 us with eight, 94.3 / 86.5 with sixteen; 22528 107.5 /
      // 99.2 against 115.4 / 105.4; 24576 125.1 / 115.4 against 123.7 / 113.4; 32768 205.5 / 190.3 against 199.6 / 182.3)
      // (Kahan: four below 22528 — N = 20480 95.0 / 88.7 us with four, 96.6 / 91.0 with eight —, eight above)
      ipt = env_int("NBODY_SYM_IPT", kahan ? (p.n_total >= 22528 ? 8 : 4) : (p.n_total >= 24576 ? 16 : (p.n_total >= 20480 ? 8 : 4)));
    } else if (ipt == 0) {
      // measured on one box, sustained load (profiles/r02_sweep_symmetric_by_n.txt, r02_tune_mid_sizes.txt): sixteen bodies
      // per lane win wherever the symmetric pass runs (N = 32768: 0.206 vs 0.211 ms with eight, 65536: 0.691 vs 0.718,
      // 131072: 2.62 vs 2.70, 2^20: 162 vs 170.5 ms); the Kahan form has no sixteen (its running compensated sums double the
      // accumulators) and runs eight
      if (!kahan && p.n_total >= 40960) ipt = 16;      // whole step, N = 32768: 0.2353 ms with eight, 0.2408 with sixteen; 40960: 0.3355 / 0.3336
      else if (p.n_total >= 24576) ipt = 8;
      else if (p.n_total >= 17408) ipt = 4;
      else ipt = 2;                                    // N = 12288: 0.0700 ms with two, 0.0713 with four; 16384: 0.0948 / 0.0964; 18432: 0.1037 / 0.1024
      ipt = env_int("NBODY_SYM_IPT", ipt);
      if (kahan && ipt == 16) ipt = 8;
      // sharded slices must be whole i-sets
      while (ipt > 2 && p.i_count != p.n_total && p.i_count % (256 * ipt) != 0) ipt /= 2;
    }
    if (ipt != 2 && ipt != 4 && ipt != 8 && ipt != 16) return;
    if (ipt == 16 && kahan) return;
  }
  const int bi = 256 * ipt;
  // workgroups the chip holds at a time: one wave of each per SIMD -> (waves per SIMD) per CU
  int cus = 256;
  { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, p.device) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount; }
  const int np = ipt / 2;
  const int wps = f64 ? (ipt == 4 ? 2 : 4) : (np == 8 ? 2 : (np == 4 ? (kahan ? 2 : 3) : (kahan && np == 2 ? 3 : 4)));
  c->sym_slots = cus * wps;
  // A strip = 1/(K * slots) of the work still to hand out.  Large K = many short strips = best balance of the force pass but
  // one i-side segment (bodies-per-i-set x 16 B, written and read back) per strip; small K = a first round of long strips.
  // Long passes want K = 6 (N = 2^20, force pass: K = 3 170.1 ms — slots of unequal speed drift apart over 100 ms —, 6
  // 164.1, 24 163.3 at three times the segments; profiles/r02_tune_guided_k_n2p20.txt).  Short passes do not care about K
  // but their update pays for every segment (whole step, profiles/r02_tune_whole_step_mid_sizes.txt: N = 65536 K = 6
  // 0.781 ms, K <= 3 0.752; N = 131072 2.780 vs 2.653; N = 32768 0.2675 vs 0.2518).
  // What decides is how long the pass runs (fp64 at N = 262144 takes 25 ms and wants K = 6: 25.5 vs 26.9 ms with 3), so K
  // follows the expected duration of this context's share: >= 15 ms 6, >= 5 ms 3, >= 0.5 ms 1.5, below 1.
  {
    const double rate = f64 ? 2.7e12 : (kahan ? 6.0e12 : 6.6e12);                      // interactions per second, measured
    const double est_ms = (double)p.n_total * (double)p.i_count / rate * 1e3;
    c->sym_k = est_ms >= 15.0 ? 6.0 : (est_ms >= 5.0 ? 3.0 : (est_ms >= 0.5 ? 1.5 : 1.0));
  }
  if (const char *e = getenv("NBODY_SYM_K")) { const int v = atoi(e); if (v >= 1) c->sym_k = v; }                     // tuning only
  if (const char *e = getenv("NBODY_SYM_K_X10")) { const int v = atoi(e); if (v >= 5) c->sym_k = v / 10.0; }         // tuning only
  // shortest strip, in 64-body subtiles: whole 256-body tiles from N = 131072, half tiles below (same table)
  // (round 3, whole steps without events, four bodies per lane: N = 20480 0.0921 ms with two subtiles, 0.0888 with one; two
  // bodies per lane — N = 16384 — do not care: 0.0737 / 0.0738)
  c->sym_min_sub = env_int("NBODY_SYM_MIN_SUB", p.n_total >= 131072 ? 4 : (!f64 && ipt == 4 ? 1 : 2));
  nbody::SymPlan *plan = new (std::nothrow) nbody::SymPlan();
  if (!plan) return;
  std::string why;
  bool planned = false;
  const bool even = even_wanted && np >= 2;
  try {
    if (even) {
      const int rounds = env_int("NBODY_SYM_EVEN_ROUNDS", (!kahan && p.n_total >= 90112) ? 2 : 1);       // items per slot (sym_even_default)
      planned = nbody::build_sym_plan_even(p.n_total, bi, std::max(1, (int)((long long)c->sym_slots * rounds * env_int("NBODY_SYM_EVEN_ITEMS_PCT", 100) / 100)), plan, &why,
                                           env_int("NBODY_SYM_EVEN_COST_SYM", 82), env_int("NBODY_SYM_EVEN_COST_ONE", 74),
                                           env_int("NBODY_SYM_EVEN_COST_MOVE", 26), env_int("NBODY_SYM_EVEN_OWN_PCT", nbody::kSymEvenOwnPct));
      if (planned) { c->sym_k = 0.0; c->sym_min_sub = 0; }
    } else
    planned = nbody::build_sym_plan(p.n_total, p.i_begin, p.i_count, bi, c->sym_slots, c->sym_k, c->sym_min_sub, f64 ? 2 : 1, plan, &why,
                                    0, env_int("NBODY_SYM_MAX_SUB", 0));
  } catch (const std::bad_alloc &) {
    why = "out of host memory";
  }
  // The partial-sum pool must fit comfortably: at most a third of the card's TOTAL memory (and 2^32 elements).  Beyond that
  // (N = 2^23 on one 288 GB card: the j-side segments alone are 137 GB) the fp32 pass runs in PHASES that share one j-side
  // area, sized so that the whole pool stays within 32 GB (sym_plan.h); fp64 has no phased form and leaves to the
  // one-sided kernel.  NBODY_SYM_POOL_BUDGET_MB forces phases at any size (tests).
  size_t free_b = 0, total_b = 0;
  const int forced_mb = f64 ? 0 : env_int("NBODY_SYM_POOL_BUDGET_MB", 0);
  const bool too_big = !planned ? why.find("2^32") != std::string::npos
                                : (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0 &&
                                   (double)plan->pool_elems * (f64 ? 32.0 : 16.0) > (double)total_b / 3.0);
  if (!f64 && !even && (forced_mb > 0 || too_big)) {
    const double cap = 32.0 * 1073741824.0 / 16.0;                                    // elements
    double budget = forced_mb > 0 ? (double)forced_mb * 1048576.0 / 16.0 : 20.0 * 1073741824.0 / 16.0;
    for (int attempt = 0; attempt < 2; ++attempt) {
      try {
        planned = nbody::build_sym_plan(p.n_total, p.i_begin, p.i_count, bi, c->sym_slots, c->sym_k, c->sym_min_sub, 1, plan, &why,
                                        (uint64_t)budget);
      } catch (const std::bad_alloc &) { why = "out of host memory"; planned = false; }
      if (!planned || forced_mb > 0 || (double)plan->pool_elems <= cap) break;
      budget -= (double)plan->pool_elems - cap;                                       // the i-side segments took more than 12 GB
      if (budget < 2.0 * 1073741824.0 / 16.0) { planned = false; why = "the i-side segments leave no room for a shared j-side area"; break; }
    }
    if (planned && forced_mb == 0 && (double)plan->pool_elems > cap) { planned = false; why = "the partial-sum pool would exceed 32 GB even in phases"; }
  } else if (planned && too_big) {
    planned = false; why = "the partial-sum pool would exceed a third of the device memory";
  }
  if (!planned) {
    g_create_error = "symmetric plan: " + why;
    delete plan;
    return;
  }
  c->plan = plan;
  c->sym_bi = bi; c->sym_np = f64 ? ipt / 2 : np; c->sym_pad = plan->n_pad; c->sym_items_n = (int)plan->items.size();
  c->sym_nsrc = plan->n_src; c->sym_pool_elems = (size_t)plan->pool_elems; c->sym_n_local = plan->n_local;
  c->sym_phase_item0 = plan->phase_item0; c->sym_n_gran = plan->n_gran;
  c->sym_even = plan->even;
  c->sym = true;
  c->wave = 0;            // the small-system one-launch step belongs to the one-sided path
}

// Fused stepping (update_sym_fused_kernel) is for fp32 symmetric contexts that own all bodies AND their position buffer:
// then nothing but this library's kernels moves a body, and the update can prepare the next pass.
bool sym_fused(const nbody_ctx *c) {
  static const bool off = [] { const char *e = getenv("NBODY_SYM_NO_FUSE"); return e && e[0] == '1'; }();   // A/B measurements only
  return !off && c->sym && c->p.precision != NBODY_PREC_F64 && c->sym_nsrc == 1 && c->own_posm && !c->posm_escaped &&
         c->sym_phase_item0.size() == 2 &&
         (c->sym_dup_table == nullptr || c->sym_dup_table2 != nullptr);
}

nbody::SymLaunch make_sym_launch(const nbody_ctx *c) {
  nbody::SymLaunch L;
  L.posm = c->posm; L.posg = c->sym_posg; L.pool = c->sym_pool; L.items = c->sym_items; L.n_items = c->sym_items_n;
  L.i_ptr = c->sym_iptr; L.i_off = c->sym_ioff; L.j_ptr = c->sym_jptr; L.j_off = c->sym_joff;
  L.send = c->sym_send; L.recv = c->sym_recv;
  L.n_total = c->p.n_total; L.n_pad = c->sym_pad; L.n_src = c->sym_nsrc;
  L.np = c->sym_np;
  L.even = c->sym_even ? 1 : 0; L.wrap = c->sym_n_gran * 64;
  L.precision = c->p.precision == NBODY_PREC_F64 ? NBODY_PREC_F64 : NBODY_PREC_F32;
  L.kahan = c->p.precision == NBODY_PREC_F32_KAHAN ? 1 : 0;
  L.G = c->p.G; L.eps2 = c->p.eps * c->p.eps;
  if (L.eps2 == 0.0 && c->p.zero_mode == NBODY_ZERO_FLOOR && c->floor_eps2 > 0.0) L.eps2 = c->floor_eps2;
  L.dup_table = c->sym_dup_table; L.dup_slots = c->sym_dup_slots;
  L.general = c->sym_general;
  L.n_local = c->sym_n_local; L.own_begin = c->p.i_begin; L.own_count = c->p.i_count;
  // the host's own finding is final while nothing but this library writes the position buffer; otherwise "not equal"
  // still is (the device word is sticky), "equal" is only the state of things at the last upload
  // (fp64 always asks the device: its test also looks for bodies out where the padding is)
  L.uni_host = c->masses_equal == 0 ? 0 : ((c->masses_equal == 1 && c->own_posm && !c->posm_escaped &&
                                            c->p.precision != NBODY_PREC_F64) ? 1 : -1);
  if (sym_fused(c)) {
    L.fused = 1;
    L.skip_prep = c->sym_posg_valid ? 1 : 0;
    if (c->sym_dup_table) {
      L.dup_table = c->sym_dup_cur ? c->sym_dup_table2 : c->sym_dup_table;
      L.dup_table_next = c->sym_dup_cur ? c->sym_dup_table : c->sym_dup_table2;
    }
  }
  L.clk = c->clk;
  return L;
}

nbody::ForceLaunch make_launch(const nbody_ctx *c) {
  nbody::ForceLaunch L;
  L.posm = c->posm; L.accp = c->accp;
  L.n_total = c->p.n_total; L.i_begin = c->p.i_begin; L.i_count = c->p.i_count;
  L.tile = c->tile; L.ipt = c->ipt; L.j_split = c->j_split; L.j_chunk = c->j_chunk;
  L.G = c->p.G; L.eps2 = c->p.eps * c->p.eps; L.precision = c->p.precision;
  L.zero_mode = (c->p.zero_mode == NBODY_ZERO_SELECT) ? 2 : 1;
  L.wave = c->wave;
  L.guarded = env_int("NBODY_SYM_GUARDED", 0) == 1 ? 1 : 0;        // A/B measurements and tests: no bare pair law anywhere
  L.dup_table = c->sym_dup_table; L.dup_slots = c->sym_dup_slots;
  if (L.eps2 == 0.0 && c->p.zero_mode == NBODY_ZERO_FLOOR && c->floor_eps2 > 0.0) L.eps2 = c->floor_eps2;
  // equal-mass form of the packed one-sided kernel: the host's scan of the uploaded state stands while nothing else writes
  // the position buffer ("not equal" always stands: the device word is sticky); otherwise the device looks before the launch
  L.general = c->sym_general;
  L.check_masses = (c->sym_general && c->masses_equal != 0 && !(c->masses_equal == 1 && c->own_posm && !c->posm_escaped)) ? 1 : 0;
  // block kernel: which form to launch — the host's finding if it stands, both (each looks at the device word) otherwise
  L.uni = (!c->sym_general || c->masses_equal == 0) ? 0 : (L.check_masses ? -1 : 1);
  L.clk = c->clk;
  return L;
}

int timer_begin(nbody_ctx *c, int which, EventPair *ev) {
  KernelTimer &t = c->timers[which];
  if (t.pool.empty()) {
    EventPair e;
    HIP_TRY(c, hipEventCreate(&e.a));
    HIP_TRY(c, hipEventCreate(&e.b));
    t.pool.push_back(e);
  }
  *ev = t.pool.back();
  t.pool.pop_back();
  HIP_TRY(c, hipEventRecord(ev->a, c->stream));
  return NBODY_OK;
}

int timer_end(nbody_ctx *c, int which, const EventPair &ev, bool counts = true) {
  HIP_TRY(c, hipEventRecord(ev.b, c->stream));
  c->timers[which].pending.push_back(ev);
  c->timers[which].pending.back().counts = counts;
  return NBODY_OK;
}

int timer_drain(nbody_ctx *c, int which) {
  KernelTimer &t = c->timers[which];
  for (const EventPair &e : t.pending) {
    HIP_TRY(c, hipEventSynchronize(e.b));
    float ms = 0.f;
    HIP_TRY(c, hipEventElapsedTime(&ms, e.a, e.b));
    t.total_ms += ms;
    t.launches += e.counts ? 1 : 0;
    t.pool.push_back(e);
  }
  t.pending.clear();
  return NBODY_OK;
}

// NBODY_ZERO_FLOOR: the smallest eps^2 for which G*m_max*(eps^2)^(-3/2) stays below FLT_MAX/8.
int ensure_floor(nbody_ctx *c) {
  if (c->p.zero_mode != NBODY_ZERO_FLOOR || c->p.eps > 0.0 || c->floor_eps2 > 0.0) return NBODY_OK;
  HIP_TRY(c, hipMemsetAsync(c->scratch, 0, 4, c->stream));
  HIP_TRY(c, nbody::launch_massmax(c->p.precision, c->posm, c->p.n_total, (unsigned int *)c->scratch, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->h_scratch, c->scratch, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  float mmax;
  memcpy(&mmax, c->h_scratch, 4);
  const double gm = std::fabs(c->p.G) * (double)mmax;
  const double lim = (c->p.precision == NBODY_PREC_F64) ? 1e300 : 3.0e38 / 8.0;
  double f = gm > 0 ? std::pow(gm / lim, 2.0 / 3.0) : 0.0;
  const double tiny = (c->p.precision == NBODY_PREC_F64) ? 1e-280 : 1e-30;
  c->floor_eps2 = f > tiny ? f : tiny;
  return NBODY_OK;
}

// Barnes-Hut state of a context, on first use.
int ensure_bh(nbody_ctx *c) {
  if (c->p.precision != NBODY_PREC_F32)
    return fail(c, NBODY_ERR_UNSUPPORTED, "theta > 0 (Barnes-Hut) needs an fp32 context");
  if (c->bh) return NBODY_OK;
  // a context that owns a slice builds the whole tree from the replicated positions and walks its own bodies (bh_common.h, WalkSlice)
  hipError_t e = nbody::bh_create(&c->bh, c->p.n_total, c->p.i_begin, c->p.i_count);
  if (e == hipSuccess) e = hipMalloc(&c->bh_acc, (size_t)c->p.i_count * 16);
  if (e != hipSuccess) {
    nbody::bh_destroy(c->bh); c->bh = nullptr;
    if (c->bh_acc) { (void)hipFree(c->bh_acc); c->bh_acc = nullptr; }
    return fail(c, NBODY_ERR_HIP, "bh_create: %s", hipGetErrorString(e));
  }
  nbody::bh_set_div_mode(c->bh, c->p.bh_div_mode);
  if (c->posm_escaped) nbody::bh_positions_external(c->bh);
  return NBODY_OK;
}

int bh_status_error(nbody_ctx *c, int status) {
  if (status == 1) return fail(c, NBODY_ERR_UNSUPPORTED, "Barnes-Hut tree deeper than 42 levels: two bodies closer than Size/2^42 (the reference's Add would recurse without bound on coincident bodies)");
  if (status == 2) return fail(c, NBODY_ERR_NOMEM, "Barnes-Hut node pool exhausted");
  if (status == 4) return fail(c, NBODY_ERR_STATE, "Barnes-Hut: the sorted path keys are out of order (an internal error of this library; the frame was not built and the state is what it was)");
  return NBODY_OK;
}

// theta > 0: queue `nsteps` whole frames (tree, walk, update), nothing waits.
int bh_enqueue(nbody_ctx *c, float dt, int nsteps, float *stage = nullptr) {
  const bool timed = c->p.time_kernels != 0;
  for (int s = 0; s < nsteps; ++s) {
    EventPair ev;
    if (timed) { int rc = timer_begin(c, NBODY_KERNEL_FORCES, &ev); if (rc) return rc; }
    HIP_TRY(c, nbody::bh_frame(c->bh, c->posm, c->vel, c->acc, c->theta, c->p.G, dt, 0, s == nsteps - 1 ? stage : nullptr, c->stream));
    if (timed) { int rc = timer_end(c, NBODY_KERNEL_FORCES, ev); if (rc) return rc; }
    // (small systems queue a whole call's frames at once and give none up: bound the number of live events; the larger systems'
    // batches of 64 never get here, so the pairs of a given-up batch are still there to be taken back)
    if (timed && nbody::bh_is_small(c->bh) && c->timers[NBODY_KERNEL_FORCES].pending.size() >= 1024) { int rc = timer_drain(c, NBODY_KERNEL_FORCES); if (rc) return rc; }
  }
  c->bh_batch.dt = dt; c->bh_batch.stage = stage; c->bh_batch.queued += nsteps; c->bh_batch.timed = timed;
  c->sym_posg_valid = false;     // bodies move without the fused all-pairs update's preparation of the next pass
  return NBODY_OK;
}

// The event pairs of the last `k` frames queued belong to frames that did nothing (the warm sort gave one up and the rest were queued
// behind it): they are taken back, so that nbody_kernel_time counts every frame once — with the events around the run that built it.
void timer_take_back(nbody_ctx *c, int which, int k) {
  KernelTimer &t = c->timers[which];
  for (; k > 0 && !t.pending.empty(); --k) { t.pool.push_back(t.pending.back()); t.pending.pop_back(); }
}

// ... and the one wait: the frames that were built count as steps; a refused frame (and all queued behind it) left the
// state where it was.  Frames the sort from the previous order gave up (kernels_bh_sort.hip: a bucket ran over — the records were
// replaced, the root box jumped) did nothing, nor did the frames queued behind them: they are queued again here, the first of them
// with the cold sorts, inside event pairs of their own.
int bh_finish(nbody_ctx *c) {
  for (;;) {
    int status = 0, frames = 0;
    HIP_TRY(c, nbody::bh_collect(c->bh, c->stream, &status, &frames));
    c->steps_done += frames;
    const int left = c->bh_batch.queued - frames;
    c->bh_batch.queued = 0;
    if (status != 3) {
      if (c->bh_batch.timed && c->timers[NBODY_KERNEL_FORCES].pending.size() >= 1024) { int rc = timer_drain(c, NBODY_KERNEL_FORCES); if (rc) return rc; }
      return bh_status_error(c, status);
    }
    if (c->bh_batch.timed) timer_take_back(c, NBODY_KERNEL_FORCES, left);
    if (int rc = bh_enqueue(c, c->bh_batch.dt, left, c->bh_batch.stage)) return rc;
  }
}

}  // namespace

// ---- for multi.hip (declared in multi.h, not part of the C-ABI): one device's share of a theta > 0 step, in two halves, so that one
// caller thread can queue every device's frames — and the all-gathers between them — before it waits for any of them.
namespace nbody {
int part_bh_queue_frame(nbody_ctx *c, float dt, bool diagnostic) {
  if (int rc = use_device(c)) return rc;
  if (int rc = ensure_bh(c)) return rc;
  if (!(dt > 0.0f)) {                                           // accelerations only (nbody_compute_forces): walk into bh_acc, then the row fold
    if (int rc = queue_forces_bh(c, diagnostic)) return rc;
    return run_update(c, 0.0f);
  }
  return bh_enqueue(c, dt, 1);
}
// status: 0 all frames built; 1 / 2 refused (the error text is the context's); 3 the warm sort gave a frame up — the frames from
// number *built on did nothing on this device (nor, the build being the same everywhere, on any other) and are the caller's to queue again
int part_bh_collect(nbody_ctx *c, int *status, int *built) {
  if (int rc = use_device(c)) return rc;
  int st = 0, frames = 0;
  HIP_TRY(c, nbody::bh_collect(c->bh, c->stream, &st, &frames));
  *status = st; *built = frames;
  if (c->bh_batch.queued > 0) c->steps_done += frames;         // (whole frames, not a diagnostic force pass)
  if (st == 3 && c->bh_batch.timed) timer_take_back(c, NBODY_KERNEL_FORCES, c->bh_batch.queued - frames);
  c->bh_batch.queued = 0;
  if (st == 1 || st == 2 || st == 4) return bh_status_error(c, st);
  return NBODY_OK;
}
// the next tree's root centre (the previous tree's CoM, OctreeSearch.cpp:77-79) of a context that has built a tree: what a checkpoint keeps
int part_bh_root(nbody_ctx *c, float out[3], int *has_root) {
  *has_root = 0;
  if (!c->bh) return NBODY_OK;
  if (int rc = use_device(c)) return rc;
  HIP_TRY(c, nbody::bh_get_root_com(c->bh, out, c->stream));
  *has_root = 1;
  return NBODY_OK;
}
}  // namespace nbody

namespace {

// Barnes-Hut force pass alone: ComputeCubeSize -> CreateOctree -> the walk, accelerations into bh_acc (run_update adds them up
// and, for nbody_step_end, moves the bodies).  diagnostic: the pass belongs to no frame (nbody_compute_forces) and leaves the
// next tree's root centre alone.
int queue_forces_bh(nbody_ctx *c, bool diagnostic) {
  { int rc = ensure_bh(c); if (rc) return rc; }
  EventPair ev;
  const bool timed = c->p.time_kernels != 0;
  if (timed) { int rc = timer_begin(c, NBODY_KERNEL_FORCES, &ev); if (rc) return rc; }
  HIP_TRY(c, nbody::bh_frame(c->bh, c->posm, nullptr, c->bh_acc, c->theta, c->p.G, 0.0f, diagnostic ? 1 : 0, nullptr, c->stream));
  if (timed) { int rc = timer_end(c, NBODY_KERNEL_FORCES, ev); if (rc) return rc; }
  return NBODY_OK;
}
int run_forces_bh(nbody_ctx *c, bool diagnostic) {
  for (;;) {
    { int rc = queue_forces_bh(c, diagnostic); if (rc) return rc; }
    int status = 0;
    HIP_TRY(c, nbody::bh_collect(c->bh, c->stream, &status, nullptr));
    if (status != 3) return bh_status_error(c, status);
    if (c->p.time_kernels) timer_take_back(c, NBODY_KERNEL_FORCES, 1);   // given up by the warm sort: once more, with the cold sorts
  }
}

// phase (SymLaunch::phase): 0 the whole pass; 1 / 2 the two goes of a sharded fp32 symmetric context (sym_two_goes)
int run_forces(nbody_ctx *c, bool diagnostic = false, int phase = 0) {
  if (c->theta > 0.0f) return run_forces_bh(c, diagnostic);
  { int rc = ensure_floor(c); if (rc) return rc; }
  EventPair ev;
  const bool timed = c->p.time_kernels != 0;
  if (timed) { int rc = timer_begin(c, NBODY_KERNEL_FORCES, &ev); if (rc) return rc; }
  if (c->sym) {
    nbody::SymLaunch L = make_sym_launch(c);
    L.phase = phase;
    // a fused context about to run the preparation kernel again (new state): its current table may hold the entries
    // the last update left for positions that are gone
    if (L.fused && !L.skip_prep && L.dup_table && L.eps2 == 0.0)
      HIP_TRY(c, hipMemsetAsync(L.dup_table, 0, (size_t)L.dup_slots * 8 + 64, c->stream));
    if (c->p.precision == NBODY_PREC_F64) {
      HIP_TRY(c, nbody::launch_forces_sym(L, c->stream));          // the fp64 launcher runs its whole pass
    } else {
      // this go's items (phase: 0 all, 1 the strips inside the own slice, 2 the others), pool phase by pool phase: a pool
      // phase's j-side sums are folded into `send` as soon as its last item has been launched, and the next one reuses
      // the area (sym_plan.h).  One pool phase and phase 0: a single call, as ever.
      const int r0 = phase == 2 ? c->sym_n_local : 0, r1 = phase == 1 ? c->sym_n_local : c->sym_items_n;
      const int n_ph = (int)c->sym_phase_item0.size() - 1;
      bool first = true;
      for (int q = 0; q < n_ph; ++q) {
        const int a = std::max(c->sym_phase_item0[(size_t)q], r0), b = std::min(c->sym_phase_item0[(size_t)q + 1], r1);
        if (a >= b && !(first && q == n_ph - 1)) continue;           // nothing of this pool phase in this go (but every go prepares)
        L.item0 = a < b ? a : r0; L.item1 = a < b ? b : r0;
        L.do_prep = first ? 1 : 0;
        // a go with nothing to launch (a plan without remote strips: every item — and with its last item every pool phase's
        // fold — went out in the first go) prepares and folds NOTHING again: a second fold of the last pool phase would add its
        // j-side sums to `send` twice.  What the go's preparation entered into the coincident-body table is cleared by hand.
        const bool folded_in_first_go = phase == 2 && c->sym_n_local >= c->sym_phase_item0[(size_t)q + 1];
        L.do_fold = (a < b && b == c->sym_phase_item0[(size_t)q + 1]) || (phase != 1 && q == n_ph - 1 && a >= b && !folded_in_first_go) ? 1 : 0;
        L.fold_accumulate = q > 0 ? 1 : 0;
        L.clear_detector = q == n_ph - 1 ? 1 : 0;
        L.j_ptr = (const unsigned int *)c->sym_jptr + (size_t)q * ((size_t)c->sym_n_gran + 1);
        HIP_TRY(c, nbody::launch_forces_sym(L, c->stream));
        if (a >= b && folded_in_first_go && L.dup_table && L.eps2 == 0.0)
          HIP_TRY(c, hipMemsetAsync(L.dup_table, 0, (size_t)L.dup_slots * 8 + 64, c->stream));
        first = false;
      }
    }
  } else {
    HIP_TRY(c, nbody::launch_forces(make_launch(c), c->stream));
  }
  if (timed) { int rc = timer_end(c, NBODY_KERNEL_FORCES, ev, phase != 1); if (rc) return rc; }   // two goes are ONE pass
  // bound the number of live events on long untimed-drain runs
  if (timed && c->timers[NBODY_KERNEL_FORCES].pending.size() >= 1024) return timer_drain(c, NBODY_KERNEL_FORCES);
  return NBODY_OK;
}

int run_update(nbody_ctx *c, float dt) {
  EventPair ev;
  const bool timed = c->p.time_kernels != 0;
  if (timed) { int rc = timer_begin(c, NBODY_KERNEL_UPDATE, &ev); if (rc) return rc; }
  if (c->theta > 0.0f) {
    HIP_TRY(c, nbody::launch_update(c->p.precision, c->posm, c->vel, c->acc, c->bh_acc, c->p.i_begin, c->p.i_count, 1, dt, c->stream));
    // bodies moved without the fused update's preparation of the next all-pairs pass: posg and the detector table are
    // those of older positions (the next theta == 0 pass runs the preparation kernel again)
    if (dt > 0.0f) c->sym_posg_valid = false;
    if (dt > 0.0f && c->bh) nbody::bh_positions_changed(c->bh);   // ... and the next Barnes-Hut frame looks at the positions for its Size
  } else if (c->sym) {
    const nbody::SymLaunch L = make_sym_launch(c);
    HIP_TRY(c, nbody::launch_update_sym(L, c->posm, c->vel, c->acc, c->p.i_begin, c->p.i_count, dt, c->stream));
    if (L.fused) { c->sym_posg_valid = true; c->sym_dup_cur ^= 1; }   // the update wrote posg and the other table
  }
  else
    HIP_TRY(c, nbody::launch_update(c->p.precision, c->posm, c->vel, c->acc, c->accp, c->p.i_begin, c->p.i_count,
                                    c->j_split, dt, c->stream));
  if (timed) { int rc = timer_end(c, NBODY_KERNEL_UPDATE, ev); if (rc) return rc; }
  if (timed && c->timers[NBODY_KERNEL_UPDATE].pending.size() >= 1024) return timer_drain(c, NBODY_KERNEL_UPDATE);
  return NBODY_OK;
}

template <typename SRC, typename DST>
void convert4(const SRC *src, DST *dst, size_t n_elems4, bool zero_w) {
  for (size_t i = 0; i < n_elems4; ++i) {
    dst[4 * i + 0] = (DST)src[4 * i + 0];
    dst[4 * i + 1] = (DST)src[4 * i + 1];
    dst[4 * i + 2] = (DST)src[4 * i + 2];
    dst[4 * i + 3] = zero_w ? (DST)0 : (DST)src[4 * i + 3];
  }
}

// A new state arrives from the host: are all masses (as the fp32 kernels will see them) equal?  The device word the
// equal-mass kernels are gated on is reset to that finding.
template <typename T>
int note_masses(nbody_ctx *c, const T *posm4) {
  if (!c->sym_general) return NBODY_OK;
  const bool ctx64 = c->p.precision == NBODY_PREC_F64;
  auto seen = [&](T v) { return ctx64 ? (double)v : (double)(float)v; };
  const double m0 = seen(posm4[3]);
  bool equal = true;
  for (size_t i = 1; i < (size_t)c->p.n_total && equal; ++i) equal = seen(posm4[4 * i + 3]) == m0;
  c->masses_equal = equal ? 1 : 0;
  HIP_TRY(c, hipMemsetAsync(c->sym_general, equal ? 0 : 0xFF, 4, c->stream));
  return NBODY_OK;
}

// Upload host SoA state given as T (float or double); converts to the context's precision.
// keep_history: the records of a running simulation edited by the host (nbody_push_particles) — the step count and the
// Barnes-Hut root centre (the previous tree's CoM) stay what they are.
template <typename T>
int upload_soa(nbody_ctx *c, const T *posm4, const T *vel4, bool keep_history = false) {
  if (int rc = use_device(c)) return rc;
  const int n = c->p.n_total, ib = c->p.i_begin, ic = c->p.i_count;
  const bool ctx64 = c->p.precision == NBODY_PREC_F64;
  const bool same = ctx64 == (sizeof(T) == 8);
  if (same) {
    HIP_TRY(c, hipMemcpyAsync(c->posm, posm4, (size_t)n * c->elem, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->vel, vel4 + 4 * (size_t)ib, (size_t)ic * c->elem, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  } else if (ctx64) {
    std::vector<double> tp((size_t)n * 4), tv((size_t)ic * 4);
    convert4(posm4, tp.data(), (size_t)n, false);
    convert4(vel4 + 4 * (size_t)ib, tv.data(), (size_t)ic, true);
    HIP_TRY(c, hipMemcpy(c->posm, tp.data(), tp.size() * 8, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->vel, tv.data(), tv.size() * 8, hipMemcpyHostToDevice));
  } else {
    std::vector<float> tp((size_t)n * 4), tv((size_t)ic * 4);
    convert4(posm4, tp.data(), (size_t)n, false);
    convert4(vel4 + 4 * (size_t)ib, tv.data(), (size_t)ic, true);
    HIP_TRY(c, hipMemcpy(c->posm, tp.data(), tp.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(c->vel, tv.data(), tv.size() * 4, hipMemcpyHostToDevice));
  }
  HIP_TRY(c, hipMemsetAsync(c->acc, 0, (size_t)ic * c->elem, c->stream));
  { const int rc = note_masses(c, posm4); if (rc) return rc; }
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->have_state = true;
  c->floor_eps2 = -1.0;
  c->sym_posg_valid = false;
  if (c->bh) nbody::bh_positions_changed(c->bh);                   // the next Barnes-Hut frame looks at the positions for its Size
  if (keep_history) return NBODY_OK;
  c->steps_done = 0;
  if (c->bh) HIP_TRY(c, nbody::bh_reset_root(c->bh, c->stream));   // a new scene: root centre starts at zero again
  return NBODY_OK;
}

// Download `count` elements starting at element `first` of a device buffer as T x 4.
template <typename T>
int download4(nbody_ctx *c, const void *dev, size_t first, size_t count, T *out) {
  const bool ctx64 = c->p.precision == NBODY_PREC_F64;
  const char *src = (const char *)dev + first * c->elem;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (ctx64 == (sizeof(T) == 8)) {
    HIP_TRY(c, hipMemcpy(out, src, count * c->elem, hipMemcpyDeviceToHost));
  } else if (ctx64) {
    std::vector<double> tmp(count * 4);
    HIP_TRY(c, hipMemcpy(tmp.data(), src, count * 32, hipMemcpyDeviceToHost));
    convert4(tmp.data(), out, count, false);
  } else {
    std::vector<float> tmp(count * 4);
    HIP_TRY(c, hipMemcpy(tmp.data(), src, count * 16, hipMemcpyDeviceToHost));
    convert4(tmp.data(), out, count, false);
  }
  return NBODY_OK;
}

// (Re)allocate the hand-off staging pair for `bytes`.
int ensure_stage(nbody_ctx *c, size_t bytes) {
  if (bytes <= c->stage_bytes) return NBODY_OK;
  if (c->d_stage) (void)hipFree(c->d_stage);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  c->d_stage = c->h_stage = nullptr; c->stage_bytes = 0;
  HIP_TRY(c, hipMalloc(&c->d_stage, bytes));
  HIP_TRY(c, hipHostMalloc(&c->h_stage, bytes, hipHostMallocDefault));
  c->stage_bytes = bytes;
  return NBODY_OK;
}

// Is [dst, dst + bytes) inside memory the caller pinned for this context?
bool in_pinned(const nbody_ctx *c, const void *dst, size_t bytes) {
  const char *d = (const char *)dst;
  for (const auto &r : c->pinned)
    if (d >= r.first && d + bytes <= r.first + r.second) return true;
  return false;
}

// The staged, packed FParticle records -> the caller's array (records `stride` bytes apart).
void unstage_particles(const nbody_ctx *c, void *aos, size_t stride, size_t count) {
  if (stride == sizeof(nbody_particle)) {
    memcpy(aos, c->h_stage, count * sizeof(nbody_particle));
    return;
  }
  char *base = (char *)aos;
  const char *src = (const char *)c->h_stage;
  for (size_t i = 0; i < count; ++i) memcpy(base + i * stride, src + i * sizeof(nbody_particle), sizeof(nbody_particle));
}

// The position buffer becomes visible to (or is replaced by) the caller: bodies may move behind the library's back from
// now on, so fused stepping and the buffer-swapping one-launch step end here, for the life of the context.  The detector
// table the two-kernel path uses may still hold what the last fused update left in it.
int posm_escapes(nbody_ctx *c) {
  if (c->posm_escaped) return NBODY_OK;
  if (c->sym_dup_table2)
    HIP_TRY(c, hipMemsetAsync(c->sym_dup_table, 0, (size_t)c->sym_dup_slots * 8 + 64, c->stream));
  c->posm_escaped = true;
  c->sym_posg_valid = false;
  if (c->bh) nbody::bh_positions_external(c->bh);
  return NBODY_OK;
}

int check_ready(nbody_ctx *c) {
  if (!c) return NBODY_ERR_INVALID;
  if (!c->have_state) return fail(c, NBODY_ERR_STATE, "no particles set (call nbody_set_particles / nbody_set_state_soa first)");
  return use_device(c);
}

}  // namespace

extern "C" {

int nbody_version(void) { return NBODY_VERSION_MAJOR * 100 + NBODY_VERSION_MINOR; }

int nbody_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int nbody_default_params(nbody_params *p) {
  if (!p) return NBODY_ERR_INVALID;
  memset(p, 0, sizeof *p);
  p->struct_size = (uint32_t)sizeof(nbody_params);
  p->precision = NBODY_PREC_F32;
  p->G = 1.0e4;      // OctreeSearch.h:104
  p->eps = 0.0;      // OctreeSearch.h:101-104: no softening
  return NBODY_OK;
}

const char *nbody_last_error(const nbody_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int nbody_create(const nbody_params *pin, nbody_ctx **out) try {
  if (!pin || !out) return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: null argument");
  *out = nullptr;
  if (pin->struct_size != sizeof(nbody_params))
    return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: struct_size %u != %zu", pin->struct_size, sizeof(nbody_params));
  nbody_params p = *pin;
  if (p.n_total <= 0) return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: n_total must be > 0");
  if (p.i_begin < 0 || p.i_begin >= p.n_total) return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: i_begin out of range");
  if (p.i_count == 0) p.i_count = p.n_total - p.i_begin;
  if (p.i_count < 0 || p.i_begin + p.i_count > p.n_total)
    return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: owned range [%d,%d) exceeds n_total %d", p.i_begin,
                p.i_begin + p.i_count, p.n_total);
  if (p.precision != NBODY_PREC_F32 && p.precision != NBODY_PREC_F32_KAHAN && p.precision != NBODY_PREC_F64)
    return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: unknown precision %d", p.precision);
  if (!(p.eps >= 0.0) || !std::isfinite(p.G)) return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: bad G/eps");
  if (!(p.theta >= 0.0f)) return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: theta must be >= 0");
  if (p.tile != 0 && p.tile != 64 && p.tile != 128 && p.tile != 256 && p.tile != 512)
    return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: tile must be 64, 128, 256 or 512");
  if (p.i_per_thread != 0 && p.i_per_thread != 1 && p.i_per_thread != 2 && p.i_per_thread != 4 && p.i_per_thread != 8 &&
      p.i_per_thread != 16)
    return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: i_per_thread must be 1, 2, 4, 8 or 16");
  if (p.i_per_thread == 16 && (p.algorithm == NBODY_ALGO_TILED || p.precision != NBODY_PREC_F32))
    return fail(nullptr, NBODY_ERR_UNSUPPORTED, "nbody_create: i_per_thread 16 exists for the plain fp32 symmetric kernel only");
  if (p.i_per_thread == 8 && (p.algorithm == NBODY_ALGO_TILED || p.precision == NBODY_PREC_F64))
    return fail(nullptr, NBODY_ERR_UNSUPPORTED, "nbody_create: i_per_thread 8 exists for the fp32 symmetric kernels only");
  if (p.j_split < 0) return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: j_split must be >= 0");
  if (p.zero_mode < 0 || p.zero_mode > NBODY_ZERO_FLOOR) return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: unknown zero_mode %d", p.zero_mode);
  if (p.algorithm < 0 || p.algorithm > NBODY_ALGO_SYMMETRIC) return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: unknown algorithm %d", p.algorithm);
  if (p.bh_div_mode != 0 && p.bh_div_mode != 1) return fail(nullptr, NBODY_ERR_INVALID, "nbody_create: bh_div_mode must be 0 or 1");

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return fail(nullptr, NBODY_ERR_NO_DEVICE, "nbody_create: no HIP device (%s); this engine has no CPU path",
                e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  if (p.device < 0 || p.device >= ndev)
    return fail(nullptr, NBODY_ERR_NO_DEVICE, "nbody_create: device %d not in [0,%d)", p.device, ndev);
  if ((e = hipSetDevice(p.device)) != hipSuccess)      // before anything that asks the device questions (free memory)
    return fail(nullptr, NBODY_ERR_HIP, "nbody_create: hipSetDevice(%d): %s", p.device, hipGetErrorString(e));

  nbody_ctx *c = new (std::nothrow) nbody_ctx();
  if (!c) return fail(nullptr, NBODY_ERR_NOMEM, "nbody_create: out of host memory");
  c->p = p;
  c->theta = p.theta;
  c->elem = (p.precision == NBODY_PREC_F64) ? 32 : 16;
  choose_geometry(c);
  choose_algorithm(c);
  if ((p.i_per_thread == 8 || p.i_per_thread == 16) && !(c->sym && c->sym_bi == 256 * p.i_per_thread)) {
    delete c;
    return fail(nullptr, NBODY_ERR_UNSUPPORTED,
                "nbody_create: i_per_thread %d needs the fp32 symmetric kernel (N >= 9216 or NBODY_ALGO_SYMMETRIC; "
                "sharded slices in multiples of %d bodies)", p.i_per_thread, 256 * p.i_per_thread);
  }
  if (p.algorithm == NBODY_ALGO_SYMMETRIC && !c->sym) {
    const std::string why = g_create_error;
    delete c;
    return fail(nullptr, NBODY_ERR_UNSUPPORTED,
                "nbody_create: NBODY_ALGO_SYMMETRIC needs fp32 (i_per_thread 2, 4, 8 or 16, zero_mode != SELECT) or "
                "fp64 (i_per_thread 2 or 4) and, when sharded, equal slices that are a multiple of 256*i_per_thread bodies%s%s",
                why.empty() ? "" : " — ", why.c_str());
  }

  auto bail = [&](hipError_t he, const char *what) {
    fail(nullptr, NBODY_ERR_HIP, "nbody_create: %s: %s", what, hipGetErrorString(he));
    nbody_destroy(c);
    return NBODY_ERR_HIP;
  };
  if ((e = hipSetDevice(p.device)) != hipSuccess) return bail(e, "hipSetDevice");
  if ((e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
  c->stream = c->own_stream;
  if ((e = hipMalloc(&c->posm, (size_t)p.n_total * c->elem)) != hipSuccess) return bail(e, "hipMalloc posm");
  c->own_posm = true;
  if ((e = hipMalloc(&c->vel, (size_t)p.i_count * c->elem)) != hipSuccess) return bail(e, "hipMalloc vel");
  c->own_vel = true;
  if ((e = hipMalloc(&c->acc, (size_t)p.i_count * c->elem)) != hipSuccess) return bail(e, "hipMalloc acc");
  c->own_acc = true;
  if (c->sym) {
    const nbody::SymPlan &P = *c->plan;
    auto up = [&](void **dst, const void *src, size_t bytes, const char *what) -> hipError_t {
      hipError_t he = hipMalloc(dst, bytes ? bytes : 4);
      if (he != hipSuccess) { fail(nullptr, NBODY_ERR_HIP, "nbody_create: hipMalloc %s: %s", what, hipGetErrorString(he)); return he; }
      if (bytes) he = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
      return he;
    };
    if ((e = hipMalloc(&c->sym_pool, c->sym_pool_elems * c->elem)) != hipSuccess) return bail(e, "hipMalloc partial-sum pool");
    if ((e = up(&c->sym_items, P.items.data(), P.items.size() * sizeof(nbody::SymItem), "work items")) != hipSuccess) return bail(e, "work items");
    if ((e = up(&c->sym_iptr, P.i_ptr.data(), P.i_ptr.size() * 4, "i-side list")) != hipSuccess) return bail(e, "i-side list");
    if ((e = up(&c->sym_ioff, P.i_off.data(), P.i_off.size() * 4, "i-side list")) != hipSuccess) return bail(e, "i-side list");
    if ((e = up(&c->sym_jptr, P.j_ptr.data(), P.j_ptr.size() * 4, "j-side list")) != hipSuccess) return bail(e, "j-side list");
    if ((e = up(&c->sym_joff, P.j_off.data(), P.j_off.size() * 4, "j-side list")) != hipSuccess) return bail(e, "j-side list");
    delete c->plan; c->plan = nullptr;
    if (p.precision != NBODY_PREC_F64 &&
        (e = hipMalloc(&c->sym_posg, (size_t)c->sym_pad * 16)) != hipSuccess) return bail(e, "hipMalloc scaled positions");
    // NBODY_SYM_GUARDED=1 (A/B measurements only): always run the guarded kernel, no coincident-body detector
    const char *guarded = getenv("NBODY_SYM_GUARDED");
    if (p.eps == 0.0 && p.zero_mode == NBODY_ZERO_EXACT && !(guarded && guarded[0] == '1')) {
      // How sparse: a body's entry is a chain of dependent device-scope compare-and-swaps (linear probing), each a round trip
      // to memory, and the update kernel ends with the LONGEST chain of the system.  At a power of two >= 2 N slots (load up to
      // 0.5; rounds 1-4) that chain was most of the fused update of a mid-size system: update kernel, slots >= 2 N / 4 N / 16 N /
      // 64 N (profiles/r05_ab_detector_table_sparsity.txt, r05_ab_update_kernel_parts.txt): N = 20480 12.6 / 10.6 / 9.8 / 11.3 us,
      // 32768 20.3 / 12.8 / 11.0 / 12.4, 65536 23.6 / 15.8 / 14.0 / 16.1 — and the table is cleared once per pass, which is what
      // large systems see: N = 262144 153 / 155 / 166 / 193 us.  Hence 16 N below 131072 bodies, 4 N from there on.
      int slots = 1024;
      const int factor = env_int("NBODY_SYM_DUP_FACTOR", p.n_total < 131072 ? 16 : 4);
      while ((long long)slots < (long long)factor * p.n_total && slots < (1 << 30)) slots *= 2;
      c->sym_dup_slots = slots;
      if ((e = hipMalloc(&c->sym_dup_table, (size_t)slots * 8 + 64)) != hipSuccess) return bail(e, "hipMalloc duplicate detector");
      if ((e = hipMemset(c->sym_dup_table, 0, (size_t)slots * 8 + 64)) != hipSuccess) return bail(e, "hipMemset duplicate detector");
      if (p.precision != NBODY_PREC_F64 && c->sym_nsrc == 1) {       // fused stepping alternates between two tables
        if ((e = hipMalloc(&c->sym_dup_table2, (size_t)slots * 8 + 64)) != hipSuccess) return bail(e, "hipMalloc duplicate detector");
        if ((e = hipMemset(c->sym_dup_table2, 0, (size_t)slots * 8 + 64)) != hipSuccess) return bail(e, "hipMemset duplicate detector");
      }
    }
    // equal-mass kernels (not with the eps floor, which is sized for G m |d|^-3, not for a bare |d|^-3).
    // NBODY_SYM_NO_UNI=1 (A/B measurements only) keeps every context on the general kernels.
    const char *no_uni = getenv("NBODY_SYM_NO_UNI");
    if (p.zero_mode != NBODY_ZERO_FLOOR && !(no_uni && no_uni[0] == '1')) {
      if ((e = hipMalloc(&c->sym_general, 64)) != hipSuccess) return bail(e, "hipMalloc equal-mass flag");
      if ((e = hipMemset(c->sym_general, 0, 64)) != hipSuccess) return bail(e, "hipMemset equal-mass flag");
    }
    if ((e = hipMalloc(&c->sym_send, (size_t)p.n_total * c->elem)) != hipSuccess) return bail(e, "hipMalloc send row");
    c->own_send = true;
    if (c->sym_nsrc > 1) {
      if ((e = hipMalloc(&c->sym_recv, (size_t)c->sym_nsrc * p.i_count * c->elem)) != hipSuccess) return bail(e, "hipMalloc recv rows");
      c->own_recv = true;
    } else {
      c->sym_recv = c->sym_send;
    }
  } else {
    if ((e = hipMalloc(&c->accp, (size_t)c->j_split * p.i_count * c->elem)) != hipSuccess) return bail(e, "hipMalloc accp");
    // packed one-sided kernel, exact d == 0: the same detector lets the tiles that hold no self pair run unguarded
    // (N = 2^20: 276.8 -> 248.8 ms).  Below N = 32768 the detector's two extra launches cost more than they save
    // (N = 8192: 27 -> 43 us).
    const char *guarded = getenv("NBODY_SYM_GUARDED");
    if (p.precision != NBODY_PREC_F64 && !c->wave && c->ipt % 2 == 0 && p.eps == 0.0 && p.zero_mode == NBODY_ZERO_EXACT &&
        p.n_total >= 32768 &&
        !(guarded && guarded[0] == '1')) {
      int slots = 1024;                                              // as sparse as the symmetric pass's (above): short chains
      const int factor = env_int("NBODY_SYM_DUP_FACTOR", p.n_total < 131072 ? 16 : 4);
      while ((long long)slots < (long long)factor * p.n_total && slots < (1 << 30)) slots *= 2;
      c->sym_dup_slots = slots;
      if ((e = hipMalloc(&c->sym_dup_table, (size_t)slots * 8 + 64)) != hipSuccess) return bail(e, "hipMalloc duplicate detector");
    }
    // equal-mass form of the packed one-sided kernel and of the block kernel (not small_pk_kernel)
    const char *no_uni = getenv("NBODY_SYM_NO_UNI");
    if (p.precision != NBODY_PREC_F64 && ((c->wave >= 2 && p.precision == NBODY_PREC_F32) || (!c->wave && c->ipt % 2 == 0)) && p.zero_mode != NBODY_ZERO_SELECT &&
        p.zero_mode != NBODY_ZERO_FLOOR && !(no_uni && no_uni[0] == '1')) {
      if ((e = hipMalloc(&c->sym_general, 64)) != hipSuccess) return bail(e, "hipMalloc equal-mass flag");
      if ((e = hipMemset(c->sym_general, 0, 64)) != hipSuccess) return bail(e, "hipMemset equal-mass flag");
    }
  }
  if ((e = hipMalloc(&c->scratch, 64)) != hipSuccess) return bail(e, "hipMalloc scratch");
  if ((e = hipMemset(c->scratch, 0, 64)) != hipSuccess) return bail(e, "hipMemset scratch");
  if (p.time_kernels) {
    // NBODY_SYM_ITEM_CLOCKS=1 (tools/even_items.py): room for every work item's own two stamps, word 2 says so
    c->clk_items = (c->sym && env_int("NBODY_SYM_ITEM_CLOCKS", 0) == 1) ? c->sym_items_n : 0;
    const size_t clk_bytes = 64 + 16 * (size_t)c->clk_items;
    if ((e = hipMalloc(&c->clk, clk_bytes)) != hipSuccess) return bail(e, "hipMalloc clock words");
    if ((e = hipMemset(c->clk, 0, clk_bytes)) != hipSuccess) return bail(e, "hipMemset clock words");
    if (c->clk_items) { const unsigned long long one = 1; if ((e = hipMemcpy(c->clk + 2, &one, 8, hipMemcpyHostToDevice)) != hipSuccess) return bail(e, "hipMemcpy clock words"); }
    (void)hipDeviceGetAttribute(&c->wall_khz, hipDeviceAttributeWallClockRate, p.device);
    (void)hipDeviceGetAttribute(&c->cus, hipDeviceAttributeMultiprocessorCount, p.device);
  }
  if ((e = hipHostMalloc(&c->h_scratch, 64, hipHostMallocDefault)) != hipSuccess) return bail(e, "hipHostMalloc");
  // the hipMemset calls above run on the null stream and return early; the context's own stream is non-blocking and would not
  // wait for them (bh_frame.hip, bh_create)
  if ((e = hipStreamSynchronize(nullptr)) != hipSuccess) return bail(e, "hipStreamSynchronize after the creation memsets");
  g_create_error.clear();   // e.g. the reason AUTO passed over the symmetric plan: not an error of this call
  *out = c;
  return NBODY_OK;
} catch (const std::bad_alloc &) {
  return fail(nullptr, NBODY_ERR_NOMEM, "nbody_create: out of host memory");
}

int nbody_create_multi(const nbody_params *pin, const int32_t *devices, int32_t n_dev, nbody_ctx **out) try {
  if (!pin || !devices || !out) return fail(nullptr, NBODY_ERR_INVALID, "nbody_create_multi: null argument");
  *out = nullptr;
  nbody_ctx *c = new (std::nothrow) nbody_ctx();
  if (!c) return fail(nullptr, NBODY_ERR_NOMEM, "nbody_create_multi: out of host memory");
  std::string why;
  const int rc = nbody::multi_create(pin, devices, n_dev, &c->multi, &why);
  if (rc) { delete c; return fail(nullptr, rc, "%s", why.c_str()); }
  c->p = *pin;
  c->p.i_begin = 0; c->p.i_count = pin->n_total; c->p.device = devices[0];
  c->theta = pin->theta;
  c->elem = (pin->precision == NBODY_PREC_F64) ? 32 : 16;
  *out = c;
  return NBODY_OK;
} catch (const std::bad_alloc &) {
  return fail(nullptr, NBODY_ERR_NOMEM, "nbody_create_multi: out of host memory");
}

void nbody_destroy(nbody_ctx *c) {
  if (c && c->multi) { nbody::multi_destroy(c->multi); delete c; return; }
  if (!c) return;
  (void)hipSetDevice(c->p.device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  for (KernelTimer &t : c->timers) {
    for (EventPair &e : t.pending) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (EventPair &e : t.pool) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  }
  if (c->own_posm && c->posm) (void)hipFree(c->posm);
  if (c->posm_alt) (void)hipFree(c->posm_alt);
  if (c->own_vel && c->vel) (void)hipFree(c->vel);
  if (c->own_acc && c->acc) (void)hipFree(c->acc);
  if (c->accp) (void)hipFree(c->accp);
  for (void *q : {c->sym_pool, c->sym_items, c->sym_iptr, c->sym_ioff, c->sym_jptr, c->sym_joff, c->sym_posg})
    if (q) (void)hipFree(q);
  delete c->plan;
  if (c->own_send && c->sym_send) (void)hipFree(c->sym_send);
  if (c->own_recv && c->sym_recv) (void)hipFree(c->sym_recv);
  if (c->sym_dup_table) (void)hipFree(c->sym_dup_table);
  if (c->sym_dup_table2) (void)hipFree(c->sym_dup_table2);
  if (c->sym_general) (void)hipFree(c->sym_general);
  if (c->bh) nbody::bh_destroy(c->bh);
  if (c->bh_acc) (void)hipFree(c->bh_acc);
  if (c->d_stage) (void)hipFree(c->d_stage);
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->clk) (void)hipFree(c->clk);
  if (c->energy_part) (void)hipFree(c->energy_part);
  if (c->h_scratch) (void)hipHostFree(c->h_scratch);
  for (const auto &r : c->pinned) (void)hipHostUnregister(r.first);   // the memory itself stays the caller's
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

int nbody_set_stream(nbody_ctx *c, void *hip_stream) {
  if (c && c->multi) return multi_unsupported(c, "nbody_set_stream");
  if (!c) return NBODY_ERR_INVALID;
  if (int rc = use_device(c)) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
  return NBODY_OK;
}

int nbody_device_ptr(nbody_ctx *c, int32_t which, void **ptr, size_t *bytes) {
  if (c && c->multi) return multi_unsupported(c, "nbody_device_ptr");
  if (!c || !ptr) return NBODY_ERR_INVALID;
  if (int rc = use_device(c)) return rc;
  switch (which) {
    case NBODY_BUF_POSM:
      *ptr = c->posm; if (bytes) *bytes = (size_t)c->p.n_total * c->elem;
      if (int rc = posm_escapes(c)) return rc;                    // the caller may write positions from now on
      break;
    case NBODY_BUF_VEL:  *ptr = c->vel;  if (bytes) *bytes = (size_t)c->p.i_count * c->elem; break;
    case NBODY_BUF_ACC:  *ptr = c->acc;  if (bytes) *bytes = (size_t)c->p.i_count * c->elem; break;
    default: return fail(c, NBODY_ERR_INVALID, "nbody_device_ptr: unknown buffer %d", which);
  }
  return NBODY_OK;
}

int nbody_bind_device_state(nbody_ctx *c, void *posm, void *vel, void *acc) {
  if (c && c->multi) return multi_unsupported(c, "nbody_bind_device_state");
  if (!c) return NBODY_ERR_INVALID;
  if (int rc = use_device(c)) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (posm) {
    if (int rc = posm_escapes(c)) return rc;
    if (c->own_posm) (void)hipFree(c->posm);
    c->posm = posm; c->own_posm = false;
    c->masses_equal = -1;                       // masses nobody here has seen
    if (c->sym_general)                         // ... every pass's preparation kernel looks
      HIP_TRY(c, hipMemsetAsync(c->sym_general, 0, 4, c->stream));
  }
  if (vel)  { if (c->own_vel) (void)hipFree(c->vel);   c->vel = vel;   c->own_vel = false; }
  if (acc)  { if (c->own_acc) (void)hipFree(c->acc);   c->acc = acc;   c->own_acc = false; }
  // the caller vouches that bound buffers hold a valid state
  if (posm && vel) c->have_state = true;
  c->floor_eps2 = -1.0;
  return NBODY_OK;
}

int nbody_synchronize(nbody_ctx *c) {
  if (c && c->multi) return multi_rc(c, nbody::multi_synchronize(c->multi));
  if (!c) return NBODY_ERR_INVALID;
  if (int rc = use_device(c)) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return NBODY_OK;
}

int nbody_set_state_soa(nbody_ctx *c, const float *posm4, const float *vel4, int32_t n) try {
  if (!c || !posm4 || !vel4) return c ? fail(c, NBODY_ERR_INVALID, "nbody_set_state_soa: null buffer") : NBODY_ERR_INVALID;
  if (n != c->p.n_total) return fail(c, NBODY_ERR_INVALID, "nbody_set_state_soa: n = %d but the context holds %d bodies", n, c->p.n_total);
  if (c->multi) { const int rc = multi_rc(c, nbody::multi_set_state_soa(c->multi, posm4, vel4, n)); if (!rc) { c->have_state = true; c->steps_done = 0; } return rc; }
  return upload_soa<float>(c, posm4, vel4);
} catch (const std::bad_alloc &) {
  return fail(c, NBODY_ERR_NOMEM, "nbody_set_state_soa: out of host memory");
}

int nbody_set_state_soa_f64(nbody_ctx *c, const double *posm4, const double *vel4, int32_t n) try {
  if (!c || !posm4 || !vel4) return c ? fail(c, NBODY_ERR_INVALID, "nbody_set_state_soa_f64: null buffer") : NBODY_ERR_INVALID;
  if (n != c->p.n_total) return fail(c, NBODY_ERR_INVALID, "nbody_set_state_soa_f64: n = %d but the context holds %d bodies", n, c->p.n_total);
  if (c->multi) { const int rc = multi_rc(c, nbody::multi_set_state_soa_f64(c->multi, posm4, vel4, n)); if (!rc) { c->have_state = true; c->steps_done = 0; } return rc; }
  return upload_soa<double>(c, posm4, vel4);
} catch (const std::bad_alloc &) {
  return fail(c, NBODY_ERR_NOMEM, "nbody_set_state_soa_f64: out of host memory");
}

}  // extern "C"

namespace {
int set_particles(nbody_ctx *c, const void *aos, size_t stride, int32_t n, bool keep_history, const char *who) {
  if (!c || !aos) return c ? fail(c, NBODY_ERR_INVALID, "%s: null buffer", who) : NBODY_ERR_INVALID;
  if (n != c->p.n_total) return fail(c, NBODY_ERR_INVALID, "%s: n = %d but the context holds %d bodies", who, n, c->p.n_total);
  if (stride < sizeof(nbody_particle)) return fail(c, NBODY_ERR_INVALID, "%s: stride %zu < %zu", who, stride, sizeof(nbody_particle));
  if (keep_history && !c->have_state) return fail(c, NBODY_ERR_STATE, "%s: no state has been set yet (nbody_set_particles first)", who);
  if (c->multi) {
    const int rc = multi_rc(c, nbody::multi_set_particles(c->multi, aos, stride, n, keep_history));
    if (!rc) { c->have_state = true; if (!keep_history) c->steps_done = 0; }
    return rc;
  }
  std::vector<float> posm((size_t)n * 4), vel((size_t)n * 4);
  const char *base = (const char *)aos;
  for (int i = 0; i < n; ++i) {
    nbody_particle q;
    memcpy(&q, base + (size_t)i * stride, sizeof q);
    posm[4 * (size_t)i + 0] = q.Position[0]; posm[4 * (size_t)i + 1] = q.Position[1];
    posm[4 * (size_t)i + 2] = q.Position[2]; posm[4 * (size_t)i + 3] = q.Mass;
    vel[4 * (size_t)i + 0] = q.Velocity[0]; vel[4 * (size_t)i + 1] = q.Velocity[1];
    vel[4 * (size_t)i + 2] = q.Velocity[2]; vel[4 * (size_t)i + 3] = 0.f;
  }
  int rc = upload_soa<float>(c, posm.data(), vel.data(), keep_history);
  if (rc) return rc;
  // carry the records' Acceleration field over too (the reference keeps whatever was there until the next force pass)
  std::vector<float> acc((size_t)c->p.i_count * 4);
  for (int i = 0; i < c->p.i_count; ++i) {
    nbody_particle q;
    memcpy(&q, base + (size_t)(c->p.i_begin + i) * stride, sizeof q);
    acc[4 * (size_t)i + 0] = q.Acceleration[0]; acc[4 * (size_t)i + 1] = q.Acceleration[1];
    acc[4 * (size_t)i + 2] = q.Acceleration[2]; acc[4 * (size_t)i + 3] = 0.f;
  }
  if (c->p.precision == NBODY_PREC_F64) {
    std::vector<double> a64(acc.begin(), acc.end());
    HIP_TRY(c, hipMemcpy(c->acc, a64.data(), a64.size() * 8, hipMemcpyHostToDevice));
  } else {
    HIP_TRY(c, hipMemcpy(c->acc, acc.data(), acc.size() * 4, hipMemcpyHostToDevice));
  }
  return NBODY_OK;
}
}  // namespace

extern "C" {

int nbody_set_particles(nbody_ctx *c, const void *aos, size_t stride, int32_t n) try {
  return set_particles(c, aos, stride, n, false, "nbody_set_particles");
} catch (const std::bad_alloc &) {
  return fail(c, NBODY_ERR_NOMEM, "nbody_set_particles: out of host memory");
}

int nbody_push_particles(nbody_ctx *c, const void *aos, size_t stride, int32_t n) try {
  return set_particles(c, aos, stride, n, true, "nbody_push_particles");
} catch (const std::bad_alloc &) {
  return fail(c, NBODY_ERR_NOMEM, "nbody_push_particles: out of host memory");
}

// A sharded symmetric context has an exchange between the force pass and the update: the caller must drive
// nbody_step_begin -> all-to-all(nbody_exchange_info) -> nbody_step_end.
static int needs_phases(nbody_ctx *c, const char *who) {
  if (c->theta > 0.0f) return NBODY_OK;
  if (c->sym && c->sym_nsrc > 1)
    return fail(c, NBODY_ERR_STATE, "%s: this sharded context uses the symmetric algorithm; drive it with "
                "nbody_step_begin / all-to-all of nbody_exchange_info buffers / nbody_step_end", who);
  return NBODY_OK;
}

int nbody_step_begin(nbody_ctx *c) {
  if (c && c->multi) return multi_unsupported(c, "nbody_step_begin");
  int rc = check_ready(c);
  if (rc) return rc;
  if (c->step_open || c->step_local) return fail(c, NBODY_ERR_STATE, "nbody_step_begin: previous step not ended");
  HIP_TRY(c, hipSetDevice(c->p.device));
  if ((rc = run_forces(c))) return rc;
  c->step_open = true;
  return NBODY_OK;
}

// Can the force pass run in two goes — the strips inside the own slice first, the rest once the other ranks' positions
// are in?  Sharded fp32 symmetric contexts (all-pairs): their preparation kernel and their plan know the cut.
static bool sym_two_goes(const nbody_ctx *c) {
  return c->theta == 0.0f && c->sym && c->sym_nsrc > 1 && c->p.precision != NBODY_PREC_F64;
}

int nbody_step_begin_local(nbody_ctx *c) {
  if (c && c->multi) return multi_unsupported(c, "nbody_step_begin_local");
  int rc = check_ready(c);
  if (rc) return rc;
  if (c->step_open || c->step_local) return fail(c, NBODY_ERR_STATE, "nbody_step_begin_local: previous step not ended");
  if (sym_two_goes(c)) { if ((rc = run_forces(c, false, 1))) return rc; }   // otherwise everything happens in the second go
  c->step_local = true;
  return NBODY_OK;
}

int nbody_step_begin_remote(nbody_ctx *c) {
  if (c && c->multi) return multi_unsupported(c, "nbody_step_begin_remote");
  int rc = check_ready(c);
  if (rc) return rc;
  if (!c->step_local) return fail(c, NBODY_ERR_STATE, "nbody_step_begin_remote: nbody_step_begin_local first");
  c->step_local = false;
  if ((rc = run_forces(c, false, sym_two_goes(c) ? 2 : 0))) return rc;
  c->step_open = true;
  return NBODY_OK;
}

int nbody_step_end(nbody_ctx *c, float dt) {
  if (c && c->multi) return multi_unsupported(c, "nbody_step_end");
  int rc = check_ready(c);
  if (rc) return rc;
  if (!c->step_open) return fail(c, NBODY_ERR_STATE, "nbody_step_end: no step begun");
  HIP_TRY(c, hipSetDevice(c->p.device));
  c->step_open = false;
  if ((rc = run_update(c, dt > 0.0f ? dt : 0.0f))) return rc;
  if (dt > 0.0f) c->steps_done += 1;
  return NBODY_OK;
}

int nbody_exchange_info(nbody_ctx *c, void **send, void **recv, size_t *bytes_per_rank, int32_t *n_ranks) {
  if (c && c->multi) { if (send) *send = nullptr; if (recv) *recv = nullptr; if (bytes_per_rank) *bytes_per_rank = 0; if (n_ranks) *n_ranks = 0; return NBODY_OK; }
  if (!c) return NBODY_ERR_INVALID;
  const bool ex = c->sym && c->sym_nsrc > 1;
  if (send) *send = ex ? c->sym_send : nullptr;
  if (recv) *recv = ex ? c->sym_recv : nullptr;
  if (bytes_per_rank) *bytes_per_rank = ex ? (size_t)c->p.i_count * c->elem : 0;
  if (n_ranks) *n_ranks = ex ? c->sym_nsrc : 0;
  return NBODY_OK;
}

int nbody_exchange_read_send(nbody_ctx *c, void *host) {
  if (c && c->multi) return multi_unsupported(c, "nbody_exchange_read_send");
  if (!c || !host) return NBODY_ERR_INVALID;
  if (!(c->sym && c->sym_nsrc > 1)) return fail(c, NBODY_ERR_STATE, "nbody_exchange_read_send: this context has no exchange step");
  if (int rc = use_device(c)) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(host, c->sym_send, (size_t)c->p.n_total * c->elem, hipMemcpyDeviceToHost));
  return NBODY_OK;
}

int nbody_exchange_write_recv(nbody_ctx *c, const void *host) {
  if (c && c->multi) return multi_unsupported(c, "nbody_exchange_write_recv");
  if (!c || !host) return NBODY_ERR_INVALID;
  if (!(c->sym && c->sym_nsrc > 1)) return fail(c, NBODY_ERR_STATE, "nbody_exchange_write_recv: this context has no exchange step");
  if (int rc = use_device(c)) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(c->sym_recv, host, (size_t)c->sym_nsrc * c->p.i_count * c->elem, hipMemcpyHostToDevice));
  return NBODY_OK;
}

int nbody_bind_exchange(nbody_ctx *c, void *send, void *recv) {
  if (c && c->multi) return multi_unsupported(c, "nbody_bind_exchange");
  if (!c) return NBODY_ERR_INVALID;
  if (!(c->sym && c->sym_nsrc > 1)) return fail(c, NBODY_ERR_STATE, "nbody_bind_exchange: this context has no exchange step");
  if (!send || !recv) return fail(c, NBODY_ERR_INVALID, "nbody_bind_exchange: null buffer");
  if (int rc = use_device(c)) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (c->own_send) (void)hipFree(c->sym_send);
  if (c->own_recv) (void)hipFree(c->sym_recv);
  c->sym_send = send; c->sym_recv = recv; c->own_send = c->own_recv = false;
  return NBODY_OK;
}

int nbody_compute_forces(nbody_ctx *c) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (c->multi) return multi_rc(c, nbody::multi_forces(c->multi, 0.0f));
  if ((rc = needs_phases(c, "nbody_compute_forces"))) return rc;
  HIP_TRY(c, hipSetDevice(c->p.device));
  if ((rc = run_forces(c, true))) return rc;
  if ((rc = run_update(c, 0.0f))) return rc;
  return NBODY_OK;
}

// Small and mid-size single-context fp32 systems (forces_block_pk_kernel): forces + update in ONE launch per step, ping-ponging
// the position buffer — not once the caller holds a pointer to one of the two buffers, and the host must know the masses.
static bool one_launch_ok(const nbody_ctx *c) {
  return c->wave != 0 && c->theta == 0.0f && c->own_posm && !c->posm_escaped && c->p.i_count == c->p.n_total &&
         (c->p.precision != NBODY_PREC_F32 || make_launch(c).uni >= 0);
}

// one such step; stage / size_bits / size_zero: the frame's mirror and ComputeCubeSize from the same launch (nbody_tick)
static int step_one_launch(nbody_ctx *c, float dt, void *stage, void *size_bits, void *size_zero) {
  int rc;
  if (!c->posm_alt) HIP_TRY(c, hipMalloc(&c->posm_alt, (size_t)c->p.n_total * c->elem));
  if ((rc = ensure_floor(c))) return rc;
  EventPair ev;
  const bool timed = c->p.time_kernels != 0;
  if (timed && (rc = timer_begin(c, NBODY_KERNEL_FORCES, &ev))) return rc;
  HIP_TRY(c, nbody::launch_step_small(make_launch(c), c->posm_alt, c->vel, c->acc, dt, c->stream, stage, size_bits, size_zero));
  if (timed && (rc = timer_end(c, NBODY_KERNEL_FORCES, ev))) return rc;
  if (timed && c->timers[NBODY_KERNEL_FORCES].pending.size() >= 1024 && (rc = timer_drain(c, NBODY_KERNEL_FORCES))) return rc;
  std::swap(c->posm, c->posm_alt);
  return NBODY_OK;
}

int nbody_step(nbody_ctx *c, float dt, int32_t nsteps) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (nsteps < 0) return fail(c, NBODY_ERR_INVALID, "nbody_step: nsteps < 0");
  if (!(dt > 0.0f)) return NBODY_OK;   // OctreeSearch.cpp:25: PhDeltaTime <= 0 freezes the physics
  if (c->multi) {
    if (c->theta > 0.0f) {                                        // whole frames on every device, one wait per batch (multi_bh_steps)
      int built = 0;
      rc = multi_rc(c, nbody::multi_bh_steps(c->multi, dt, nsteps, &built));
      c->steps_done += built;
      return rc;
    }
    for (int s = 0; s < nsteps; ++s) {
      if ((rc = multi_rc(c, nbody::multi_forces(c->multi, dt)))) return rc;
      c->steps_done += 1;
    }
    return NBODY_OK;
  }
  if ((rc = needs_phases(c, "nbody_step"))) return rc;
  if (nsteps > 1 && c->p.i_count != c->p.n_total)
    return fail(c, NBODY_ERR_STATE, "nbody_step: a sharded context advances one step per call (all-gather NBODY_BUF_POSM in between)");
  HIP_TRY(c, hipSetDevice(c->p.device));
  if (c->theta > 0.0f) {
    if ((rc = ensure_bh(c))) return rc;
    // every frame queued, one wait per call — the larger systems' in batches of 64: a frame their warm sort gives up takes the
    // frames queued behind it along (bh_collect queues them again), and that should not be hundreds
    const int batch = nbody::bh_is_small(c->bh) ? nsteps : 64;
    for (int done = 0; done < nsteps; done += batch) {
      if ((rc = bh_enqueue(c, dt, std::min(batch, nsteps - done)))) return rc;
      if ((rc = bh_finish(c))) return rc;
    }
    return NBODY_OK;
  }
  const bool one_launch = one_launch_ok(c);
  for (int s = 0; s < nsteps; ++s) {
    if (one_launch) {
      if ((rc = step_one_launch(c, dt, nullptr, nullptr, nullptr))) return rc;
      continue;
    }
    if ((rc = run_forces(c))) return rc;
    if ((rc = run_update(c, dt))) return rc;
  }
  c->steps_done += nsteps;
  return NBODY_OK;
}

int nbody_get_bounds(nbody_ctx *c, float *size) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (!size) return fail(c, NBODY_ERR_INVALID, "nbody_get_bounds: null output");
  if (c->multi) return multi_rc(c, nbody::multi_get_bounds(c->multi, size));
  HIP_TRY(c, hipMemsetAsync(c->scratch, 0, 4, c->stream));
  HIP_TRY(c, nbody::launch_bounds(c->p.precision, c->posm, c->p.i_begin, c->p.i_count, (unsigned int *)c->scratch, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->h_scratch, c->scratch, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  memcpy(size, c->h_scratch, 4);
  return NBODY_OK;
}

int nbody_energy(nbody_ctx *c, double *ke, double *pe) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (c->multi) return multi_rc(c, nbody::multi_energy(c->multi, ke, pe));
  if (!c->energy_part)
    HIP_TRY(c, hipMalloc(&c->energy_part, nbody::energy_partials(c->p.n_total, c->p.i_count) * sizeof(double)));
  HIP_TRY(c, nbody::launch_energy(c->p.precision, c->posm, c->vel, c->p.n_total, c->p.i_begin, c->p.i_count, c->p.G,
                                  c->p.eps * c->p.eps, (double *)c->energy_part, (double *)c->scratch, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->h_scratch, c->scratch, 16, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  double v[2];
  memcpy(v, c->h_scratch, 16);
  if (ke) *ke = v[0];
  if (pe) *pe = v[1];
  return NBODY_OK;
}

int nbody_get_positions(nbody_ctx *c, float *xyz, size_t stride, int32_t first, int32_t count) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (!xyz || stride < 12) return fail(c, NBODY_ERR_INVALID, "nbody_get_positions: null buffer or stride < 12");
  if (first < 0 || count < 0 || first + count > c->p.n_total) return fail(c, NBODY_ERR_INVALID, "nbody_get_positions: range out of bounds");
  if (count == 0) return NBODY_OK;
  if (c->multi) return multi_rc(c, nbody::multi_get_positions(c->multi, xyz, stride, first, count));
  // one repack kernel + one pinned D2H copy (OctreeSearch.cpp:41 reads Position of every body each frame)
  const size_t bytes = (size_t)count * 12;
  if ((rc = ensure_stage(c, bytes))) return rc;
  HIP_TRY(c, nbody::launch_pack_positions(c->p.precision, c->posm, (float *)c->d_stage, first, count, c->stream));
  const bool direct = stride == 12 && in_pinned(c, xyz, bytes);     // caller's pinned buffer: DMA straight into it
  HIP_TRY(c, hipMemcpyAsync(direct ? (void *)xyz : c->h_stage, c->d_stage, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (direct) return NBODY_OK;
  if (stride == 12) {
    memcpy(xyz, c->h_stage, bytes);
  } else {
    char *o = (char *)xyz;
    const char *src = (const char *)c->h_stage;
    for (int i = 0; i < count; ++i) memcpy(o + (size_t)i * stride, src + (size_t)i * 12, 12);
  }
  return NBODY_OK;
}

int nbody_get_state_soa(nbody_ctx *c, float *posm4, float *vel4, float *acc4) try {
  int rc = check_ready(c);
  if (rc) return rc;
  if (c->multi) return multi_rc(c, nbody::multi_get_state_soa(c->multi, posm4, vel4, acc4));
  if (posm4 && (rc = download4<float>(c, c->posm, (size_t)c->p.i_begin, (size_t)c->p.i_count, posm4))) return rc;
  if (vel4 && (rc = download4<float>(c, c->vel, 0, (size_t)c->p.i_count, vel4))) return rc;
  if (acc4 && (rc = download4<float>(c, c->acc, 0, (size_t)c->p.i_count, acc4))) return rc;
  return NBODY_OK;
} catch (const std::bad_alloc &) {
  return fail(c, NBODY_ERR_NOMEM, "nbody_get_state_soa: out of host memory");
}

int nbody_get_state_soa_f64(nbody_ctx *c, double *posm4, double *vel4, double *acc4) try {
  int rc = check_ready(c);
  if (rc) return rc;
  if (c->multi) return multi_rc(c, nbody::multi_get_state_soa_f64(c->multi, posm4, vel4, acc4));
  if (posm4 && (rc = download4<double>(c, c->posm, (size_t)c->p.i_begin, (size_t)c->p.i_count, posm4))) return rc;
  if (vel4 && (rc = download4<double>(c, c->vel, 0, (size_t)c->p.i_count, vel4))) return rc;
  if (acc4 && (rc = download4<double>(c, c->acc, 0, (size_t)c->p.i_count, acc4))) return rc;
  return NBODY_OK;
} catch (const std::bad_alloc &) {
  return fail(c, NBODY_ERR_NOMEM, "nbody_get_state_soa_f64: out of host memory");
}

int nbody_get_particles(nbody_ctx *c, void *aos, size_t stride) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (!aos || stride < sizeof(nbody_particle)) return fail(c, NBODY_ERR_INVALID, "nbody_get_particles: null buffer or stride < 40");
  if (c->multi) return multi_rc(c, nbody::multi_get_particles(c->multi, aos, stride));
  const size_t ic = (size_t)c->p.i_count;
  const size_t bytes = ic * sizeof(nbody_particle);
  if ((rc = ensure_stage(c, bytes))) return rc;
  HIP_TRY(c, nbody::launch_pack_particles(c->p.precision, c->posm, c->vel, c->acc, (float *)c->d_stage, c->p.i_begin,
                                          c->p.i_count, c->stream));
  const bool direct = stride == sizeof(nbody_particle) && in_pinned(c, aos, bytes);
  HIP_TRY(c, hipMemcpyAsync(direct ? aos : c->h_stage, c->d_stage, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (!direct) unstage_particles(c, aos, stride, ic);
  return NBODY_OK;
}

// One whole frame of the actor with ONE host synchronisation: ComputeCubeSize of the current positions, the Tick body,
// the FParticle mirror.
int nbody_tick(nbody_ctx *c, float dt, float *size, void *aos, size_t stride) {
  int rc = check_ready(c);
  if (rc) return rc;
  if (aos && stride < sizeof(nbody_particle)) return fail(c, NBODY_ERR_INVALID, "nbody_tick: stride < 40");
  if (c->multi) {                                                // the same frame as three calls: every device has its own stream to wait for
    if (dt > 0.0f && size && (rc = nbody_get_bounds(c, size))) return rc;
    if (dt > 0.0f && (rc = nbody_step(c, dt, 1))) return rc;
    return aos ? nbody_get_particles(c, aos, stride) : nbody_synchronize(c);
  }
  if ((rc = needs_phases(c, "nbody_tick"))) return rc;
  HIP_TRY(c, hipSetDevice(c->p.device));
  const bool live = dt > 0.0f;                                   // OctreeSearch.cpp:25
  // the eps floor of NBODY_ZERO_FLOOR is computed through the same 64-byte scratch the bounds travel in: settle it
  // before the bounds are queued, or the first frame would return the largest mass as Size
  if (live && c->theta == 0.0f && (rc = ensure_floor(c))) return rc;
  // small systems at theta > 0: the tree's own launch computes Size (it is the root's half-width) — the frame is queued
  // here and its verdict collected by the frame's one wait
  bool bh_frame = false;
  if (live && c->theta > 0.0f) {
    if ((rc = ensure_bh(c))) return rc;
    bh_frame = true;
  }
  const size_t ic = (size_t)c->p.i_count, bytes = ic * sizeof(nbody_particle);
  // systems on the one-launch step (theta == 0, up to 16384 bodies): the same launch leaves Size (of the positions before
  // the update, as .cpp:26 has it) and the frame's FParticle records — one kernel, the copies, one wait
  if (live && c->theta == 0.0f && (size || aos) && c->p.precision == NBODY_PREC_F32 && one_launch_ok(c)) {
    // the records go straight into page-locked host memory — the caller's own mirror if it pinned it (nbody_pin_host_buffer),
    // the context's staging buffer otherwise: no copy to wait for (profiles/r03_tick_parts_n2000.txt: the 80 KB copy of the
    // shipped scene's mirror cost 12.7 us of a 32.7 us frame)
    bool direct = false;
    void *stage = nullptr;
    if (aos) {
      if ((rc = ensure_stage(c, bytes))) return rc;
      direct = stride == sizeof(nbody_particle) && in_pinned(c, aos, bytes) && ((uintptr_t)aos & 15u) == 0;
      HIP_TRY(c, hipHostGetDevicePointer(&stage, direct ? aos : c->h_stage, 0));
    }
    unsigned int *words = (unsigned int *)c->scratch + 8;        // two words that take turns: this frame's (zero), the next one's
    unsigned int *cur = words + c->tick_word, *nxt = words + (c->tick_word ^ 1);
    if ((rc = step_one_launch(c, dt, stage, size ? cur : nullptr, size ? nxt : nullptr))) return rc;
    c->steps_done += 1;
    if (size) { c->tick_word ^= 1; HIP_TRY(c, hipMemcpyAsync(c->h_scratch, cur, 4, hipMemcpyDeviceToHost, c->stream)); }
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (size) memcpy(size, c->h_scratch, 4);
    if (aos && !direct) unstage_particles(c, aos, stride, ic);
    return NBODY_OK;
  }
  bool bh_direct = false;
  if (bh_frame) {                                                // the walk writes the frame's records itself, Size rides with the verdict
    // ... straight into page-locked host memory: the caller's mirror if it is pinned, the staging buffer otherwise
    void *stage = nullptr;
    if (aos) {
      if ((rc = ensure_stage(c, bytes))) return rc;
      bh_direct = stride == sizeof(nbody_particle) && in_pinned(c, aos, bytes);
      HIP_TRY(c, hipHostGetDevicePointer(&stage, bh_direct ? aos : c->h_stage, 0));
    }
    if ((rc = bh_enqueue(c, dt, 1, (float *)stage))) return rc;
  } else if (live && size) {                                     // .cpp:26, 47-56: bounds of the positions BEFORE the step
    HIP_TRY(c, hipMemsetAsync(c->scratch, 0, 4, c->stream));
    HIP_TRY(c, nbody::launch_bounds(c->p.precision, c->posm, c->p.i_begin, c->p.i_count, (unsigned int *)c->scratch, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->h_scratch, c->scratch, 4, hipMemcpyDeviceToHost, c->stream));
  }
  if (live && !bh_frame && (rc = nbody_step(c, dt, 1))) return rc;   // .cpp:27-31
  bool direct = false;
  if (aos) {                                                     // .cpp:33,41: what the frame draws
    if ((rc = ensure_stage(c, bytes))) return rc;
    if (!bh_frame)
      HIP_TRY(c, nbody::launch_pack_particles(c->p.precision, c->posm, c->vel, c->acc, (float *)c->d_stage, c->p.i_begin,
                                              c->p.i_count, c->stream));
    direct = stride == sizeof(nbody_particle) && in_pinned(c, aos, bytes);
    if (!bh_frame) HIP_TRY(c, hipMemcpyAsync(direct ? aos : c->h_stage, c->d_stage, bytes, hipMemcpyDeviceToHost, c->stream));
  }
  int frame_rc = NBODY_OK;
  if (bh_frame) frame_rc = bh_finish(c);                         // the frame's one wait
  else HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (bh_frame && frame_rc != NBODY_OK && aos) {                 // a refused frame wrote no records: deliver the untouched state
    HIP_TRY(c, nbody::launch_pack_particles(c->p.precision, c->posm, c->vel, c->acc, (float *)c->d_stage, c->p.i_begin,
                                            c->p.i_count, c->stream));
    HIP_TRY(c, hipMemcpyAsync(direct ? aos : c->h_stage, c->d_stage, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
  }
  if (bh_frame) { if (size) *size = nbody::bh_last_size(c->bh); }
  else if (live && size) memcpy(size, c->h_scratch, 4);
  if (aos && !direct) unstage_particles(c, aos, stride, ic);
  return frame_rc;
}

int nbody_pin_host_buffer(nbody_ctx *c, void *host, size_t bytes) {
  if (!c) return NBODY_ERR_INVALID;
  if (!host || bytes == 0) return fail(c, NBODY_ERR_INVALID, "nbody_pin_host_buffer: null buffer or zero size");
  if (c->multi) return NBODY_OK;   // an optimisation only: a multi-device context delivers through each device's staging buffer
  for (const auto &r : c->pinned)
    if ((char *)host < r.first + r.second && r.first < (char *)host + bytes)
      return fail(c, NBODY_ERR_INVALID, "nbody_pin_host_buffer: overlaps a range that is already pinned");
  HIP_TRY(c, hipSetDevice(c->p.device));
  HIP_TRY(c, hipHostRegister(host, bytes, hipHostRegisterDefault));
  c->pinned.emplace_back((char *)host, bytes);
  return NBODY_OK;
}

int nbody_unpin_host_buffer(nbody_ctx *c, void *host) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) return NBODY_OK;
  for (size_t k = 0; k < c->pinned.size(); ++k)
    if (c->pinned[k].first == (char *)host) {
      if (int rc = use_device(c)) return rc;
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      c->pinned.erase(c->pinned.begin() + (long)k);
      HIP_TRY(c, hipHostUnregister(host));
      return NBODY_OK;
    }
  return fail(c, NBODY_ERR_INVALID, "nbody_unpin_host_buffer: not a buffer pinned through this context");
}

// ---- checkpoint / resume (nothing in the reference to mirror: its state is not even a UPROPERTY) ----
namespace {
struct CkptHeader {
  char magic[8];          // "NBDYCKP2"
  uint32_t header_bytes;
  int32_t n_total, i_begin, i_count;
  int32_t elem_bytes;     // 4 (fp32 state) or 8 (fp64 state)
  int32_t has_root;       // Barnes-Hut cross-frame state present: root_com is the next tree's root centre
  int64_t steps_done;
  double G, eps;
  float theta;            // opening angle in force when the file was written (the reference ships 1.0, OctreeSearch.cpp:85)
  float root_com[3];      // previous tree's centre of mass (OctreeSearch.cpp:77-79)
};
}  // namespace

int nbody_save_checkpoint(nbody_ctx *c, const char *path) try {
  int rc = check_ready(c);
  if (rc) return rc;
  if (!path) return fail(c, NBODY_ERR_INVALID, "nbody_save_checkpoint: null path");
  const bool f64 = c->p.precision == NBODY_PREC_F64;
  const size_t eb = f64 ? 8 : 4, n = (size_t)c->p.n_total, ic = (size_t)c->p.i_count;
  std::vector<char> posm(n * 4 * eb), vel(ic * 4 * eb), acc(ic * 4 * eb);
  CkptHeader h;
  memset(&h, 0, sizeof h);
  if (c->multi) {
    // a multi-device context writes the file a single context of the whole system would write
    rc = f64 ? nbody_get_state_soa_f64(c, (double *)posm.data(), (double *)vel.data(), (double *)acc.data())
             : nbody_get_state_soa(c, (float *)posm.data(), (float *)vel.data(), (float *)acc.data());
    if (rc) return rc;
    int has_root = 0;
    if ((rc = multi_rc(c, nbody::multi_bh_root(c->multi, h.root_com, &has_root)))) return rc;
    h.has_root = has_root;
  } else {
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(posm.data(), c->posm, posm.size(), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(vel.data(), c->vel, vel.size(), hipMemcpyDeviceToHost));
    HIP_TRY(c, hipMemcpy(acc.data(), c->acc, acc.size(), hipMemcpyDeviceToHost));
    if (c->bh) {                                                   // the next frame's tree is rooted at this frame's CoM
      HIP_TRY(c, nbody::bh_get_root_com(c->bh, h.root_com, c->stream));
      h.has_root = 1;
    }
  }
  memcpy(h.magic, "NBDYCKP2", 8);
  h.header_bytes = (uint32_t)sizeof h;
  h.n_total = c->p.n_total; h.i_begin = c->p.i_begin; h.i_count = c->p.i_count;
  h.elem_bytes = (int32_t)eb; h.steps_done = c->steps_done; h.G = c->p.G; h.eps = c->p.eps; h.theta = c->theta;
  FILE *f = fopen(path, "wb");
  if (!f) return fail(c, NBODY_ERR_INVALID, "nbody_save_checkpoint: cannot open %s for writing", path);
  const bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(posm.data(), 1, posm.size(), f) == posm.size() &&
                  fwrite(vel.data(), 1, vel.size(), f) == vel.size() && fwrite(acc.data(), 1, acc.size(), f) == acc.size();
  if (fclose(f) != 0 || !ok) return fail(c, NBODY_ERR_INVALID, "nbody_save_checkpoint: short write to %s", path);
  return NBODY_OK;
} catch (const std::bad_alloc &) {
  return fail(c, NBODY_ERR_NOMEM, "nbody_save_checkpoint: out of host memory");
}

int nbody_load_checkpoint(nbody_ctx *c, const char *path, int64_t *steps_done) try {
  if (!c || !path) return c ? fail(c, NBODY_ERR_INVALID, "nbody_load_checkpoint: null path") : NBODY_ERR_INVALID;
  if (c->multi) {                                                  // every device reads its slice of the same file
    int64_t n = 0;
    const int rc = multi_rc(c, nbody::multi_load_checkpoint(c->multi, path, &n));
    if (rc) return rc;
    c->have_state = true; c->steps_done = n;
    (void)nbody_get_theta(nbody::multi_part(c->multi, 0), &c->theta);   // the file's opening angle: every device took it over
    if (steps_done) *steps_done = n;
    return NBODY_OK;
  }
  CkptHeader h;
  const bool f64 = c->p.precision == NBODY_PREC_F64;
  const size_t eb = f64 ? 8 : 4, n = (size_t)c->p.n_total, ic = (size_t)c->p.i_count;
  int rc = NBODY_OK;
  std::vector<char> posm(n * 4 * eb), vel(ic * 4 * eb), acc(ic * 4 * eb);
  FILE *f = fopen(path, "rb");
  if (!f) return fail(c, NBODY_ERR_INVALID, "nbody_load_checkpoint: cannot open %s", path);
  // format 1 (no Barnes-Hut state: the header ends after eps) still loads, as a theta = 0 file without a tree root
  constexpr size_t v1_bytes = offsetof(CkptHeader, theta);
  memset(&h, 0, sizeof h);
  bool head_ok = fread(&h, v1_bytes, 1, f) == 1;
  if (head_ok && memcmp(h.magic, "NBDYCKP2", 8) == 0)
    head_ok = h.header_bytes == sizeof h && fread((char *)&h + v1_bytes, sizeof h - v1_bytes, 1, f) == 1;
  else if (head_ok && memcmp(h.magic, "NBDYCKP1", 8) == 0) {
    head_ok = h.header_bytes == v1_bytes;
    h.has_root = 0; h.theta = 0.0f;
  } else head_ok = false;
  if (!head_ok)
    rc = fail(c, NBODY_ERR_INVALID, "nbody_load_checkpoint: %s is not a checkpoint of this engine (formats NBDYCKP1/2)", path);
  // the file's owned range must contain the context's: a whole-system file also feeds the slices of a sharded job
  else if (h.n_total != c->p.n_total || h.elem_bytes != (int32_t)eb || h.i_begin > c->p.i_begin ||
           h.i_begin + h.i_count < c->p.i_begin + c->p.i_count)
    rc = fail(c, NBODY_ERR_INVALID, "nbody_load_checkpoint: layout mismatch (file n=%d [%d,+%d) %d-byte, context n=%d [%d,+%d) %zu-byte)",
              h.n_total, h.i_begin, h.i_count, h.elem_bytes, c->p.n_total, c->p.i_begin, c->p.i_count, eb);
  else if (h.G != c->p.G || h.eps != c->p.eps)
    rc = fail(c, NBODY_ERR_INVALID, "nbody_load_checkpoint: the file was written with G = %.17g, eps = %.17g but the context has G = %.17g, "
              "eps = %.17g: resuming would not continue the same trajectory", h.G, h.eps, c->p.G, c->p.eps);
  else {
    const long skip = (long)((size_t)(c->p.i_begin - h.i_begin) * 4 * eb), rest = (long)(((size_t)h.i_count - ic) * 4 * eb) - skip;
    if (fread(posm.data(), 1, posm.size(), f) != posm.size() || fseek(f, skip, SEEK_CUR) != 0 ||
        fread(vel.data(), 1, vel.size(), f) != vel.size() || fseek(f, rest + skip, SEEK_CUR) != 0 ||
        fread(acc.data(), 1, acc.size(), f) != acc.size())
      rc = fail(c, NBODY_ERR_INVALID, "nbody_load_checkpoint: %s is truncated", path);
  }
  fclose(f);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->p.device));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(c->posm, posm.data(), posm.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->vel, vel.data(), vel.size(), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->acc, acc.data(), acc.size(), hipMemcpyHostToDevice));
  { const int rc2 = f64 ? note_masses(c, (const double *)posm.data()) : note_masses(c, (const float *)posm.data()); if (rc2) return rc2; }
  c->have_state = true; c->floor_eps2 = -1.0; c->step_open = false; c->step_local = false; c->sym_posg_valid = false;
  if (c->bh) nbody::bh_positions_changed(c->bh);
  c->steps_done = h.steps_done;
  // Barnes-Hut: the opening angle and the root of the next tree (the previous tree's CoM, OctreeSearch.cpp:77-79) are
  // part of the trajectory.  Only contexts that can run the walk take them over (a slice of a sharded job as well: it builds
  // the whole tree).
  if (c->p.precision == NBODY_PREC_F32) {
    c->theta = h.theta;
    if (h.has_root || c->bh) {
      { const int rc2 = ensure_bh(c); if (rc2) return rc2; }
      const float zero[3] = {0.f, 0.f, 0.f};
      HIP_TRY(c, nbody::bh_set_root_com(c->bh, h.has_root ? h.root_com : zero, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
    }
  }
  if (steps_done) *steps_done = h.steps_done;
  return NBODY_OK;
} catch (const std::bad_alloc &) {
  return fail(c, NBODY_ERR_NOMEM, "nbody_load_checkpoint: out of host memory");
}

int nbody_set_theta(nbody_ctx *c, float theta) {
  if (!c) return NBODY_ERR_INVALID;
  if (!(theta >= 0.0f)) return fail(c, NBODY_ERR_INVALID, "nbody_set_theta: theta must be >= 0");
  if (c->multi) {
    const int rc = multi_rc(c, nbody::multi_set_theta(c->multi, theta));
    if (!rc) c->theta = theta;
    return rc;
  }
  if (theta > 0.0f && c->p.precision != NBODY_PREC_F32)
    return fail(c, NBODY_ERR_UNSUPPORTED, "nbody_set_theta: Barnes-Hut needs an fp32 context");
  if (theta != c->theta) c->sym_posg_valid = false;   // the other force pass moves bodies without preparing the next all-pairs pass
  if (theta != c->theta && c->bh) nbody::bh_positions_changed(c->bh);   // ... nor leaving the next Barnes-Hut frame's Size
  c->theta = theta;
  return NBODY_OK;
}

int nbody_get_theta(nbody_ctx *c, float *theta) {
  if (!c || !theta) return NBODY_ERR_INVALID;
  *theta = c->theta;
  return NBODY_OK;
}

int nbody_bh_stats(nbody_ctx *c, int32_t *nodes, int32_t *levels, float root_com[3]) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) return multi_rc(c, nbody::multi_bh_stats(c->multi, nodes, levels, root_com));   // every device holds the whole tree
  if (!c->bh) return fail(c, NBODY_ERR_STATE, "nbody_bh_stats: no tree has been built on this context");
  if (int rc = use_device(c)) return rc;
  int n = 0, l = 0;
  HIP_TRY(c, nbody::bh_stats(c->bh, c->stream, &n, &l));
  if (nodes) *nodes = n;
  if (levels) *levels = l;
  if (root_com) HIP_TRY(c, nbody::bh_get_tree_com(c->bh, root_com, c->stream));
  return NBODY_OK;
}

int nbody_bh_leaf_boxes(nbody_ctx *c, float *boxes, size_t stride) {
  if (!c || !boxes || stride < 16) return c ? fail(c, NBODY_ERR_INVALID, "nbody_bh_leaf_boxes: null buffer or stride < 16") : NBODY_ERR_INVALID;
  if (c->multi) return multi_rc(c, nbody::multi_bh_leaf_boxes(c->multi, boxes, stride));
  if (!c->bh) return fail(c, NBODY_ERR_STATE, "nbody_bh_leaf_boxes: no tree has been built on this context (theta == 0?)");
  if (int rc0 = use_device(c)) return rc0;
  const size_t bytes = (size_t)c->p.n_total * 16;
  int rc = ensure_stage(c, bytes);
  if (rc) return rc;
  HIP_TRY(c, nbody::bh_leaf_boxes(c->bh, c->d_stage, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->h_stage, c->d_stage, bytes, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  if (stride == 16) memcpy(boxes, c->h_stage, bytes);
  else for (int i = 0; i < c->p.n_total; ++i) memcpy((char *)boxes + (size_t)i * stride, (const char *)c->h_stage + (size_t)i * 16, 16);
  return NBODY_OK;
}

int nbody_bh_leaf_order(nbody_ctx *c, int32_t *order) {
  if (!c || !order) return c ? fail(c, NBODY_ERR_INVALID, "nbody_bh_leaf_order: null buffer") : NBODY_ERR_INVALID;
  if (c->multi) return multi_rc(c, nbody::multi_bh_leaf_order(c->multi, order));
  if (!c->bh) return fail(c, NBODY_ERR_STATE, "nbody_bh_leaf_order: no tree has been built on this context (theta == 0?)");
  if (int rc = use_device(c)) return rc;
  HIP_TRY(c, nbody::bh_leaf_order(c->bh, order, c->stream));
  return NBODY_OK;
}

// Not in include/nbody.h: tuning aid of tools/even_items.py — the reference-clock stamps (start, end) every work item of the
// last symmetric force launch left (contexts created with time_kernels under NBODY_SYM_ITEM_CLOCKS=1), and the clock's rate.
__attribute__((visibility("default"))) int nbody_debug_sym_item_clocks(nbody_ctx *c, unsigned long long *out, int32_t cap, int32_t *n_items,
                                                                       int32_t *khz) {
  if (!c || c->multi || !c->clk || c->clk_items <= 0) return NBODY_ERR_INVALID;
  if (int rc = use_device(c)) return rc;
  if (n_items) *n_items = c->clk_items;
  if (khz) *khz = c->wall_khz;
  if (!out) return NBODY_OK;
  if (cap < 2 * c->clk_items) return NBODY_ERR_INVALID;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(out, c->clk + 8, 16 * (size_t)c->clk_items, hipMemcpyDeviceToHost));
  return NBODY_OK;
}

// Not in include/nbody.h: tuning aid of tools/bh_phases.py (meaningful in -DNBODY_BH_PHASE_CLOCKS builds only).
__attribute__((visibility("default"))) int nbody_debug_bh_clocks(nbody_ctx *c, long long out[16 + 3 * 512]) {
  if (!c || c->multi || !c->bh || !out) return NBODY_ERR_INVALID;
  if (int rc = use_device(c)) return rc;
  HIP_TRY(c, nbody::bh_debug_clocks(c->bh, out, c->stream));
  return NBODY_OK;
}

// Not in include/nbody.h either: how many frames of the larger systems were sorted starting from the previous frame's order, and how
// often a frame given up by that sort (a bucket ran over) was queued again with the cold sorts (tests/test_bh_gpu.py).
__attribute__((visibility("default"))) int nbody_debug_bh_sort_counts(nbody_ctx *c, long long *warm_frames, long long *retries) {
  if (!c || c->multi || !c->bh || !warm_frames || !retries) return NBODY_ERR_INVALID;
  nbody::bh_debug_sort_counts(c->bh, warm_frames, retries);
  return NBODY_OK;
}

// Not in include/nbody.h: fault injection for tests/test_bh_gpu.py — the words the creation memsets clear, set to something else between
// two frames (what a fill that ran late, or never, would leave): kind 1 = the warm sort's bucket counts := 3 each.
__attribute__((visibility("default"))) int nbody_debug_bh_poison(nbody_ctx *c, int kind) {
  if (!c || c->multi || !c->bh) return NBODY_ERR_INVALID;
  if (int rc = use_device(c)) return rc;
  HIP_TRY(c, nbody::bh_debug_poison(c->bh, kind, c->stream));
  return NBODY_OK;
}

int nbody_steps_done(nbody_ctx *c, int64_t *steps) {
  if (!c || !steps) return NBODY_ERR_INVALID;
  *steps = c->steps_done;
  return NBODY_OK;
}

int nbody_kernel_time(nbody_ctx *c, int32_t which, double *total_ms, int64_t *launches) {
  if (!c || which < 0 || which > 1) return NBODY_ERR_INVALID;
  if (c->multi) return multi_rc(c, nbody::multi_kernel_time(c->multi, which, total_ms, launches));
  if (int rc0 = use_device(c)) return rc0;
  int rc = timer_drain(c, which);
  if (rc) return rc;
  if (total_ms) *total_ms = c->timers[which].total_ms;
  if (launches) *launches = c->timers[which].launches;
  return NBODY_OK;
}

int nbody_kernel_time_reset(nbody_ctx *c) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) return multi_rc(c, nbody::multi_kernel_time_reset(c->multi));
  if (int rc0 = use_device(c)) return rc0;
  for (int w = 0; w < 2; ++w) {
    int rc = timer_drain(c, w);
    if (rc) return rc;
    c->timers[w].total_ms = 0.0;
    c->timers[w].launches = 0;
  }
  if (c->clk) { HIP_TRY(c, hipMemsetAsync(c->clk, 0, 16, c->stream)); HIP_TRY(c, hipStreamSynchronize(c->stream)); }
  return NBODY_OK;
}

int nbody_kernel_clock(nbody_ctx *c, double *shader_mhz, int32_t *compute_units) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) return multi_rc(c, nbody::multi_kernel_clock(c->multi, shader_mhz, compute_units));
  if (int rc0 = use_device(c)) return rc0;
  if (shader_mhz) *shader_mhz = 0.0;
  if (compute_units) *compute_units = c->cus;
  if (!c->clk) return fail(c, NBODY_ERR_STATE, "nbody_kernel_clock: the context was created without time_kernels");
  unsigned long long w[2] = {0, 0};
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipMemcpy(w, c->clk, sizeof w, hipMemcpyDeviceToHost));
  if (shader_mhz && w[1] != 0ull) *shader_mhz = (double)w[0] / (double)w[1] * (double)c->wall_khz * 1e-3;
  return NBODY_OK;
}

const char *nbody_force_kernel_name(const nbody_ctx *c) {
  if (!c) return "";
  if (c->multi) return nbody_force_kernel_name(nbody::multi_part(c->multi, 0));
  if (c->theta > 0.0f) return c->p.n_total <= 4096 ? "bh_walk_compact_kernel (+ bh_small_build_kernel)" : "bh_walk_lane_kernel (+ tree build)";
  if (c->sym) return c->p.precision == NBODY_PREC_F64 ? "forces_sym_f64_kernel" : "forces_sym_pk_kernel";
  if (c->wave) return c->p.precision == NBODY_PREC_F32 ? "forces_block_pk_kernel" : "forces_block_kernel";
  if (c->p.precision != NBODY_PREC_F64 && c->ipt % 2 == 0 && (c->p.eps > 0.0 || c->p.zero_mode != NBODY_ZERO_SELECT))
    return "forces_tile_pk_kernel";
  return "forces_tile_kernel";
}

int32_t nbody_block_pairs_describe(int32_t n_total, int32_t compute_units) {
  return n_total > 0 ? block_pairs(n_total, compute_units) : 0;
}

int nbody_get_algorithm(nbody_ctx *c, int32_t *algorithm, int32_t *super_tile) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) return nbody_get_algorithm(nbody::multi_part(c->multi, 0), algorithm, super_tile);
  if (algorithm) *algorithm = c->sym ? NBODY_ALGO_SYMMETRIC : NBODY_ALGO_TILED;
  if (super_tile) *super_tile = c->sym ? c->sym_bi : 0;
  return NBODY_OK;
}

int32_t nbody_sym_plan_is_even(const nbody_ctx *c) {
  if (!c) return 0;
  if (c->multi) return nbody_sym_plan_is_even(nbody::multi_part(c->multi, 0));
  return c->sym && c->sym_even ? 1 : 0;
}

int nbody_sym_pool_info(nbody_ctx *c, uint64_t *pool_bytes, int32_t *phases) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) return nbody_sym_pool_info(nbody::multi_part(c->multi, 0), pool_bytes, phases);
  if (pool_bytes) *pool_bytes = c->sym ? (uint64_t)c->sym_pool_elems * c->elem : 0;
  if (phases) *phases = c->sym ? (int32_t)c->sym_phase_item0.size() - 1 : 0;
  return NBODY_OK;
}

int nbody_equal_mass_form(nbody_ctx *c, int32_t *in_use) {
  if (!c || !in_use) return NBODY_ERR_INVALID;
  if (c->multi) return nbody_equal_mass_form(nbody::multi_part(c->multi, 0), in_use);
  *in_use = 0;
  if (!c->sym_general || c->theta > 0.0f) return NBODY_OK;
  HIP_TRY(c, hipSetDevice(c->p.device));
  HIP_TRY(c, hipMemcpyAsync(c->h_scratch, c->sym_general, 4, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  int32_t word;
  memcpy(&word, c->h_scratch, 4);
  *in_use = word == 0 ? 1 : 0;
  return NBODY_OK;
}

int nbody_get_launch_config(nbody_ctx *c, int32_t *tile, int32_t *i_per_thread, int32_t *j_split, int32_t *blocks,
                            int32_t *threads) {
  if (!c) return NBODY_ERR_INVALID;
  if (c->multi) return nbody_get_launch_config(nbody::multi_part(c->multi, 0), tile, i_per_thread, j_split, blocks, threads);   // per device
  int b = 0, t = 0;
  nbody::forces_geometry(make_launch(c), &b, &t);
  if (c->sym) { b = c->sym_items_n; t = 256; }
  if (tile) *tile = c->tile;
  if (i_per_thread) *i_per_thread = c->sym ? c->sym_bi / 256 : c->ipt;
  if (j_split) *j_split = c->j_split;
  if (blocks) *blocks = b;
  if (threads) *threads = t;
  return NBODY_OK;
}

}  // extern "C"

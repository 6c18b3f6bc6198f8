// Symmetric all-pairs force kernel for gfx950: every unordered pair {i, j} is evaluated ONCE and feeds both
// accelerations (a_i += G m_j s d, a_j -= G m_i s d with s = |d|^-3, d = r_j - r_i) — the same pair law,
// OctreeSearch.h:101-104, at 16 packed ops + 2 v_rsq_f32 per two pairs (four interactions) instead of 14 + 2 per
// two interactions; 14 + 2 when all bodies have the same mass (the equal-mass form, UNI: the common G m is taken out of
// the sums and applied by the update — which form runs is decided on the device before every pass).
//
// Structure (the plan — who evaluates what, where partial sums go, how they are added up — is sym_plan.h):
//   * a workgroup takes one work item: an i-set of 256*IPT bodies (register pairs, as in kernels.hip) against a strip
//     of 64-body subtiles, staged four at a time (a 256-body j tile) in LDS by DMA straight from HBM
//     (global_load_lds_dwordx4, double-buffered: the next tile lands under this tile's arithmetic, no registers held);
//   * symmetric step: every wave walks the tile's subtiles; at step k lane l meets body (l - k) & 63 of the subtile — a
//     per-lane ds_read_b128 from a doubled subtile image — and the body's running j-side sum travels with it from lane
//     to lane (v_mov_b32_dpp wave_ror:1; ds_add_f32 on LDS was measured ~190 cycles per wave instruction and is not
//     used).  After 64 steps the sums are home; the four waves' sums are added in wave order into the item's j-side
//     segment;
//   * strips inside the i-set's own block: the register pairs above the subtile's own pair meet it symmetrically (their
//     bodies are other bodies of the block), its own pair one-sided (every ordered pair of its 512 bodies from both
//     ends, d == 0 skipped), the pairs below are idle — they met these bodies when their own subtiles came up;
//   * results go to item-private segments of the partial-sum pool.  reduce_j_kernel folds the j-side segments into one
//     row per body (the send buffer of the all-to-all when the bodies are sharded over GPUs); update_sym_kernel adds a
//     body's i-side segments and the received rows in a fixed order.  No global atomics: bit-reproducible.
#include "kernels.h"

#include <cstdlib>

#include "../../include/nbody.h"
#include "pk_common.h"
#include "sym_common.h"

namespace nbody {

namespace {

#ifndef NBODY_SYM_UNROLL
#define NBODY_SYM_UNROLL 4
#endif
#ifndef NBODY_SYM_AHEAD
#define NBODY_SYM_AHEAD 0     // A/B builds only (tools/ab_read_ahead.sh): 1 / 2 read the next step's body from LDS one step ahead
#endif
#ifndef NBODY_SYM_UNROLL4
#define NBODY_SYM_UNROLL4 2   // four and more register pairs per lane: two steps in flight, not four
#endif

// BARE = symmetric strips without any d == 0 handling (two packed ops per register pair cheaper than Z_CLAMP).  A
// symmetric strip never contains a self pair (i-set and strip are disjoint), so d == 0 there means two DIFFERENT
// bodies on one point.  sym_prep_kernel looks for that before every pass and leaves the verdict in *dup_flag: the BARE
// launch runs only when there is none (run_if_dup == 0), the guarded launch only when there is one (run_if_dup == 1);
// dup_flag == nullptr runs unconditionally.  Results are those of the guarded kernel either way.
// KAHAN: blocked compensated summation.  Inside a 64-step subtile every sum is a plain packed-FMA chain, exactly the
// plain kernel's loop: the i-side register pairs collect 128 terms per component, a travelling j-side sum at most
// 64 * 2 * NP.  The rounding error of such a short chain (~sqrt(terms) * 2^-24 of a partial sum that is itself one of
// thousands) vanishes in what follows, which IS compensated: the i-side partials are Kahan-added to the lane's running
// sums after every subtile (Acc3pk<true>::fold), the four waves' j-side sums are added in double, and reduce_j_kernel /
// update_sym_kernel add the segments with compensation.  Each travelling sum is a register PAIR (lo: what the
// lanes' first bodies contributed, hi: the second bodies'), fed by three v_pk_fma_f32 per register pair and folded
// once, after the 64 steps: six v_mov_b32_dpp per step shared by the lane's NP register pairs.
// waves per SIMD: packed ops are 4-cycle, two waves keep a SIMD within 2 % of four.
constexpr int sym_waves(int np, bool kahan) {
  return np == 8 ? 2 : (np == 4 ? (kahan ? 2 : 3) : (kahan && np == 2 ? 3 : 4));
}

// One 64-body subtile against the lane's register pairs P0 .. NP-1.  At step k lane l meets body (l - k) & 63 of the
// subtile (sp points at the lane's entry in the upper copy of the doubled image); that body's running j-side sum sits in
// the same lane and moves on with it (wave_ror:1: lane l+1 takes lane l's value) after every step; after 64 moves lane l
// holds the sum of body l again (ox, oy, oz).
//   ONE = false, P0 = 0: a symmetric strip — every register pair meets the body symmetrically.
//   ONE = true: the subtile lies in the i-set's OWN block, in register pair P0's slots.  Pairs above P0 hold other bodies of
//   the block: symmetric, each unordered pair once (the pairs below P0 met these bodies when THEIR subtiles came up);
//   pair P0 itself runs one-sided — every ordered pair inside a register pair's 512 bodies is evaluated from both
//   ends, the self pair (d == 0) dropped by the guard — and credits nothing to the j side.
//   UNI: every body has the same mass.  The lane sums s d = |d|^-3 d on both sides — no mass factor per pair: 14 packed
//   ops per register pair and step instead of 16, and no -G m_i registers; the common G m is applied once per body
//   by the update.  The j-side sums travel with the i side's sign and are negated when they come home.
//   EVEN (even-share plans, sym_plan.h): the steps ka <= k < kb of the subtile only (multiples of four; the other steps are
//   another item's).  A sum that starts at step ka in lane l is body (l - ka)'s; after step kb - 1 and its move, lane l
//   holds body (l - kb)'s sum — the caller stores it there.
template <int NP, int P0, bool ONE, int ZMODE, bool BARE, bool UNI, bool EVEN = false>
__device__ __forceinline__ void sym_subtile(const f2 (&xi)[NP], const f2 (&yi)[NP], const f2 (&zi)[NP], const f2 (&nmi)[NP],
                                            Acc3pk<false> (&acc)[NP], const float4 *sp, f2 zp2, f2 one2, float &ox, float &oy,
                                            float &oz, int ka = 0, int kb = 64) {
  constexpr int NA = NP - P0;                                     // active register pairs
  // steps in flight: four for one or two register pairs, two from four pairs up; the own-block forms (a small share
  // of the work) keep one step in flight where many pairs are active, so that they never raise the kernel's register need
#ifndef NBODY_SYM_OWN_UNROLL_LOW
#define NBODY_SYM_OWN_UNROLL_LOW 2     // A/B builds: steps in flight of an own-block subtile with one or two active register pairs
#endif
  constexpr int kUnroll = ONE ? (NA >= 5 ? 1 : (NA >= 3 ? 2 : NBODY_SYM_OWN_UNROLL_LOW)) : ((NA >= 4) ? NBODY_SYM_UNROLL4 : NBODY_SYM_UNROLL);
  f2 qx = splat2(0.f), qy = splat2(0.f), qz = splat2(0.f);        // (lo, hi) partial sums
  // Reading the next step's body one step ahead (pinned with a scheduling barrier; the last read, sp[-64], is the image's
  // first copy of the lane's own entry: in bounds, unused) halves the wave-cycles parked on LDS (10.4 % -> 4.8 %) and
  // buys nothing: two waves per SIMD already cover the latency (same-box A/B, N = 2^20: 150.65 vs 150.30 ms,
  // profiles/r02_ab_lds_read_ahead.txt).  Off.
  constexpr bool kAhead = !EVEN && (NBODY_SYM_AHEAD == 2 ? !(NP == 8 && !BARE && !UNI) : (NBODY_SYM_AHEAD == 1 && UNI));
  float4 pj_next = sp[0];
  auto step = [&](int k) __attribute__((always_inline)) {
    const float4 pj = kAhead ? pj_next : sp[-k];
    if (kAhead) {
      pj_next = sp[-k - 1];
      __builtin_amdgcn_sched_barrier(0x47F);                      // everything may cross but LDS accesses: the read is issued here
    }
    f2 dx[NA], dy[NA], dz[NA], w[NA], u[NA];
#pragma unroll
    for (int a = 0; a < NA; ++a) { dx[a] = splat2(pj.x) - xi[P0 + a]; dy[a] = splat2(pj.y) - yi[P0 + a]; dz[a] = splat2(pj.z) - zi[P0 + a]; }
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      if (ZMODE == Z_SOFT) w[a] = fma2(dz[a], dz[a], zp2);
      else                 w[a] = dz[a] * dz[a];
      w[a] = fma2(dy[a], dy[a], w[a]);
      w[a] = fma2(dx[a], dx[a], w[a]);
    }
    if (ZMODE == Z_CLAMP) {
#pragma unroll
      for (int a = 0; a < NA; ++a)
        if (!BARE || (ONE && a == 0)) {                           // the one-sided pair holds the self pair: always guarded
          f2 nf;
          asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(nf) : "v"(w[a]), "v"(zp2), "v"(one2));
          w[a] = w[a] + nf;
        }
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) u[a] = f2{rsq_dev(w[a].x), rsq_dev(w[a].y)};
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      w[a] = u[a] * u[a];
      w[a] = w[a] * u[a];                                         // |d|^-3 (ordinary ops between rsq and the asm)
      if (!UNI) {
        if (!(ONE && a == 0)) u[a] = w[a] * nmi[P0 + a];          // -G m_i |d|^-3
        w[a] = mul_bcast_hi(w[a], f2{pj.z, pj.w});                //  G m_j |d|^-3
      }
    }
#pragma unroll
    for (int a = 0; a < NA; ++a) {
      acc[P0 + a].add(w[a], dx[a], dy[a], dz[a]);
      if (!(ONE && a == 0)) {
        const f2 sj = UNI ? w[a] : u[a];
        qx = fma2(sj, dx[a], qx); qy = fma2(sj, dy[a], qy); qz = fma2(sj, dz[a], qz);
      }
    }
    if (NA > (ONE ? 1 : 0)) {                                     // the sums move on with their body
      qx = f2{wave_ror1(qx.x), wave_ror1(qx.y)}; qy = f2{wave_ror1(qy.x), wave_ror1(qy.y)};
      qz = f2{wave_ror1(qz.x), wave_ror1(qz.y)};
    }
  };
  if constexpr (EVEN) {
    // ka, kb and kb - ka are multiples of four, kUnroll divides four: the steps in flight are written out (a loop with a
    // trip count known only at run time is not unrolled around the cross-lane moves — they are convergent operations)
    for (int k = ka; k < kb; k += kUnroll) {
#pragma unroll
      for (int u = 0; u < kUnroll; ++u) step(k + u);
    }
  } else {
#pragma unroll kUnroll
    for (int k = 0; k < 64; ++k) step(k);
  }
  ox = qx.x + qx.y; oy = qy.x + qy.y; oz = qz.x + qz.y;
  if (UNI) { ox = -ox; oy = -oy; oz = -oz; }
}

// own-block subtile in register pair pc's slots: pick the instantiation (pc is wave-uniform)
template <int NP, int PC, int ZMODE, bool BARE, bool UNI, bool EVEN = false>
__device__ __forceinline__ void own_block_subtile(int pc, const f2 (&xi)[NP], const f2 (&yi)[NP], const f2 (&zi)[NP],
                                                  const f2 (&nmi)[NP], Acc3pk<false> (&acc)[NP], const float4 *sp, f2 zp2,
                                                  f2 one2, float &ox, float &oy, float &oz, int ka = 0, int kb = 64) {
  if (pc == PC) sym_subtile<NP, PC, true, ZMODE, BARE, UNI, EVEN>(xi, yi, zi, nmi, acc, sp, zp2, one2, ox, oy, oz, ka, kb);
  else if constexpr (PC + 1 < NP) own_block_subtile<NP, PC + 1, ZMODE, BARE, UNI, EVEN>(pc, xi, yi, zi, nmi, acc, sp, zp2, one2, ox, oy, oz, ka, kb);
}

using lds_f4 = __attribute__((address_space(3))) float4;
using glb_f4 = const __attribute__((address_space(1))) float4;

// UNI = the equal-mass form (sym_subtile).  Whether the bodies' masses are all equal is sym_prep_kernel's finding,
// *general (0 = equal): the UNI launch runs only when it is clear (run_if_general == 0), the general launch only when it
// is raised; general == nullptr runs unconditionally (the host already knows).
// run_if_dup == -1 (where registers allow it: up to four register pairs per lane, and the equal-mass form at eight too —
// it has no -G m_i registers to keep): ONE launch holds both loops and the detector's verdict picks — systems of
// 12288 ... 24576 bodies step in ~100 us, and a twin that returns at its first instruction still costs a launch (4-6 us
// of kernel + the gap in front of it).  BARE is then the form that runs when no two bodies coincide.
// EVEN: the items are an even-share plan's (sym_plan.h): a run of subtiles in the row's ring order — whether a subtile lies in
// the i-set's own block is asked subtile by subtile, past the system's last granule (`wrap` bodies) the run goes on at body
// 0 — of which the first starts at step k0 and the last ends at step 64 - k_skip.
template <int NP, int ZMODE, bool BARE, bool KAHAN, bool UNI, bool EVEN = false>
__global__ __launch_bounds__(kBlock)
__attribute__((amdgpu_waves_per_eu(sym_waves(NP, KAHAN), sym_waves(NP, KAHAN))))
void forces_sym_pk_kernel(const float4 *__restrict__ posg, float4 *__restrict__ pool, const SymItem *__restrict__ items,
                          float zp, const int *__restrict__ dup_flag, int run_if_dup, const int *__restrict__ general,
                          int run_if_general, unsigned long long *__restrict__ clk, int wrap = 0) {
  constexpr bool kCanMerge = BARE && (NP <= 4 || UNI || EVEN);   // the general form at NP = 8 would spill (52 B of scratch); EVEN: see launch_forces_sym
  const bool merged = kCanMerge && run_if_dup < 0;
  if (!merged && dup_flag != nullptr && ((*dup_flag != 0) ? 1 : 0) != run_if_dup) return;
  if (general != nullptr && ((*general != 0) ? 1 : 0) != run_if_general) return;
  const bool guard_all = merged && *dup_flag != 0;               // merged launch, coincident bodies: the guarded loops
  const ClockStamp stamp = clock_begin(clk);
  __shared__ float4 sh_pos[2][4][128];   // double-buffered subtile images, doubled: entries l and l+64 hold body l
  __shared__ float sh_acc[4][3][kJT];    // per-WAVE j-side sums of the tile (private: no ordering between waves needed)

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const SymItem *__restrict__ itp = items + blockIdx.x;          // wave-uniform: scalar loads
  const int i0 = itp->i0, j0 = itp->j0, n_sub = itp->n_sub;
  const unsigned int slot_i = itp->slot_i, slot_j = itp->slot_j;
  const bool own_strip = (itp->flags & kSymOneSided) != 0;      // the strip lies inside the i-set's own block
  const int k_first = EVEN ? itp->k0 : 0, k_last = EVEN ? 64 - itp->k_skip : 64;
  const int n_tiles = (n_sub + 3) >> 2;

  // Tile c of the strip -> LDS buffer c & 1: wave w brings subtile w, twice (the doubled image), 1 KiB per DMA.
  auto stage = [&](int c, int wave, int lane) {
    if (4 * c + wave < n_sub) {
      int jb = j0 + (4 * c + wave) * 64;
      if (EVEN && jb >= wrap) jb -= wrap;
      glb_f4 *src = (glb_f4 *)(posg + jb + lane);
      __builtin_amdgcn_global_load_lds(src, (lds_f4 *)&sh_pos[c & 1][wave][0], 16, 0, 0);
      __builtin_amdgcn_global_load_lds(src, (lds_f4 *)&sh_pos[c & 1][wave][64], 16, 0, 0);
    }
  };
  stage(0, wave, lane);

  // the d == 0 guard's two constants live in VGPRs (packed ops take no literals).  The BARE kernels need them only in
  // own-block strips: there they are made on the spot, so that the symmetric strips have four registers more
  f2 zp2 = splat2(zp), one2 = splat2(1.0f);
  if (!BARE || kCanMerge) asm volatile("" : "+v"(zp2), "+v"(one2));

  f2 xi[NP], yi[NP], zi[NP], nmi[NP];
  // i-side sums.  Plain: `a` runs through the whole strip.  KAHAN: `a` collects one subtile (64 steps, 128 terms per
  // component) with plain packed FMAs — the loop is the plain kernel's — and is then folded, compensated, into `ka`:
  // twelve packed ops per register pair and subtile instead of nine more per STEP.
  Acc3pk<false> a[NP];
  Acc3pk<true> ka[KAHAN ? NP : 1];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const float4 pa = posg[i0 + t + (2 * p) * kBlock], pb = posg[i0 + t + (2 * p + 1) * kBlock];
    xi[p] = f2{pa.x, pb.x}; yi[p] = f2{pa.y, pb.y}; zi[p] = f2{pa.z, pb.z};
    nmi[p] = UNI ? splat2(0.f) : f2{-pa.w, -pb.w};             // -G m_i: the j side gets a_j -= G m_i s d
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    asm volatile("" ::"v"(xi[p]), "v"(yi[p]), "v"(zi[p]));
    if (!UNI) asm volatile("" ::"v"(nmi[p]));
  }
  __syncthreads();                                               // tile 0 has landed (the barrier waits for the DMA)

  for (int c = 0; c < n_tiles; ++c) {
    const int buf = c & 1;
    const int nsub = min(4, n_sub - 4 * c);
    // lane and wave are derived afresh from the thread id in every tile (opaque to the optimiser): hoisted out of the
    // loop, the addresses computed from them would sit in VGPRs across the arithmetic and push it into scratch
    int tc = threadIdx.x;
    asm volatile("" : "+v"(tc));
    const int lane = tc & 63, wave = tc >> 6;
    if (c + 1 < n_tiles) stage(c + 1, wave, lane);               // in flight under this tile's arithmetic

    for (int sub = 0; sub < nsub; ++sub) {
      const float4 *sp = &sh_pos[buf][sub][lane + 64];
      float ox, oy, oz;
      // EVEN: this subtile's place in the ring, whether it lies in the own block, and which of its steps are this item's
      int off = j0 - i0 + (4 * c + sub) * 64;                      // offset from the i-set's first body (wave-uniform)
      if (EVEN && j0 + (4 * c + sub) * 64 >= wrap) off -= wrap;
      const bool own_block = EVEN ? (unsigned int)off < (unsigned int)(NP * 512) : own_strip;
      const int kfrom = (EVEN && 4 * c + sub == 0) ? k_first : 0, kto = (EVEN && 4 * c + sub == n_sub - 1) ? k_last : 64;
      if (kCanMerge && guard_all) {
        if (!own_block)
          sym_subtile<NP, 0, false, ZMODE, false, UNI, EVEN>(xi, yi, zi, nmi, a, sp, zp2, one2, ox, oy, oz, kfrom, kto);
        else
          own_block_subtile<NP, 0, ZMODE, false, UNI, EVEN>(off >> 9, xi, yi, zi, nmi, a, sp, zp2, one2, ox, oy, oz, kfrom, kto);
      } else if (!own_block)
        sym_subtile<NP, 0, false, ZMODE, BARE, UNI, EVEN>(xi, yi, zi, nmi, a, sp, zp2, one2, ox, oy, oz, kfrom, kto);
      else {   // the subtile's bodies sit in the slots of register pair (offset from the i-set's first body) / 512
        f2 zq = zp2, oq = one2;
        if (BARE && !kCanMerge) { zq = splat2(zp); oq = splat2(1.0f); asm volatile("" : "+v"(zq), "+v"(oq)); }
        own_block_subtile<NP, 0, ZMODE, BARE, UNI, EVEN>(off >> 9, xi, yi, zi, nmi, a, sp, zq, oq, ox, oy, oz, kfrom, kto);
      }
      if (KAHAN) {
#pragma unroll
        for (int p = 0; p < NP; ++p) { ka[p].fold(a[p]); a[p] = Acc3pk<false>(); }
      }
      // (EVEN: after step kto - 1 the lane holds the sum of body lane - kto of the subtile)
      const int home = EVEN ? sub * 64 + ((lane - kto) & 63) : sub * 64 + lane;
      sh_acc[wave][0][home] = ox; sh_acc[wave][1][home] = oy; sh_acc[wave][2][home] = oz;
    }
    __syncthreads();   // the four waves' tile sums are complete; the next tile has landed
    {
      // thread e adds body e's sums (waves in fixed order) and stores them in the item's j-side segment
      int e = threadIdx.x;
      asm volatile("" : "+v"(e));   // recomputed here rather than kept in a register across the tile
      if (e < nsub * 64) {
        float4 o;
        if (KAHAN) {                // four terms, summed in double and rounded once
          o.x = (float)(((double)sh_acc[0][0][e] + (double)sh_acc[1][0][e]) + ((double)sh_acc[2][0][e] + (double)sh_acc[3][0][e]));
          o.y = (float)(((double)sh_acc[0][1][e] + (double)sh_acc[1][1][e]) + ((double)sh_acc[2][1][e] + (double)sh_acc[3][1][e]));
          o.z = (float)(((double)sh_acc[0][2][e] + (double)sh_acc[1][2][e]) + ((double)sh_acc[2][2][e] + (double)sh_acc[3][2][e]));
        } else {
          o.x = ((sh_acc[0][0][e] + sh_acc[1][0][e]) + sh_acc[2][0][e]) + sh_acc[3][0][e];
          o.y = ((sh_acc[0][1][e] + sh_acc[1][1][e]) + sh_acc[2][1][e]) + sh_acc[3][1][e];
          o.z = ((sh_acc[0][2][e] + sh_acc[1][2][e]) + sh_acc[2][2][e]) + sh_acc[3][2][e];
        }
        o.w = 0.f;
        pool[(size_t)slot_j + (size_t)c * kJT + e] = o;
      }
      __syncthreads(); // sh_acc is rewritten by the next tile
    }
  }

  // i-side sums -> the item's i-side segment (indices recomputed from the thread id: kept alive across the strip they
  // would cost the registers the compiler otherwise spills)
  int te = threadIdx.x;
  asm volatile("" : "+v"(te));
  float4 *__restrict__ Pi = pool + (size_t)slot_i + te;
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const f2 sx = KAHAN ? ka[KAHAN ? p : 0].x : a[p].x, sy = KAHAN ? ka[KAHAN ? p : 0].y : a[p].y, sz = KAHAN ? ka[KAHAN ? p : 0].z : a[p].z;
    Pi[(2 * p) * kBlock] = make_float4(sx.x, sy.x, sz.x, 0.f);
    Pi[(2 * p + 1) * kBlock] = make_float4(sx.y, sy.y, sz.y, 0.f);
  }
  clock_end(clk, stamp, (int)blockIdx.x);
}

// one force launch: the even-share form of the kernel for an even-share plan (plain fp32, two and more register pairs)
template <int NPV, int ZM, bool BARE, bool KH, bool UNI>
void launch_sym_kernel(const SymLaunch &L, dim3 grid, dim3 block, hipStream_t s, const SymItem *items, float zp, const int *flag,
                       int run_if, const int *gate) {
  if constexpr (NPV >= 2) {
    if (L.even) {
      hipLaunchKernelGGL((forces_sym_pk_kernel<NPV, ZM, BARE, KH, UNI, true>), grid, block, 0, s, (const float4 *)L.posg,
                         (float4 *)L.pool, items, zp, flag, run_if, gate, UNI ? 0 : 1, (unsigned long long *)L.clk, L.wrap);
      return;
    }
  }
  hipLaunchKernelGGL((forces_sym_pk_kernel<NPV, ZM, BARE, KH, UNI, false>), grid, block, 0, s, (const float4 *)L.posg,
                     (float4 *)L.pool, items, zp, flag, run_if, gate, UNI ? 0 : 1, (unsigned long long *)L.clk, 0);
}

}  // namespace

hipError_t launch_forces_sym64(const SymLaunch &L, hipStream_t s);   // kernels_sym64.hip

hipError_t launch_forces_sym(const SymLaunch &L, hipStream_t s) {
  if (L.precision == NBODY_PREC_F64) return launch_forces_sym64(L, s);
  if (L.n_total <= 0 || L.n_items <= 0 || L.n_pad < L.n_total || !L.posg || !L.pool || !L.items) return hipErrorInvalidValue;
  if (L.np != 1 && L.np != 2 && L.np != 4 && L.np != 8) return hipErrorInvalidValue;
  if (L.np == 8 && L.kahan) return hipErrorInvalidValue;
  if (L.even && (L.np < 2 || L.wrap <= 0 || L.wrap % 64 != 0 || L.phase != 0)) return hipErrorInvalidValue;
  if (L.phase < 0 || L.phase > 2 || (L.phase != 0 && (L.fused || L.n_local < 0 || L.n_local > L.n_items))) return hipErrorInvalidValue;
  // which items this call launches, and which bodies it prepares (SymLaunch::phase)
  int item0 = L.phase == 2 ? L.n_local : 0, item1 = L.phase == 1 ? L.n_local : L.n_items;
  if (L.item1 >= 0) { item0 = L.item0; item1 = L.item1; }
  if (item0 < 0 || item1 < item0 || item1 > L.n_items) return hipErrorInvalidValue;
  const bool fold = L.do_fold < 0 ? L.phase != 1 : L.do_fold != 0;
  const SymItem *items = (const SymItem *)L.items + item0;
  dim3 grid(item1 - item0), block(kBlock), pgrid((L.n_pad + kBlock - 1) / kBlock);
  const bool detect = L.eps2 == 0.0 && L.dup_table != nullptr;
  int *flag_all = detect ? (int *)((unsigned long long *)L.dup_table + L.dup_slots) : nullptr;
  int *flag_own = detect ? flag_all + 2 : nullptr;                // the own slice's verdict (phase 1): words 2, 3 behind the table
  int *flag = L.phase == 1 ? flag_own : flag_all;                 // what this call's force kernels look at
  const int b0 = L.phase == 0 ? 0 : L.own_begin, b1 = L.phase == 0 ? L.n_pad : L.own_begin + L.own_count;
  const int inside = L.phase == 2 ? 0 : 1, check_mass = L.phase == 0 ? 1 : 2, mass_ref = L.phase == 0 ? 0 : L.own_begin;
  // positions -> (x, y, z, G m) with far-away zero-mass padding; the coincident-body detector rides along
  // equal masses (L.general: the device's finding, raised by the preparation kernel; L.uni_host: what the host knows —
  // 1 equal and nobody else can write the buffer, 0 not equal / not applicable, -1 ask the device)
  int *general = (int *)L.general;
  const bool run_uni = general != nullptr && L.uni_host != 0, run_gen = general == nullptr || L.uni_host != 1;
  const int *gate = (run_uni && run_gen) ? general : nullptr;     // both forms launched: each looks at the finding
  if (L.skip_prep || !L.do_prep) {
    // the previous update_sym_fused_kernel left posg and the detector's verdict for exactly these positions (skip_prep), or
    // an earlier call of this pass has prepared (do_prep = 0)
  } else if (detect) {      // the table and its flag words are zero: cleared at creation and by every pass's fold
    hipLaunchKernelGGL(sym_prep_kernel<true>, pgrid, block, 0, s, (const float4 *)L.posm, (float4 *)L.posg, L.n_total,
                       L.n_pad, (float)L.G, (unsigned long long *)L.dup_table, (unsigned int)(L.dup_slots - 1), flag_all, general,
                       b0, b1, inside, check_mass, L.phase == 1 ? flag_own : (int *)nullptr, mass_ref);
  } else {
    hipLaunchKernelGGL(sym_prep_kernel<false>, pgrid, block, 0, s, (const float4 *)L.posm, (float4 *)L.posg, L.n_total,
                       L.n_pad, (float)L.G, (unsigned long long *)nullptr, 0u, (int *)nullptr, general, b0, b1, inside,
                       check_mass, (int *)nullptr, mass_ref);
  }
  if (grid.x == 0) {                                              // nothing to launch in this go (phase 2 of a plan without remote strips)
  } else {
#define NBODY_SYM_K(NPV, ZM, BARE, KH, UNI, ZP, FLAG, RUNIF)                                                     \
  launch_sym_kernel<NPV, ZM, BARE, KH, UNI>(L, grid, block, s, items, (float)(ZP), (const int *)(FLAG), RUNIF, gate)
  bool do_uni = run_uni, do_gen = run_gen;                         // which forms the next NBODY_SYM_NP launches
#define NBODY_SYM_U(NPV, ZM, BARE, KH, ZP, FLAG, RUNIF)                                                          \
  do {                                                                                                           \
    if (do_uni) NBODY_SYM_K(NPV, ZM, BARE, KH, true, ZP, FLAG, RUNIF);                                           \
    if (do_gen) NBODY_SYM_K(NPV, ZM, BARE, KH, false, ZP, FLAG, RUNIF);                                          \
  } while (0)
#define NBODY_SYM(NPV, ZM, BARE, ZP, FLAG, RUNIF)                                                                \
  do { if (L.kahan) NBODY_SYM_U(NPV, ZM, BARE, true, ZP, FLAG, RUNIF); else NBODY_SYM_U(NPV, ZM, BARE, false, ZP, FLAG, RUNIF); } while (0)
#define NBODY_SYM_NP(ZM, BARE, ZP, FLAG, RUNIF)                                                                  \
  do {                                                                                                           \
    if (L.np == 1) NBODY_SYM(1, ZM, BARE, ZP, FLAG, RUNIF);                                                      \
    else if (L.np == 2) NBODY_SYM(2, ZM, BARE, ZP, FLAG, RUNIF);                                                 \
    else if (L.np == 4) NBODY_SYM(4, ZM, BARE, ZP, FLAG, RUNIF);                                                 \
    else NBODY_SYM_U(8, ZM, BARE, false, ZP, FLAG, RUNIF);                                                       \
  } while (0)
  if (L.eps2 > 0.0) {
    NBODY_SYM_NP(Z_SOFT, false, L.eps2, nullptr, 0);
  } else if (detect) {
    // exact d == 0 semantics at the unguarded kernel's price: both forms are launched — exactly one of them runs (the
    // other returns at its first instruction)
    if (L.np <= 4 || L.even) {
      NBODY_SYM_NP(Z_CLAMP, true, -0x1p126, flag, -1);            // one launch holds both loops (see the kernel)
    } else {
      do_gen = false;                                             // equal-mass form: one launch at any size
      NBODY_SYM_NP(Z_CLAMP, true, -0x1p126, flag, -1);
      do_uni = false; do_gen = run_gen;                           // general form: twins
      NBODY_SYM_NP(Z_CLAMP, true, -0x1p126, flag, 0);
      NBODY_SYM_NP(Z_CLAMP, false, -0x1p126, flag, 1);
    }
  } else {
    NBODY_SYM_NP(Z_CLAMP, false, -0x1p126, nullptr, 0);
  }
#undef NBODY_SYM_NP
#undef NBODY_SYM
#undef NBODY_SYM_U
#undef NBODY_SYM_K
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || L.fused || !fold) return e;              // fused: update_sym_fused_kernel folds the j-side rows
  dim3 rgrid((L.n_total + 63) / 64);                               // one workgroup per 64-body granule (sym_common.h, row folds)
  if (L.kahan)
    hipLaunchKernelGGL((reduce_j_kernel<float, true>), rgrid, block, 0, s, (const float4 *)L.pool, (float4 *)L.send,
                       (const unsigned int *)L.j_ptr, (const unsigned int *)L.j_off, L.n_total,
                       (unsigned long long *)L.dup_table, detect && L.clear_detector ? L.dup_slots + 8 : 0, L.fold_accumulate);
  else
    hipLaunchKernelGGL((reduce_j_kernel<float, false>), rgrid, block, 0, s, (const float4 *)L.pool, (float4 *)L.send,
                       (const unsigned int *)L.j_ptr, (const unsigned int *)L.j_off, L.n_total,
                       (unsigned long long *)L.dup_table, detect && L.clear_detector ? L.dup_slots + 8 : 0, L.fold_accumulate);
  return hipGetLastError();
}

hipError_t launch_update_sym(const SymLaunch &L, void *posm, void *vel, void *acc, int i_begin, int i_count, float dt,
                             hipStream_t s) {
  if (i_count <= 0) return hipErrorInvalidValue;
  dim3 grid((i_count + 63) / 64), block(kBlock);                  // one workgroup per own granule
  const unsigned int *ip = (const unsigned int *)L.i_ptr, *io = (const unsigned int *)L.i_off;
  if (L.fused) {
    if (L.precision == NBODY_PREC_F64 || i_begin != 0 || i_count != L.n_total || L.n_src != 1) return hipErrorInvalidValue;
    const bool detect = L.eps2 == 0.0 && L.dup_table != nullptr && L.dup_table_next != nullptr;
#define NBODY_FUSED(KH, DT)                                                                                       \
    hipLaunchKernelGGL((update_sym_fused_kernel<KH, DT>), grid, dim3(2 * kBlock), 0, s, (float4 *)posm, (float4 *)vel, (float4 *)acc, \
                       (float4 *)L.posg, (const float4 *)L.pool, ip, io, (const unsigned int *)L.j_ptr,             \
                       (const unsigned int *)L.j_off, L.n_total, (float)L.G, dt, dt > 0.0f ? 1 : 0,                  \
                       (unsigned long long *)L.dup_table_next, (unsigned int)(L.dup_slots - 1),                      \
                       (unsigned long long *)L.dup_table, L.dup_slots + 8, (const int *)L.general)
    if (L.kahan) { if (detect) NBODY_FUSED(true, true); else NBODY_FUSED(true, false); }
    else { if (detect) NBODY_FUSED(false, true); else NBODY_FUSED(false, false); }
#undef NBODY_FUSED
    return hipGetLastError();
  }
  if (L.precision == NBODY_PREC_F64)
    hipLaunchKernelGGL((update_sym_kernel<double, false>), grid, block, 0, s, (double4 *)posm, (double4 *)vel, (double4 *)acc,
                       (const double4 *)L.pool, ip, io, (const double4 *)L.recv, i_begin, i_count, L.n_src, (double)dt,
                       dt > 0.0f ? 1 : 0, (const int *)L.general, L.G);
  else if (L.kahan)
    hipLaunchKernelGGL((update_sym_kernel<float, true>), grid, block, 0, s, (float4 *)posm, (float4 *)vel, (float4 *)acc,
                       (const float4 *)L.pool, ip, io, (const float4 *)L.recv, i_begin, i_count, L.n_src, dt,
                       dt > 0.0f ? 1 : 0, (const int *)L.general, (float)L.G);
  else
    hipLaunchKernelGGL((update_sym_kernel<float, false>), grid, block, 0, s, (float4 *)posm, (float4 *)vel, (float4 *)acc,
                       (const float4 *)L.pool, ip, io, (const float4 *)L.recv, i_begin, i_count, L.n_src, dt,
                       dt > 0.0f ? 1 : 0, (const int *)L.general, (float)L.G);
  return hipGetLastError();
}

}  // namespace nbody

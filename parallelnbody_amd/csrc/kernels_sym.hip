// Symmetric all-pairs force kernel for gfx950: every unordered pair {i, j} is evaluated ONCE and feeds both
// accelerations (a_i += G m_j s d, a_j -= G m_i s d with s = |d|^-3, d = r_j - r_i) — the same pair law,
// OctreeSearch.h:101-104, at 18 packed ops + 2 v_rsq_f32 per two pairs (four interactions) instead of 14 + 2 per
// two interactions.
//
// Structure:
//   * bodies are cut into T super tiles of S bodies.  Super-tile pair {a, b} belongs to a if b lies in the forward
//     half of the ring from a (sym_assigned below) — a circulant assignment, so every super tile (and every rank,
//     which owns a contiguous run of them) gets the same number of pairs.  Workgroup (si, sj) owns every body
//     pair between si (as the i side) and sj;
//   * it walks i-sets of 256*IPT bodies of si (register pairs, as in kernels.hip) against 256-body j tiles of sj
//     staged in LDS; tiles wholly after the i-set (always, when si < sj) run the SYMMETRIC step, the tiles that
//     overlap the i-set's own range run the plain one-sided step (all ordered pairs, d == 0 skipped), tiles
//     before it are skipped (they were handled when their bodies were the i-set);
//   * symmetric step: in round r wave w takes 64-body subtile (r + w) & 3; at step k lane l meets body
//     (l - k) & 63 of it — a per-lane ds_read_b128 from a doubled subtile image — and the body's running j-side
//     sum travels with it from lane to lane (three v_mov_b32_dpp wave_ror:1 per step; ds_add_f32 on LDS was
//     measured ~190 cycles per wave instruction and is not used).  After 64 steps the sums are home and are added
//     to the tile's LDS accumulators; within a round no two waves touch the same subtile and rounds are separated
//     by a barrier, so the summation order is fixed;
//   * results go to workgroup-private rows: i-side sums of the rank's own bodies to part_i[sj], j-side sums (of any
//     body) to part_j[si] (read-modify-write by the same thread every time).  reduce_j_kernel folds the rank's
//     j-side rows into one row per destination rank (the send buffer of the all-to-all when the bodies are
//     sharded over GPUs); update_sym_kernel adds a body's i-side rows and the received rows in a fixed order.
//     No global atomics: bit-reproducible for a given number of ranks.
#include "kernels.h"

#include <cstdlib>

#include "../../include/nbody.h"
#include "pk_common.h"
#include "sym_common.h"

namespace nbody {

namespace {

#ifndef NBODY_SYM_WAVES
#define NBODY_SYM_WAVES 4
#endif
#ifndef NBODY_SYM_UNROLL
#define NBODY_SYM_UNROLL 4
#endif
#ifndef NBODY_SYM_UNROLL4
#define NBODY_SYM_UNROLL4 2   // four register pairs per lane: 128 VGPRs hold two steps in flight, not four
#endif
// Zero-mass padding bodies sit far outside any scene when the symmetric tiles run without a d == 0 guard (BARE):
// a pad at the origin would coincide with a body at the origin — the reference pins body 0 there — and 0 * inf = NaN.
// At 1e18 every pad-to-body term is |d|^-3 = 1e-55 -> 0 times a zero mass, exactly 0.
constexpr float kPadFar = 1.0e18f;

// BARE = symmetric tiles without any d == 0 handling (two packed ops per register pair cheaper than Z_CLAMP).  The
// symmetric tiles never contain a self pair (i-set and j tile are disjoint), so d == 0 there means two DIFFERENT
// bodies on one point.  dup_detect_kernel looks for that before every pass and leaves the verdict in *dup_flag:
// the BARE launch runs only when there is none (run_if_dup == 0), the guarded launch only when there is one
// (run_if_dup == 1); dup_flag == nullptr runs unconditionally.  Results are those of the guarded kernel either way.
// KAHAN: every accumulation is compensated — the i-side register pairs (Acc3pk<true>), the running j-side sums (the
// compensation term travels with the sum: six DPP moves per step instead of three) and the diagonal one-sided tiles.
//
// The j side of the plain (not KAHAN) kernel: each travelling sum is a register PAIR (lo: what the lanes' first bodies
// contributed, hi: the second bodies'), fed by three v_pk_fma_f32 per register pair and folded once, after the 64 steps.
// Six v_mov_b32_dpp per step instead of three, but no scalar FMAs: a wave64 v_fmac_f32 only issues at its 2-cycle rate
// next to another wave's 2-cycle op — in this packed instruction stream it costs ~3.5 cycles, twelve of them per step
// more than six packed FMAs (tools/microbench6.hip; profiles/r01_microbench_sym_inner_loop.txt).
// waves per SIMD: 4 (128 VGPRs); the Kahan form needs 164 VGPRs with two register pairs per lane (3 waves) and ~230
// with four (2 waves — packed ops are 4-cycle, two waves keep the SIMD within 2 % of four)
constexpr int sym_waves(int np, bool kahan) {
  return np == 8 ? 2 : (!kahan ? NBODY_SYM_WAVES : (np == 4 ? 2 : (np == 2 ? 3 : NBODY_SYM_WAVES)));
}

template <int NP, int ZMODE, bool BARE, bool KAHAN, bool JPK>
__global__ __launch_bounds__(kBlock)
__attribute__((amdgpu_waves_per_eu(sym_waves(NP, KAHAN), sym_waves(NP, KAHAN))))
void forces_sym_pk_kernel(const float4 *__restrict__ posm, float4 *__restrict__ part_i, float4 *__restrict__ part_j,
                          const int2 *__restrict__ pairs, int n_total, int S, int n_pad, int own_tile0, int n_own_pad,
                          float gscale, float zp, const int *__restrict__ dup_flag, int run_if_dup) {
  if (dup_flag != nullptr && ((*dup_flag != 0) ? 1 : 0) != run_if_dup) return;
  const float4 pad = BARE ? make_float4(kPadFar, kPadFar, kPadFar, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
  constexpr int IPT = 2 * NP;
  constexpr int BI = kBlock * IPT;
  constexpr int kUnroll = (NP >= 4) ? NBODY_SYM_UNROLL4 : NBODY_SYM_UNROLL;
  __shared__ float4 sh_pos[2][4][128];   // double-buffered subtile images, doubled: entries l and l+64 hold body l
  __shared__ float sh_acc[4][3][kJT];    // per-WAVE j-side sums of the tile (private: no ordering between waves needed)

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int2 pr = pairs[blockIdx.x];
  const int si = pr.x, sj = pr.y;
  const bool diag_super = si == sj;
  const int own0 = own_tile0 * S;                                                   // first body this rank owns
  float4 *__restrict__ Pi = part_i + (size_t)sj * n_own_pad;                         // i-side sums, index i - own0
  float4 *__restrict__ Pj = part_j + (size_t)(si - own_tile0) * n_pad;               // j-side sums, index j

  // This workgroup's j-side row segment starts from zero; element e is only ever touched by thread e % 256.
  for (int e = t; e < S; e += kBlock) Pj[(size_t)sj * S + e] = make_float4(0.f, 0.f, 0.f, 0.f);

  f2 zp2 = splat2(zp), one2 = splat2(1.0f);
  asm volatile("" : "+v"(zp2), "+v"(one2));

  const int tiles_in_super = S / kJT;
  int c_end = (n_total - sj * S + kJT - 1) / kJT;               // tiles of sj that hold at least one body
  if (c_end > tiles_in_super) c_end = tiles_in_super;

  for (int b = 0; b < S / BI; ++b) {
    const int i0 = si * S + b * BI;
    if (i0 >= n_total) break;
    f2 xi[NP], yi[NP], zi[NP], nmi[NP];
    Acc3pk<KAHAN> a[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      // one 16-byte load from a clamped index, then a select: `ia < n_total ? posm[ia] : pad` compiles to four
      // predicated dword loads and keeps ia alive for the epilogue
      const int ia = i0 + t + (2 * p) * kBlock, ib = ia + kBlock;
      float4 pa = posm[min(ia, n_total - 1)], pb = posm[min(ib, n_total - 1)];
      if (ia >= n_total) pa = pad;
      if (ib >= n_total) pb = pad;
      xi[p] = f2{pa.x, pb.x}; yi[p] = f2{pa.y, pb.y}; zi[p] = f2{pa.z, pb.z};
      nmi[p] = f2{-gscale * pa.w, -gscale * pb.w};             // -G m_i: the j side gets a_j -= G m_i s d
    }
#pragma unroll
    for (int p = 0; p < NP; ++p) asm volatile("" ::"v"(xi[p]), "v"(yi[p]), "v"(zi[p]), "v"(nmi[p]));

    // tiles before the i-set (same super tile only) were handled when their bodies were the i-set
    const int c_begin = diag_super ? b * (BI / kJT) : 0;
    auto fetch = [&](int c) {                                   // thread t owns body j0 + t of the tile
      const int j = sj * S + c * kJT + t;
      return (j < n_total) ? posm[j] : pad;                               // zero-mass padding
    };
    auto stage = [&](int buf, float4 q) {
      q.w *= gscale;
      sh_pos[buf][wave][lane] = q;
      sh_pos[buf][wave][lane + 64] = q;
    };
    if (c_begin < c_end) stage(c_begin & 1, fetch(c_begin));
    __syncthreads();

    for (int c = c_begin; c < c_end; ++c) {
      const int buf = c & 1;
      const int j0 = sj * S + c * kJT;
      const bool sym = !diag_super || j0 >= i0 + BI;
      const bool more = c + 1 < c_end;
      float4 nxt;
      if (more) nxt = fetch(c + 1);                             // in flight under this tile's arithmetic

      if (sym) {
        for (int r = 0; r < 4; ++r) {
          const int sub = (r + wave) & 3;                       // waves start on different subtiles
          // At step k lane l meets body (l - k) & 63 of the subtile; that body's running j-side sum sits in the
          // same lane and moves on with it (wave_ror:1: lane l+1 takes lane l's value) after every step.
          const float4 *sp = &sh_pos[buf][sub][lane + 64];
          float jx = 0.f, jy = 0.f, jz = 0.f, kx = 0.f, ky = 0.f, kz = 0.f;   // k*: Kahan compensation of j*
          f2 qx = splat2(0.f), qy = splat2(0.f), qz = splat2(0.f);            // JPK: (lo, hi) partial sums
          f2 cx = splat2(0.f), cy = splat2(0.f), cz = splat2(0.f);            // JPK && KAHAN: their compensation
#pragma unroll kUnroll
          for (int k = 0; k < 64; ++k) {
            const float4 pj = sp[-k];
            f2 dx[NP], dy[NP], dz[NP], w[NP], u[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) { dx[p] = splat2(pj.x) - xi[p]; dy[p] = splat2(pj.y) - yi[p]; dz[p] = splat2(pj.z) - zi[p]; }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
              if (ZMODE == Z_SOFT && !BARE) w[p] = fma2(dz[p], dz[p], zp2);
              else                          w[p] = dz[p] * dz[p];
              w[p] = fma2(dy[p], dy[p], w[p]);
              w[p] = fma2(dx[p], dx[p], w[p]);
            }
            if (ZMODE == Z_CLAMP && !BARE) {
#pragma unroll
              for (int p = 0; p < NP; ++p) {
                f2 nf;
                asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(nf) : "v"(w[p]), "v"(zp2), "v"(one2));
                w[p] = w[p] + nf;
              }
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) u[p] = f2{rsq_dev(w[p].x), rsq_dev(w[p].y)};
#pragma unroll
            for (int p = 0; p < NP; ++p) {
              w[p] = u[p] * u[p];
              w[p] = w[p] * u[p];                                     // |d|^-3 (ordinary ops between rsq and the asm)
              if (JPK) u[p] = w[p] * nmi[p];                          // -G m_i |d|^-3
              else     u[p] = mul_swap(w[p], nmi[p]);                 // the same with its halves swapped
              w[p] = mul_bcast_hi(w[p], f2{pj.z, pj.w});              //  G m_j |d|^-3
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) {
              a[p].add(w[p], dx[p], dy[p], dz[p]);
              if (KAHAN && JPK) {
                Acc3pk<true>::kadd(qx, cx, u[p], dx[p]); Acc3pk<true>::kadd(qy, cy, u[p], dy[p]);
                Acc3pk<true>::kadd(qz, cz, u[p], dz[p]);
              } else if (KAHAN) {
                auto kadd = [](float &sum, float &c, float sc, float d) {
                  const float yv = fmaf(sc, d, -c);
                  const float tt = sum + yv;
                  c = (tt - sum) - yv;
                  sum = tt;
                };
                kadd(jx, kx, u[p].x, dx[p].y); kadd(jy, ky, u[p].x, dy[p].y); kadd(jz, kz, u[p].x, dz[p].y);
                kadd(jx, kx, u[p].y, dx[p].x); kadd(jy, ky, u[p].y, dy[p].x); kadd(jz, kz, u[p].y, dz[p].x);
              } else if (JPK) {
                qx = fma2(u[p], dx[p], qx); qy = fma2(u[p], dy[p], qy); qz = fma2(u[p], dz[p], qz);
              } else {
                // both of the lane's bodies act on the same j: scalar FMAs straight into its running sum
                jx = fmaf(u[p].x, dx[p].y, jx); jy = fmaf(u[p].x, dy[p].y, jy); jz = fmaf(u[p].x, dz[p].y, jz);
                jx = fmaf(u[p].y, dx[p].x, jx); jy = fmaf(u[p].y, dy[p].x, jy); jz = fmaf(u[p].y, dz[p].x, jz);
              }
            }
            if (!JPK) {                                                      // the sum moves on with its body
              jx = wave_ror1(jx); jy = wave_ror1(jy); jz = wave_ror1(jz);
              if (KAHAN) { kx = wave_ror1(kx); ky = wave_ror1(ky); kz = wave_ror1(kz); }
            } else {
              qx = f2{wave_ror1(qx.x), wave_ror1(qx.y)}; qy = f2{wave_ror1(qy.x), wave_ror1(qy.y)};
              qz = f2{wave_ror1(qz.x), wave_ror1(qz.y)};
              if (KAHAN) {
                cx = f2{wave_ror1(cx.x), wave_ror1(cx.y)}; cy = f2{wave_ror1(cy.x), wave_ror1(cy.y)};
                cz = f2{wave_ror1(cz.x), wave_ror1(cz.y)};
              }
            }
          }
          if (JPK && KAHAN) {   // fold: (lo - its compensation) + (hi - its compensation)
            jx = (qx.x - cx.x) + (qx.y - cx.y); jy = (qy.x - cy.x) + (qy.y - cy.y); jz = (qz.x - cz.x) + (qz.y - cz.y);
          } else if (JPK) { jx = qx.x + qx.y; jy = qy.x + qy.y; jz = qz.x + qz.y; }
          // after 64 moves lane l holds the sum of body l of the subtile again
          sh_acc[wave][0][sub * 64 + lane] = jx; sh_acc[wave][1][sub * 64 + lane] = jy; sh_acc[wave][2][sub * 64 + lane] = jz;
        }
      } else {
        // one-sided step on the tiles that overlap the i-set: every ordered pair, self pairs dropped by ZMODE
        constexpr int JB = (NP == 1) ? 4 : (NP == 2 ? 2 : 1);   // j-bodies in flight in the one-sided tiles
        for (int q = 0; q < 4; ++q) {
#pragma unroll 2
          for (int k = 0; k < 64; k += JB) {
            float4 pj[JB];
#pragma unroll
            for (int g = 0; g < JB; ++g) pj[g] = sh_pos[buf][q][k + g];
            pair_group_pk<NP, JB, ZMODE, KAHAN>(xi, yi, zi, pj, zp2, one2, a);
          }
        }
      }
      if (more) stage(buf ^ 1, nxt);
      __syncthreads();   // the four waves' tile sums are complete; the next tile is staged
      if (sym) {
        // thread t adds body j0 + t's sum (waves in fixed order) to this workgroup's private row
        float4 *dst = &Pj[j0 + t];
        float4 o = *dst;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) { o.x += sh_acc[wv][0][t]; o.y += sh_acc[wv][1][t]; o.z += sh_acc[wv][2][t]; }
        *dst = o;
        __syncthreads(); // sh_acc is rewritten by the next tile
      }
    }

    // the body indices are recomputed from the thread id here (opaque to the optimiser) instead of staying alive in
    // VGPRs across the whole i-set: they were what the compiler spilled to scratch
    int te = threadIdx.x;
    asm volatile("" : "+v"(te));
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      const int ia = i0 + te + (2 * p) * kBlock, ib = ia + kBlock;
      if (ia < n_pad) Pi[ia - own0] = make_float4(a[p].x.x, a[p].y.x, a[p].z.x, 0.f);
      if (ib < n_pad) Pi[ib - own0] = make_float4(a[p].x.y, a[p].y.y, a[p].z.y, 0.f);
    }
  }
}

}  // namespace

bool sym_pair_assigned(int a, int b, int T) { return sym_assigned(a, b, T); }

hipError_t launch_forces_sym64(const SymLaunch &L, hipStream_t s);   // kernels_sym64.hip

hipError_t launch_forces_sym(const SymLaunch &L, hipStream_t s) {
  if (L.precision == NBODY_PREC_F64) return launch_forces_sym64(L, s);
  if (L.n_total <= 0 || L.n_pairs <= 0 || L.S <= 0 || L.T <= 0 || L.tiles_own <= 0) return hipErrorInvalidValue;
  if (L.S % (kBlock * 2 * L.np) != 0 || L.S % kJT != 0) return hipErrorInvalidValue;
  dim3 grid(L.n_pairs), block(kBlock);
#define NBODY_SYM_K(NPV, ZM, BARE, KH, JP, ZP, FLAG, RUNIF)                                                        \
  hipLaunchKernelGGL((forces_sym_pk_kernel<NPV, ZM, BARE, KH, JP>), grid, block, 0, s, (const float4 *)L.posm,  \
                     (float4 *)L.part_i, (float4 *)L.part_j, (const int2 *)L.pairs, L.n_total, L.S, L.n_pad,    \
                     L.own_tile0, L.tiles_own * L.S, (float)L.G, (float)(ZP), (const int *)(FLAG), RUNIF)
#define NBODY_SYM(NPV, ZM, BARE, ZP, FLAG, RUNIF)                                                                \
  do {                                                                                                           \
    if (L.kahan && jpk) NBODY_SYM_K(NPV, ZM, BARE, true, true, ZP, FLAG, RUNIF);                                 \
    else if (L.kahan) NBODY_SYM_K(NPV, ZM, BARE, true, false, ZP, FLAG, RUNIF);                                  \
    else if (jpk) NBODY_SYM_K(NPV, ZM, BARE, false, true, ZP, FLAG, RUNIF);                                      \
    else NBODY_SYM_K(NPV, ZM, BARE, false, false, ZP, FLAG, RUNIF);                                              \
  } while (0)
#define NBODY_SYM_NP(ZM, BARE, ZP, FLAG, RUNIF)                                                                  \
  do {                                                                                                           \
    if (L.np == 1) NBODY_SYM(1, ZM, BARE, ZP, FLAG, RUNIF);                                                      \
    else if (L.np == 2) NBODY_SYM(2, ZM, BARE, ZP, FLAG, RUNIF);                                                 \
    else if (L.np == 8) NBODY_SYM_K(8, ZM, BARE, false, true, ZP, FLAG, RUNIF);                                  \
    else if (L.kahan) NBODY_SYM_K(4, ZM, BARE, true, true, ZP, FLAG, RUNIF);                                     \
    else NBODY_SYM_K(4, ZM, BARE, false, true, ZP, FLAG, RUNIF);                                                 \
  } while (0)
  if (L.np != 1 && L.np != 2 && L.np != 4 && L.np != 8) return hipErrorInvalidValue;
  if (L.np == 8 && L.kahan) return hipErrorInvalidValue;
  // j-side sums: packed pairs (JPK) unless NBODY_SYM_JSCALAR=1 asks for the scalar form (A/B measurements, np <= 2)
  static const bool jscalar = [] { const char *e = getenv("NBODY_SYM_JSCALAR"); return e && e[0] == '1'; }();
  const bool jpk = !jscalar;
  if (L.eps2 > 0.0) {
    NBODY_SYM_NP(Z_SOFT, false, L.eps2, nullptr, 0);
  } else if (L.dup_table != nullptr) {
    // exact d == 0 semantics at the unguarded kernel's price: look for coincident bodies first, then launch both
    // forms — exactly one of them runs (the other returns at its first instruction)
    hipError_t e0 = hipMemsetAsync(L.dup_table, 0, (size_t)L.dup_slots * 8 + 8, s);   // slots + {flag, near-origin count}
    if (e0 != hipSuccess) return e0;
    int *flag = (int *)((unsigned long long *)L.dup_table + L.dup_slots);
    hipLaunchKernelGGL(dup_detect_kernel<float>, dim3((L.n_total + kBlock - 1) / kBlock), block, 0, s, (const float4 *)L.posm,
                       L.n_total, (unsigned long long *)L.dup_table, (unsigned int)(L.dup_slots - 1), flag);
    NBODY_SYM_NP(Z_CLAMP, true, -0x1p126, flag, 0);
    NBODY_SYM_NP(Z_CLAMP, false, -0x1p126, flag, 1);
  } else {
    NBODY_SYM_NP(Z_CLAMP, false, -0x1p126, nullptr, 0);
  }
#undef NBODY_SYM_NP
#undef NBODY_SYM
#undef NBODY_SYM_K
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (L.kahan)
    hipLaunchKernelGGL((reduce_j_kernel<float, true>), dim3((L.n_total + kBlock - 1) / kBlock), block, 0, s,
                       (const float4 *)L.part_j, (float4 *)L.send, L.n_total, L.S, L.T, L.n_pad, L.own_tile0, L.tiles_own);
  else
  hipLaunchKernelGGL((reduce_j_kernel<float, false>), dim3((L.n_total + kBlock - 1) / kBlock), block, 0, s, (const float4 *)L.part_j,
                     (float4 *)L.send, L.n_total, L.S, L.T, L.n_pad, L.own_tile0, L.tiles_own);
  return hipGetLastError();
}

hipError_t launch_update_sym(const SymLaunch &L, void *posm, void *vel, void *acc, int i_begin, int i_count, float dt,
                             hipStream_t s) {
  if (i_count <= 0) return hipErrorInvalidValue;
  dim3 grid((i_count + kBlock - 1) / kBlock), block(kBlock);
  if (L.precision == NBODY_PREC_F64)
    hipLaunchKernelGGL((update_sym_kernel<double, false>), grid, block, 0, s, (double4 *)posm, (double4 *)vel, (double4 *)acc,
                       (const double4 *)L.part_i, (const double4 *)L.recv, i_begin, i_count, L.S, L.T, L.tiles_own * L.S,
                       L.n_src, (double)dt, dt > 0.0f ? 1 : 0);
  else if (L.kahan)
    hipLaunchKernelGGL((update_sym_kernel<float, true>), grid, block, 0, s, (float4 *)posm, (float4 *)vel, (float4 *)acc,
                       (const float4 *)L.part_i, (const float4 *)L.recv, i_begin, i_count, L.S, L.T, L.tiles_own * L.S,
                       L.n_src, dt, dt > 0.0f ? 1 : 0);
  else
    hipLaunchKernelGGL((update_sym_kernel<float, false>), grid, block, 0, s, (float4 *)posm, (float4 *)vel, (float4 *)acc,
                       (const float4 *)L.part_i, (const float4 *)L.recv, i_begin, i_count, L.S, L.T, L.tiles_own * L.S,
                       L.n_src, dt, dt > 0.0f ? 1 : 0);
  return hipGetLastError();
}

}  // namespace nbody

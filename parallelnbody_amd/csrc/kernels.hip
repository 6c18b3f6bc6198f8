// gfx950 (MI355X, CDNA4) kernels of the N-body hot path.  wave = 64 lanes; no MFMA: the pair law is
// scalar fp32/fp64 FMA work, bounded by the vector-ALU issue rate.
//
// Reference lines restated on the device (paths relative to /root/reference/Source/NBody/):
//   forces_tile_kernel  <- the pair law OctreeSearch.h:101-104 summed over all j, i.e. the loop
//                          OctreeSearch.cpp:83-86 at theta = 0
//   update_kernel       <- OctreeSearch.cpp:28-31 (v += dt*a; x += dt*v)
//   bounds_kernel       <- OctreeSearch.cpp:47-56 (ComputeCubeSize)
#include "kernels.h"

#include <type_traits>

#include "../../include/nbody.h"
#include "pk_common.h"
#include "sym_common.h"

namespace nbody {

namespace {

template <typename T> struct V4;
template <> struct V4<float> { using type = float4; };
template <> struct V4<double> { using type = double4; };

// 1/sqrt(x): v_rsq_f64 seed (~2^-26 relative) and one third-order step, y (1 + e/2 + 3/8 e^2) with e = 1 - x y^2
// (error ~e^3, far below 2^-53; five ops where two Newton steps take seven) — as in kernels_sym64.hip
__device__ __forceinline__ double rsq_dev(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * y), y, 1.0);
  const double q = e * fma(e, 0.375, 0.5);
  return fma(y, q, y);
}

// Plain or Kahan-compensated 3-vector accumulator.
template <typename T, bool KAHAN> struct Acc3;
template <typename T> struct Acc3<T, false> {
  T x = 0, y = 0, z = 0;
  __device__ __forceinline__ void add(T s, T dx, T dy, T dz) {
    x = fma(s, dx, x); y = fma(s, dy, y); z = fma(s, dz, z);
  }
};
template <typename T> struct Acc3<T, true> {
  T x = 0, y = 0, z = 0, cx = 0, cy = 0, cz = 0;
  static __device__ __forceinline__ void kadd(T &sum, T &c, T s, T d) {
    const T yv = fma(s, d, -c);
    const T t = sum + yv;
    c = (t - sum) - yv;
    sum = t;
  }
  __device__ __forceinline__ void add(T s, T dx, T dy, T dz) {
    kadd(x, cx, s, dx); kadd(y, cy, s, dy); kadd(z, cz, s, dz);
  }
};

__device__ __forceinline__ float clamp01(float x) { return __builtin_amdgcn_fmed3f(x, 0.0f, 1.0f); }   // folds into the clamp bit

// The pair law for a group of JB j-bodies against the lane's IPT i-bodies, written stage by stage (all
// differences, then all r^2, then all rsq, ...) so that each instruction's consumers sit JB*IPT issue slots
// behind it: no dependent back-to-back VALU pairs and no wait states behind the quarter-rate v_rsq_f32.
// mj already carries G.  `zp` is eps2 (Z_SOFT) or -2^126 (Z_CLAMP).
template <typename T, int IPT, int JB, int ZMODE, bool KAHAN, typename V>
__device__ __forceinline__ void interact_group(const T (&xi)[IPT], const T (&yi)[IPT], const T (&zi)[IPT],
                                               const V (&pj)[JB], T zp, Acc3<T, KAHAN> (&a)[IPT]) {
  T dx[JB][IPT], dy[JB][IPT], dz[JB][IPT], w[JB][IPT];
#pragma unroll
  for (int b = 0; b < JB; ++b)
#pragma unroll
    for (int k = 0; k < IPT; ++k) { dx[b][k] = pj[b].x - xi[k]; dy[b][k] = pj[b].y - yi[k]; dz[b][k] = pj[b].z - zi[k]; }
#pragma unroll
  for (int b = 0; b < JB; ++b)
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      if (ZMODE == Z_SOFT) w[b][k] = fma(dz[b][k], dz[b][k], zp);
      else                 w[b][k] = dz[b][k] * dz[b][k];
    }
#pragma unroll
  for (int b = 0; b < JB; ++b)
#pragma unroll
    for (int k = 0; k < IPT; ++k) w[b][k] = fma(dy[b][k], dy[b][k], w[b][k]);
#pragma unroll
  for (int b = 0; b < JB; ++b)
#pragma unroll
    for (int k = 0; k < IPT; ++k) w[b][k] = fma(dx[b][k], dx[b][k], w[b][k]);
  if (ZMODE == Z_CLAMP && sizeof(T) == 4) {
    T nf[JB][IPT];
#pragma unroll
    for (int b = 0; b < JB; ++b)
#pragma unroll
      for (int k = 0; k < IPT; ++k) nf[b][k] = (T)clamp01((float)fma(w[b][k], zp, T(1)));
#pragma unroll
    for (int b = 0; b < JB; ++b)
#pragma unroll
      for (int k = 0; k < IPT; ++k) w[b][k] = w[b][k] + nf[b][k];
  }
  T rinv[JB][IPT];
#pragma unroll
  for (int b = 0; b < JB; ++b)
#pragma unroll
    for (int k = 0; k < IPT; ++k) {
      rinv[b][k] = rsq_dev(w[b][k]);
      if (ZMODE == Z_SELECT || (ZMODE == Z_CLAMP && sizeof(T) == 8)) rinv[b][k] = (w[b][k] > T(0)) ? rinv[b][k] : T(0);
    }
#pragma unroll
  for (int b = 0; b < JB; ++b)
#pragma unroll
    for (int k = 0; k < IPT; ++k) w[b][k] = rinv[b][k] * rinv[b][k];
#pragma unroll
  for (int b = 0; b < JB; ++b)
#pragma unroll
    for (int k = 0; k < IPT; ++k) rinv[b][k] = pj[b].w * rinv[b][k];
#pragma unroll
  for (int b = 0; b < JB; ++b)
#pragma unroll
    for (int k = 0; k < IPT; ++k) w[b][k] = rinv[b][k] * w[b][k];
#pragma unroll
  for (int b = 0; b < JB; ++b)
#pragma unroll
    for (int k = 0; k < IPT; ++k) a[k].add(w[b][k], dx[b][k], dy[b][k], dz[b][k]);
}

// All-pairs partial accelerations.
//   grid.x : i-blocks of kBlock*IPT owned bodies (lane t holds bodies ibase + t + k*kBlock: coalesced)
//   grid.y : j chunks [c*j_chunk, min((c+1)*j_chunk, n_total)); each writes its own partial row
//   LDS    : double-buffered tile of TILE bodies (x,y,z,G*m); every lane reads the same address
//            (broadcast ds_read_b128), one barrier per tile
template <typename T, int IPT, int TILE, int ZMODE, bool KAHAN>
__global__ __launch_bounds__(kBlock) void forces_tile_kernel(const typename V4<T>::type *__restrict__ posm,
                                                             typename V4<T>::type *__restrict__ accp, int n_total,
                                                             int i_begin, int i_count, int j_chunk, T gscale, T zp) {
  using V = typename V4<T>::type;
  constexpr int LPT = (TILE + kBlock - 1) / kBlock;   // tile elements loaded per lane
  __shared__ V sh[2][TILE];

  const int t = threadIdx.x;
  const int ibase = blockIdx.x * (kBlock * IPT);
  const int c = blockIdx.y;
  const int j0 = c * j_chunk;
  const int j1 = min(j0 + j_chunk, n_total);
  const int ntiles = (j1 > j0) ? (j1 - j0 + TILE - 1) / TILE : 0;

  T xi[IPT], yi[IPT], zi[IPT];
  Acc3<T, KAHAN> a[IPT];
#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const int il = min(ibase + t + k * kBlock, i_count - 1);
    const V p = posm[i_begin + il];
    xi[k] = p.x; yi[k] = p.y; zi[k] = p.z;
  }
  // Consume the i-body loads here, so that their s_waitcnt sits in front of the loops and not at the
  // registers' first use inside the j loop (where vmcnt(0) would also drain the next tile's prefetch).
#pragma unroll
  for (int k = 0; k < IPT; ++k) asm volatile("" ::"v"(xi[k]), "v"(yi[k]), "v"(zi[k]));

  V r[LPT];
  auto load_tile = [&](int tile) {
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
      const int e = t + l * kBlock;
      if (e < TILE) {
        const int j = j0 + tile * TILE + e;
        if (j < j1) r[l] = posm[j];
        else        { r[l].x = 0; r[l].y = 0; r[l].z = 0; r[l].w = 0; }   // zero-mass padding
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
      const int e = t + l * kBlock;
      if (e < TILE) { V q = r[l]; q.w *= gscale; sh[buf][e] = q; }   // G folded into the mass here, after the wait
    }
  };

  if (ntiles > 0) { load_tile(0); store_tile(0); }
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const int buf = tile & 1;
    const bool more = tile + 1 < ntiles;
    if (more) load_tile(tile + 1);          // global loads in flight under the tile's arithmetic
    constexpr int JB = (IPT >= 4) ? 2 : (8 / (IPT * (sizeof(T) / 4)) > 0 ? 8 / (IPT * (int)(sizeof(T) / 4)) : 1);
#pragma unroll 2
    for (int jj = 0; jj < TILE; jj += JB) {
      V pj[JB];
#pragma unroll
      for (int b = 0; b < JB; ++b) pj[b] = sh[buf][jj + b];
      interact_group<T, IPT, JB, ZMODE, KAHAN, V>(xi, yi, zi, pj, zp, a);
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int k = 0; k < IPT; ++k) {
    const int il = ibase + t + k * kBlock;
    if (il < i_count) {
      V o; o.x = a[k].x; o.y = a[k].y; o.z = a[k].z; o.w = 0;
      accp[(size_t)c * i_count + il] = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// Packed-fp32 force kernel (the fp32 production path).
//
// Measured on MI355X (tools/microbench*.hip, DESIGN.md "VALU issue model"): v_fma/v_mul/v_add/v_sub issue in
// ~2.2 cycles per wave64, v_rsq_f32 in 8.2; a 3-source op whose three VGPRs all have the same register-number
// parity, and ANY op with an SGPR operand, takes 4.2; v_max/v_min/v_cmp/v_med3 take 4.2 and v_cndmask far
// more.  v_pk_{fma,mul,add}_f32 take 4.2 for two results and, reading even-aligned register PAIRS, can never
// hit the parity conflict.  So each lane carries its i-bodies two by two in register pairs and the whole pair
// law runs on packed instructions with every constant in VGPRs: 12 packed ops + 2 v_rsq_f32 per two pairs.
// ---------------------------------------------------------------------------------------------------------
// NP register pairs of i-bodies per lane (IPT = 2*NP); JB j-bodies per staged group.
// dup_flag (Z_CLAMP only): verdict of dup_detect_kernel on this pass's positions.  With no two bodies on one point
// (*dup_flag == 0) d == 0 can only be a self pair, so full tiles that do not overlap the workgroup's own i-range run
// the pair law bare — 5 packed ops + 1 v_rsq_f32 per pair-lane instead of 7 + 1.  The results are the guarded ones.
#ifndef NBODY_TILE_UNI
#define NBODY_TILE_UNI 1      // 0 compiles the equal-mass branch out (A/B builds: tools/ab_tile_uni.sh)
#endif
template <int NP, int TILE, int ZMODE, bool KAHAN>
__global__ __launch_bounds__(kBlock) void forces_tile_pk_kernel(const float4 *__restrict__ posm,
                                                                float4 *__restrict__ accp, int n_total, int i_begin,
                                                                int i_count, int j_chunk, float gscale, float zp,
                                                                const int *__restrict__ dup_flag,
                                                                const int *__restrict__ general,
                                                                unsigned long long *__restrict__ clk) {
  constexpr int IPT = 2 * NP;
  const ClockStamp stamp = clock_begin(clk);
  // equal-mass form (see forces_sym_pk_kernel, UNI): *general == 0 says every body has body 0's mass — found by the host
  // in the state it uploaded, or by mass_check_kernel before this launch when somebody else can write the buffer.  Then
  // the pair loop carries no mass factor (11 packed ops per register pair and j instead of 12), the padding of ragged tiles
  // goes far away instead of to the origin (there is no zero mass to hide it behind), and the sums get the common G m
  // on their way out.  One kernel, a wave-uniform branch: small systems cannot afford a second launch.
  const bool uni = NBODY_TILE_UNI && general != nullptr && *general == 0;
  const float padc = uni ? kPadFar : 0.f;
  constexpr int LPT = (TILE + kBlock - 1) / kBlock;
  __shared__ float4 sh[2][TILE];

  const int t = threadIdx.x;
  const int ibase = blockIdx.x * (kBlock * IPT);
  const int c = blockIdx.y;
  const int j0 = c * j_chunk;
  const int j1 = min(j0 + j_chunk, n_total);
  const int ntiles = (j1 > j0) ? (j1 - j0 + TILE - 1) / TILE : 0;

  f2 xi[NP], yi[NP], zi[NP];
  // KAHAN = blocked compensated summation (as in kernels_sym.hip): a tile's TILE terms per component are a plain packed-FMA
  // chain in `a`, which is then Kahan-added to the running sums `ka`; update_kernel adds the chunks with compensation
  Acc3pk<false> a[NP];
  Acc3pk<true> ka[KAHAN ? NP : 1];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const float4 p0 = posm[i_begin + min(ibase + t + (2 * p) * kBlock, i_count - 1)];
    const float4 p1 = posm[i_begin + min(ibase + t + (2 * p + 1) * kBlock, i_count - 1)];
    xi[p] = f2{p0.x, p1.x}; yi[p] = f2{p0.y, p1.y}; zi[p] = f2{p0.z, p1.z};
  }
  // every loop-invariant operand in VGPRs (an SGPR operand halves the issue rate), loads consumed before the loops
  f2 zp2 = splat2(zp), one2 = splat2(1.0f);
  asm volatile("" : "+v"(zp2), "+v"(one2));
#pragma unroll
  for (int p = 0; p < NP; ++p) asm volatile("" ::"v"(xi[p]), "v"(yi[p]), "v"(zi[p]));

  float4 r[LPT];
  auto load_tile = [&](int tile) {
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
      const int e = t + l * kBlock;
      if (e < TILE) {
        const int j = j0 + tile * TILE + e;
        if (j < j1) r[l] = posm[j];
        else        r[l] = make_float4(padc, padc, padc, 0.f);   // zero-mass padding
      }
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int l = 0; l < LPT; ++l) {
      const int e = t + l * kBlock;
      if (e < TILE) { float4 q = r[l]; q.w *= gscale; sh[buf][e] = q; }
    }
  };

  const bool bare_ok = ZMODE == Z_CLAMP && dup_flag != nullptr && *dup_flag == 0;
  const int own_lo = i_begin + ibase, own_hi = own_lo + kBlock * IPT;       // this workgroup's i-bodies (global indices)
  if (ntiles > 0) { load_tile(0); store_tile(0); }
  __syncthreads();
  for (int tile = 0; tile < ntiles; ++tile) {
    const int buf = tile & 1;
    const bool more = tile + 1 < ntiles;
    if (more) load_tile(tile + 1);
    constexpr int JB = (NP == 1) ? 4 : 2;
    const int t_lo = j0 + tile * TILE, t_hi = t_lo + TILE;
    // no self pair in the tile and no zero-mass padding (pads sit on the origin, where a body may be)
    auto walk = [&](auto zm, auto un) {
#pragma unroll 2
      for (int jj = 0; jj < TILE; jj += JB) {
        float4 pj[JB];
#pragma unroll
        for (int b = 0; b < JB; ++b) pj[b] = sh[buf][jj + b];
        pair_group_pk<NP, JB, decltype(zm)::value, false, decltype(un)::value>(xi, yi, zi, pj, zp2, one2, a);
      }
    };
    using std::integral_constant;
    if (ZMODE == Z_CLAMP && bare_ok && t_hi <= j1 && (t_hi <= own_lo || t_lo >= own_hi)) {
      if (uni) walk(integral_constant<int, Z_BARE>{}, integral_constant<bool, true>{});
      else     walk(integral_constant<int, Z_BARE>{}, integral_constant<bool, false>{});
    } else {
      if (uni) walk(integral_constant<int, ZMODE>{}, integral_constant<bool, true>{});
      else     walk(integral_constant<int, ZMODE>{}, integral_constant<bool, false>{});
    }
    if (KAHAN) {
#pragma unroll
      for (int p = 0; p < NP; ++p) { ka[p].fold(a[p]); a[p] = Acc3pk<false>(); }
    }
    if (more) store_tile(buf ^ 1);
    __syncthreads();
  }

#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const int il0 = ibase + t + (2 * p) * kBlock, il1 = il0 + kBlock;
    f2 sx = KAHAN ? ka[KAHAN ? p : 0].x : a[p].x, sy = KAHAN ? ka[KAHAN ? p : 0].y : a[p].y, sz = KAHAN ? ka[KAHAN ? p : 0].z : a[p].z;
    if (uni) { const f2 gm = splat2(posm[0].w * gscale); sx = sx * gm; sy = sy * gm; sz = sz * gm; }
    if (il0 < i_count) accp[(size_t)c * i_count + il0] = make_float4(sx.x, sy.x, sz.x, 0.f);
    if (il1 < i_count) accp[(size_t)c * i_count + il1] = make_float4(sx.y, sy.y, sz.y, 0.f);
  }
  clock_end(clk, stamp);
}

// HIP's __fmul_rn/__fadd_rn are plain * and + and get contracted into FMAs under the default
// -ffp-contract=fast; the pragma keeps the two roundings of the reference's operators (FVector's * and +).
template <typename T> __device__ __forceinline__ T mul_add_sep(T a, T b, T c) {
#pragma clang fp contract(off)
  const T p = a * b;
  return c + p;
}

// ---------------------------------------------------------------------------------------------------------
// The block kernel's idea (kernels_block.hip: a workgroup owns a few bodies, its 256 lanes split the j range, a body's
// whole sum is finished inside the workgroup, so the update rides along and a step is ONE launch with no partial rows)
// for the other two precisions, on the scalar staged pair law above: fp64, and fp32 with Kahan-compensated accumulation.
// Small systems only — there one lane per i-body leaves the chip empty (N = 2000: 27.6 us per step with the tile kernel
// + update in either precision; the packed fp32 block kernel: 5.3).
//   per lane    NB running sums over its j-bodies (Acc3: plain for fp64, compensated for KAHAN), LD loads in flight
//   the sums    every lane's value as a double (KAHAN: sum - compensation), added over the wave by six DPP steps in a fixed
//               order and over the four waves in order, all in double, rounded ONCE to the working precision: the fp32
//               result is the correctly rounded sum of the lanes' compensated sums
//   thread t    finishes body t: acc, and with integrate != 0 the reference's update (OctreeSearch.cpp:29-30, multiply
//               and add kept apart) into the OTHER position buffer
// Padding beyond n_total: zero-mass bodies on the origin — every ZMODE here guards d == 0.
// ---------------------------------------------------------------------------------------------------------
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_add64(double v) {
  const long long i = __builtin_bit_cast(long long, v);
  const int lo = (int)(i & 0xffffffffll), hi = (int)(i >> 32);
  const int lo2 = __builtin_amdgcn_update_dpp(ROW_MASK == 0xf ? lo : 0, lo, CTRL, ROW_MASK, 0xf, false);
  const int hi2 = __builtin_amdgcn_update_dpp(ROW_MASK == 0xf ? hi : 0, hi, CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(double, ((long long)hi2 << 32) | (long long)(unsigned int)lo2);
}

// sum over the 64 lanes, left in lane 63 (the order of kernels_block.hip's wave_sum_to_lane63)
template <int NV> __device__ __forceinline__ void wave_sum64_to_lane63(double (&v)[NV]) {
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = dpp_add64<0xB1, 0xf>(v[q]);    // quad_perm:[1,0,3,2]
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = dpp_add64<0x4E, 0xf>(v[q]);    // quad_perm:[2,3,0,1]
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = dpp_add64<0x141, 0xf>(v[q]);   // row_half_mirror
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = dpp_add64<0x140, 0xf>(v[q]);   // row_mirror
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = dpp_add64<0x142, 0xa>(v[q]);   // row_bcast:15 into rows 1 and 3
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = dpp_add64<0x143, 0xc>(v[q]);   // row_bcast:31 into rows 2 and 3
}

template <typename T, int NB, int ZMODE, bool KAHAN>
__global__ __launch_bounds__(kBlock) void forces_block_kernel(const typename V4<T>::type *__restrict__ posm,
                                                              typename V4<T>::type *__restrict__ posm_out,
                                                              typename V4<T>::type *__restrict__ vel,
                                                              typename V4<T>::type *__restrict__ acc_out, int n_total,
                                                              int i_begin, int i_count, T gscale, T zp, T dt, int integrate) {
  using V = typename V4<T>::type;
  constexpr int LD = 4;                      // loads in flight per lane
  __shared__ double red[kBlock / 64][3 * NB];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ia = blockIdx.x * NB;
  if (ia >= i_count) return;                 // uniform per workgroup

  T xi[NB], yi[NB], zi[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    const V p = posm[i_begin + min(ia + k, i_count - 1)];
    xi[k] = p.x; yi[k] = p.y; zi[k] = p.z;
  }
  const int il = ia + t;
  const bool finisher = t < NB && il < i_count;
  V vv, x;
  vv.x = vv.y = vv.z = vv.w = 0; x = vv;
  if (finisher && integrate) { vv = vel[il]; x = posm[i_begin + il]; }
#pragma unroll
  for (int k = 0; k < NB; ++k) asm volatile("" : "+v"(xi[k]), "+v"(yi[k]), "+v"(zi[k]));   // workgroup-uniform: keep them in VGPRs
  asm volatile("" : "+v"(zp));

  Acc3<T, KAHAN> a[NB];
  const int trips = (n_total + kBlock - 1) / kBlock;         // lane t meets bodies t, t + 256, ...
  auto load_one = [&](int trip) {
    const int j = trip * kBlock + t;
    V q = posm[min(j, n_total - 1)];
    if (j >= n_total) { q.x = 0; q.y = 0; q.z = 0; q.w = 0; }
    return q;
  };
  V cur[LD];
#pragma unroll
  for (int l = 0; l < LD; ++l) cur[l] = load_one(min(l, trips - 1));
  for (int g = 0; g < trips; g += LD) {
#pragma unroll
    for (int l = 0; l < LD; ++l) {
      V pj[1] = {cur[l]};
      const bool live = g + l < trips;       // uniform
      if (g + LD + l < trips) cur[l] = load_one(g + LD + l);   // the next round's body: in flight under this round's arithmetic
      if (live) {
        pj[0].w *= gscale;
        interact_group<T, NB, 1, ZMODE, KAHAN, V>(xi, yi, zi, pj, zp, a);
      }
    }
  }

  double v[3 * NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) {
    if constexpr (KAHAN) {                   // the running sum minus what it still owes
      v[3 * k] = (double)a[k].x - (double)a[k].cx; v[3 * k + 1] = (double)a[k].y - (double)a[k].cy; v[3 * k + 2] = (double)a[k].z - (double)a[k].cz;
    } else {
      v[3 * k] = (double)a[k].x; v[3 * k + 1] = (double)a[k].y; v[3 * k + 2] = (double)a[k].z;
    }
  }
  wave_sum64_to_lane63(v);
  if (lane == 63) {
#pragma unroll
    for (int q = 0; q < 3 * NB; ++q) red[wave][q] = v[q];
  }
  __syncthreads();
  if (!finisher) return;
  const T ax = (T)(((red[0][3 * t] + red[1][3 * t]) + red[2][3 * t]) + red[3][3 * t]);
  const T ay = (T)(((red[0][3 * t + 1] + red[1][3 * t + 1]) + red[2][3 * t + 1]) + red[3][3 * t + 1]);
  const T az = (T)(((red[0][3 * t + 2] + red[1][3 * t + 2]) + red[2][3 * t + 2]) + red[3][3 * t + 2]);
  V ao; ao.x = ax; ao.y = ay; ao.z = az; ao.w = 0;
  acc_out[il] = ao;
  if (!integrate) return;
  vv.x = mul_add_sep(dt, ax, vv.x); vv.y = mul_add_sep(dt, ay, vv.y); vv.z = mul_add_sep(dt, az, vv.z);
  x.x = mul_add_sep(dt, vv.x, x.x); x.y = mul_add_sep(dt, vv.y, x.y); x.z = mul_add_sep(dt, vv.z, x.z);
  vel[il] = vv;
  posm_out[i_begin + il] = x;
}

// Combine the j-chunk partials in chunk order (deterministic), store the acceleration, and — when
// integrate != 0 — apply the reference's update with separate multiply and add (no FMA), exactly
// as FVector's operators do: v = v + dt*a; x = x + dt*v   (OctreeSearch.cpp:29-30).
template <typename T, bool KAHAN>
__global__ __launch_bounds__(kBlock) void update_kernel(typename V4<T>::type *__restrict__ posm,
                                                        typename V4<T>::type *__restrict__ vel,
                                                        typename V4<T>::type *__restrict__ acc,
                                                        const typename V4<T>::type *__restrict__ accp, int i_begin,
                                                        int i_count, int j_split, T dt, int integrate) {
  using V = typename V4<T>::type;
  const int il = blockIdx.x * kBlock + threadIdx.x;
  if (il >= i_count) return;
  T ax = 0, ay = 0, az = 0, cx = 0, cy = 0, cz = 0;
  // rows are added in chunk order (deterministic); eight loads in flight: small systems have few threads and many rows
#pragma unroll 8
  for (int c = 0; c < j_split; ++c) {
    const V p = accp[(size_t)c * i_count + il];
    if (KAHAN) {
      T yv = p.x - cx; T tt = ax + yv; cx = (tt - ax) - yv; ax = tt;
      yv = p.y - cy; tt = ay + yv; cy = (tt - ay) - yv; ay = tt;
      yv = p.z - cz; tt = az + yv; cz = (tt - az) - yv; az = tt;
    } else {
      ax = ax + p.x; ay = ay + p.y; az = az + p.z;
    }
  }
  V a; a.x = ax; a.y = ay; a.z = az; a.w = 0;
  acc[il] = a;
  if (integrate) {
    V v = vel[il];
    V x = posm[i_begin + il];
    v.x = mul_add_sep(dt, ax, v.x); v.y = mul_add_sep(dt, ay, v.y); v.z = mul_add_sep(dt, az, v.z);
    x.x = mul_add_sep(dt, v.x, x.x); x.y = mul_add_sep(dt, v.y, x.y); x.z = mul_add_sep(dt, v.z, x.z);
    vel[il] = v;
    posm[i_begin + il] = x;
  }
}

template <typename T, bool MASS>
__global__ __launch_bounds__(kBlock) void bounds_kernel(const typename V4<T>::type *__restrict__ posm, int i_begin,
                                                        int i_count, unsigned int *__restrict__ out_bits,
                                                        unsigned int *__restrict__ zero_word) {
  using V = typename V4<T>::type;
  // callers that alternate between two result words let this launch clear the one the NEXT call will use (no memset launch)
  if (zero_word != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *zero_word = 0u;
  float m = 0.0f;
  for (int il = blockIdx.x * kBlock + threadIdx.x; il < i_count; il += gridDim.x * kBlock) {
    const V p = posm[i_begin + il];
    // GetAbsMax: max(max(|X|,|Y|),|Z|); compared in fp32 like the reference's float Size
    const float v = MASS ? fabsf((float)p.w) : fmaxf(fmaxf(fabsf((float)p.x), fabsf((float)p.y)), fabsf((float)p.z));
    m = fmaxf(m, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  // one atomic per workgroup (8192 same-address atomics — one per wave of a 2048-block grid — cost 80 us at N = 2^20)
  __shared__ float wave_max[kBlock / 64];
  if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < kBlock / 64; ++w) m = fmaxf(m, wave_max[w]);
    atomicMax(out_bits, __float_as_uint(m));                              // non-negative floats order as uints
  }
}

// fp64 energy diagnostic.  Per workgroup: part[2 slot] = sum_i 1/2 m_i v_i^2 (only by blockIdx.y == 0),
// part[2 slot + 1] = sum_i 1/2 m_i phi_i with phi_i = -G sum_j m_j / sqrt(d^2 + eps2), d^2+eps2 == 0 skipped.
template <typename T>
__global__ __launch_bounds__(kBlock) void energy_kernel(const typename V4<T>::type *__restrict__ posm,
                                                        const typename V4<T>::type *__restrict__ vel, int n_total,
                                                        int i_begin, int i_count, int j_chunk, double G, double eps2,
                                                        double *__restrict__ part) {
  using V = typename V4<T>::type;
  __shared__ double4 sh[kBlock];
  __shared__ double red[2][kBlock / 64];
  const int t = threadIdx.x;
  const int il = blockIdx.x * kBlock + t;
  const bool live = il < i_count;
  const int ig = i_begin + min(il, i_count - 1);
  const V pi = posm[ig];
  const double xi = pi.x, yi = pi.y, zi = pi.z, mi = pi.w;
  const int j0 = blockIdx.y * j_chunk;
  const int j1 = min(j0 + j_chunk, n_total);
  double phi = 0.0;
  for (int jt = j0; jt < j1; jt += kBlock) {
    const int j = jt + t;
    double4 q; q.x = 0; q.y = 0; q.z = 0; q.w = 0;
    if (j < j1) { const V p = posm[j]; q.x = p.x; q.y = p.y; q.z = p.z; q.w = p.w; }
    __syncthreads();
    sh[t] = q;
    __syncthreads();
#pragma unroll 4
    for (int jj = 0; jj < kBlock; ++jj) {
      const double4 pj = sh[jj];
      const double dx = pj.x - xi, dy = pj.y - yi, dz = pj.z - zi;
      const double r2 = fma(dx, dx, fma(dy, dy, fma(dz, dz, eps2)));
      double rinv = rsq_dev(r2);
      rinv = (r2 > 0.0 && jt + jj != ig) ? rinv : 0.0;   // no self term, coincident pairs skipped
      phi = fma(pj.w, rinv, phi);
    }
  }
  double pe = live ? -0.5 * G * mi * phi : 0.0;
  double ke = 0.0;
  if (live && blockIdx.y == 0) {
    const V v = vel[il];
    const double vx = v.x, vy = v.y, vz = v.z;
    ke = 0.5 * mi * (vx * vx + vy * vy + vz * vz);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { ke += __shfl_xor(ke, off, 64); pe += __shfl_xor(pe, off, 64); }
  if ((t & 63) == 0) { red[0][t >> 6] = ke; red[1][t >> 6] = pe; }
  __syncthreads();
  if (t == 0) {
    double k = 0, p = 0;
    for (int w = 0; w < kBlock / 64; ++w) { k += red[0][w]; p += red[1][w]; }
    // one slot per workgroup, added up in a fixed order by energy_fold_kernel: no atomics, the same bits every run
    const size_t slot = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    part[2 * slot] = k;
    part[2 * slot + 1] = p;
  }
}

// out[0] = sum of the workgroups' kinetic parts, out[1] = of their potential parts: lane t adds slots t, t + 256, ... in
// order, then the fixed shuffle tree and the four waves in order.
__global__ __launch_bounds__(kBlock) void energy_fold_kernel(const double *__restrict__ part, int slots, double *__restrict__ out) {
  __shared__ double red[2][kBlock / 64];
  const int t = threadIdx.x;
  double k = 0.0, p = 0.0;
  for (int q = t; q < slots; q += kBlock) { k += part[2 * q]; p += part[2 * q + 1]; }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { k += __shfl_xor(k, off, 64); p += __shfl_xor(p, off, 64); }
  if ((t & 63) == 0) { red[0][t >> 6] = k; red[1][t >> 6] = p; }
  __syncthreads();
  if (t == 0) {
    double ks = 0, ps = 0;
    for (int w = 0; w < kBlock / 64; ++w) { ks += red[0][w]; ps += red[1][w]; }
    out[0] = ks;
    out[1] = ps;
  }
}

template <typename T, int IPT, int TILE, bool KAHAN>
hipError_t launch_forces_t(const ForceLaunch &L, hipStream_t s) {
  using V = typename V4<T>::type;
  const int iblocks = (L.i_count + kBlock * IPT - 1) / (kBlock * IPT);
  dim3 grid(iblocks, L.j_split), block(kBlock);
  const T gscale = (T)L.G;
  if constexpr (sizeof(T) == 4 && IPT % 2 == 0) {
    // production fp32 path: packed kernel (compare+select only exists in the scalar kernel, for A/B)
    if (L.eps2 > 0.0 || L.zero_mode != Z_SELECT) {
#define NBODY_LAUNCH_PK(ZM, ZP, FLAG)                                                                            \
  hipLaunchKernelGGL((forces_tile_pk_kernel<IPT / 2, TILE, ZM, KAHAN>), grid, block, 0, s, (const float4 *)L.posm, \
                     (float4 *)L.accp, L.n_total, L.i_begin, L.i_count, L.j_chunk, (float)L.G, (float)(ZP),        \
                     (const int *)(FLAG), (const int *)L.general, (unsigned long long *)L.clk)
      // equal masses?  The host's finding stands while only this library writes the buffer; otherwise the device looks
      if (L.general != nullptr && L.check_masses)
        hipLaunchKernelGGL(mass_check_kernel<float>, dim3((L.n_total + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                           (const float4 *)L.posm, L.n_total, (int *)L.general);
      if (L.eps2 > 0.0) {
        NBODY_LAUNCH_PK(Z_SOFT, L.eps2, nullptr);
      } else if (L.dup_table != nullptr) {
        hipError_t e0 = hipMemsetAsync(L.dup_table, 0, (size_t)L.dup_slots * 8 + 64, s);   // slots + {flag, near-origin count}
        if (e0 != hipSuccess) return e0;
        int *flag = (int *)((unsigned long long *)L.dup_table + L.dup_slots);
        hipLaunchKernelGGL(dup_detect_kernel<float>, dim3((L.n_total + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                           (const float4 *)L.posm, L.n_total, (unsigned long long *)L.dup_table,
                           (unsigned int)(L.dup_slots - 1), flag);
        NBODY_LAUNCH_PK(Z_CLAMP, -0x1p126, flag);
      } else {
        NBODY_LAUNCH_PK(Z_CLAMP, -0x1p126, nullptr);
      }
#undef NBODY_LAUNCH_PK
      return hipGetLastError();
    }
  }
#define NBODY_LAUNCH(ZM, ZP)                                                                                   \
  hipLaunchKernelGGL((forces_tile_kernel<T, IPT, TILE, ZM, KAHAN>), grid, block, 0, s, (const V *)L.posm,      \
                     (V *)L.accp, L.n_total, L.i_begin, L.i_count, L.j_chunk, gscale, (T)(ZP))
  if (L.eps2 > 0.0) NBODY_LAUNCH(Z_SOFT, L.eps2);
  else if (L.zero_mode == Z_SELECT || sizeof(T) == 8) NBODY_LAUNCH(Z_SELECT, 0.0);
  else NBODY_LAUNCH(Z_CLAMP, -0x1p126);
#undef NBODY_LAUNCH
  return hipGetLastError();
}

template <typename T, int IPT, bool KAHAN>
hipError_t launch_forces_tile(const ForceLaunch &L, hipStream_t s) {
  switch (L.tile) {
    case 64:  return launch_forces_t<T, IPT, 64, KAHAN>(L, s);
    case 128: return launch_forces_t<T, IPT, 128, KAHAN>(L, s);
    case 256: return launch_forces_t<T, IPT, 256, KAHAN>(L, s);
    case 512: return launch_forces_t<T, IPT, 512, KAHAN>(L, s);
    default:  return hipErrorInvalidValue;
  }
}

template <typename T, bool KAHAN>
hipError_t launch_forces_ipt(const ForceLaunch &L, hipStream_t s) {
  switch (L.ipt) {
    case 1: return launch_forces_tile<T, 1, KAHAN>(L, s);
    case 2: return launch_forces_tile<T, 2, KAHAN>(L, s);
    case 4: return launch_forces_tile<T, 4, KAHAN>(L, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

// L.wave >= 2: the block kernel (kernels_block.hip); dt > 0 makes it the whole Tick body
static BlockLaunch block_launch(const ForceLaunch &L, void *posm_out, void *vel, void *acc, float dt) {
  BlockLaunch b;
  b.posm = L.posm; b.posm_out = posm_out; b.vel = vel; b.acc = acc;
  b.n_total = L.n_total; b.i_begin = L.i_begin; b.i_count = L.i_count;
  b.np = L.wave; b.G = L.G; b.eps2 = L.eps2; b.dt = dt; b.uni = L.uni; b.general = L.general; b.optimistic = L.guarded ? 0 : 1;
  return b;
}

// L.wave != 0 in the other two precisions: forces_block_kernel, L.wave bodies per workgroup (4 or 8)
template <typename T, bool KAHAN>
static hipError_t launch_block_generic(const ForceLaunch &L, void *posm_out, void *vel, void *acc, float dt, hipStream_t s) {
  using V = typename V4<T>::type;
  if (L.wave != 4 && L.wave != 8) return hipErrorInvalidValue;
  dim3 grid((L.i_count + L.wave - 1) / L.wave), block(kBlock);
  const int integrate = dt > 0.f ? 1 : 0;
#define NBODY_LAUNCH_BLOCK(NB, ZM, ZP)                                                                                    \
  hipLaunchKernelGGL((forces_block_kernel<T, NB, ZM, KAHAN>), grid, block, 0, s, (const V *)L.posm, (V *)posm_out, (V *)vel, \
                     (V *)acc, L.n_total, L.i_begin, L.i_count, (T)L.G, (T)(ZP), (T)dt, integrate)
  if (L.eps2 > 0.0) { if (L.wave == 4) NBODY_LAUNCH_BLOCK(4, Z_SOFT, L.eps2); else NBODY_LAUNCH_BLOCK(8, Z_SOFT, L.eps2); }
  else if (sizeof(T) == 8) { if (L.wave == 4) NBODY_LAUNCH_BLOCK(4, Z_SELECT, 0.0); else NBODY_LAUNCH_BLOCK(8, Z_SELECT, 0.0); }
  else { if (L.wave == 4) NBODY_LAUNCH_BLOCK(4, Z_CLAMP, -0x1p126); else NBODY_LAUNCH_BLOCK(8, Z_CLAMP, -0x1p126); }
#undef NBODY_LAUNCH_BLOCK
  return hipGetLastError();
}

static hipError_t launch_forces_wave(const ForceLaunch &L, hipStream_t s) {
  if (L.precision == NBODY_PREC_F64) return launch_block_generic<double, false>(L, nullptr, nullptr, L.accp, 0.f, s);
  if (L.precision == NBODY_PREC_F32_KAHAN) return launch_block_generic<float, true>(L, nullptr, nullptr, L.accp, 0.f, s);
  if (L.wave < 2) return hipErrorInvalidValue;
  if (L.general != nullptr && L.check_masses)     // somebody else may have written the buffer: the device looks at the masses
    hipLaunchKernelGGL(mass_check_kernel<float>, dim3((L.n_total + kBlock - 1) / kBlock), dim3(kBlock), 0, s,
                       (const float4 *)L.posm, L.n_total, (int *)L.general);
  return launch_block(block_launch(L, nullptr, nullptr, L.accp, 0.f), s);
}

// One whole Tick body of a small or mid-size single-context system in one launch: forces + kick-drift into posm_out.
hipError_t launch_step_small(const ForceLaunch &L, void *posm_out, void *vel, void *acc, float dt, hipStream_t s, void *stage,
                             void *size_bits, void *size_zero) {
  if (L.wave == 0 || L.i_begin != 0 || L.i_count != L.n_total || !(dt > 0.f)) return hipErrorInvalidValue;
  if (L.precision != NBODY_PREC_F32) {
    if (stage != nullptr || size_bits != nullptr) return hipErrorInvalidValue;       // the mirror rides with the packed kernel only
    return L.precision == NBODY_PREC_F64 ? launch_block_generic<double, false>(L, posm_out, vel, acc, dt, s)
                                         : launch_block_generic<float, true>(L, posm_out, vel, acc, dt, s);
  }
  if (L.wave < 2 || L.uni < 0) return hipErrorInvalidValue;
  BlockLaunch b = block_launch(L, posm_out, vel, acc, dt);
  b.stage = stage; b.size_bits = size_bits; b.size_zero = size_zero;
  return launch_block(b, s);
}

hipError_t launch_forces(const ForceLaunch &L, hipStream_t s) {
  if (L.i_count <= 0 || L.n_total <= 0 || L.j_split <= 0 || L.j_chunk <= 0) return hipErrorInvalidValue;
  if (L.wave != 0) {
    if (L.j_split != 1) return hipErrorInvalidValue;
    return launch_forces_wave(L, s);
  }
  if (L.j_chunk % L.tile != 0 && L.j_split > 1) return hipErrorInvalidValue;
  switch (L.precision) {
    case NBODY_PREC_F32:       return launch_forces_ipt<float, false>(L, s);
    case NBODY_PREC_F32_KAHAN: return launch_forces_ipt<float, true>(L, s);
    case NBODY_PREC_F64:       return launch_forces_ipt<double, false>(L, s);
    default: return hipErrorInvalidValue;
  }
}

void forces_geometry(const ForceLaunch &L, int *blocks, int *threads) {
  if (L.wave != 0) {
    const int per = L.precision == NBODY_PREC_F32 ? 2 * L.wave : L.wave;   // bodies of a workgroup
    if (blocks) *blocks = (L.i_count + per - 1) / per;
    if (threads) *threads = kBlock;
    return;
  }
  const int iblocks = (L.i_count + kBlock * L.ipt - 1) / (kBlock * L.ipt);
  if (blocks) *blocks = iblocks * L.j_split;
  if (threads) *threads = kBlock;
}

hipError_t launch_update(int precision, void *posm, void *vel, void *acc, const void *accp, int i_begin, int i_count,
                         int j_split, float dt, hipStream_t s) {
  if (i_count <= 0) return hipErrorInvalidValue;
  dim3 grid((i_count + kBlock - 1) / kBlock), block(kBlock);
  const int integrate = dt > 0.0f ? 1 : 0;
  switch (precision) {
    case NBODY_PREC_F32:
      hipLaunchKernelGGL((update_kernel<float, false>), grid, block, 0, s, (float4 *)posm, (float4 *)vel, (float4 *)acc,
                         (const float4 *)accp, i_begin, i_count, j_split, dt, integrate);
      break;
    case NBODY_PREC_F32_KAHAN:
      hipLaunchKernelGGL((update_kernel<float, true>), grid, block, 0, s, (float4 *)posm, (float4 *)vel, (float4 *)acc,
                         (const float4 *)accp, i_begin, i_count, j_split, dt, integrate);
      break;
    case NBODY_PREC_F64:
      hipLaunchKernelGGL((update_kernel<double, false>), grid, block, 0, s, (double4 *)posm, (double4 *)vel,
                         (double4 *)acc, (const double4 *)accp, i_begin, i_count, j_split, (double)dt, integrate);
      break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

hipError_t launch_bounds(int precision, const void *posm, int i_begin, int i_count, unsigned int *out_bits,
                         hipStream_t s, unsigned int *zero_word) {
  if (i_count <= 0) return hipErrorInvalidValue;
  int blocks = (i_count + kBlock - 1) / kBlock;
  if (blocks > 512) blocks = 512;
  if (precision == NBODY_PREC_F64)
    hipLaunchKernelGGL((bounds_kernel<double, false>), dim3(blocks), dim3(kBlock), 0, s, (const double4 *)posm, i_begin,
                       i_count, out_bits, zero_word);
  else
    hipLaunchKernelGGL((bounds_kernel<float, false>), dim3(blocks), dim3(kBlock), 0, s, (const float4 *)posm, i_begin,
                       i_count, out_bits, zero_word);
  return hipGetLastError();
}

hipError_t launch_massmax(int precision, const void *posm, int n_total, unsigned int *out_bits, hipStream_t s) {
  if (n_total <= 0) return hipErrorInvalidValue;
  int blocks = (n_total + kBlock - 1) / kBlock;
  if (blocks > 512) blocks = 512;
  if (precision == NBODY_PREC_F64)
    hipLaunchKernelGGL((bounds_kernel<double, true>), dim3(blocks), dim3(kBlock), 0, s, (const double4 *)posm, 0, n_total,
                       out_bits, (unsigned int *)nullptr);
  else
    hipLaunchKernelGGL((bounds_kernel<float, true>), dim3(blocks), dim3(kBlock), 0, s, (const float4 *)posm, 0, n_total,
                       out_bits, (unsigned int *)nullptr);
  return hipGetLastError();
}

// Renderer hand-off: repack the owned slice into FParticle records (OctreeSearch.h:8-18, 40 bytes) on the device,
// so that one D2H copy feeds what DrawDebugPoint reads (OctreeSearch.cpp:41) instead of three SoA copies.
template <typename T>
__global__ __launch_bounds__(kBlock) void pack_particles_kernel(const typename V4<T>::type *__restrict__ posm,
                                                                const typename V4<T>::type *__restrict__ vel,
                                                                const typename V4<T>::type *__restrict__ acc,
                                                                float *__restrict__ out, int i_begin, int i_count) {
  const int il = blockIdx.x * kBlock + threadIdx.x;
  if (il >= i_count) return;
  const auto p = posm[i_begin + il];
  const auto v = vel[il];
  const auto a = acc[il];
  float *o = out + (size_t)il * 10;
  o[0] = (float)p.w; o[1] = (float)p.x; o[2] = (float)p.y; o[3] = (float)p.z;
  o[4] = (float)v.x; o[5] = (float)v.y; o[6] = (float)v.z;
  o[7] = (float)a.x; o[8] = (float)a.y; o[9] = (float)a.z;
}

// xyz of bodies [first, first+count) of the whole system as packed float3.
template <typename T>
__global__ __launch_bounds__(kBlock) void pack_positions_kernel(const typename V4<T>::type *__restrict__ posm,
                                                                float *__restrict__ out, int first, int count) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= count) return;
  const auto p = posm[first + i];
  out[3 * (size_t)i] = (float)p.x; out[3 * (size_t)i + 1] = (float)p.y; out[3 * (size_t)i + 2] = (float)p.z;
}

hipError_t launch_pack_particles(int precision, const void *posm, const void *vel, const void *acc, float *out,
                                 int i_begin, int i_count, hipStream_t s) {
  if (i_count <= 0) return hipErrorInvalidValue;
  dim3 grid((i_count + kBlock - 1) / kBlock), block(kBlock);
  if (precision == NBODY_PREC_F64)
    hipLaunchKernelGGL((pack_particles_kernel<double>), grid, block, 0, s, (const double4 *)posm, (const double4 *)vel,
                       (const double4 *)acc, out, i_begin, i_count);
  else
    hipLaunchKernelGGL((pack_particles_kernel<float>), grid, block, 0, s, (const float4 *)posm, (const float4 *)vel,
                       (const float4 *)acc, out, i_begin, i_count);
  return hipGetLastError();
}

hipError_t launch_pack_positions(int precision, const void *posm, float *out, int first, int count, hipStream_t s) {
  if (count <= 0) return hipErrorInvalidValue;
  dim3 grid((count + kBlock - 1) / kBlock), block(kBlock);
  if (precision == NBODY_PREC_F64)
    hipLaunchKernelGGL((pack_positions_kernel<double>), grid, block, 0, s, (const double4 *)posm, out, first, count);
  else
    hipLaunchKernelGGL((pack_positions_kernel<float>), grid, block, 0, s, (const float4 *)posm, out, first, count);
  return hipGetLastError();
}

namespace {
void energy_geometry(int n_total, int i_count, int *iblocks, int *j_split, int *j_chunk) {
  *iblocks = (i_count + kBlock - 1) / kBlock;
  int js = 1;
  while (*iblocks * js < 2048 && n_total / (js * 2) >= 4 * kBlock) js *= 2;
  int chunk = (n_total + js - 1) / js;
  chunk = (chunk + kBlock - 1) / kBlock * kBlock;
  *j_split = (n_total + chunk - 1) / chunk;
  *j_chunk = chunk;
}
}  // namespace

size_t energy_partials(int n_total, int i_count) {
  int iblocks, j_split, j_chunk;
  energy_geometry(n_total, i_count > 0 ? i_count : 1, &iblocks, &j_split, &j_chunk);
  return (size_t)iblocks * j_split * 2;
}

hipError_t launch_energy(int precision, const void *posm, const void *vel, int n_total, int i_begin, int i_count,
                         double G, double eps2, double *partials, double *out, hipStream_t s) {
  if (i_count <= 0) return hipErrorInvalidValue;
  int iblocks, j_split, j_chunk;
  energy_geometry(n_total, i_count, &iblocks, &j_split, &j_chunk);
  dim3 grid(iblocks, j_split), block(kBlock);
  if (precision == NBODY_PREC_F64)
    hipLaunchKernelGGL((energy_kernel<double>), grid, block, 0, s, (const double4 *)posm, (const double4 *)vel, n_total,
                       i_begin, i_count, j_chunk, G, eps2, partials);
  else
    hipLaunchKernelGGL((energy_kernel<float>), grid, block, 0, s, (const float4 *)posm, (const float4 *)vel, n_total,
                       i_begin, i_count, j_chunk, G, eps2, partials);
  hipLaunchKernelGGL(energy_fold_kernel, dim3(1), block, 0, s, partials, iblocks * j_split, out);
  return hipGetLastError();
}

}  // namespace nbody

// Shared device pieces of the symmetric force kernels (kernels_sym.hip: fp32 packed; kernels_sym64.hip: fp64).
#pragma once
#include <hip/hip_runtime.h>

#include "pk_common.h"
#include "sym_plan.h"

namespace nbody {
namespace {

constexpr int kJT = 256;    // j tile (bodies), 4 subtiles of 64

// segment loads in flight per wave of a row fold (fold_way below).  Same-box A/B of whole steps with one thread per body
// (profiles/r02_ab_fold_unroll.txt): update 59 -> 47 us at N = 65536 and 37 -> 31 us at N = 32768 with eight in flight;
// sixteen no better.
#ifndef NBODY_SYM_FOLD_UNROLL
#define NBODY_SYM_FOLD_UNROLL 8
#endif

// Zero-mass padding bodies sit far outside any scene: the symmetric tiles may run without a d == 0 guard, and a pad at
// the origin would coincide with a body at the origin — the reference pins body 0 there — and 0 * inf = NaN.
// fp32: at 1e30 the squared distance from a pad to any body overflows to +inf, v_rsq_f32(+inf) = +0, so |d|^-3 is
// exactly 0 and so is the term — whatever it is multiplied with (G m_j = 0 in the general kernels, nothing at all in the
// equal-mass kernels, which have no per-body factor to hide a pad behind).  The only float within reach of the pad
// point is the pad point itself (ulp(1e30) = 7.6e22); a body exactly there selects the guarded kernel (dup_detect).
// fp64: 1e120 — the squared distance stays finite (the Newton step of rsq64 would turn inf into NaN) and |d|^-3 =
// 1e-361 underflows to exactly 0.  Doubles within ~6e107 of that point exist (ulp = 1.5e104); mass_check_kernel sends
// a scene with a body out there (all three coordinates > 9e119) to the general kernels, where the pad's zero mass does it.
constexpr float kPadFar = 1.0e30f;
constexpr double kPadFar64 = 1.0e120;
template <typename T> __device__ __forceinline__ T pad_far();
template <> __device__ __forceinline__ float pad_far<float>() { return kPadFar; }
template <> __device__ __forceinline__ double pad_far<double>() { return kPadFar64; }

// lane l+1 <- lane l, lane 0 <- lane 63 (v_mov_b32_dpp wave_ror:1; a half-rate VALU op on gfx950)
__device__ __forceinline__ float wave_ror1(float v) {
  const int i = __builtin_bit_cast(int, v);   // every lane is written, so `old` is irrelevant: pass the source (no v_mov to seed it)
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x13C, 0xf, 0xf, false));
}

__device__ __forceinline__ double wave_ror1(double v) {          // a double moves as two dwords
  const long long i = __builtin_bit_cast(long long, v);
  const int lo = (int)(i & 0xffffffffll), hi = (int)(i >> 32);
  const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, 0x13C, 0xf, 0xf, false);
  const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, 0x13C, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((long long)hi2 << 32) | (long long)(unsigned int)lo2);
}

template <typename T> struct SymVec;
template <> struct SymVec<float> { using type = float4; };
template <> struct SymVec<double> { using type = double4; };

__device__ __forceinline__ unsigned long long coord_bits(float v) { return (unsigned long long)__float_as_uint(v + 0.0f); }
__device__ __forceinline__ unsigned long long coord_bits(double v) { return (unsigned long long)__double_as_longlong(v + 0.0); }

// Do two different bodies share a position?  Every body inserts a 64-bit hash of its three coordinates into an
// open-addressing table (pre-zeroed, >= 2n slots); meeting its own hash again sets *flag.  A hash collision between
// different positions also sets it — that only selects the guarded kernel for this pass, never a wrong result.
// flag2 (optional): a second verdict word that takes every finding of this call too — the verdict on the bodies inserted
// so far, for a pass that inserts them in two goes (own slice first, the rest once their positions have arrived).
template <typename T, typename V>
__device__ __forceinline__ void dup_detect(const V p, unsigned long long *__restrict__ table, unsigned int mask,
                                           int *__restrict__ flag, int *__restrict__ flag2 = nullptr) {
  auto raise = [&]() { atomicExch(flag, 1); if (flag2) atomicExch(flag2, 1); };
  // d == 0 also happens for DIFFERENT positions when every squared difference underflows: only possible if both
  // bodies sit within ~1e-12 of the origin on all three axes (elsewhere two distinct floats differ by >= 1 ulp of
  // their own size).  Two or more bodies in that cube -> guarded kernel.  flag[1] counts them.
  if (fabs((double)p.x) < 1e-12 && fabs((double)p.y) < 1e-12 && fabs((double)p.z) < 1e-12) {
    if (atomicAdd(flag + 1, 1) >= 1) atomicExch(flag, 1);
    if (flag2 && atomicAdd(flag2 + 1, 1) >= 1) atomicExch(flag2, 1);
  }
  // the unguarded kernels park their zero-mass padding at pad_far on all three axes: a body exactly there would meet it at d == 0
  if (p.x == pad_far<T>() && p.y == pad_far<T>() && p.z == pad_far<T>()) raise();
  // coord_bits adds +0 first: -0 and +0 are the same position
  unsigned long long h = coord_bits(p.x) * 0x9E3779B97F4A7C15ull;
  h = (h ^ (h >> 29)) + coord_bits(p.y) * 0xBF58476D1CE4E5B9ull;
  h = (h ^ (h >> 31)) + coord_bits(p.z) * 0x94D049BB133111EBull;
  h ^= h >> 32;
  if (h == 0ull) h = 1ull;                                  // 0 marks an empty slot
  unsigned int slot = (unsigned int)(h * 0xD6E8FEB86659FD93ull >> 32) & mask;
  for (unsigned int probe = 0; probe <= mask; ++probe) {
    const unsigned long long old = atomicCAS(&table[slot], 0ull, h);
    if (old == 0ull) return;                                // inserted
    if (old == h) { raise(); return; }                      // somebody with the same position (or hash) is already in
    slot = (slot + 1) & mask;
  }
  raise();                                                  // table full (cannot happen at >= 2n slots): be safe
}

template <typename T>
__global__ __launch_bounds__(kBlock) void dup_detect_kernel(const typename SymVec<T>::type *__restrict__ posm, int n,
                                                            unsigned long long *__restrict__ table,
                                                            unsigned int mask, int *__restrict__ flag) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  dup_detect<T>(posm[i], table, mask, flag);
}

// fp64 (no preparation kernel there): the equal-mass test on its own — *general is raised (sticky) when a body's mass
// differs from body 0's or a body sits where the far-away padding is.
template <typename T>
__global__ __launch_bounds__(kBlock) void mass_check_kernel(const typename SymVec<T>::type *__restrict__ posm, int n,
                                                            int *__restrict__ general) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const typename SymVec<T>::type p = posm[i];
  const T edge = pad_far<T>() * (T)0.9;
  if (!(p.w == posm[0].w) || (p.x > edge && p.y > edge && p.z > edge)) *general = 1;
}

// fp32: what the force kernel reads is not posm but posg = (x, y, z, G*m), n_pad entries, zero-mass padding at kPadFar
// beyond n_total — so that its loads need neither bounds checks nor a multiply, and whole j tiles can go from HBM to
// LDS by DMA.  One pass over the positions per force pass (16 B read + 16 B written per body); the coincident-body
// detector rides along (DETECT), and so does the equal-mass test: a body whose mass differs from body 0's raises
// *general (sticky until the host resets it on a new state), which selects the general kernels and tells the update not
// to scale (see forces_sym_pk_kernel, UNI).
// A sharded context may prepare in two goes (SymLaunch::phase): the bodies of [b0, b1) — its own slice, whose positions its
// own update has just written — with inside = 1, and the others, once the all-gather has delivered them, with inside = 0.
// Each go looks at the masses of the bodies it prepares and at nothing else — the first go runs while the all-gather may still
// be writing the other slices' records, and a record in flight is not read, not even for a component the gather never changes
// (check_mass = 2; a pass in one go, check_mass = 1, looks at every body).  The reference mass is body `ref`'s: body 0 for a
// pass in one go, the own slice's first body otherwise — its record is the rank's own.  The verdict word is sticky between
// uploads and starts from the host's scan of the WHOLE uploaded state, so the first go's strips see every difference the
// upload held; a mass changed later through another rank's bound buffer raises the word in that step's second go, i.e. it
// takes effect one pass later: masses are immutable between uploads for sharded contexts (include/nbody.h).
// The first go leaves its coincident-body verdict in flag2 as well: the verdict on the own slice, which is all the strips
// inside that slice need.
template <bool DETECT>
__global__ __launch_bounds__(kBlock) void sym_prep_kernel(const float4 *__restrict__ posm, float4 *__restrict__ posg,
                                                          int n_total, int n_pad, float gscale,
                                                          unsigned long long *__restrict__ table, unsigned int mask,
                                                          int *__restrict__ flag, int *__restrict__ general,
                                                          int b0, int b1, int inside, int check_mass, int *__restrict__ flag2,
                                                          int ref) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_pad) return;
  const bool mine = (i >= b0 && i < b1) == (inside != 0);
  if (i >= n_total) { if (mine) posg[i] = make_float4(kPadFar, kPadFar, kPadFar, 0.f); return; }
  if (!mine && !(check_mass == 1 && general != nullptr)) return;
  float4 p = posm[i];
  if (check_mass && general != nullptr && !(p.w == posm[ref].w)) *general = 1;   // every writer writes the same value
  if (!mine) return;
  if (DETECT) dup_detect<float>(p, table, mask, flag, flag2);
  p.w *= gscale;
  posg[i] = p;
}

// ---- row folds ------------------------------------------------------------------------------------------------------
// A body's acceleration is the sum of the partial-sum segments that cover its 64-body granule, named in item order by a
// CSR list (sym_plan.h).  One thread per body walking that list is bound by load latency wherever the lists are long and
// the bodies few (N = 65536: ~150 segments per body, one wave per SIMD).  So a granule is folded by a WORKGROUP of four
// waves: wave w adds the list entries k = w, w + 4, w + 8, ... in order (NBODY_SYM_FOLD_UNROLL loads in flight each), the
// four partial sums meet in LDS and wave 0 adds them ((s0 + s1) + s2) + s3.  That — not the plain sequential sum — is the
// summation order of the symmetric pass; reduce_j_kernel, update_sym_kernel and update_sym_fused_kernel all use it, so
// the two-kernel and the fused path still agree in every bit.  KAHAN: every one of those additions is compensated.
constexpr int kFoldWays = kBlock / 64;

template <typename R, bool KAHAN> __device__ __forceinline__ void fold_add(R &sum, R &c, R v) {
  if (KAHAN) { const R yv = v - c; const R tt = sum + yv; c = (tt - sum) - yv; sum = tt; }
  else sum += v;
}

// this wave's share of granule g's list: entries ptr[g] + way, + kFoldWays, ...
template <typename R, bool KAHAN>
__device__ __forceinline__ void fold_way(const typename SymVec<R>::type *__restrict__ pool, const unsigned int *__restrict__ ptr,
                                         const unsigned int *__restrict__ off, int g, int l, int way, R &sx, R &sy, R &sz) {
  using V = typename SymVec<R>::type;
  R cx = 0, cy = 0, cz = 0;
  sx = 0; sy = 0; sz = 0;
  const unsigned int k1 = ptr[g + 1];
#pragma unroll NBODY_SYM_FOLD_UNROLL
  for (unsigned int k = ptr[g] + (unsigned int)way; k < k1; k += kFoldWays) {
    const V p = pool[(size_t)off[k] + l];
    fold_add<R, KAHAN>(sx, cx, p.x); fold_add<R, KAHAN>(sy, cy, p.y); fold_add<R, KAHAN>(sz, cz, p.z);
  }
}

// the four waves' partial sums -> wave 0 (all threads of the workgroup must call this; the result is valid for way 0)
template <typename R, bool KAHAN>
__device__ __forceinline__ void fold_meet(R (&sh)[kFoldWays][3][64], int l, int way, R &sx, R &sy, R &sz, R &cx, R &cy, R &cz) {
  sh[way][0][l] = sx; sh[way][1][l] = sy; sh[way][2][l] = sz;
  __syncthreads();
  cx = 0; cy = 0; cz = 0;
  if (way == 0) {
#pragma unroll
    for (int w = 1; w < kFoldWays; ++w) {
      fold_add<R, KAHAN>(sx, cx, sh[w][0][l]); fold_add<R, KAHAN>(sy, cy, sh[w][1][l]); fold_add<R, KAHAN>(sz, cz, sh[w][2][l]);
    }
  }
}

// send[b] = sum of the j-side segments that cover body b: what this rank's pairs contribute to b's acceleration as the
// "other" body.  One workgroup per 64-body granule (grid = granules of the whole system).
// It also clears the coincident-body detector's table and flag words for the NEXT pass (they were last read by this
// pass's force kernels, which precede this launch on the stream): a hipMemsetAsync per pass costs a launch plus ~6 us of
// host time in front of it, which at N = 32768 is 4 % of the step.  The table is zeroed once at creation; a pass that dies
// between the detector and this kernel leaves stale entries, which can only select the guarded kernel — never a wrong sum.
template <typename R, bool KAHAN>
__global__ __launch_bounds__(kBlock) void reduce_j_kernel(const typename SymVec<R>::type *__restrict__ pool,
                                                          typename SymVec<R>::type *__restrict__ send,
                                                          const unsigned int *__restrict__ j_ptr,
                                                          const unsigned int *__restrict__ j_off, int n_total,
                                                          unsigned long long *__restrict__ dup_table, int dup_words,
                                                          int accumulate) {
  using V = typename SymVec<R>::type;
  __shared__ R sh[kFoldWays][3][64];
  const int t = threadIdx.x, l = t & 63, way = t >> 6, g = blockIdx.x, b = g * 64 + l;
  for (int w = g * kBlock + t; w < dup_words; w += gridDim.x * kBlock) dup_table[w] = 0ull;
  R sx, sy, sz, cx, cy, cz;
  fold_way<R, KAHAN>(pool, j_ptr, j_off, g, l, way, sx, sy, sz);
  fold_meet<R, KAHAN>(sh, l, way, sx, sy, sz, cx, cy, cz);
  if (way != 0 || b >= n_total) return;
  V o; o.x = sx; o.y = sy; o.z = sz; o.w = 0;
  if (accumulate) {      // a later phase of a pass whose j-side segments share one pool area (sym_plan.h): on top of the earlier ones
    const V prev = send[b];
    o.x = prev.x + o.x; o.y = prev.y + o.y; o.z = prev.z + o.z;
  }
  send[b] = o;
}

template <typename T> __device__ __forceinline__ T mul_add_sep2(T a, T b, T c) {
#pragma clang fp contract(off)
  const T p = a * b;
  return c + p;
}

// Own body bl: acc = its i-side segments + the rows received from every rank (rank order), times the common G m if
// the equal-mass kernels ran (`general` non-null and clear); then optionally the reference's update
// (OctreeSearch.cpp:29-30), multiply and add kept apart.  One workgroup per own granule.
template <typename R, bool KAHAN>
__global__ __launch_bounds__(kBlock) void update_sym_kernel(typename SymVec<R>::type *__restrict__ posm,
                                                            typename SymVec<R>::type *__restrict__ vel,
                                                            typename SymVec<R>::type *__restrict__ acc,
                                                            const typename SymVec<R>::type *__restrict__ pool,
                                                            const unsigned int *__restrict__ i_ptr,
                                                            const unsigned int *__restrict__ i_off,
                                                            const typename SymVec<R>::type *__restrict__ recv, int i_begin,
                                                            int i_count, int n_src, R dt, int integrate,
                                                            const int *__restrict__ general, R gscale) {
  using V = typename SymVec<R>::type;
  __shared__ R sh[kFoldWays][3][64];
  const int t = threadIdx.x, l = t & 63, way = t >> 6, g = blockIdx.x, bl = g * 64 + l;
  R ax, ay, az, cx, cy, cz;
  fold_way<R, KAHAN>(pool, i_ptr, i_off, g, l, way, ax, ay, az);
  fold_meet<R, KAHAN>(sh, l, way, ax, ay, az, cx, cy, cz);
  if (way != 0 || bl >= i_count) return;
#pragma unroll 4
  for (int q = 0; q < n_src; ++q) {
    const V p = recv[(size_t)q * i_count + bl];
    fold_add<R, KAHAN>(ax, cx, p.x); fold_add<R, KAHAN>(ay, cy, p.y); fold_add<R, KAHAN>(az, cz, p.z);
  }
  if (general != nullptr && *general == 0) {      // the equal-mass kernels summed |d|^-3 d: the common G m comes in here
    // (body 0's owner may be storing its new position at this moment: the mass word it stores is the one already there)
    const R gm = posm[0].w * gscale;
    ax *= gm; ay *= gm; az *= gm;
  }
  V ao; ao.x = ax; ao.y = ay; ao.z = az; ao.w = 0;
  acc[bl] = ao;
  if (integrate) {
    V v = vel[bl], x = posm[i_begin + bl];
    v.x = mul_add_sep2(dt, ax, v.x); v.y = mul_add_sep2(dt, ay, v.y); v.z = mul_add_sep2(dt, az, v.z);
    x.x = mul_add_sep2(dt, v.x, x.x); x.y = mul_add_sep2(dt, v.y, x.y); x.z = mul_add_sep2(dt, v.z, x.z);
    vel[bl] = v;
    posm[i_begin + bl] = x;
  }
}

// The single-device fp32 form of the two kernels above in ONE launch, for contexts whose positions nobody else writes:
// acc = i-side segments + (j-side segments folded on their own, exactly as reduce_j_kernel would, then added as one
// term: the same bits as the two-kernel path), the reference's update, and the NEXT pass's preparation while the new
// position is in registers — posg[i] = (x, y, z, G m) and, DETECT, the body's entry in the next pass's coincident-body
// table (`next`; this pass's table `cur`, last read by this pass's force kernels, is cleared for the pass after).  A
// stepping loop then runs force kernel + this kernel per step: no sym_prep_kernel, no reduce_j_kernel, no memset.
template <bool KAHAN, bool DETECT>
__global__ __launch_bounds__(2 * kBlock) void update_sym_fused_kernel(float4 *__restrict__ posm, float4 *__restrict__ vel,
                                                                      float4 *__restrict__ acc, float4 *__restrict__ posg,
                                                                      const float4 *__restrict__ pool,
                                                                      const unsigned int *__restrict__ i_ptr,
                                                                      const unsigned int *__restrict__ i_off,
                                                                      const unsigned int *__restrict__ j_ptr,
                                                                      const unsigned int *__restrict__ j_off, int n_total,
                                                                      float gscale, float dt, int integrate,
                                                                      unsigned long long *__restrict__ next, unsigned int mask,
                                                                      unsigned long long *__restrict__ cur, int cur_words,
                                                                      const int *__restrict__ general) {
  // Eight waves per granule: waves 0-3 fold the j-side list (reduce_j_kernel's sum, same four ways, same order), waves 4-7
  // the i-side list at the same time — the two folds are chains of dependent loads, and small systems have few granules.
  __shared__ float sh[2][kFoldWays][3][64];
  __shared__ float sj[3][64];
  const int t = threadIdx.x, l = t & 63, way = (t >> 6) & (kFoldWays - 1), side = t >> 8, g = blockIdx.x, bl = g * 64 + l;
#if defined(NBODY_UPD_EXPERIMENT) && NBODY_UPD_EXPERIMENT == 1      // A/B builds only (tools/ab_update_parts.sh): what the launch alone costs
  return;
#endif
  if (DETECT) for (int w = g * 2 * kBlock + t; w < cur_words; w += gridDim.x * 2 * kBlock) cur[w] = 0ull;
  float ax, ay, az, cx, cy, cz;
  if (side == 0) fold_way<float, KAHAN>(pool, j_ptr, j_off, g, l, way, ax, ay, az);
  else           fold_way<float, KAHAN>(pool, i_ptr, i_off, g, l, way, ax, ay, az);
  sh[side][way][0][l] = ax; sh[side][way][1][l] = ay; sh[side][way][2][l] = az;
  __syncthreads();
#if defined(NBODY_UPD_EXPERIMENT) && NBODY_UPD_EXPERIMENT == 2      // ... and the folds
  if (ax != 12345.678f) return;
#endif
  cx = 0; cy = 0; cz = 0;
  if (way == 0) {                                                 // ((s0 + s1) + s2) + s3, on both sides (fold_meet)
#pragma unroll
    for (int w = 1; w < kFoldWays; ++w) {
      fold_add<float, KAHAN>(ax, cx, sh[side][w][0][l]); fold_add<float, KAHAN>(ay, cy, sh[side][w][1][l]);
      fold_add<float, KAHAN>(az, cz, sh[side][w][2][l]);
    }
    if (side == 0) { sj[0][l] = ax; sj[1][l] = ay; sj[2][l] = az; }
  }
  __syncthreads();
  if (side != 1 || way != 0 || bl >= n_total) return;
  fold_add<float, KAHAN>(ax, cx, sj[0][l]); fold_add<float, KAHAN>(ay, cy, sj[1][l]); fold_add<float, KAHAN>(az, cz, sj[2][l]);
  if (general != nullptr && *general == 0) {      // equal-mass kernels: the common G m comes in here (update_sym_kernel)
    const float gm = posm[0].w * gscale;
    ax *= gm; ay *= gm; az *= gm;
  }
  acc[bl] = make_float4(ax, ay, az, 0.f);
  float4 x = posm[bl];
  if (integrate) {
    float4 v = vel[bl];
    v.x = mul_add_sep2(dt, ax, v.x); v.y = mul_add_sep2(dt, ay, v.y); v.z = mul_add_sep2(dt, az, v.z);
    x.x = mul_add_sep2(dt, v.x, x.x); x.y = mul_add_sep2(dt, v.y, x.y); x.z = mul_add_sep2(dt, v.z, x.z);
    vel[bl] = v;
    posm[bl] = x;
  }
#if !(defined(NBODY_UPD_EXPERIMENT) && NBODY_UPD_EXPERIMENT == 3)   // ... and everything but the detector's entry
  if (DETECT) dup_detect<float>(x, next, mask, (int *)(next + (size_t)mask + 1));
#endif
  x.w *= gscale;
  posg[bl] = x;
}

}  // namespace
}  // namespace nbody

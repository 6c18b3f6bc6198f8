// Shared device pieces of the symmetric force kernels (kernels_sym.hip: fp32 packed; kernels_sym64.hip: fp64).
#pragma once
#include <hip/hip_runtime.h>

#include "pk_common.h"
#include "sym_plan.h"

namespace nbody {
namespace {

constexpr int kJT = 256;    // j tile (bodies), 4 subtiles of 64

// segment loads in flight per thread in the row folds (reduce_j_kernel, update_sym_kernel, update_sym_fused_kernel): one
// thread per body walks its granule's segment list, so at mid sizes (N = 65536: ~150 segments per body, 256 workgroups)
// the fold is bound by load latency, not bandwidth.  Same-box A/B of whole steps (profiles/r02_ab_fold_unroll.txt): update
// 59 -> 47 us at N = 65536 and 37 -> 31 us at N = 32768 with eight in flight (step 0.686 -> 0.673 ms, 0.221 -> 0.214 ms);
// sixteen no better.  The adds stay one dependent chain in list order: the bits do not change.
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
template <typename T, typename V>
__device__ __forceinline__ void dup_detect(const V p, unsigned long long *__restrict__ table, unsigned int mask,
                                           int *__restrict__ flag) {
  // d == 0 also happens for DIFFERENT positions when every squared difference underflows: only possible if both
  // bodies sit within ~1e-12 of the origin on all three axes (elsewhere two distinct floats differ by >= 1 ulp of
  // their own size).  Two or more bodies in that cube -> guarded kernel.  flag[1] counts them.
  if (fabs((double)p.x) < 1e-12 && fabs((double)p.y) < 1e-12 && fabs((double)p.z) < 1e-12)
    if (atomicAdd(flag + 1, 1) >= 1) atomicExch(flag, 1);
  // the unguarded kernels park their zero-mass padding at pad_far on all three axes: a body exactly there would meet it at d == 0
  if (p.x == pad_far<T>() && p.y == pad_far<T>() && p.z == pad_far<T>()) atomicExch(flag, 1);
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
    if (old == h) { atomicExch(flag, 1); return; }          // somebody with the same position (or hash) is already in
    slot = (slot + 1) & mask;
  }
  atomicExch(flag, 1);                                      // table full (cannot happen at >= 2n slots): be safe
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
template <bool DETECT>
__global__ __launch_bounds__(kBlock) void sym_prep_kernel(const float4 *__restrict__ posm, float4 *__restrict__ posg,
                                                          int n_total, int n_pad, float gscale,
                                                          unsigned long long *__restrict__ table, unsigned int mask,
                                                          int *__restrict__ flag, int *__restrict__ general) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_pad) return;
  if (i >= n_total) { posg[i] = make_float4(kPadFar, kPadFar, kPadFar, 0.f); return; }
  float4 p = posm[i];
  if (DETECT) dup_detect<float>(p, table, mask, flag);
  if (general != nullptr && !(p.w == posm[0].w)) *general = 1;     // every writer writes the same value
  p.w *= gscale;
  posg[i] = p;
}

// send[b] = sum, in item order, of the j-side segments that cover body b: what this rank's pairs contribute to b's
// acceleration as the "other" body.  One wave per 64-body granule; the granule's segment list is CSR (sym_plan.h).
// KAHAN: segments are added with a compensated sum.
// It also clears the coincident-body detector's table and flag words for the NEXT pass (they were last read by this
// pass's force kernels, which precede this launch on the stream): a hipMemsetAsync per pass costs a launch plus ~6 us of
// host time in front of it, which at N = 32768 is 4 % of the step.  The table is zeroed once at creation; a pass that dies
// between the detector and this kernel leaves stale entries, which can only select the guarded kernel — never a wrong sum.
template <typename R, bool KAHAN>
__global__ __launch_bounds__(kBlock) void reduce_j_kernel(const typename SymVec<R>::type *__restrict__ pool,
                                                          typename SymVec<R>::type *__restrict__ send,
                                                          const unsigned int *__restrict__ j_ptr,
                                                          const unsigned int *__restrict__ j_off, int n_total,
                                                          unsigned long long *__restrict__ dup_table, int dup_words) {
  using V = typename SymVec<R>::type;
  const int b = blockIdx.x * kBlock + threadIdx.x;
  for (int w = b; w < dup_words; w += gridDim.x * kBlock) dup_table[w] = 0ull;
  if (b >= n_total) return;
  const int g = b >> 6, l = b & 63;
  R sx = 0, sy = 0, sz = 0, cx = 0, cy = 0, cz = 0;
  auto add = [](R &sum, R &c, R v) {
    if (KAHAN) { const R yv = v - c; const R tt = sum + yv; c = (tt - sum) - yv; sum = tt; }
    else sum += v;
  };
  const unsigned int k1 = j_ptr[g + 1];
#pragma unroll NBODY_SYM_FOLD_UNROLL
  for (unsigned int k = j_ptr[g]; k < k1; ++k) {
    const V p = pool[(size_t)j_off[k] + l];
    add(sx, cx, p.x); add(sy, cy, p.y); add(sz, cz, p.z);
  }
  V o; o.x = sx; o.y = sy; o.z = sz; o.w = 0;
  send[b] = o;
}

template <typename T> __device__ __forceinline__ T mul_add_sep2(T a, T b, T c) {
#pragma clang fp contract(off)
  const T p = a * b;
  return c + p;
}

// Own body bl: acc = its i-side segments (item order) + the rows received from every rank (rank order), times the
// common G m if the equal-mass kernels ran (`general` non-null and clear); then optionally the reference's update (OctreeSearch.cpp:29-30), multiply and add kept apart.
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
  const int bl = blockIdx.x * kBlock + threadIdx.x;
  if (bl >= i_count) return;
  const int g = bl >> 6, l = bl & 63;
  R ax = 0, ay = 0, az = 0, cx = 0, cy = 0, cz = 0;
  auto add = [](R &sum, R &c, R v) {
    if (KAHAN) { const R yv = v - c; const R tt = sum + yv; c = (tt - sum) - yv; sum = tt; }
    else sum += v;
  };
  const unsigned int k1 = i_ptr[g + 1];
#pragma unroll NBODY_SYM_FOLD_UNROLL
  for (unsigned int k = i_ptr[g]; k < k1; ++k) {
    const V p = pool[(size_t)i_off[k] + l];
    add(ax, cx, p.x); add(ay, cy, p.y); add(az, cz, p.z);
  }
#pragma unroll NBODY_SYM_FOLD_UNROLL
  for (int q = 0; q < n_src; ++q) {
    const V p = recv[(size_t)q * i_count + bl];
    add(ax, cx, p.x); add(ay, cy, p.y); add(az, cz, p.z);
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
// acc = i-side segments + (j-side segments summed on their own, exactly as reduce_j_kernel would, then added as one
// term: the same bits as the two-kernel path), the reference's update, and the NEXT pass's preparation while the new
// position is in registers — posg[i] = (x, y, z, G m) and, DETECT, the body's entry in the next pass's coincident-body
// table (`next`; this pass's table `cur`, last read by this pass's force kernels, is cleared for the pass after).  A
// stepping loop then runs force kernel + this kernel per step: no sym_prep_kernel, no reduce_j_kernel, no memset.
template <bool KAHAN, bool DETECT>
__global__ __launch_bounds__(kBlock) void update_sym_fused_kernel(float4 *__restrict__ posm, float4 *__restrict__ vel,
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
  const int bl = blockIdx.x * kBlock + threadIdx.x;
  if (DETECT) for (int w = bl; w < cur_words; w += gridDim.x * kBlock) cur[w] = 0ull;
  if (bl >= n_total) return;
  const int g = bl >> 6, l = bl & 63;
  auto add = [](float &sum, float &c, float v) {
    if (KAHAN) { const float yv = v - c; const float tt = sum + yv; c = (tt - sum) - yv; sum = tt; }
    else sum += v;
  };
  float sx = 0, sy = 0, sz = 0, dx = 0, dy = 0, dz = 0;           // the j-side row of this body (reduce_j_kernel's sum)
  {
    const unsigned int k1 = j_ptr[g + 1];
#pragma unroll NBODY_SYM_FOLD_UNROLL
    for (unsigned int k = j_ptr[g]; k < k1; ++k) {
      const float4 p = pool[(size_t)j_off[k] + l];
      add(sx, dx, p.x); add(sy, dy, p.y); add(sz, dz, p.z);
    }
  }
  float ax = 0, ay = 0, az = 0, cx = 0, cy = 0, cz = 0;
  {
    const unsigned int k1 = i_ptr[g + 1];
#pragma unroll NBODY_SYM_FOLD_UNROLL
    for (unsigned int k = i_ptr[g]; k < k1; ++k) {
      const float4 p = pool[(size_t)i_off[k] + l];
      add(ax, cx, p.x); add(ay, cy, p.y); add(az, cz, p.z);
    }
  }
  add(ax, cx, sx); add(ay, cy, sy); add(az, cz, sz);
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
  if (DETECT) dup_detect<float>(x, next, mask, (int *)(next + (size_t)mask + 1));
  x.w *= gscale;
  posg[bl] = x;
}

}  // namespace
}  // namespace nbody

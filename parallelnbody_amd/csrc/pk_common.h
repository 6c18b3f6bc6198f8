// Shared device helpers of the packed-fp32 force kernels (kernels.hip, kernels_sym.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace nbody {
namespace {

constexpr int kBlock = 256;   // 4 waves of 64 lanes: one per SIMD of a CU

// How a pair at distance exactly 0 (the self pair, or two bodies on one point) is kept out of the sum —
// the reference's `if (d == 0) return;` (OctreeSearch.h:102):
//   Z_SOFT   : eps2 > 0 is added to r^2; rsq stays finite and the pair contributes s*0 = 0.
//   Z_CLAMP  : r2' = r2 + clamp01(1 - r2*2^126): exactly r2 for every normal r2 > 0, exactly 1 for r2 == 0
//              (then s = G*m is finite and s*0 = 0).  Two full-rate VALU ops (v_fma ... clamp, v_add).
//   Z_SELECT : rinv = r2 > 0 ? rsq(r2) : 0 — v_cmp + v_cndmask, both half-rate on gfx950; kept for A/B.
//   Z_BARE   : no handling at all — only for tiles that hold no self pair, on inputs proven free of coincident
//              bodies (dup_detect_kernel, sym_common.h).
enum { Z_SOFT = 0, Z_CLAMP = 1, Z_SELECT = 2, Z_BARE = 3 };

__device__ __forceinline__ float rsq_dev(float x) { return __builtin_amdgcn_rsqf(x); }   // v_rsq_f32, 1 ulp

// Which clock did the kernel run at?  The force loops are power-limited: the clock a box holds under them differs by several
// per cent from box to box (2.13 - 2.33 GHz seen), so a time alone cannot tell a slower box from slower code.  A wave can read two
// counters: s_memtime counts SHADER-clock cycles, s_memrealtime a fixed reference (100 MHz: hipDeviceAttributeWallClockRate) —
// profiles/r04_microbench_clock_counters.txt shows the first following the load, the second not.  With clk != nullptr
// (nbody_params.time_kernels) every workgroup adds its own two intervals to clk[0] / clk[1]; their ratio is the clock the
// kernel's workgroups saw, weighted by how long each ran (nbody_kernel_clock).  Two scalar reads at either end of a workgroup
// that runs for tens of microseconds to milliseconds, two atomics per workgroup; nothing when clk is null.
struct ClockStamp { unsigned long long t, r; };
__device__ __forceinline__ ClockStamp clock_begin(const unsigned long long *clk) {
  ClockStamp s{0ull, 0ull};
  if (clk != nullptr) { s.t = __builtin_amdgcn_s_memtime(); s.r = __builtin_amdgcn_s_memrealtime(); }
  return s;
}
// item >= 0 and clk[2] != 0 (tools/even_items.py; NBODY_SYM_ITEM_CLOCKS=1 at creation sized the buffer for it): the
// workgroup also leaves its own two reference-clock stamps in clk[8 + 2 item], clk[9 + 2 item].
__device__ __forceinline__ void clock_end(unsigned long long *clk, const ClockStamp &s, int item = -1) {
  if (clk != nullptr) {
    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t = __builtin_amdgcn_s_memtime() - s.t, r = now - s.r;
    if (threadIdx.x == 0) {
      atomicAdd(clk, t); atomicAdd(clk + 1, r);
      if (item >= 0 && clk[2] != 0ull) { clk[8 + 2 * item] = s.r; clk[9 + 2 * item] = now; }
    }
  }
}

typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 splat2(float v) { return f2{v, v}; }
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }

// a * b.y in both halves: one v_pk_mul_f32 reading the (z, G*m) half of the LDS quad in place (hipcc would
// first copy the mass down with a v_mov_b32)
__device__ __forceinline__ f2 mul_bcast_hi(f2 a, f2 b) {
  f2 o;
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(o) : "v"(a), "v"(b));
  return o;
}

template <bool KAHAN> struct Acc3pk;
template <> struct Acc3pk<false> {
  f2 x = splat2(0.f), y = splat2(0.f), z = splat2(0.f);
  __device__ __forceinline__ void add(f2 s, f2 dx, f2 dy, f2 dz) { x = fma2(s, dx, x); y = fma2(s, dy, y); z = fma2(s, dz, z); }
};
template <> struct Acc3pk<true> {
  f2 x = splat2(0.f), y = splat2(0.f), z = splat2(0.f), cx = splat2(0.f), cy = splat2(0.f), cz = splat2(0.f);
  static __device__ __forceinline__ void kadd(f2 &sum, f2 &c, f2 s, f2 d) {
    const f2 yv = fma2(s, d, -c);
    const f2 t = sum + yv;
    c = (t - sum) - yv;
    sum = t;
  }
  __device__ __forceinline__ void add(f2 s, f2 dx, f2 dy, f2 dz) { kadd(x, cx, s, dx); kadd(y, cy, s, dy); kadd(z, cz, s, dz); }
  // sum += v with compensation (v: a short plain partial sum)
  static __device__ __forceinline__ void kadd1(f2 &sum, f2 &c, f2 v) {
    const f2 yv = v - c;
    const f2 t = sum + yv;
    c = (t - sum) - yv;
    sum = t;
  }
  __device__ __forceinline__ void fold(const Acc3pk<false> &p) { kadd1(x, cx, p.x); kadd1(y, cy, p.y); kadd1(z, cz, p.z); }
};


// The pair law for JB j-bodies (x, y, z, G*m as read from the LDS tile) against NP register pairs of i-bodies,
// stage by stage (all differences, then all r^2, ...) so that no instruction's consumer is adjacent to it.
// zp2 = eps^2 (Z_SOFT) or -2^126 (Z_CLAMP) in both halves; one2 = (1, 1); both live in VGPRs.
// UNI: every body has the same mass — no G*m_j factor here; the caller multiplies the finished sums by the common G*m.
template <int NP, int JB, int ZMODE, bool KAHAN, bool UNI = false>
__device__ __forceinline__ void pair_group_pk(const f2 (&xi)[NP], const f2 (&yi)[NP], const f2 (&zi)[NP],
                                              const float4 (&pj)[JB], f2 zp2, f2 one2, Acc3pk<KAHAN> (&a)[NP]) {
  f2 dx[JB][NP], dy[JB][NP], dz[JB][NP], w[JB][NP], u[JB][NP];
#pragma unroll
  for (int b = 0; b < JB; ++b)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      dx[b][p] = splat2(pj[b].x) - xi[p]; dy[b][p] = splat2(pj[b].y) - yi[p]; dz[b][p] = splat2(pj[b].z) - zi[p];
    }
#pragma unroll
  for (int b = 0; b < JB; ++b)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      if (ZMODE == Z_SOFT) w[b][p] = fma2(dz[b][p], dz[b][p], zp2);
      else                 w[b][p] = dz[b][p] * dz[b][p];
      w[b][p] = fma2(dy[b][p], dy[b][p], w[b][p]);
      w[b][p] = fma2(dx[b][p], dx[b][p], w[b][p]);
    }
  if (ZMODE == Z_CLAMP) {
#pragma unroll
    for (int b = 0; b < JB; ++b)
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        f2 nf;   // clamp01(1 - r2*2^126): 1 for r2 == 0, 0 for every normal r2 > 0
        asm("v_pk_fma_f32 %0, %1, %2, %3 clamp" : "=v"(nf) : "v"(w[b][p]), "v"(zp2), "v"(one2));
        w[b][p] = w[b][p] + nf;
      }
  }
#pragma unroll
  for (int b = 0; b < JB; ++b)
#pragma unroll
    for (int p = 0; p < NP; ++p) u[b][p] = f2{rsq_dev(w[b][p].x), rsq_dev(w[b][p].y)};
#pragma unroll
  for (int b = 0; b < JB; ++b)
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      // The inline-asm multiply must not read a v_rsq_f32 result directly: hipcc pads the trans->VALU
      // hazard only for its own instructions.  It therefore takes rinv^3, produced by two ordinary ops.
      w[b][p] = u[b][p] * u[b][p];
      w[b][p] = w[b][p] * u[b][p];
      if (!UNI) w[b][p] = mul_bcast_hi(w[b][p], f2{pj[b].z, pj[b].w});   // * G*m_j
    }
#pragma unroll
  for (int b = 0; b < JB; ++b)
#pragma unroll
    for (int p = 0; p < NP; ++p) a[p].add(w[b][p], dx[b][p], dy[b][p], dz[b][p]);
}

}  // namespace
}  // namespace nbody

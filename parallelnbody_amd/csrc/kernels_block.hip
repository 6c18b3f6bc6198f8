// gfx950 (MI355X, CDNA4): the one-launch step of small and mid-size fp32 systems.
//
// Reference lines restated on the device (paths relative to /root/reference/Source/NBody/):
//   forces_block_pk_kernel <- the pair law OctreeSearch.h:101-104 summed over all j (the loop OctreeSearch.cpp:83-86 at
//                             theta = 0) and, when integrating, the Tick's update OctreeSearch.cpp:28-31 for the same bodies
//
// One lane per i-body gives the shipped N = 2000 eight workgroups on a 256-CU chip, and the tile kernels fill the chip only by
// cutting the j range into chunks whose partial rows a second launch has to add.  Round 2 had a kernel for the smallest
// systems in which a workgroup owned ONE register pair of bodies and its lanes split the j range: right at N = 2000, but
// every lane then loads 16 bytes per two pair evaluations and a few thousand bodies more are bound by the L1/L2 paths, not
// by the arithmetic.  Here a workgroup owns NP register pairs (2 NP bodies, the same values in every lane), its 256 lanes
// still split the j range with coalesced loads — one 16-byte load per 2 NP pair evaluations —, the wave sums are six DPP
// adds per value and one LDS hop joins the four waves.  Because a body's whole sum is finished inside its workgroup, the
// update rides along (new positions into a second buffer: other workgroups still read the first) and a step is ONE launch
// with no partial rows — where the tile kernels need a second launch just to add their j chunks.
//
// Exact d == 0 (OctreeSearch.h:102) without a detector: the guard (Z_CLAMP, pk_common.h) changes nothing for a normal
// r^2 > 0, so the kernel first runs the BARE pair law on every group of j-bodies that does not hold the workgroup's own
// bodies (the self pairs are the only d == 0 of a scene without coincident bodies).  Where the bet was wrong — r^2 zero
// or subnormal for two different bodies — v_rsq_f32 returns +inf, |d|^-3 d is inf or NaN, and a sum that has met one
// stays non-finite: the workgroup sees it in its finished sums and walks the j range again with the guard everywhere.
// Either way the stored bits are the guarded kernel's.  (A sum that overflows on its own merits is redone too, to the
// same inf.)  With equal masses (known to the host: nobody else writes the buffer while this kernel steps it) the mass
// factor leaves the loop as well.
#include "kernels.h"

#include <type_traits>

#include "../../include/nbody.h"
#include "pk_common.h"
#include "sym_common.h"

namespace nbody {

namespace {

// v + (v as seen through one DPP control), in the enabled rows; the other rows add 0
template <int CTRL, int ROW_MASK> __device__ __forceinline__ float dpp_add(float v) {
  const int i = __builtin_bit_cast(int, v);
  const int m = __builtin_amdgcn_update_dpp(ROW_MASK == 0xf ? i : 0, i, CTRL, ROW_MASK, 0xf, false);
  return v + __builtin_bit_cast(float, m);
}

// Sum of a value over the 64 lanes, left in lane 63, always in this order: neighbours, pairs of neighbours, the two
// quads of a half row, the two halves of a row (quad_perm, quad_perm, row_half_mirror, row_mirror: every lane of a row
// then holds the row's sum), row 0 into row 1 and row 2 into row 3 (row_bcast:15), rows 0+1 into rows 2, 3 (row_bcast:31).
// The first step goes through the builtin (hipcc turns it into a copy, a v_mov_b32_dpp and an add, and pads the hazard
// between whatever wrote the value and the DPP read); the other five are ONE v_add_f32_dpp each, written out.
// A DPP read needs two wait states after the VALU write of its operand, and LLVM's hazard recogniser does not look inside
// inline asm, so the distance is built in rather than left to the scheduler: the asms are `volatile` (they keep their source
// order among themselves — a value's steps are then NV >= 8 instructions apart), every first-step result is consumed by an
// empty volatile asm before ONE `s_nop 1`, and only then do the written-out steps begin.  tools/isa_hazards.py checks the
// built object's disassembly for exactly this (tests/test_isa_hazards.py).
// (Rows a row_mask leaves out keep their value; nobody reads them afterwards.)
#ifdef NBODY_BLOCK_DPP_UNORDERED   // round 3's form, for the A/B timing only (tools/ab_dpp_order.sh): order left to the scheduler
#define NBODY_DPP_ADD(v, ctrl) asm("v_add_f32_dpp %0, %0, %0 " ctrl : "+v"(v))
#define NBODY_DPP_SETTLE(v, nv) do { } while (0)
#else
#define NBODY_DPP_ADD(v, ctrl) asm volatile("v_add_f32_dpp %0, %0, %0 " ctrl : "+v"(v))
#define NBODY_DPP_SETTLE(v, nv) do { _Pragma("unroll") for (int q_ = 0; q_ < (nv); ++q_) asm volatile("" : "+v"((v)[q_])); asm volatile("s_nop 1"); } while (0)
#endif
template <int NV> __device__ __forceinline__ void wave_sum_to_lane63(float (&v)[NV]) {
  static_assert(NV >= 8, "the written-out steps rely on NV instructions between a value's steps");
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = dpp_add<0xB1, 0xf>(v[q]);    // quad_perm:[1,0,3,2]
  NBODY_DPP_SETTLE(v, NV);   // all first-step results exist (an empty volatile asm has consumed each) and are two wait states old
#pragma unroll
  for (int q = 0; q < NV; ++q) NBODY_DPP_ADD(v[q], "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf");
#pragma unroll
  for (int q = 0; q < NV; ++q) NBODY_DPP_ADD(v[q], "row_half_mirror row_mask:0xf bank_mask:0xf");
#pragma unroll
  for (int q = 0; q < NV; ++q) NBODY_DPP_ADD(v[q], "row_mirror row_mask:0xf bank_mask:0xf");
#pragma unroll
  for (int q = 0; q < NV; ++q) NBODY_DPP_ADD(v[q], "row_bcast:15 row_mask:0xa bank_mask:0xf");
#pragma unroll
  for (int q = 0; q < NV; ++q) NBODY_DPP_ADD(v[q], "row_bcast:31 row_mask:0xc bank_mask:0xf");
}
#undef NBODY_DPP_ADD
#undef NBODY_DPP_SETTLE

#ifndef NBODY_BLOCK_JL_BIG
#define NBODY_BLOCK_JL_BIG 4      // loads in flight per lane at five and more register pairs; 6 and 8 measured no better
                                  // (profiles/r03_ab_block_kernel_loads_in_flight.txt: N = 16384 68.3 / 70.4 / 80.0 us)
#endif
// j-bodies per staged group of the pair law (NP * JB independent chains) and loads in flight per lane (a multiple of JB)
constexpr int block_jb(int np) { return np >= 5 ? 1 : (np >= 3 ? 2 : 4); }
constexpr int block_jl(int np) { return np >= 5 ? NBODY_BLOCK_JL_BIG : 8; }

// grid.x = ceil(i_count / (2 NP)) workgroups of 256 lanes.
//   posm      all n_total bodies (x, y, z, m), read only
//   posm_out  integrate != 0: the bodies' new (x, y, z, m) go here (a second buffer — other workgroups still read posm)
//   acc_out   [i_count] accelerations
//   gate      optional device word (see below)
//   stage     integrate != 0, optional: the frame's FParticle records (OctreeSearch.h:8-18: Mass, Position, Velocity,
//             Acceleration, 10 floats per body) as they stand after the update — what the actor's Tick mirrors
//   size_bits integrate != 0, optional: pre-zeroed word that takes the bit pattern of max_i max(|x|, |y|, |z|) over the
//             positions BEFORE the update (ComputeCubeSize, OctreeSearch.cpp:47-56, runs first in the Tick); size_zero: a
//             second word, cleared for the frame after
//   optimistic (Z_CLAMP only)  first the bare pair law outside the own group, the guarded walk only if a sum came out non-finite
template <int NP, int ZMODE, bool UNI>
__global__ __launch_bounds__(kBlock) void forces_block_pk_kernel(const float4 *__restrict__ posm, float4 *__restrict__ posm_out,
                                                                 float4 *__restrict__ vel, float4 *__restrict__ acc_out,
                                                                 int n_total, int i_begin, int i_count, float gscale, float zp,
                                                                 float dt, int integrate, int optimistic,
                                                                 const int *__restrict__ gate, int gate_want,
                                                                 float *__restrict__ stage, unsigned int *__restrict__ size_bits,
                                                                 unsigned int *__restrict__ size_zero) {
  constexpr int B = 2 * NP;                  // bodies of a workgroup
  constexpr int JB = block_jb(NP), JL = block_jl(NP);
  __shared__ float red[kBlock / 64][6 * NP];
  __shared__ int redo;
  __shared__ unsigned int s_size;
  __shared__ __attribute__((aligned(16))) float s_rec[10 * B];   // the workgroup's FParticle records, contiguous (stage)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ia = blockIdx.x * B;
  if (ia >= i_count) return;                 // uniform per workgroup
  // twin launches (the host cannot vouch for the masses): *gate != 0 says "masses differ"; only the form it names runs
  if (gate != nullptr && ((*gate != 0) ? 1 : 0) != gate_want) return;
  if (t == 0) { redo = 0; s_size = 0u; }

  f2 xi[NP], yi[NP], zi[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const float4 p0 = posm[i_begin + min(ia + 2 * p, i_count - 1)];
    const float4 p1 = posm[i_begin + min(ia + 2 * p + 1, i_count - 1)];
    xi[p] = f2{p0.x, p1.x}; yi[p] = f2{p0.y, p1.y}; zi[p] = f2{p0.z, p1.z};
  }
  // thread t finishes body t of the workgroup: what its update needs is on its way from here on
  const int il = ia + t;
  const bool finisher = t < B && il < i_count;
  float4 vv = make_float4(0.f, 0.f, 0.f, 0.f), x = vv;
  if (finisher && integrate) { vv = vel[il]; x = posm[i_begin + il]; }
  // workgroup-uniform values and loop constants: in VGPRs (an SGPR operand halves the issue rate)
#pragma unroll
  for (int p = 0; p < NP; ++p) asm volatile("" : "+v"(xi[p]), "+v"(yi[p]), "+v"(zi[p]));
  f2 zp2 = splat2(zp), one2 = splat2(1.0f);
  asm volatile("" : "+v"(zp2), "+v"(one2));

  // lane t meets bodies t, t + 256, ... ("trips"), JL trips to a group; the workgroup's own bodies sit in these trips
  const int own_lo = (i_begin + ia) / kBlock, own_hi = (i_begin + min(ia + B, i_count) - 1) / kBlock;
  const int full_groups = n_total / (kBlock * JL);          // groups that need no bounds check
  const bool has_tail = full_groups * (kBlock * JL) < n_total;
  // The ragged end (at most one group) is loaded here, once, and kept: zero-mass bodies far outside any scene stand in for
  // what is missing (sym_common.h, kPadFar: |d|^2 = inf, rsq = 0, the term is exactly 0 with or without a guard and with or
  // without a mass factor).  Clamped index + per-component selects: a select of pointers would park the pad in scratch.
  float4 tail[JL];
#pragma unroll
  for (int l = 0; l < JL; ++l) {
    const int j = (full_groups * JL + l) * kBlock + t;
    const float4 q = posm[min(j, n_total - 1)];
    const bool in = j < n_total;
    tail[l] = make_float4(in ? q.x : kPadFar, in ? q.y : kPadFar, in ? q.z : kPadFar, in ? q.w : 0.f);
  }

  float ax, ay, az;
  // one walk over all j and the workgroup's sums -> (ax, ay, az) of body t (threads t < B); `bare` is a compile-time constant
  auto pass = [&](auto bare_c) {
    constexpr bool bare = decltype(bare_c)::value;
    Acc3pk<false> a[NP];
    float4 cur[JL];
    // groups [ga, gb) of the full ones, one after the other: JL / JB staged sub-groups each; a sub-group's registers are
    // reloaded with the NEXT group's bodies as soon as the pair law has read them (the last group loads itself again: no
    // branch), so JL loads are in flight under a whole group's arithmetic without a second set of registers
    auto run = [&](auto zm, int ga, int gb) {
#pragma unroll 1
      for (int g = ga; g < gb; ++g) {
        const float4 *__restrict__ nx = posm + (size_t)(min(g + 1, full_groups - 1) * JL) * kBlock + t;
#pragma unroll
        for (int b = 0; b < JL; b += JB) {
          float4 pj[JB];
#pragma unroll
          for (int k = 0; k < JB; ++k) pj[k] = cur[b + k];
          pair_group_pk<NP, JB, decltype(zm)::value, false, UNI>(xi, yi, zi, pj, zp2, one2, a);
#pragma unroll
          for (int k = 0; k < JB; ++k) cur[b + k] = nx[(b + k) * kBlock];
          __builtin_amdgcn_sched_barrier(0);   // one staged sub-group at a time: interleaving them only costs registers
        }
      }
    };

    if (full_groups > 0) {
#pragma unroll
      for (int l = 0; l < JL; ++l) cur[l] = posm[l * kBlock + t];
      if (ZMODE == Z_CLAMP && bare) {          // only the groups that hold the own bodies are guarded
        const int ga = min(own_lo / JL, full_groups), gb = min(own_hi / JL + 1, full_groups);
        run(std::integral_constant<int, Z_BARE>{}, 0, ga);
        run(std::integral_constant<int, ZMODE>{}, ga, gb);
        run(std::integral_constant<int, Z_BARE>{}, gb, full_groups);
      } else {
        run(std::integral_constant<int, ZMODE>{}, 0, full_groups);
      }
    }
    if (has_tail) {                            // the ragged group: always with the guard (it is one group)
#pragma unroll
      for (int b = 0; b < JL; b += JB) {
        float4 pj[JB];
#pragma unroll
        for (int k = 0; k < JB; ++k) pj[k] = tail[b + k];
        pair_group_pk<NP, JB, ZMODE, false, UNI>(xi, yi, zi, pj, zp2, one2, a);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    float v[6 * NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      v[6 * p + 0] = a[p].x.x; v[6 * p + 1] = a[p].y.x; v[6 * p + 2] = a[p].z.x;
      v[6 * p + 3] = a[p].x.y; v[6 * p + 4] = a[p].y.y; v[6 * p + 5] = a[p].z.y;
    }
    wave_sum_to_lane63(v);
    if (lane == 63) {
#pragma unroll
      for (int q = 0; q < 6 * NP; ++q) red[wave][q] = v[q];
    }
    __syncthreads();
    const int tb = t < B ? t : 0;
    ax = ((red[0][3 * tb] + red[1][3 * tb]) + red[2][3 * tb]) + red[3][3 * tb];
    ay = ((red[0][3 * tb + 1] + red[1][3 * tb + 1]) + red[2][3 * tb + 1]) + red[3][3 * tb + 1];
    az = ((red[0][3 * tb + 2] + red[1][3 * tb + 2]) + red[2][3 * tb + 2]) + red[3][3 * tb + 2];
  };

  if (ZMODE == Z_CLAMP && optimistic != 0) {
    pass(std::true_type{});
    // the bet: did any body of the workgroup meet a d == 0 outside its own group?  (uniform: every thread reads `redo`)
    if (finisher && !(fabsf(ax) <= 3.0e38f && fabsf(ay) <= 3.0e38f && fabsf(az) <= 3.0e38f)) redo = 1;
    __syncthreads();
    if (redo != 0) pass(std::false_type{});    // everybody has read red[] and redo: walk again, guarded everywhere
  } else {
    pass(std::false_type{});
  }

  if (wave != 0) return;                       // what is left is the first wave's (the finishing threads are its lanes 0 .. B-1)
  // the loop summed m_j |d|^-3 d (equal masses: |d|^-3 d): G (and the common m) come in here
  const float gm = UNI ? posm[0].w * gscale : gscale;
  ax *= gm; ay *= gm; az *= gm;
  if (finisher) acc_out[il] = make_float4(ax, ay, az, 0.f);
  if (!integrate) return;
  const float4 x0 = x;
  vv.x = mul_add_sep2(dt, ax, vv.x); vv.y = mul_add_sep2(dt, ay, vv.y); vv.z = mul_add_sep2(dt, az, vv.z);
  x.x = mul_add_sep2(dt, vv.x, x.x); x.y = mul_add_sep2(dt, vv.y, x.y); x.z = mul_add_sep2(dt, vv.z, x.z);
  if (finisher) {
    vel[il] = vv;
    posm_out[i_begin + il] = x;
  }
  if (stage != nullptr) {
    // the records leave as one contiguous piece per workgroup (40 B x bodies, 16-byte stores): `stage` may be host memory —
    // the caller's pinned mirror — where ten scattered 4-byte stores per body would each be a transaction of their own
    if (finisher) {
      float *o = s_rec + 10 * t;
      o[0] = x.w; o[1] = x.x; o[2] = x.y; o[3] = x.z; o[4] = vv.x; o[5] = vv.y; o[6] = vv.z; o[7] = ax; o[8] = ay; o[9] = az;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    const int nrec = min(B, i_count - ia), nfl = 10 * nrec;                 // floats of this workgroup's records
    float *dst = stage + (size_t)ia * 10;                                   // ia is even: 16-byte aligned
    if (4 * lane + 3 < nfl) *(float4 *)(dst + 4 * lane) = *(const float4 *)(s_rec + 4 * lane);
    else if (4 * lane < nfl) { for (int q = 4 * lane; q < nfl; ++q) dst[q] = s_rec[q]; }
  }
  if (!finisher) return;
  if (size_bits != nullptr) {
    // the finishing threads are lanes of one wave: their LDS maximum is complete when lane 0 reads it back
    atomicMax(&s_size, __float_as_uint(fmaxf(fmaxf(fabsf(x0.x), fabsf(x0.y)), fabsf(x0.z))));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    if (t == 0) {
      atomicMax(size_bits, s_size);            // non-negative floats order as their bit patterns; one atomic per workgroup
      if (blockIdx.x == 0 && size_zero != nullptr) *size_zero = 0u;
    }
  }
}

template <int NP, bool UNI>
hipError_t launch_block_np(const BlockLaunch &L, hipStream_t s) {
  const int B = 2 * NP;
  dim3 grid((L.i_count + B - 1) / B), block(kBlock);
  const int integrate = L.dt > 0.f ? 1 : 0;
  const int *gate = L.uni < 0 ? (const int *)L.general : nullptr;
  if (L.eps2 > 0.0)
    hipLaunchKernelGGL((forces_block_pk_kernel<NP, Z_SOFT, UNI>), grid, block, 0, s, (const float4 *)L.posm, (float4 *)L.posm_out,
                       (float4 *)L.vel, (float4 *)L.acc, L.n_total, L.i_begin, L.i_count, (float)L.G, (float)L.eps2, L.dt, integrate, 0,
                       gate, UNI ? 0 : 1, (float *)L.stage, (unsigned int *)L.size_bits, (unsigned int *)L.size_zero);
  else
    hipLaunchKernelGGL((forces_block_pk_kernel<NP, Z_CLAMP, UNI>), grid, block, 0, s, (const float4 *)L.posm, (float4 *)L.posm_out,
                       (float4 *)L.vel, (float4 *)L.acc, L.n_total, L.i_begin, L.i_count, (float)L.G, -0x1p126f, L.dt, integrate,
                       L.optimistic, gate, UNI ? 0 : 1, (float *)L.stage, (unsigned int *)L.size_bits, (unsigned int *)L.size_zero);
  return hipGetLastError();
}

template <bool UNI>
hipError_t launch_block_uni(const BlockLaunch &L, hipStream_t s) {
  switch (L.np) {
    case 2: return launch_block_np<2, UNI>(L, s);
    case 3: return launch_block_np<3, UNI>(L, s);
    case 4: return launch_block_np<4, UNI>(L, s);
    case 5: return launch_block_np<5, UNI>(L, s);
    case 6: return launch_block_np<6, UNI>(L, s);
    case 7: return launch_block_np<7, UNI>(L, s);
    case 8: return launch_block_np<8, UNI>(L, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

hipError_t launch_block(const BlockLaunch &L, hipStream_t s) {
  if (L.i_count <= 0 || L.n_total <= 0 || L.i_begin < 0 || L.i_begin + L.i_count > L.n_total) return hipErrorInvalidValue;
  if (L.dt > 0.f && (L.posm_out == nullptr || L.vel == nullptr || L.posm_out == L.posm)) return hipErrorInvalidValue;
  if (!(L.dt > 0.f) && (L.stage != nullptr || L.size_bits != nullptr)) return hipErrorInvalidValue;
  if (L.uni < 0) {                             // both forms; the device word decides which one does the work
    if (L.general == nullptr) return hipErrorInvalidValue;
    const hipError_t e = launch_block_uni<true>(L, s);
    return e != hipSuccess ? e : launch_block_uni<false>(L, s);
  }
  return L.uni ? launch_block_uni<true>(L, s) : launch_block_uni<false>(L, s);
}

}  // namespace nbody

// Octree::ComputeForces (OctreeSearch.h:99-108) on the compact tree, and the Tick's update behind every walk (OctreeSearch.cpp:28-31)
// — see bh_common.h.
#include "bh_common.h"

namespace nbody {
namespace bh {

template <typename T> __device__ __forceinline__ T mul_add_sep(T a, T b, T c) {
#pragma clang fp contract(off)
  const T p = a * b;
  return c + p;
}

// One accepted node's term of Octree::ComputeForces (.h:104): float(G * M / pow(d, 3)) * (CoM - Pos), d = Dist.
// The walks are bound by the instructions of this term (N = 2^20: 16 000 per wave), so:
//  * (CoM - Pos) is taken as -(Pos - CoM), the difference the squared distance was made of: a - b and -(b - a) agree in every bit
//    except that equal operands give +0 and -0 — and such a term goes into a sum that started at +0 and therefore never is -0, so
//    adding either zero leaves every bit of it alone;
//  * the correctly rounded square root is v_sqrt_f32 (one ulp) put right by the two fused residuals the compiler's own sqrtf uses,
//    without its scaling for arguments below 2^-96 and its special cases, and the double-precision division likewise without its
//    scaling and special cases: a wave with an argument below 2^-96, an infinite / NaN one or a mass that is not finite in any of
//    its lanes takes sqrtf and the division themselves.
__device__ __forceinline__ void force_term(float cx, float cy, float cz, float M, const float4 &p, double G, float &tx, float &ty,
                                           float &tz) {
#pragma clang fp contract(off)
  const float ex = p.x - cx, ey = p.y - cy, ez = p.z - cz;
  float d2 = ex * ex + ey * ey;
  d2 = d2 + ez * ez;
  float d, s;                                                  // FVector::Dist, .h:101 (correctly rounded); the scale factor
  if (__any(!(d2 >= 0x1p-96f) || d2 == __builtin_inff() || !(fabsf(M) <= 0x1.fffffep127f))) {
    d = sqrtf(d2);
    const double dd = (double)d;
    s = (float)(G * (double)M / ((dd * dd) * dd));             // (d*d)*d in double = the correctly rounded cube
  } else {
    const float r = __builtin_amdgcn_sqrtf(d2);
    const float below = __uint_as_float(__float_as_uint(r) - 1u), above = __uint_as_float(__float_as_uint(r) + 1u);
    const float eb = __builtin_fmaf(-below, r, d2), ea = __builtin_fmaf(-above, r, d2);
    d = eb <= 0.0f ? below : r;
    d = ea > 0.0f ? above : d;
    // ... and the correctly rounded double quotient is the compiler's own sequence — reciprocal, two Newton steps, quotient, one
    // residual step — without the operand scaling and the special cases that cannot occur here: d in [2^-48, 2^64), so d^3 in
    // [2^-144, 2^192), G M finite: every value on the way is a normal double (a mass of +-0 gives +0 where the division gives the
    // mass's sign: a term of +-0 either way, which changes no sum).
    const double dd = (double)d, den = (dd * dd) * dd, num = G * (double)M;
    double rc = __builtin_amdgcn_rcp(den);
    rc = __builtin_fma(rc, __builtin_fma(-den, rc, 1.0), rc);
    rc = __builtin_fma(rc, __builtin_fma(-den, rc, 1.0), rc);
    const double q0 = num * rc;
    s = (float)__builtin_fma(__builtin_fma(-den, q0, num), rc, q0);
  }
  tx = s * -ex; ty = s * -ey; tz = s * -ez;
}

// own[]: a count per block of kB sorted positions, then every block adds up the counts before it and ranks its own bodies (two
// launches of a few microseconds; the list has exactly `count` entries whatever the order is)
__global__ __launch_bounds__(kB) void bh_own_count_kernel(const unsigned int *__restrict__ sidx, int n, unsigned int lo, unsigned int cnt,
                                                          const int *__restrict__ status, unsigned int *__restrict__ blk) {
  __shared__ unsigned int s_w[kB / 64];
  if (*status != 0) return;                                    // a frame refused or given up: there is no order
  const int i = blockIdx.x * kB + threadIdx.x;
  const bool mine = i < n && sidx[i] - lo < cnt;
  const unsigned long long bm = __ballot(mine);
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = (unsigned int)__popcll(bm);
  __syncthreads();
  if (threadIdx.x == 0) blk[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
__global__ __launch_bounds__(kB) void bh_own_list_kernel(const unsigned int *__restrict__ sidx, int n, unsigned int lo, unsigned int cnt,
                                                         const int *__restrict__ status, const unsigned int *__restrict__ blk,
                                                         unsigned int *__restrict__ own) {
  __shared__ unsigned int s_w[kB / 64], s_c[kB / 64];
  if (*status != 0) return;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  unsigned int before = 0;                                     // own bodies in the blocks before this one
  for (int q = t; q < (int)blockIdx.x; q += kB) before += blk[q];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
  const int i = blockIdx.x * kB + t;
  const bool mine = i < n && sidx[i] - lo < cnt;
  const unsigned long long bm = __ballot(mine);
  if (lane == 0) { s_w[wave] = before; s_c[wave] = (unsigned int)__popcll(bm); }
  __syncthreads();
  unsigned int base = s_w[0] + s_w[1] + s_w[2] + s_w[3];
  for (int w = 0; w < wave; ++w) base += s_c[w];
  if (mine) own[base + (unsigned int)__popcll(bm & ((1ull << lane) - 1ull))] = (unsigned int)i;
}

// value of lane (l - N) mod 16 of the same 16-lane row (v_mov_b32_dpp row_ror:N: the row rotates right)
template <int N> __device__ __forceinline__ int row_ror(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x120 + N, 0xf, 0xf, false); }
__device__ __forceinline__ int row_or(int v) { v |= row_ror<8>(v); v |= row_ror<4>(v); v |= row_ror<2>(v); v |= row_ror<1>(v); return v; }
__device__ __forceinline__ int row_max(int v) {
  v = max(v, row_ror<8>(v)); v = max(v, row_ror<4>(v)); v = max(v, row_ror<2>(v)); v = max(v, row_ror<1>(v));
  return v;
}

// Octree::ComputeForces (.h:99-108) on the compact tree, bodies in key order, one 16-lane row per body.  With dt > 0 the
// row's first lane goes on to the Tick's update of its body (.cpp:28-31) — nobody else reads that body's position:
// leaves carry copies.
// In LDS a node is (CoM, threshold) + a 16-bit "node after the subtree" + M.  The nodes stand in the walk's own order
// (preorder), so the walk is not followed node by node: the row looks at SIXTEEN consecutive nodes at once, lane j at node
// w + j.  Each lane decides for its node alone — taken (.h:103: d2 >= threshold; a leaf's is 0), d == 0 (.h:102), or
// descend — and a node is visited by the reference's recursion exactly when no ancestor of it was taken or had d == 0.
// The ancestors of a window's nodes that lie before the window are on the path to its first node, hence descended; those
// inside it announce the nodes they cover as a bit mask, and one OR over the row tells every lane whether its node is
// visited.  The visited taken nodes' terms (.h:104) are worked out by their lanes side by side and added by the first lane
// in lane order = the walk's order = the reference's order of additions; the next window starts behind whatever the
// window's taken nodes cover.
// The walk of one row (see the kernels below).  LDS_TREE: the nodes are the LDS arrays s_a / s_past / s_m; otherwise they
// are read from the tree's global arrays (coalesced: a window is sixteen consecutive nodes) and the threshold comes from the
// level.  list / term: the row's own LDS slices.  The row's first lane ends up with the acceleration.
template <bool LDS_TREE, typename LIST_T>
__device__ __forceinline__ void walk_windows(const SmallTree &T, const float4 *s_a, const float *s_m, const unsigned short *s_past,
                                             const float *s_thr, LIST_T *list, float4 *term, int nodes, bool valid,
                                             const float4 &p, double G, int g, int row_shift, float &ax, float &ay, float &az) {
#pragma clang fp contract(off)
  float sum = 0.f;                                             // lanes 0, 1, 2 of the row: the x, y, z sums (ZeroVector, .cpp:84)
  int w0 = valid ? 0 : nodes;                                  // first node of the window (the same in all lanes of the row)
  for (;;) {
    // ---- the walk: windows of sixteen nodes until the row's list cannot take another window's worth
    int cnt = 0;
    for (;;) {
      const bool open = w0 < nodes && cnt + kWalkG <= kWalkK;
      if (!__any(open)) break;
      const int my = w0 + g;
      const bool in = open && my < nodes;
      float4 a;
      int past;
      if (LDS_TREE) {
        a = s_a[in ? my : 0];
        past = s_past[in ? my : 0];
      } else {
        const float4 c = T.com[in ? my : 0];
        const unsigned int w = T.meta[in ? my : 0];
        const bool leaf = (w & kLeafBit) != 0u;
        a = make_float4(c.x, c.y, c.z, leaf ? 0.0f : s_thr[(w >> kLevelShift) & 63u]);
        past = leaf ? my + 1 : (int)(w & kLinkMask);
      }
      const float ex = p.x - a.x, ey = p.y - a.y, ez = p.z - a.z;
      float d2 = ex * ex + ey * ey;
      d2 = d2 + ez * ez;
      const bool take = in && d2 >= a.w;                       // .h:103: Size / d < Theta, or an occupied leaf
      const bool zero = in && d2 == 0.f;                       // .h:102: d == 0 adds nothing and ends the subtree
      const bool ends = take || zero;                          // the recursion does not go below this node
      // nodes of this window below mine: window offsets g + 1 .. past - w0 - 1
      const int rel = min(past - w0, kWalkG);
      const int cover = (ends && rel > g + 1) ? (((1 << rel) - 1) & ~((2 << g) - 1)) : 0;
      const int dead = row_or(cover);
      const bool adds = take && !zero && ((dead >> g) & 1) == 0;
      const unsigned long long am = __ballot(adds);
      const int row = (int)((am >> row_shift) & 0xFFFFull);    // this row's lanes whose node adds a term
      if (adds) list[cnt + __popc(row & ((1 << g) - 1))] = (LIST_T)my;   // lane order = the walk's order
      cnt += __popc(row);
      const int next = max(min(w0 + kWalkG, nodes), row_max(ends ? past : 0));
      w0 = open ? next : w0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (!__any(cnt > 0)) break;                                // every walk of this wave has ended, nothing left to add
    // ---- the listed nodes' terms (.h:104), sixteen at a time
    for (int e = g; e < cnt; e += kWalkG) {
      const int nd = (int)list[e];
      float tx, ty, tz;
      if (LDS_TREE) { const float4 a = s_a[nd]; force_term(a.x, a.y, a.z, s_m[nd], p, G, tx, ty, tz); }
      else { const float4 c = T.com[nd]; force_term(c.x, c.y, c.z, c.w, p, G, tx, ty, tz); }
      term[e] = make_float4(tx, ty, tz, 0.f);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // ---- added in the walk's order, one lane per component, eight loads in flight
    if (g < 3) {
      const float *col = (const float *)term + g;
      for (int e = 0; e < cnt; e += 8) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = col[4 * min(e + q, kWalkK - 1)];
#pragma unroll
        for (int q = 0; q < 8; ++q) sum = (e + q < cnt) ? sum + v[q] : sum;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  ay = __builtin_bit_cast(float, row_ror<15>(__builtin_bit_cast(int, sum)));    // lane 0 takes lane 1's and lane 2's sums
  az = __builtin_bit_cast(float, row_ror<14>(__builtin_bit_cast(int, sum)));
  ax = sum;
}

// What follows a body's walk: the acceleration, and with dt > 0 the Tick's update of the body (.cpp:28-31: v += dt*a; x += dt*v,
// multiply and add kept apart) — nobody else reads that body's position: the tree's leaves carry copies.  A walk kernel returns at
// once on a refused frame (hdr[3]), so the state of a refused frame — and of everything queued behind it — stays what it was.
// stage (optional): the frame's FParticle record (.h:8-18), for the renderer hand-off; it may be page-locked HOST memory
// (nbody_tick hands the caller's pinned mirror over).
// Sixteen lanes per body: the row's first lane lays the ten floats out in the row's LDS slice `rec` and ten lanes store them with
// ONE instruction — 40 contiguous bytes per body — instead of ten scattered 4-byte stores.
// own: the context's slice of the bodies (WalkSlice below) — vel, acc and stage hold the slice's bodies only, posm all of them.
__device__ __forceinline__ void walk_row_tail(bool valid, int g, unsigned int body, const float4 &p, float ax, float ay, float az,
                                              float4 *__restrict__ posm, float4 *__restrict__ vel, float4 *__restrict__ acc, float dt,
                                              float *__restrict__ stage, float *rec, int off, unsigned int *__restrict__ next_size = nullptr,
                                              float4 *__restrict__ pos_sorted = nullptr, int k = 0) {
  const unsigned int lb = body - (unsigned int)off;             // the body's place in the slice's arrays
  if (next_size != nullptr) {                                  // (larger systems, dt > 0: every lane of the wave comes by here)
    float nx = 0.f, ny = 0.f, nz = 0.f;                         // where the body is about to go (the same arithmetic as below)
    if (valid && g == 0) {
      const float4 u = vel[lb];
      nx = mul_add_sep(dt, mul_add_sep(dt, ax, u.x), p.x); ny = mul_add_sep(dt, mul_add_sep(dt, ay, u.y), p.y);
      nz = mul_add_sep(dt, mul_add_sep(dt, az, u.z), p.z);
    }
    note_next_size(next_size, valid && g == 0, nx, ny, nz);
  }
  if (!valid || (g != 0 && stage == nullptr)) return;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f), x = p;
  if (g == 0) {
    acc[lb] = make_float4(ax, ay, az, 0.f);
    if ((dt > 0.f || stage != nullptr) && vel != nullptr) v = vel[lb];   // (not `vel ? vel[lb] : v`: a select of addresses parks v in scratch)
    if (dt > 0.f) {                                            // v += dt*a; x += dt*v, separate multiply and add
      v.x = mul_add_sep(dt, ax, v.x); v.y = mul_add_sep(dt, ay, v.y); v.z = mul_add_sep(dt, az, v.z);
      x.x = mul_add_sep(dt, v.x, x.x); x.y = mul_add_sep(dt, v.y, x.y); x.z = mul_add_sep(dt, v.z, x.z);
      vel[lb] = v;
      posm[body] = x;
    }
    if (pos_sorted != nullptr) pos_sorted[k] = x;              // the positions in key order, for the next frame's key kernel (larger systems)
  }
  if (stage != nullptr) {
    if (g == 0) { rec[0] = x.w; rec[1] = x.x; rec[2] = x.y; rec[3] = x.z; rec[4] = v.x; rec[5] = v.y; rec[6] = v.z; rec[7] = ax; rec[8] = ay; rec[9] = az; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (g < 10) stage[(size_t)lb * 10 + g] = rec[g];
  }
}
// ... and the same for a walk with one lane per body
__device__ __forceinline__ void walk_lane_tail(bool valid, unsigned int body, const float4 &p, float ax, float ay, float az,
                                               float4 *__restrict__ posm, float4 *__restrict__ vel, float4 *__restrict__ acc, float dt,
                                               float *__restrict__ stage, int off, unsigned int *__restrict__ next_size,
                                               float4 *__restrict__ pos_sorted, int k) {
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f), x = p;
  const unsigned int lb = body - (unsigned int)off;
  if (valid) {
    acc[lb] = make_float4(ax, ay, az, 0.f);
    if ((dt > 0.f || stage != nullptr) && vel != nullptr) v = vel[lb];
    if (dt > 0.f) {
      v.x = mul_add_sep(dt, ax, v.x); v.y = mul_add_sep(dt, ay, v.y); v.z = mul_add_sep(dt, az, v.z);
      x.x = mul_add_sep(dt, v.x, x.x); x.y = mul_add_sep(dt, v.y, x.y); x.z = mul_add_sep(dt, v.z, x.z);
      vel[lb] = v;
      posm[body] = x;
    }
    if (pos_sorted != nullptr) pos_sorted[k] = x;              // the positions in key order, for the next frame's key kernel: one coalesced store (uniform: not on a slice)
  }
  note_next_size(next_size, valid, x.x, x.y, x.z);             // (every lane of the wave comes by here)
  if (!valid) return;
  if (stage != nullptr) {
    float *o = stage + (size_t)lb * 10;
    o[0] = x.w; o[1] = x.x; o[2] = x.y; o[3] = x.z; o[4] = v.x; o[5] = v.y; o[6] = v.z; o[7] = ax; o[8] = ay; o[9] = az;
  }
}

__global__ __launch_bounds__(kWalkT) void bh_walk_compact_kernel(SmallTree T, float4 *__restrict__ posm, float4 *__restrict__ vel,
                                                                 float4 *__restrict__ acc, int n, float theta, double G, float dt,
                                                                 float *__restrict__ stage, WalkSlice S) {
  static_assert(kWalkG == 16, "one DPP row per body");
  constexpr int kGroups = kWalkT / kWalkG;
  __shared__ float4 s_a[kSmNodesLds];
  __shared__ float s_m[kSmNodesLds];
  __shared__ unsigned short s_past[kSmNodesLds];
  __shared__ float s_thr[kMaxLevels + 2];
  __shared__ unsigned int s_list[kGroups][kWalkK];
  __shared__ float4 s_term[kGroups][kWalkK];
  (void)theta;
  hand_verdict(T);
  if (T.hdr[3] != 0) return;                                   // the frame was refused: nothing moves
  BH_WALK_CLOCK(9);
  BH_WG_STAMP(0);
  const int t = threadIdx.x;
  const int nodes = T.hdr[0];
  const bool in_lds = nodes <= kSmNodesLds;
  if (t <= kMaxLevels) s_thr[t] = T.thr[t];
  __syncthreads();
  if (in_lds) {
#pragma unroll 4
    for (int m = t; m < nodes; m += kWalkT) {
      const float4 c = T.com[m];
      const unsigned int w = T.meta[m];
      const bool leaf = (w & kLeafBit) != 0u;
      s_a[m] = make_float4(c.x, c.y, c.z, leaf ? 0.0f : s_thr[(w >> kLevelShift) & 63u]);
      s_m[m] = c.w;
      s_past[m] = (unsigned short)(leaf ? m + 1 : (int)(w & kLinkMask));
    }
    __syncthreads();
  }
  const int group = t / kWalkG, g = t % kWalkG;
  const int k = blockIdx.x * kGroups + group;
  const bool valid = k < n;                                    // (n: the bodies this context walks — all, or its slice's)
  const int place = valid ? walk_place(S, k) : 0;               // the body's sorted position
  const unsigned int body = valid ? T.sidx[place] : 0u;
  const float4 p = posm[body];
  float ax = 0.f, ay = 0.f, az = 0.f;                          // Acceleration = ZeroVector, .cpp:84
  BH_WALK_CLOCK(10);
  BH_WG_STAMP(1);
  if (in_lds)
    walk_windows<true>(T, s_a, s_m, s_past, s_thr, s_list[group], s_term[group], nodes, valid, p, G, g, (t & 63) - g, ax, ay, az);
  else   // a tree too large for LDS (deep chains of single-child cells): the same windows on the global arrays
    walk_windows<false>(T, s_a, s_m, s_past, s_thr, s_list[group], s_term[group], nodes, valid, p, G, g, (t & 63) - g, ax, ay, az);
  BH_WALK_CLOCK(11);
  BH_WG_STAMP(2);
#ifdef NBODY_BH_PHASE_CLOCKS
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) {       // the shader clock under this load: s_sleep 127 = 127 * 64 cycles
    const long long c0 = wall_clock64();
    for (int q = 0; q < 16; ++q) __builtin_amdgcn_s_sleep(127);
    T.clocks[15] = wall_clock64() - c0;
  }
#endif
  walk_row_tail(valid, g, body, p, ax, ay, az, posm, vel, acc, dt, stage, (float *)s_term[group], S.off);
}

// ---------------------------------------------------------------------------------------------------------------------
// The same walk with a whole WAVE per body (round 4): the window is sixty-four consecutive nodes.  A body's walk is a chain of
// dependent round trips — to LDS on the small systems' tree, to L2 on the larger ones' — one per window; sixteen-node windows made
// it ~27 links long for a typical body of the shipped scene, sixty-four-node windows make it ~10, every branch is wave-uniform
// (w0 and the list count are the same in all lanes), and the listed terms are worked out sixty-four at a time.
// Which nodes of a window the reference's recursion visits: node j is skipped iff some earlier node i of the window ended the
// recursion (taken, or d == 0) and covers it, i.e. past_i > j — subtrees nest, so that is "the largest past among the ended
// nodes before j exceeds j": ONE exclusive max-scan over the wave (six DPP steps) instead of an OR of cover masks per row.
template <int CTRL, int RM> __device__ __forceinline__ int dpp_max0(int v) {     // max(v, v as seen through the control; 0 where nothing arrives)
  return max(v, __builtin_amdgcn_update_dpp(0, v, CTRL, RM, 0xf, false));
}
__device__ __forceinline__ int wave_incl_max(int v) {             // inclusive maximum over lanes 0 .. own (values >= 0)
  v = dpp_max0<0x111, 0xf>(v); v = dpp_max0<0x112, 0xf>(v); v = dpp_max0<0x114, 0xf>(v); v = dpp_max0<0x118, 0xf>(v);   // row_shr:1,2,4,8
  v = dpp_max0<0x142, 0xa>(v);                                   // row_bcast:15 into rows 1, 3
  v = dpp_max0<0x143, 0xc>(v);                                   // row_bcast:31 into rows 2, 3
  return v;
}

// K: nodes a body lists before their terms (.h:104) are worked out and added — in the walk's order, the reference's own order of
// additions — by lanes 0, 1, 2 (x, y, z).  list / term: the wave's own LDS slices (term: 3 K floats, 16-byte aligned; K a multiple of 8).
template <bool LDS_TREE, int K, typename LIST_T>
__device__ __forceinline__ void walk_wave(const SmallTree &T, const float4 *s_a, const float *s_m, const unsigned short *s_past,
                                          const float *s_thr, LIST_T *list, float *term, int nodes, bool valid, const float4 &p,
                                          double G, int lane, float &ax, float &ay, float &az) {
#pragma clang fp contract(off)
  float sum = 0.f;                                             // lanes 0, 1, 2: the x, y, z sums (ZeroVector, .cpp:84)
  int w0 = valid ? 0 : nodes;                                  // first node of the window: the same in every lane
  const unsigned long long below = (1ull << lane) - 1ull;
  for (;;) {
    int cnt = 0;
    while (w0 < nodes && cnt + 64 <= K) {
      const int my = w0 + lane;
      const bool in = my < nodes;
      float4 a;
      int past;
      if (LDS_TREE) {
        a = s_a[in ? my : 0];
        past = s_past[in ? my : 0];
      } else {
        const float4 c = T.com[in ? my : 0];
        const unsigned int w = T.meta[in ? my : 0];
        const bool leaf = (w & kLeafBit) != 0u;
        a = make_float4(c.x, c.y, c.z, leaf ? 0.0f : s_thr[(w >> kLevelShift) & 63u]);
        past = leaf ? my + 1 : (int)(w & kLinkMask);
      }
      const float ex = p.x - a.x, ey = p.y - a.y, ez = p.z - a.z;
      float d2 = ex * ex + ey * ey;
      d2 = d2 + ez * ez;
      const bool take = in && d2 >= a.w;                       // .h:103: Size / d < Theta, or an occupied leaf
      const bool zero = in && d2 == 0.f;                       // .h:102: d == 0 adds nothing and ends the subtree
      const int reach = (take || zero) ? past : 0;             // the recursion does not go below this node: nothing before `past`
      const int incl = wave_incl_max(reach);
      const int excl = __builtin_amdgcn_update_dpp(0, incl, 0x138, 0xf, 0xf, false);   // wave_shr:1 — the ended nodes BEFORE mine
      const bool adds = take && !zero && excl <= my;           // visited (no earlier ended node covers it), taken, d != 0
      const unsigned long long am = __ballot(adds);
      if (adds) list[cnt + __popcll(am & below)] = (LIST_T)my; // lane order = the walk's order
      cnt += (int)__popcll(am);
      w0 = max(min(w0 + 64, nodes), __builtin_amdgcn_readlane(incl, 63));   // behind whatever the window's ended nodes cover
    }
    if (cnt == 0) break;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // the listed nodes' terms (.h:104), sixty-four at a time, into three columns (x, y, z); the list's last eight-term group is
    // filled up with +0: a sum that started at +0 never is -0, so adding +0 leaves every bit of it alone — and the adding loop
    // below needs no bounds
    const int cnt8 = (cnt + 7) & ~7;
    for (int e = lane; e < cnt8; e += 64) {
      float tx = 0.f, ty = 0.f, tz = 0.f;
      if (e < cnt) {
        const int nd = (int)list[e];
        if (LDS_TREE) { const float4 a = s_a[nd]; force_term(a.x, a.y, a.z, s_m[nd], p, G, tx, ty, tz); }
        else { const float4 c = T.com[nd]; force_term(c.x, c.y, c.z, c.w, p, G, tx, ty, tz); }
      }
      term[e] = tx; term[K + e] = ty; term[2 * K + e] = tz;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane < 3) {                                            // added in the walk's order, one lane per component, eight terms a go
      const float4 *col = (const float4 *)(term + lane * K);
      for (int e = 0; e < cnt8; e += 8) {
        const float4 u = col[e >> 2], v = col[(e >> 2) + 1];
        sum = sum + u.x; sum = sum + u.y; sum = sum + u.z; sum = sum + u.w;
        sum = sum + v.x; sum = sum + v.y; sum = sum + v.z; sum = sum + v.w;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
  ax = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sum), 0));
  ay = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sum), 1));
  az = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, sum), 2));
}

__global__ __launch_bounds__(kWvT) void bh_walk_wave_compact_kernel(SmallTree T, float4 *__restrict__ posm, float4 *__restrict__ vel,
                                                                    float4 *__restrict__ acc, int n, double G, float dt,
                                                                    float *__restrict__ stage, WalkSlice S) {
  constexpr int kWaves = kWvT / 64;
  __shared__ float4 s_a[kSmNodesLds];
  __shared__ float s_m[kSmNodesLds];
  __shared__ unsigned short s_past[kSmNodesLds];
  __shared__ float s_thr[kMaxLevels + 2];
  __shared__ unsigned short s_list[kWaves][kWvK];
  __shared__ __attribute__((aligned(16))) float s_term[kWaves][3 * kWvK];
  // A fresh kernel's first look at anything is a trip to memory the build's workgroup wrote (~1 us), and this kernel is a handful of
  // such trips and a walk in LDS: the verdict, the node count, the thresholds, the body's number AND the first 2048 nodes go out
  // together (the arrays are there whatever the verdict, and hold T.cap nodes whatever the count; nothing is written before the verdict).
  const int t = threadIdx.x;
  const int wave = t >> 6, lane = t & 63;
  const int k = blockIdx.x * kWaves + wave;
  const bool valid = k < n;                                    // (n: the bodies this context walks — all, or its slice's)
  const int place = valid ? walk_place(S, k) : 0;               // the body's sorted position
  const unsigned int body = valid ? T.sidx[place] : 0u;
  const int status = T.hdr[3], nodes = T.hdr[0];
  const float thr_pre = T.thr[min(t, kMaxLevels)];
  float4 c0[4];
  unsigned int w0[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { const int m = min(t + u * kWvT, T.cap - 1); c0[u] = T.com[m]; w0[u] = T.meta[m]; }
  hand_verdict(T);
  if (status != 0) return;                                     // the frame was refused: nothing moves
  BH_WG_STAMP(0);
  const float4 p = posm[body];
  const bool in_lds = nodes <= kSmNodesLds;
  if (t <= kMaxLevels) s_thr[t] = thr_pre;
  __syncthreads();
  if (in_lds) {
    auto put = [&](int m, const float4 &c, unsigned int w) {
      const bool leaf = (w & kLeafBit) != 0u;
      s_a[m] = make_float4(c.x, c.y, c.z, leaf ? 0.0f : s_thr[(w >> kLevelShift) & 63u]);
      s_m[m] = c.w;
      s_past[m] = (unsigned short)(leaf ? m + 1 : (int)(w & kLinkMask));
    };
    float4 c[4];
    unsigned int w[4];
    const bool more = 4 * kWvT < nodes;                         // (uniform) the next nodes' loads go out before the first ones are put away
    if (more) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int m = min(t + (4 + u) * kWvT, nodes - 1); c[u] = T.com[m]; w[u] = T.meta[m]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int m = t + u * kWvT; if (m < nodes) put(m, c0[u], w0[u]); }
    if (more) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int m = t + (4 + u) * kWvT; if (m < nodes) put(m, c[u], w[u]); }
    }
    for (int m0 = t + 8 * kWvT; m0 < nodes; m0 += 4 * kWvT) {   // four nodes' loads in flight per thread: the fill is round trips to L2
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int m = min(m0 + u * kWvT, nodes - 1); c[u] = T.com[m]; w[u] = T.meta[m]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int m = m0 + u * kWvT; if (m < nodes) put(m, c[u], w[u]); }
    }
    __syncthreads();
  }
  float ax = 0.f, ay = 0.f, az = 0.f;                          // Acceleration = ZeroVector, .cpp:84
  BH_WG_STAMP(1);
  if (in_lds)
    walk_wave<true, kWvK>(T, s_a, s_m, s_past, s_thr, s_list[wave], s_term[wave], nodes, valid, p, G, lane, ax, ay, az);
  else   // a tree too large for LDS (deep chains of single-child cells): the same windows on the global arrays
    walk_wave<false, kWvK>(T, s_a, s_m, s_past, s_thr, s_list[wave], s_term[wave], nodes, valid, p, G, lane, ax, ay, az);
  BH_WG_STAMP(2);
#ifdef NBODY_BH_PHASE_CLOCKS
  if (threadIdx.x == 0 && blockIdx.x == gridDim.x - 1) {       // the shader clock under this load: s_sleep 127 = 127 * 64 cycles
    const long long c0 = wall_clock64();
    for (int q = 0; q < 16; ++q) __builtin_amdgcn_s_sleep(127);
    T.clocks[15] = wall_clock64() - c0;
  }
#endif
  walk_row_tail(valid, lane, body, p, ax, ay, az, posm, vel, acc, dt, stage, s_term[wave], S.off);
}

// ... and on the larger systems' tree in its global arrays: a window is sixty-four consecutive nodes — one coalesced 1 KB load
__global__ __launch_bounds__(kWvGT) void bh_walk_wave_rows_kernel(SmallTree T, float4 *__restrict__ posm, float4 *__restrict__ vel,
                                                                  float4 *__restrict__ acc, int n, double G, float dt,
                                                                  float *__restrict__ stage, unsigned int *__restrict__ next_size,
                                                                  float4 *__restrict__ pos_sorted, WalkSlice S) {
  constexpr int kWaves = kWvGT / 64;
  __shared__ float s_thr[kMaxLevels + 2];
  __shared__ unsigned int s_list[kWaves][kWvGK];
  __shared__ __attribute__((aligned(16))) float s_term[kWaves][3 * kWvGK];
  // (a fresh kernel's first look at anything is a trip to memory other XCDs wrote, ~1 us: the body's number, the node count and the
  // thresholds go out together with the verdict, not behind it)
  const int t = threadIdx.x;
  const int wave = t >> 6, lane = t & 63;
  const int k = blockIdx.x * kWaves + wave;
  const bool valid = k < n;                                    // (n: the bodies this context walks — all, or its slice's)
  const int place = valid ? walk_place(S, k) : 0;               // the body's sorted position
  const unsigned int body = valid ? T.sidx[place] : 0u;
  const int status = T.hdr[3], nodes = T.hdr[0];
  const float thr_pre = T.thr[min(t, kMaxLevels)];
  hand_verdict(T);
  if (status != 0) return;
  if (t <= kMaxLevels) s_thr[t] = thr_pre;
  __syncthreads();
  const float4 p = posm[body];
  float ax = 0.f, ay = 0.f, az = 0.f;
  walk_wave<false, kWvGK>(T, (const float4 *)nullptr, (const float *)nullptr, (const unsigned short *)nullptr, s_thr, s_list[wave],
                          s_term[wave], nodes, valid, p, G, lane, ax, ay, az);
  walk_row_tail(valid, lane, body, p, ax, ay, az, posm, vel, acc, dt, stage, s_term[wave], S.off, next_size, pos_sorted, place);
}

// The same walk for systems whose tree does not go into LDS but that have too few bodies to keep the chip busy with one lane
// each (bh_walk_lane_kernel): rows of sixteen lanes on the global arrays, no tree copy.
__global__ __launch_bounds__(kWalkT) void bh_walk_rows_kernel(SmallTree T, float4 *__restrict__ posm, float4 *__restrict__ vel,
                                                              float4 *__restrict__ acc, int n, double G, float dt, float *__restrict__ stage,
                                                              unsigned int *__restrict__ next_size, float4 *__restrict__ pos_sorted,
                                                              WalkSlice S) {
  constexpr int kGroups = kWalkT / kWalkG;
  __shared__ float s_thr[kMaxLevels + 2];
  __shared__ unsigned int s_list[kGroups][kWalkK];
  __shared__ float4 s_term[kGroups][kWalkK];
  const int t = threadIdx.x;
  const int group = t / kWalkG, g = t % kWalkG;
  const int k = blockIdx.x * kGroups + group;
  const bool valid = k < n;                                    // (n: the bodies this context walks — all, or its slice's)
  const int place = valid ? walk_place(S, k) : 0;               // the body's sorted position
  const unsigned int body = valid ? T.sidx[place] : 0u;         // (asked for together with the verdict, as in bh_walk_wave_rows_kernel)
  const int status = T.hdr[3], nodes = T.hdr[0];
  const float thr_pre = T.thr[min(t, kMaxLevels)];
  hand_verdict(T);
  if (status != 0) return;                                     // the frame was refused: nothing moves
  if (t <= kMaxLevels) s_thr[t] = thr_pre;
  __syncthreads();
  const float4 p = posm[body];
  float ax = 0.f, ay = 0.f, az = 0.f;
  walk_windows<false>(T, (const float4 *)nullptr, (const float *)nullptr, (const unsigned short *)nullptr, s_thr, s_list[group],
                      s_term[group], nodes, valid, p, G, g, (t & 63) - g, ax, ay, az);
  walk_row_tail(valid, g, body, p, ax, ay, az, posm, vel, acc, dt, stage, (float *)s_term[group], S.off, next_size, pos_sorted, place);
}

// What DrawOctreeBoxes hands to DrawDebugBox (.cpp:39-40) from the compact tree: the leaf's box follows from the body's
// path digits (the keys of the tree that was built, not the body's position now: the update may have moved it since).
__global__ __launch_bounds__(kB) void bh_small_leaf_boxes_kernel(SmallTree T, int n, float4 *__restrict__ out) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  const unsigned long long h = T.khi[i];
  const int level = T.leaf_level[i];
  const unsigned long long l = level > kLevelsPerKey ? second_word(T, i) : 0ull;
  float o[3] = {T.root[0], T.root[1], T.root[2]};
  float size = T.root[3];
  for (int lev = 0; lev < level; ++lev) {
    const int c = lev < kLevelsPerKey ? (int)((h >> (3 * (kLevelsPerKey - 1 - lev))) & 7ull)
                                      : (int)((l >> (3 * (kMaxLevels - 1 - lev))) & 7ull);
    float no[3], ns;
    child_box(o, size, c, no, &ns);
    o[0] = no[0]; o[1] = no[1]; o[2] = no[2]; size = ns;
  }
  out[T.sidx[i]] = make_float4(o[0], o[1], o[2], size);
}



// Octree::ComputeForces (.h:99-108) on the compact tree, one lane per body in key order, node by node: a 16-byte and an
// 8-byte load (CoM and mass; the hop word), the squared distance and a compare per node (accept_threshold); root, double-precision
// factor and the three multiply-adds only where a term is added.
// (Fetching the NEXT node of the preorder while a node is looked at — the walk goes there whenever it descends or the node is a
// leaf, two steps in three — was tried in round 4: slower at every size, N = 32768 200 us a frame against 185, 65536 213 / 197,
// 2^18 316 / 284, 2^20 843 / 710.)
template <bool TWO>
__global__ __launch_bounds__(kB) void bh_walk_lane_kernel(SmallTree T, float4 *__restrict__ posm, float4 *__restrict__ vel,
                                                          float4 *__restrict__ acc, int n, double G, float dt, float *__restrict__ stage,
                                                          unsigned int *__restrict__ next_size, float4 *__restrict__ pos_sorted,
                                                          WalkSlice S) {
#pragma clang fp contract(off)
  // A fresh kernel's first look at anything is a trip to memory other XCDs wrote (~1 us): the body's number and the root's words are
  // asked for together with the frame's verdict, not behind it (the arrays are there whatever the verdict; nothing is written before it).
  const int wg = xcd_run_block();
  const int k = wg * kB + threadIdx.x;
  const bool valid = k < n;                                    // (n: the bodies this context walks — all, or its slice's)
  const int place = valid ? walk_place(S, k) : 0;               // the body's sorted position
  const unsigned int body = valid ? T.sidx[place] : 0u;
  const int status = T.hdr[3], nodes_all = T.hdr[0];
#if !defined(NBODY_BH_LANE_NO_PIPELINE) && !defined(NBODY_BH_LANE_META_WORD)
  float4 cm = T.com[0];
  uint2 h = T.hop[0];
#endif
  hand_verdict(T);
  if (status != 0) return;                                     // the frame was refused: nothing moves
#if defined(NBODY_BH_LANE_NO_PIPELINE) || defined(NBODY_BH_LANE_META_WORD)
  __shared__ float s_thr[kMaxLevels + 2];
  if (threadIdx.x <= kMaxLevels) s_thr[threadIdx.x] = T.thr[threadIdx.x];
  __syncthreads();
#endif
  const int nodes = valid ? nodes_all : 0;
  const float4 p = posm[body];
  float ax = 0.f, ay = 0.f, az = 0.f;                          // Acceleration = ZeroVector, .cpp:84
  int node = 0;
#ifdef NBODY_BH_LANE_NO_PIPELINE                               // round 4's loop, for A/B builds (make variant)
  while (node < nodes) {
    const float4 cm = T.com[node];
    const unsigned int w = T.meta[node];
    const bool leaf = (w & kLeafBit) != 0u;
    const int past = leaf ? node + 1 : (int)(w & kLinkMask);
    const float ex = p.x - cm.x, ey = p.y - cm.y, ez = p.z - cm.z;
    float d2 = ex * ex + ey * ey;
    d2 = d2 + ez * ez;
    const bool take = leaf || d2 >= s_thr[(w >> kLevelShift) & 63u];
    if (take && d2 != 0.f) {
      float tx, ty, tz;
      force_term(cm.x, cm.y, cm.z, cm.w, p, G, tx, ty, tz);
      ax = ax + tx; ay = ay + ty; az = az + tz;
    }
    node = (take || d2 == 0.f) ? past : node + 1;
  }
#elif defined(NBODY_BH_LANE_META_WORD)                         // this round's first loop (the next node's load under the term; the node's
                                                               // word unpacked and its level's threshold read from LDS inside the step), for A/B builds
  float4 cm = T.com[0];
  unsigned int w = T.meta[0];
  while (node < nodes) {
    const bool leaf = (w & kLeafBit) != 0u;
    const int past = leaf ? node + 1 : (int)(w & kLinkMask);
    const float ex = p.x - cm.x, ey = p.y - cm.y, ez = p.z - cm.z;
    float d2 = ex * ex + ey * ey;
    d2 = d2 + ez * ez;
    const bool take = leaf || d2 >= s_thr[(w >> kLevelShift) & 63u];
    const int next = (take || d2 == 0.f) ? past : node + 1;
    const int fetch = min(next, nodes - 1);
    const float4 cm_next = T.com[fetch];
    const unsigned int w_next = T.meta[fetch];
    if (take && d2 != 0.f) {
      float tx, ty, tz;
      force_term(cm.x, cm.y, cm.z, cm.w, p, G, tx, ty, tz);
      ax = ax + tx; ay = ay + ty; az = az + tz;
    }
    cm = cm_next; w = w_next; node = next;
  }
#else
  // Where the walk goes next follows from the node's test alone — a compare —, not from its term: the NEXT node's load is issued
  // before the term (root, double-precision factor: ~100 dependent instructions) is worked out, and is in flight under it.  At
  // the sizes where a SIMD holds one or two of these waves (N up to ~131072: the walk is a chain of ~220 dependent loads per body,
  // DESIGN 4.5) that takes the term off the chain; at 2^20, where the waves queue for the VALU anyway, it changes nothing.
  // (Not round 4's speculative fetch of node + 1 — wrong one step in three, and slower: this is the node the walk does visit.
  // Nor the node AND its successor per round trip with the second step taken from registers where the walk goes there: a wave
  // takes as many round trips as its slowest lane — 256 became 218, not the 155 of a lane alone — and pays both steps' tests and
  // terms every time: N = 65536 175 us a frame against 152, profiles/r05_ab_lane_walk_node_and_successor.txt.)
  // What is left ON the chain between a node's arrival and the next node's address is kept short: the node's hop word (T.hop,
  // written by bh_nodes_kernel next to the node's packed word) carries the node to go to and the level's threshold ready-made, so
  // a step has no unpacking, no LDS round trip for the threshold and no branch around it: eight fp32 operations, three compares.
  // One step: the node in (CA, HA) is looked at, the node the walk goes to is asked for into (CB, HB).  Two steps to a turn of the
  // loop with the two register sets changing places, so that no step ends with the moves of "next becomes current" (two 64-bit moves
  // behind the loads' arrival, 4 % of the walk's instructions where the waves queue for the VALU).
#define BH_LANE_STEP(CA, HA, CB, HB)                                                                                                   \
  {                                                                                                                                    \
    const float ex = p.x - CA.x, ey = p.y - CA.y, ez = p.z - CA.z;                                                                     \
    float d2 = ex * ex + ey * ey;                                                                                                      \
    d2 = d2 + ez * ez;                                                                                                                 \
    /* .h:103: a leaf, or Size / d < Theta as a threshold on d2 (a NaN distance: leaves only, as there) */                            \
    const bool take = (int)HA.x < 0 || d2 >= __uint_as_float(HA.y);                                                                    \
    /* .h:102: d == 0 adds nothing and ends the subtree; children 0..7 otherwise */                                                   \
    const int next = (take || d2 == 0.f) ? (int)(HA.x & ~kLeafBit) : node + 1;                                                         \
    const unsigned int fetch = (unsigned int)min(next, nodes - 1);   /* (the last step fetches a node nobody looks at) */            \
    CB = *(const float4 *)((const char *)T.com + (fetch << 4));      /* 32-bit byte offsets on the arrays' bases: at most 2^25 nodes */ \
    HB = *(const uint2 *)((const char *)T.hop + (fetch << 3));                                                                        \
    if (take && d2 != 0.f) {                                                                                                           \
      float tx, ty, tz;                                                                                                                \
      force_term(CA.x, CA.y, CA.z, CA.w, p, G, tx, ty, tz);                                                                            \
      ax = ax + tx; ay = ay + ty; az = az + tz;                                                                                        \
    }                                                                                                                                  \
    node = next;                                                                                                                       \
  }
  // (TWO: from 131072 bodies on, where the waves queue for the VALU — N = 2^20 568.8 -> 561.6 us a frame, 2^18 210.6 -> 207.7; below, the
  // second step's own end-of-walk test costs what the moves cost: N = 65536 146.3 -> 147.8.  profiles/r05_ab_lane_walk_two_steps_a_turn.txt)
  if constexpr (!TWO) {
    while (node < nodes) {
      float4 cm2; uint2 h2;
      BH_LANE_STEP(cm, h, cm2, h2)
      cm = cm2; h = h2;
    }
  } else {
    float4 cm2 = cm;
    uint2 h2 = h;
    while (node < nodes) {
      BH_LANE_STEP(cm, h, cm2, h2)
      if (node < nodes) BH_LANE_STEP(cm2, h2, cm, h)
    }
  }
#undef BH_LANE_STEP
#endif
  walk_lane_tail(valid, body, p, ax, ay, az, posm, vel, acc, dt, stage, S.off, next_size, pos_sorted, place);
}


template __global__ void bh_walk_lane_kernel<false>(SmallTree, float4 *, float4 *, float4 *, int, double, float, float *, unsigned int *, float4 *, WalkSlice);
template __global__ void bh_walk_lane_kernel<true>(SmallTree, float4 *, float4 *, float4 *, int, double, float, float *, unsigned int *, float4 *, WalkSlice);

}  // namespace bh
}  // namespace nbody

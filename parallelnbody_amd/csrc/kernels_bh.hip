// GPU Barnes-Hut force pass with the reference's own tree and opening rule.
//
// Restated on the device, operation by operation in the reference's types (paths relative to
// /root/reference/Source/NBody/):
//   class Octree           OctreeSearch.h:21-109   region octree, <= 1 body per leaf, 8 children per split
//   Octree::Add            .h:60-81    -> bh_keys_kernel + radix sort + bh_split_kernel.  The tree the reference
//                                          builds depends only on the SET of positions and on the root box, not on
//                                          the insertion order: a cell is internal iff it holds >= 2 bodies.  A
//                                          body's path (octant = 4[x>=ox] + 2[y>=oy] + [z>=oz] per level, child
//                                          centre = centre +- Size*0.5 evaluated as float(double + double)) is
//                                          computed exactly as Add walks it, packed 3 bits per level into two
//                                          64-bit keys (42 levels), sorted (rocPRIM radix sort), and cells are split level by level.
//   Octree::ComputeMass    .h:83-97    -> bh_upsweep_kernel, children 0..7 in order, fp32, /= as reciprocal multiply (or division: div_mode)
//   Octree::ComputeForces  .h:99-108   -> bh_walk_kernel: depth-first, children 0..7, `Size/d < Theta || leaf`,
//                                          d == 0 skips (also a whole subtree whose CoM coincides with the body),
//                                          scale factor 1e4*M/d^3 in double rounded once to float, separate fp32
//                                          multiply and add.  (d*d)*d in double is the correctly rounded d^3: d*d
//                                          is exact for a float d.
//   CreateOctree root rule .cpp:77-79  root centre = previous tree's CoM (zero the first time), half-width = Size
//                                          from ComputeCubeSize (.cpp:47-56, about the WORLD origin — bodies may lie
//                                          outside the root box; octant tests do not care).
// Every thread follows the reference's arithmetic exactly, so accelerations agree with a CPU restatement of the same
// lines bit for bit (tests/test_bh_gpu.py).  This is latency/divergence-bound integer+fp work, not the FMA-bound
// all-pairs path; it is the drop-in for the reference's SHIPPED configuration (theta = 1.0).
#include <hip/hip_runtime.h>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "kernels.h"

namespace nbody {

namespace {

constexpr int kB = 256;
constexpr int kLevelsPerKey = 21;
constexpr int kMaxLevels = 2 * kLevelsPerKey;   // 42

// child centre and size exactly as Octree::Add computes them (.h:71-74)
__device__ __forceinline__ void child_box(const float o[3], float size, int c, float out[3], float *csize) {
  out[0] = (float)((double)o[0] + (double)size * ((c & 4) ? 0.5 : -0.5));
  out[1] = (float)((double)o[1] + (double)size * ((c & 2) ? 0.5 : -0.5));
  out[2] = (float)((double)o[2] + (double)size * ((c & 1) ? 0.5 : -0.5));
  *csize = (float)(0.5 * (double)size);
}

__global__ __launch_bounds__(kB) void bh_keys_kernel(const float4 *__restrict__ posm, int n,
                                                     const float *__restrict__ root /* ox,oy,oz,size */,
                                                     unsigned long long *__restrict__ key_hi,
                                                     unsigned long long *__restrict__ key_lo,
                                                     unsigned int *__restrict__ idx) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  const float4 p = posm[i];
  float o[3] = {root[0], root[1], root[2]};
  float size = root[3];
  unsigned long long hi = 0, lo = 0;
  for (int l = 0; l < kMaxLevels; ++l) {
    int c = 0;                                            // GetOctant, .h:50-56
    if (p.x >= o[0]) c |= 4;
    if (p.y >= o[1]) c |= 2;
    if (p.z >= o[2]) c |= 1;
    if (l < kLevelsPerKey) hi = (hi << 3) | (unsigned long long)c;
    else lo = (lo << 3) | (unsigned long long)c;
    float no[3], ns;
    child_box(o, size, c, no, &ns);
    o[0] = no[0]; o[1] = no[1]; o[2] = no[2]; size = ns;
  }
  key_hi[i] = hi; key_lo[i] = lo; idx[i] = (unsigned int)i;
}

template <typename T>
__global__ __launch_bounds__(kB) void bh_gather_kernel(const T *__restrict__ src, const unsigned int *__restrict__ idx,
                                                       T *__restrict__ dst, int n) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i < n) dst[i] = src[idx[i]];
}

// Node storage (SoA).  link = (first_child or -1, particle or -1, skip, level); range = [lo, hi) in sorted order.
struct Nodes {
  float4 *box;     // ox, oy, oz, size
  float4 *com;     // cx, cy, cz, M
  int4 *link;
  int2 *range;
};

__global__ void bh_root_kernel(Nodes nd, const float *__restrict__ root, const float4 *__restrict__ posm,
                               const unsigned int *__restrict__ sidx, int n, int *__restrict__ counters,
                               int *__restrict__ frontier) {
  // counters: [0] nodes used, [1] next-frontier size, [2] error flag, [3] current frontier size
  nd.box[0] = make_float4(root[0], root[1], root[2], root[3]);
  nd.range[0] = make_int2(0, n);
  counters[0] = 1; counters[1] = 0; counters[2] = 0;
  if (n >= 2) {
    nd.link[0] = make_int4(-1, -1, -1, 0);
    nd.com[0] = make_float4(0.f, 0.f, 0.f, 0.f);          // TotalMass(0), CenterOfMass(ZeroVector): ctor .h:33
    frontier[0] = 0; counters[3] = 1;
  } else {
    const unsigned int b = sidx[0];
    const float4 p = posm[b];
    nd.link[0] = make_int4(-1, (int)b, -1, 0);
    nd.com[0] = p;                                         // leaf: CoM = Position, TotalMass = Mass (.h:85-88)
    counters[3] = 0;
  }
}

// Create the 8 children of cell `me`, which holds >= 2 bodies (.h:68-75) — EIGHT consecutive lanes per cell, lane c
// makes child c: the eight binary searches (where does key digit > c start?) run side by side instead of one after
// the other (88 dependent loads for the root of a 2000-body tree), the lower bound comes from the neighbour lane.
// All eight lanes of a group must call this together.  node_counter / next_counter / err: global memory or LDS.
__device__ __forceinline__ void split_cell8(const Nodes &nd, const unsigned long long *khi,
                                            const unsigned long long *__restrict__ klo,
                                            const unsigned int *__restrict__ sidx, const float4 *__restrict__ posm, int me,
                                            int c, int *__restrict__ nxt, int *node_counter, int *next_counter, int *err,
                                            int node_cap) {
  const int4 lk = nd.link[me];
  const int level = lk.w;
  const int2 rg = nd.range[me];
  if (level >= kMaxLevels) { if (c == 0) atomicExch(err, 1); return; }  // bodies closer than Size/2^42: the reference would recurse on
  const unsigned long long *kw = level < kLevelsPerKey ? khi : klo;      // only the word that holds this level's digit
  const int ksh = 3 * (kLevelsPerKey - 1 - (level < kLevelsPerKey ? level : level - kLevelsPerKey));
  int base = 0;
  if (c == 0) base = atomicAdd(node_counter, 8);
  base = __shfl(base, 0, 8);
  if (base + 8 > node_cap) { if (c == 0) atomicExch(err, 2); return; }
  const float4 bx = nd.box[me];
  // bodies of this cell are sorted by key, so those of child c are contiguous: find where digit > c starts
  int a = rg.x, b = rg.y;
  while (a < b) {
    const int m = (a + b) >> 1;
    if ((int)((kw[m] >> ksh) & 7ull) <= c) a = m + 1; else b = m;
  }
  const int hi = a;
  int lo = __shfl_up(hi, 1, 8);
  if (c == 0) lo = rg.x;
  if (c == 0) nd.link[me] = make_int4(base, -1, lk.z, level);
  const int cnt = hi - lo, id = base + c;
  const float o[3] = {bx.x, bx.y, bx.z};
  float co[3], cs;
  child_box(o, bx.w, c, co, &cs);
  nd.box[id] = make_float4(co[0], co[1], co[2], cs);
  nd.range[id] = make_int2(lo, hi);
  const int skip = (c < 7) ? id + 1 : lk.z;              // next node of a depth-first walk that does not descend
  if (cnt >= 2) {
    nd.link[id] = make_int4(-1, -1, skip, level + 1);
    nd.com[id] = make_float4(0.f, 0.f, 0.f, 0.f);
    nxt[atomicAdd(next_counter, 1)] = id;
  } else if (cnt == 1) {
    const unsigned int body = sidx[lo];
    nd.link[id] = make_int4(-1, (int)body, skip, level + 1);
    nd.com[id] = posm[body];
  } else {
    nd.link[id] = make_int4(-1, -1, skip, level + 1);
    nd.com[id] = make_float4(0.f, 0.f, 0.f, 0.f);        // empty leaf: TotalMass 0, CoM ZeroVector
  }
}

// Reserve `v` slots for every active lane of the wave with ONE atomicAdd on *counter (hundreds of thousands of cells
// per level would otherwise queue on a single address); returns this lane's first slot.
__device__ __forceinline__ int wave_reserve(int *counter, int v) {
  const unsigned long long active = __ballot(1);
  const int lane = threadIdx.x & 63;
  int before = 0, total = 0;
  for (unsigned long long m = active; m != 0ull; m &= m - 1ull) {      // lanes in ascending order
    const int l = __ffsll((long long)m) - 1;
    const int vl = __shfl(v, l, 64);
    if (l < lane) before += vl;
    total += vl;
  }
  const int leader = __ffsll((long long)active) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, total);
  return __shfl(base, leader, 64) + before;
}

// The same for a constant `v` per lane: no scan needed.
__device__ __forceinline__ int wave_reserve_const(int *counter, int v) {
  const unsigned long long active = __ballot(1);
  const int lane = threadIdx.x & 63;
  const int leader = __ffsll((long long)active) - 1;
  int base = 0;
  if (lane == leader) base = atomicAdd(counter, v * __popcll(active));
  return __shfl(base, leader, 64) + v * __popcll(active & ((1ull << lane) - 1ull));
}

// Create the 8 children of cell `me` by ONE lane (levels with many cells: bandwidth- rather than latency-bound).
__device__ __forceinline__ void split_cell1(const Nodes &nd, const unsigned long long *__restrict__ khi,
                                            const unsigned long long *__restrict__ klo,
                                            const unsigned int *__restrict__ sidx, const float4 *__restrict__ posm, int me,
                                            int *__restrict__ nxt, int *node_counter, int *next_counter, int *err,
                                            int node_cap) {
  const int4 lk = nd.link[me];
  const int level = lk.w;
  const int2 rg = nd.range[me];
  if (level >= kMaxLevels) { atomicExch(err, 1); return; }
  const unsigned long long *kw = level < kLevelsPerKey ? khi : klo;
  const int ksh = 3 * (kLevelsPerKey - 1 - (level < kLevelsPerKey ? level : level - kLevelsPerKey));
  const int base = wave_reserve_const(node_counter, 8);
  if (base + 8 > node_cap) { atomicExch(err, 2); return; }
  nd.link[me] = make_int4(base, -1, lk.z, level);
  const float4 bx = nd.box[me];
  const float o[3] = {bx.x, bx.y, bx.z};
  int bound[9];
  bound[0] = rg.x;
  int splitting = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    int a = bound[c], b = rg.y;
    while (a < b) {
      const int m = (a + b) >> 1;
      if ((int)((kw[m] >> ksh) & 7ull) <= c) a = m + 1; else b = m;
    }
    bound[c + 1] = a;
    splitting += (a - bound[c] >= 2) ? 1 : 0;
  }
  int slot = wave_reserve(next_counter, splitting);
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int lo = bound[c], hi = bound[c + 1], cnt = hi - lo, id = base + c;
    float co[3], cs;
    child_box(o, bx.w, c, co, &cs);
    nd.box[id] = make_float4(co[0], co[1], co[2], cs);
    nd.range[id] = make_int2(lo, hi);
    const int skip = (c < 7) ? id + 1 : lk.z;
    if (cnt >= 2) {
      nd.link[id] = make_int4(-1, -1, skip, level + 1);
      nd.com[id] = make_float4(0.f, 0.f, 0.f, 0.f);
      nxt[slot++] = id;
    } else if (cnt == 1) {
      const unsigned int body = sidx[lo];
      nd.link[id] = make_int4(-1, (int)body, skip, level + 1);
      nd.com[id] = posm[body];
    } else {
      nd.link[id] = make_int4(-1, -1, skip, level + 1);
      nd.com[id] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
}

// Eight threads per cell of the current level that holds >= 2 bodies (LANES = 8), or one (LANES = 1).
template <int LANES>
__global__ __launch_bounds__(kB) void bh_split_kernel(Nodes nd, const unsigned long long *__restrict__ khi,
                                                      const unsigned long long *__restrict__ klo,
                                                      const unsigned int *__restrict__ sidx,
                                                      const float4 *__restrict__ posm, const int *__restrict__ cur,
                                                      int ncur, int *__restrict__ nxt, int *__restrict__ counters,
                                                      int node_cap) {
  const int g = blockIdx.x * kB + threadIdx.x;
  const int f = LANES == 8 ? g >> 3 : g;
  if (f >= ncur) return;
  if (LANES == 8) split_cell8(nd, khi, klo, sidx, posm, cur[f], g & 7, nxt, &counters[0], &counters[1], &counters[2], node_cap);
  else            split_cell1(nd, khi, klo, sidx, posm, cur[f], nxt, &counters[0], &counters[1], &counters[2], node_cap);
}

// Octree::ComputeMass of one cell whose children are done, .h:89-95.
// div_mode: the reading of `CenterOfMass /= TotalMass` (.h:95) — 0: FVector::operator/=(float) multiplies by the fp32
// reciprocal (UE4's implementation as remembered; the engine is not vendored), 1: three divisions.
__device__ __forceinline__ void upsweep_cell(const Nodes &nd, int me, int div_mode) {
#pragma clang fp contract(off)
  const int base = nd.link[me].x;
  float M = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
  for (int c = 0; c < 8; ++c) {
    const float4 ch = nd.com[base + c];
    M = M + ch.w;
    cx = cx + ch.w * ch.x; cy = cy + ch.w * ch.y; cz = cz + ch.w * ch.z;
  }
  if (M != 0.f) {
    if (div_mode == 0) {
      const float rv = 1.0f / M;                           // FVector::operator/=(float): multiply by the reciprocal
      cx = cx * rv; cy = cy * rv; cz = cz * rv;
    } else {
      cx = cx / M; cy = cy / M; cz = cz / M;               // correctly rounded fp32 divisions
    }
  } else {
    const float4 bx = nd.box[me];
    cx = bx.x; cy = bx.y; cz = bx.z;
  }
  nd.com[me] = make_float4(cx, cy, cz, M);
}

// The cells of one level (deepest level first).
__global__ __launch_bounds__(kB) void bh_upsweep_kernel(Nodes nd, const int *__restrict__ cells, int ncells, int div_mode) {
  const int f = blockIdx.x * kB + threadIdx.x;
  if (f >= ncells) return;
  upsweep_cell(nd, cells[f], div_mode);
}

// Small systems (the reference ships N = 2000): the whole Add + ComputeMass — root, every level's splits, the upsweep
// from the deepest level, the next frame's root centre — by ONE workgroup in one launch, levels separated by
// barriers instead of a launch and a host round trip each (N = 2000: 12 levels).
// counters out: [0] nodes, [1] levels, [2] error (0 ok, 1 depth limit, 2 node pool full).
constexpr int kSmallThreads = 1024;
constexpr int kLdsKeys = 8192;
__global__ __launch_bounds__(kSmallThreads) void bh_build_small_kernel(Nodes nd, const float *__restrict__ root,
                                                                       const float4 *__restrict__ posm,
                                                                       const unsigned long long *khi,
                                                                       const unsigned long long *__restrict__ klo,
                                                                       const unsigned int *__restrict__ sidx, int n,
                                                                       int *__restrict__ frontier, int *__restrict__ counters,
                                                                       int node_cap, float *__restrict__ prev_com,
                                                                       int div_mode) {
  __shared__ int s_off[kMaxLevels + 2], s_cnt[kMaxLevels + 2];
  __shared__ int s_nodes, s_next, s_err;
  __shared__ unsigned long long s_khi[kLdsKeys];          // the first 21 levels' digits: binary searches at LDS latency
  const int t = threadIdx.x;
  if (n <= kLdsKeys) {
    for (int i = t; i < n; i += kSmallThreads) s_khi[i] = khi[i];
    khi = s_khi;                                            // generic pointer: flat loads resolve to LDS
  }
  if (t == 0) {
    nd.box[0] = make_float4(root[0], root[1], root[2], root[3]);
    nd.range[0] = make_int2(0, n);
    s_nodes = 1; s_next = 0; s_err = 0;
    if (n >= 2) {
      nd.link[0] = make_int4(-1, -1, -1, 0);
      nd.com[0] = make_float4(0.f, 0.f, 0.f, 0.f);
      frontier[0] = 0;
      s_cnt[0] = 1;
    } else {
      const unsigned int b = sidx[0];
      nd.link[0] = make_int4(-1, (int)b, -1, 0);
      nd.com[0] = posm[b];
      s_cnt[0] = 0;
    }
  }
  __threadfence();
  __syncthreads();
  int levels = 0, cur_off = 0, ncur = s_cnt[0];
  bool failed = false;
  while (ncur > 0) {
    if (levels > kMaxLevels) { failed = true; if (t == 0) s_err = 1; break; }
    int *cur = frontier + cur_off, *nxt = cur + ncur;
    for (int g = t; g < 8 * ncur; g += kSmallThreads)      // eight lanes per cell
      split_cell8(nd, khi, klo, sidx, posm, cur[g >> 3], g & 7, nxt, &s_nodes, &s_next, &s_err, node_cap);
    __threadfence();
    __syncthreads();
    if (s_err != 0) { failed = true; break; }
    if (t == 0) { s_off[levels] = cur_off; s_cnt[levels] = ncur; }
    ++levels;
    cur_off += ncur;
    ncur = s_next;
    __syncthreads();
    if (t == 0) s_next = 0;
    __syncthreads();
  }
  if (!failed) {
    for (int l = levels - 1; l >= 0; --l) {                  // ComputeMass: children before parents
      const int off = s_off[l], cnt = s_cnt[l];
      for (int f = t; f < cnt; f += kSmallThreads) upsweep_cell(nd, frontier[off + f], div_mode);
      __threadfence();
      __syncthreads();
    }
  }
  if (t == 0) {
    if (!failed) { const float4 c = nd.com[0]; prev_com[0] = c.x; prev_com[1] = c.y; prev_com[2] = c.z; }   // .cpp:78
    counters[0] = s_nodes; counters[1] = levels; counters[2] = s_err; counters[3] = 0;
  }
}

// Octree::ComputeForces for every body (.h:99-108), bodies taken in key order for coherence.
__global__ __launch_bounds__(kB) void bh_walk_kernel(Nodes nd, const float4 *__restrict__ posm,
                                                     const unsigned int *__restrict__ sidx, int n, float theta, double G,
                                                     float4 *__restrict__ acc) {
#pragma clang fp contract(off)
  const int k = blockIdx.x * kB + threadIdx.x;
  if (k >= n) return;
  const unsigned int body = sidx[k];
  const float4 p = posm[body];
  float ax = 0.f, ay = 0.f, az = 0.f;                      // Acceleration = ZeroVector, .cpp:84
  int node = 0;
  while (node >= 0) {
    const int4 lk = nd.link[node];
    const bool leaf = lk.x < 0;
    if (leaf && lk.y < 0) { node = lk.z; continue; }                      // .h:100
    const float4 cm = nd.com[node];
    const float ex = p.x - cm.x, ey = p.y - cm.y, ez = p.z - cm.z;
    float d2 = ex * ex + ey * ey;
    d2 = d2 + ez * ez;
    const float d = sqrtf(d2);   // FVector::Dist, .h:101 — correctly rounded (hipcc default); __fsqrt_rn is the 1-ulp native op
    if (d == 0.f) { node = lk.z; continue; }                              // .h:102
    const float size = nd.box[node].w;
    if (size / d < theta || lk.y >= 0) {                                  // .h:103
      const double dd = (double)d;
      const float s = (float)(G * (double)cm.w / ((dd * dd) * dd));       // .h:104
      ax = ax + s * (cm.x - p.x); ay = ay + s * (cm.y - p.y); az = az + s * (cm.z - p.z);
      node = lk.z;
    } else if (!leaf) {
      node = lk.x;                                                        // children 0..7, .h:105-107
    } else {
      node = lk.z;
    }
  }
  acc[body] = make_float4(ax, ay, az, 0.f);
}

// root = (previous CoM, Size): Size arrives as the bit pattern bounds_kernel leaves; the CoM is the last tree's root.
__global__ void bh_set_root_kernel(float *__restrict__ root, const float *__restrict__ prev_com,
                                   const unsigned int *__restrict__ size_bits) {
  root[0] = prev_com[0]; root[1] = prev_com[1]; root[2] = prev_com[2];
  root[3] = __uint_as_float(*size_bits);
}

// What DrawOctreeBoxes hands to DrawDebugBox (OctreeSearch.cpp:39-40): the box (Origin, Size) of the leaf that
// holds each body, written at the body's index.
__global__ __launch_bounds__(kB) void bh_leaf_boxes_kernel(Nodes nd, int nodes, float4 *__restrict__ out) {
  const int k = blockIdx.x * kB + threadIdx.x;
  if (k >= nodes) return;
  const int4 lk = nd.link[k];
  if (lk.x < 0 && lk.y >= 0) out[lk.y] = nd.box[k];
}

__global__ void bh_save_com_kernel(Nodes nd, float *__restrict__ prev_com) {
  const float4 c = nd.com[0];
  prev_com[0] = c.x; prev_com[1] = c.y; prev_com[2] = c.z;
}

}  // namespace

constexpr int kCoopCells = 32768;     // levels with fewer cells than this split with eight lanes per cell
constexpr int kSmallBodies = 16384;   // up to here one workgroup builds the whole tree (bh_build_small_kernel)

struct BhState {
  int n = 0, node_cap = 0;
  unsigned long long *khi = nullptr, *klo = nullptr, *khi2 = nullptr, *klo2 = nullptr;
  unsigned int *idx = nullptr, *idx2 = nullptr;
  void *sort_tmp = nullptr;
  size_t sort_tmp_bytes = 0;
  Nodes nd{};
  int *frontier = nullptr;     // all levels' internal cells, level after level
  int *counters = nullptr;     // device: nodes used, next-frontier size, error, current size
  int *h_counters = nullptr;   // pinned
  float *root = nullptr;       // ox, oy, oz, size
  float *prev_com = nullptr;   // the previous tree's root CoM (zero before the first frame)
  int last_nodes = 0, last_levels = 0;
  int div_mode = 0;            // reading of `/=` in ComputeMass (upsweep_cell)
};

#define BH_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

hipError_t bh_create(BhState **out, int n) {
  BhState *b = new BhState();
  b->n = n;
  b->node_cap = 8 * (4 * n + 1024) + 1;
  BH_TRY(hipMalloc(&b->khi, sizeof(unsigned long long) * n));
  BH_TRY(hipMalloc(&b->klo, sizeof(unsigned long long) * n));
  BH_TRY(hipMalloc(&b->khi2, sizeof(unsigned long long) * n));
  BH_TRY(hipMalloc(&b->klo2, sizeof(unsigned long long) * n));
  BH_TRY(hipMalloc(&b->idx, sizeof(unsigned int) * n));
  BH_TRY(hipMalloc(&b->idx2, sizeof(unsigned int) * n));
  size_t bytes = 0;
  BH_TRY(rocprim::radix_sort_pairs(nullptr, bytes, b->klo, b->klo2, b->idx, b->idx2, (unsigned int)n));
  b->sort_tmp_bytes = bytes;
  BH_TRY(hipMalloc(&b->sort_tmp, bytes));
  BH_TRY(hipMalloc(&b->nd.box, sizeof(float4) * b->node_cap));
  BH_TRY(hipMalloc(&b->nd.com, sizeof(float4) * b->node_cap));
  BH_TRY(hipMalloc(&b->nd.link, sizeof(int4) * b->node_cap));
  BH_TRY(hipMalloc(&b->nd.range, sizeof(int2) * b->node_cap));
  BH_TRY(hipMalloc(&b->frontier, sizeof(int) * (b->node_cap / 8 + 8)));
  BH_TRY(hipMalloc(&b->counters, sizeof(int) * 4));
  BH_TRY(hipHostMalloc(&b->h_counters, sizeof(int) * 4, hipHostMallocDefault));
  BH_TRY(hipMalloc(&b->root, sizeof(float) * 4));
  BH_TRY(hipMalloc(&b->prev_com, sizeof(float) * 3));
  BH_TRY(hipMemset(b->prev_com, 0, sizeof(float) * 3));    // FVector t = ZeroVector, .cpp:77
  *out = b;
  return hipSuccess;
}

void bh_destroy(BhState *b) {
  if (!b) return;
  void *ptrs[] = {b->khi, b->klo, b->khi2, b->klo2, b->idx, b->idx2, b->sort_tmp, b->nd.box, b->nd.com, b->nd.link,
                  b->nd.range, b->frontier, b->counters, b->root, b->prev_com};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  if (b->h_counters) (void)hipHostFree(b->h_counters);
  delete b;
}

hipError_t bh_reset_root(BhState *b, hipStream_t s) { return hipMemsetAsync(b->prev_com, 0, sizeof(float) * 3, s); }

// One CreateOctree (.cpp:74-89) on the device.  size_bits: device word holding Size as left by the bounds kernel.
// *status: 0 ok, 1 depth limit (bodies closer than Size/2^42 — the reference would keep recursing), 2 node pool full.
hipError_t bh_forces(BhState *b, const void *posm_v, void *acc_v, const unsigned int *size_bits, float theta, double G,
                     hipStream_t s, int *status) {
  const float4 *posm = (const float4 *)posm_v;
  float4 *acc = (float4 *)acc_v;
  const int n = b->n;
  const dim3 blk(kB), grd((n + kB - 1) / kB);
  *status = 0;
  hipLaunchKernelGGL(bh_set_root_kernel, dim3(1), dim3(1), 0, s, b->root, b->prev_com, size_bits);
  hipLaunchKernelGGL(bh_keys_kernel, grd, blk, 0, s, posm, n, b->root, b->khi, b->klo, b->idx);
  // stable LSD sort over the 126-bit key: low word first, then the high word
  size_t tb = b->sort_tmp_bytes;
  BH_TRY(rocprim::radix_sort_pairs(b->sort_tmp, tb, b->klo, b->klo2, b->idx, b->idx2, (unsigned int)n, 0u, 63u, s));
  hipLaunchKernelGGL((bh_gather_kernel<unsigned long long>), grd, blk, 0, s, b->khi, b->idx2, b->khi2, n);
  tb = b->sort_tmp_bytes;
  BH_TRY(rocprim::radix_sort_pairs(b->sort_tmp, tb, b->khi2, b->khi, b->idx2, b->idx, (unsigned int)n, 0u, 63u, s));
  // b->khi / b->idx are final; bring the low words (b->klo is still in body order) into the same order
  hipLaunchKernelGGL((bh_gather_kernel<unsigned long long>), grd, blk, 0, s, b->klo, b->idx, b->klo2, n);

  if (n <= kSmallBodies) {
    // one launch builds and sweeps the tree; the walk follows at once, the verdict is read after it (a refused tree
    // is still walkable: unsplit cells look like empty leaves)
    hipLaunchKernelGGL(bh_build_small_kernel, dim3(1), dim3(kSmallThreads), 0, s, b->nd, b->root, posm, b->khi, b->klo2,
                       b->idx, n, b->frontier, b->counters, b->node_cap, b->prev_com, b->div_mode);
    hipLaunchKernelGGL(bh_walk_kernel, grd, blk, 0, s, b->nd, posm, b->idx, n, theta, G, acc);
    BH_TRY(hipMemcpyAsync(b->h_counters, b->counters, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
    BH_TRY(hipStreamSynchronize(s));
    b->last_nodes = b->h_counters[0];
    b->last_levels = b->h_counters[1];
    *status = b->h_counters[2];
    return hipGetLastError();
  }
  hipLaunchKernelGGL(bh_root_kernel, dim3(1), dim3(1), 0, s, b->nd, b->root, posm, b->idx, n, b->counters, b->frontier);
  // level by level; the frontier of level l sits at frontier[off[l] .. off[l] + cnt[l])
  int off[kMaxLevels + 2], cnt[kMaxLevels + 2];
  int levels = 0, cur_off = 0;
  BH_TRY(hipMemcpyAsync(b->h_counters, b->counters, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
  BH_TRY(hipStreamSynchronize(s));
  int ncur = b->h_counters[3];
  b->last_nodes = 1;
  while (ncur > 0) {
    if (levels > kMaxLevels) { *status = 1; return hipSuccess; }
    off[levels] = cur_off; cnt[levels] = ncur; ++levels;
    int *cur = b->frontier + cur_off, *nxt = cur + ncur;
    if (ncur < kCoopCells)     // few cells: latency-bound, eight lanes per cell; many: one lane each
      hipLaunchKernelGGL(bh_split_kernel<8>, dim3((8 * ncur + kB - 1) / kB), blk, 0, s, b->nd, b->khi, b->klo2, b->idx, posm,
                         cur, ncur, nxt, b->counters, b->node_cap);
    else
      hipLaunchKernelGGL(bh_split_kernel<1>, dim3((ncur + kB - 1) / kB), blk, 0, s, b->nd, b->khi, b->klo2, b->idx, posm,
                         cur, ncur, nxt, b->counters, b->node_cap);
    BH_TRY(hipMemcpyAsync(b->h_counters, b->counters, sizeof(int) * 4, hipMemcpyDeviceToHost, s));
    BH_TRY(hipMemsetAsync(b->counters + 1, 0, sizeof(int), s));
    BH_TRY(hipStreamSynchronize(s));
    if (b->h_counters[2] != 0) { *status = b->h_counters[2]; return hipSuccess; }
    cur_off += ncur;
    ncur = b->h_counters[1];
    b->last_nodes = b->h_counters[0];
  }
  b->last_levels = levels;
  for (int l = levels - 1; l >= 0; --l)                      // ComputeMass: children before parents
    hipLaunchKernelGGL(bh_upsweep_kernel, dim3((cnt[l] + kB - 1) / kB), blk, 0, s, b->nd, b->frontier + off[l], cnt[l], b->div_mode);
  hipLaunchKernelGGL(bh_save_com_kernel, dim3(1), dim3(1), 0, s, b->nd, b->prev_com);   // next frame's root centre, .cpp:78
  hipLaunchKernelGGL(bh_walk_kernel, grd, blk, 0, s, b->nd, posm, b->idx, n, theta, G, acc);
  return hipGetLastError();
}

hipError_t bh_leaf_boxes(BhState *b, void *out, hipStream_t s) {
  if (b->last_nodes <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bh_leaf_boxes_kernel, dim3((b->last_nodes + kB - 1) / kB), dim3(kB), 0, s, b->nd, b->last_nodes,
                     (float4 *)out);
  return hipGetLastError();
}

void bh_set_div_mode(BhState *b, int div_mode) { b->div_mode = div_mode ? 1 : 0; }

// The bodies in the order DrawOctreeBoxes meets their leaves (OctreeSearch.cpp:36-45: depth first, children 0..7): the
// path keys are the octant digits root to leaf, so key order IS that order.
hipError_t bh_leaf_order(BhState *b, int *out_host, hipStream_t s) {
  if (b->last_nodes <= 0) return hipErrorInvalidValue;
  BH_TRY(hipStreamSynchronize(s));
  return hipMemcpy(out_host, b->idx, sizeof(unsigned int) * (size_t)b->n, hipMemcpyDeviceToHost);
}

void bh_stats(const BhState *b, int *nodes, int *levels) {
  if (nodes) *nodes = b->last_nodes;
  if (levels) *levels = b->last_levels;
}

hipError_t bh_get_root_com(BhState *b, float out[3], hipStream_t s) {
  BH_TRY(hipStreamSynchronize(s));
  return hipMemcpy(out, b->prev_com, sizeof(float) * 3, hipMemcpyDeviceToHost);
}

hipError_t bh_set_root_com(BhState *b, const float in[3], hipStream_t s) {
  BH_TRY(hipStreamSynchronize(s));
  return hipMemcpy(b->prev_com, in, sizeof(float) * 3, hipMemcpyHostToDevice);
}

}  // namespace nbody

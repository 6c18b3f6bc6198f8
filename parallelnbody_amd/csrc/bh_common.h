// bh_common.h — what the translation units of the theta > 0 path share (round 5 split csrc/kernels_bh.hip, 2 800 lines and 24 kernels in
// one file, into these; no code changed):
//   kernels_bh_small.hip   systems up to 4096 bodies: bh_small_build_kernel — the whole CreateOctree head by one workgroup in LDS
//   kernels_bh_sort.hip    the larger systems' path keys and their order: the sort from the previous frame's order, the cold sorts
//   kernels_bh_build.hip   shared digits + node numbering, node words and leaves, ComputeMass (chunks + top, or a launch per level)
//   kernels_bh_walk.hip    Octree::ComputeForces: the walks (wave / sixteen lanes / one lane per body), the Tick's update behind them
//   bh_frame.hip           the host side: BhState, one frame queued on the stream (bh_frame), the verdict (bh_collect)
// Here: the constants and the compact tree (SmallTree) the kernels and the host agree on, the device helpers more than one file
// uses — each a restatement of a line of the reference, cited where it stands — and the kernels' declarations.
//
// GPU Barnes-Hut force pass with the reference's own tree and opening rule.
//
// Restated on the device, operation by operation in the reference's types (paths relative to
// /root/reference/Source/NBody/):
//   class Octree           OctreeSearch.h:21-109   region octree, <= 1 body per leaf, 8 children per split
//   Octree::Add            .h:60-81    The tree the reference builds depends only on the SET of positions and on the root box,
//                                          not on the insertion order: a cell is internal iff it holds >= 2 bodies.  A body's
//                                          path (octant = 4[x>=ox] + 2[y>=oy] + [z>=oz] per level, child centre = centre +-
//                                          Size*0.5 evaluated as float(double + double)) is computed exactly as Add walks it
//                                          and packed 3 bits per level into two 64-bit keys (42 levels); the sorted keys say
//                                          which cells exist, and number them in depth-first order (the compact tree below).
//   Octree::ComputeMass    .h:83-97    -> sweep_compact_cell, children in octant order, fp32, /= as reciprocal multiply (or division: div_mode)
//   Octree::ComputeForces  .h:99-108   -> the walks: depth-first, children 0..7, `Size/d < Theta || leaf` (as a threshold on
//                                          the squared distance: accept_threshold), d == 0 skips (also a whole subtree whose
//                                          CoM coincides with the body), scale factor 1e4*M/d^3 in double rounded once to
//                                          float, separate fp32 multiply and add.  (d*d)*d in double is the correctly rounded
//                                          d^3: d*d is exact for a float d.
//   CreateOctree root rule .cpp:77-79  root centre = previous tree's CoM (zero the first time), half-width = Size
//                                          from ComputeCubeSize (.cpp:47-56, about the WORLD origin — bodies may lie
//                                          outside the root box; octant tests do not care).
// Every thread follows the reference's arithmetic exactly, so accelerations agree with a CPU restatement of the same
// lines bit for bit (tests/test_bh_gpu.py).  This is latency/divergence-bound integer+fp work, not the FMA-bound
// all-pairs path; it is the drop-in for the reference's SHIPPED configuration (theta = 1.0).
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace nbody {
namespace bh {

constexpr int kB = 256;
constexpr int kLevelsPerKey = 21;
constexpr int kMaxLevels = 2 * kLevelsPerKey;   // 42

// A workgroup barrier for data handed over in LDS: it waits for this wave's LDS traffic only.  (__syncthreads() also waits for every
// global store and load the wave has in flight — a round trip to L2, ~1 us, at each barrier behind a store; the stores of these
// kernels are read by later launches, and loads fetched ahead are meant to stay in flight.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// child centre and size exactly as Octree::Add computes them (.h:71-74)
__device__ __forceinline__ void child_box(const float o[3], float size, int c, float out[3], float *csize) {
  out[0] = (float)((double)o[0] + (double)size * ((c & 4) ? 0.5 : -0.5));
  out[1] = (float)((double)o[1] + (double)size * ((c & 2) ? 0.5 : -0.5));
  out[2] = (float)((double)o[2] + (double)size * ((c & 1) ? 0.5 : -0.5));
  *csize = (float)(0.5 * (double)size);
}

// ComputeCubeSize (.cpp:47-56) of a frame of the larger systems stands in kSizeSlots words (bit patterns of non-negative floats,
// which order as unsigned integers): the bounds kernel leaves it in the first of them; a walk that moves the bodies (dt > 0) leaves
// the NEXT frame's there, every wave that finishes raising its workgroup's slot where its body reaches further out — the frame
// that follows then needs no pass over the positions of its own.
constexpr int kSizeSlots = 64;
__device__ __forceinline__ float frame_size(const unsigned int *__restrict__ size_bits) {
  unsigned int v = size_bits[threadIdx.x & (kSizeSlots - 1)];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = max(v, (unsigned int)__shfl_xor((int)v, off, 64));
  return __uint_as_float(v);
}
// (x, y, z): the position a lane's body has moved to (mine: the lane has one); the wave's largest |coordinate| goes to the workgroup's slot
__device__ __forceinline__ void note_next_size(unsigned int *__restrict__ next_size, bool mine, float x, float y, float z) {
  if (next_size == nullptr) return;                            // (uniform)
  float m = mine ? fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z)) : 0.0f;   // GetAbsMax (bounds_kernel)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) {
    // (an atomic whose answer nobody waits for: asking the slot first would put an L2 round trip at the end of every wave)
    if (m > 0.0f) atomicMax(next_size + (blockIdx.x & (kSizeSlots - 1)), __float_as_uint(m));
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Small systems (the reference ships N = 2000; up to kSmBodies): the whole CreateOctree head — ComputeCubeSize, the root
// rule, Add, ComputeMass (.cpp:47-56, 77-81; .h:60-97) — by ONE workgroup in ONE launch, and a COMPACT tree for the walk.
//
// Which cells exist follows from the sorted path keys alone: with lcp(i) = the number of leading octant digits bodies i
// and i+1 (key order) share, the cells holding >= 2 bodies whose first body is i are those of levels
// lcp(i-1)+1 .. lcp(i) — every one of them, chains of single-child cells included (the reference creates them, each
// with its own CoM rounding and its own Size in the opening test).  In depth-first order, children 0..7 — the order
// ComputeForces recurses in — the nodes are: for body i, those cells by level, then the body's leaf.  So an exclusive
// scan of (cells opened at i) + 1 numbers all nodes in PREORDER with no level-by-level construction and no atomics:
// "descend" is node + 1, "do not descend" is the node after the cell's last body (a binary search on the keys), and
// the empty leaves of the reference's 8-way split (.h:100: a walk returns from them at once; ComputeMass adds +0 for
// them, which cannot change a sum that started at +0) are simply not there.  A node is 20 bytes: (CoM, M) and one word
// {leaf, level, skip link | body}; Size comes from the level (halved exactly per level, .h:74).  Up to kSmNodesLds nodes
// the tree lives in LDS while it is built and swept, and the walk's workgroups read it from LDS too.
constexpr int kSmT = 1024;                 // threads of the build workgroup
constexpr int kSmBodies = 4096;            // bodies the in-LDS sort holds
constexpr int kSmNodesLds = 6656;          // compact nodes that fit in LDS: the walk keeps 22 B of each (146 KB)
// LDS of the build, bytes.  While the structure is found: two (key word, body) buffers the merge sort ping-pongs between,
// the second key word by body; afterwards the nodes' (CoM, M) take that space.  Behind it the node words and the list of
// cells by level, which must not overlap anything the node-word phase still reads.
constexpr int kSmBuf = kSmBodies * (8 + 2);                    // one sort buffer: hi[P], idx[P]
constexpr int kSmRegionA = 2 * kSmBuf + kSmBodies * 8;         // 114688
static_assert(kSmRegionA >= kSmNodesLds * 16, "the CoMs must fit where the sort ran");
constexpr int kSmLds = kSmRegionA + kSmNodesLds * 4 + kSmNodesLds * 2;
constexpr unsigned int kLeafBit = 0x80000000u;
constexpr int kLevelShift = 25;
constexpr unsigned int kLinkMask = (1u << kLevelShift) - 1u;
constexpr int kSmGlobalWalkN = 2560;       // small systems from here on walk the tree in global memory (bh_frame)
constexpr int kWalkT = 256;                // threads per workgroup of the compact walk
constexpr int kWalkG = 16;                 // lanes per body there
constexpr int kWalkK = 48;                 // taken nodes a body lists before their terms are worked out and added
constexpr int kWaveMaxN = 12288;           // ... and up to here with a whole wave per body (frames: N = 8192 157 us against 168, 16384 192 / 186: profiles/r04_bh_walk_ab.txt)
constexpr int kLaneTwoStepsMinN = 131072;      // a lane per body takes two steps to a turn of its loop from here on (the waves queue for the VALU)
constexpr int kRowsMaxN = 12288;           // larger systems up to here walk with windows on the global tree, above with a lane per body.  (Round 3:
                                           // 20480 — sixteen lanes per body between kWaveMaxN and there; frames: N = 8192 208 us against 275 with a lane per body,
                                           // 16384 231 / 270, 32768 321 / 272.  Since round 5's hop word the lane walk wins from ~11 000 bodies in Plummer spheres —
                                           // N = 12288 130.7 against 138.6 us, 16384 133.6 / 148.1, 20480 136.1 / 167.9 — and loses 3 % at 16384 in the reference's box
                                           // scene, whose runaway bodies make the walks long: profiles/r05_bh_walk_by_size.txt.  The sixteen-lane walk stays for
                                           // NBODY_BH_ROWS_MAX_N / NBODY_BH_WALK=rows and the tests that drive it.)

struct SmallTree {
  float4 *com;                  // [cap] preorder nodes: centre of mass, total mass
  unsigned int *meta;           // [cap] leaf bit | level << 25 | (internal: the node after the subtree; leaf: the body)
  uint2 *hop;                   // [cap] what one step of the lane walk needs besides the CoM, in one 8-byte load: x = leaf bit | the node a walk
                                // that does not descend goes to (a leaf's: the next node), y = the acceptance threshold of the node's level
                                // (thr[level]; +0 for a leaf).  Only systems that walk with a lane per body hold it (nullptr otherwise).
  unsigned long long *khi, *klo;   // [n] sorted path keys (what the leaf boxes are rebuilt from).  klo: the second key words — in key
                                   // order when klo_by_body == 0 (small systems), in BODY order otherwise (larger systems: only the
                                   // first words go through the sort, and the second ones are looked at only where two first
                                   // words agree or a cell lies below level 21: second_word())
  int klo_by_body;
  unsigned int *sidx;           // [n] body at each sorted position = DrawOctreeBoxes' order
  unsigned char *leaf_level;    // [n] level of the leaf of the body at each sorted position
  int *verdict;                 // page-locked HOST memory: the frame's last kernel copies header words 0 .. 7 there (bh_collect reads them after its wait)
  int *hdr;                     // [0] compact nodes, [1] cells with >= 2 bodies, [2] levels, [3] status (sticky), [4] frames built
  float *root;                  // ox, oy, oz, Size of the current tree
  float *prev_com;              // the previous tree's root CoM = the next tree's root centre (.cpp:77-79)
  float *thr;                   // [kMaxLevels + 2] a cell of level l is accepted (.h:103) iff the squared distance >= thr[l]
  int *lvl;                     // larger systems, [2][64]: how many chunks leave a cell of level l to ComputeMass' second launch, and how many of
                                // them leave one of level l - 1 as well (bh_sweep_chunks_kernel counts, bh_sweep_top_kernel skips barriers by them)
  long long *clocks;            // build with -DNBODY_BH_PHASE_CLOCKS: wall_clock64 at the kernels' phase boundaries
  int cap;
};

// second key word of the body at sorted position i
__device__ __forceinline__ unsigned long long second_word(const SmallTree &T, int i) {
  return T.klo_by_body ? T.klo[T.sidx[i]] : T.klo[i];
}

// The frame's verdict for the host: every walk kernel — a frame's last launch, refused or not — has its first workgroup copy the
// header's first eight words (node count, cells, levels, status, frames built, ..., Size) into page-locked host memory, so that
// bh_collect's one wait needs no copy queued behind the frame (~4 us of an actor-style frame).
__device__ __forceinline__ void hand_verdict(const SmallTree &T) {
  if (blockIdx.x == 0 && threadIdx.x < 8) T.verdict[threadIdx.x] = T.hdr[threadIdx.x];
}

constexpr int kHdrDeep = 8, kDeepSlots = 1024;   // header words [8, 1032): the deepest level, one word per slot (larger systems):
                                                 // same-address atomics queue up (N = 2^20, 4096 workgroups: the lcp kernel: 71 us with 64 slots)
constexpr int kHdrWords = kHdrDeep + kDeepSlots;
constexpr int kDbgClocks = 16 + 3 * 512;   // tuning builds: 16 phase stamps + (start, fill end, end) of up to 512 walk workgroups
#ifdef NBODY_BH_PHASE_CLOCKS
#define BH_CLOCK(k) do { if (threadIdx.x == 0) T.clocks[k] = wall_clock64(); } while (0)
#define BH_WALK_CLOCK(k) do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) T.clocks[k] = wall_clock64(); } while (0)
#define BH_WALK_COUNT(k, v) do { if (threadIdx.x == 0 && blockIdx.x == gridDim.x / 2) T.clocks[k] = (v); } while (0)
#define BH_WG_STAMP(slot) do { if (threadIdx.x == 0 && blockIdx.x < 512) T.clocks[16 + 3 * blockIdx.x + (slot)] = wall_clock64(); } while (0)
#else
#define BH_CLOCK(k) do { } while (0)
#define BH_WALK_CLOCK(k) do { } while (0)
#define BH_WALK_COUNT(k, v) do { } while (0)
#define BH_WG_STAMP(slot) do { } while (0)
#endif

// Octree::ComputeForces accepts a cell when `Size / d < Theta` (.h:103) with d = FVector::Dist = sqrtf(d2) (.h:101), both
// correctly rounded.  Quotient and root are monotone in their argument, so for every Size there is ONE float D with
// (Size / sqrtf(d2) < Theta)  <=>  d2 >= D.  It is found with the very operations the test itself uses — a guess, a short
// scan over neighbouring bit patterns, bisection over all of them should the guess be far off — and the walk then decides
// with a compare; it needs the root only where a term is added.
template <typename PRED>
__device__ __forceinline__ unsigned int first_true(unsigned int lo, unsigned int hi, unsigned int guess, PRED pred) {
  // smallest pattern in [lo, hi] for which pred holds (pred is monotone: false ... false true ... true, true at hi)
  if (guess > lo + 4u && guess + 4u < hi && !pred(guess - 4u) && pred(guess + 4u)) {
    unsigned int g = guess - 3u;
    while (!pred(g)) ++g;
    return g;
  }
  unsigned int a = lo, b = hi;
  while (a < b) {
    const unsigned int mid = a + ((b - a) >> 1);
    if (pred(mid)) b = mid; else a = mid + 1u;
  }
  return a;
}

__device__ __forceinline__ float accept_threshold(float size, float theta) {
  // smallest d > 0 with size / d < theta (d = +inf: 0 < theta), then the smallest d2 >= 0 with sqrtf(d2) >= that d
  const unsigned int db = first_true(1u, 0x7F800000u, __float_as_uint(size / theta),
                                     [&](unsigned int v) { return size / __uint_as_float(v) < theta; });
  const float dmin = __uint_as_float(db);
  const unsigned int d2b = first_true(0u, 0x7F800000u, __float_as_uint(dmin * dmin),
                                      [&](unsigned int v) { return sqrtf(__uint_as_float(v)) >= dmin; });
  return __uint_as_float(d2b);
}

// One level of Octree::Add's descent (.h:50-56, 68-75): the octant of p in the cell (o, size), then the child's box.
// float(double(o) +- double(size) * 0.5) equals the plain fp32 o +- 0.5f * size whenever 0.5f * size is exact (the sum
// of two floats rounds once to float either way: it is exact in double unless the smaller one is below a 2^-29th of an
// ulp of the larger — tests/cpp/child_centre_equiv.c checks the claim on random operand pairs); only sizes down in the
// denormal range take the double path.
__device__ __forceinline__ int descend_level(const float4 &p, float o[3], float &size) {
  int c = 0;                                                   // GetOctant
  if (p.x >= o[0]) c |= 4;
  if (p.y >= o[1]) c |= 2;
  if (p.z >= o[2]) c |= 1;
  if (size >= 0x1p-100f) {
#pragma clang fp contract(off)
    const float h = 0.5f * size;
    o[0] = (c & 4) ? o[0] + h : o[0] - h;
    o[1] = (c & 2) ? o[1] + h : o[1] - h;
    o[2] = (c & 1) ? o[2] + h : o[2] - h;
    size = h;
  } else {
    float no[3], ns;
    child_box(o, size, c, no, &ns);
    o[0] = no[0]; o[1] = no[1]; o[2] = no[2]; size = ns;
  }
  return c;
}

// Do the keys share their first l octant digits (0 < l <= 42)?
__device__ __forceinline__ bool same_prefix(unsigned long long ha, unsigned long long la, unsigned long long hb,
                                            unsigned long long lb, int l) {
  if (l <= kLevelsPerKey) return (ha >> (3 * (kLevelsPerKey - l))) == (hb >> (3 * (kLevelsPerKey - l)));
  return ha == hb && (la >> (3 * (kMaxLevels - l))) == (lb >> (3 * (kMaxLevels - l)));
}

// The end of Octree::ComputeMass for one cell (.h:94-95): CenterOfMass /= TotalMass, or the cell's own box origin when it
// holds no mass.  (M, cx, cy, cz): the children's masses and mass-weighted centres, summed in octant order.
// (meta / com may be arrays that hold the nodes from number `off` on: a chunk's nodes in LDS)
__device__ __forceinline__ float4 cell_com_from_sums(float M, float cx, float cy, float cz, const unsigned int *meta, int m, int l,
                                                     int div_mode, const float4 *__restrict__ posm, const float *root, int off = 0) {
#pragma clang fp contract(off)
  if (M != 0.f) {
    if (div_mode == 0) { const float rv = 1.0f / M; cx = cx * rv; cy = cy * rv; cz = cz * rv; }
    else { cx = cx / M; cy = cy / M; cz = cz / M; }
  } else {                                                     // CenterOfMass = Origin (.h:95 else branch): the cell's own box
    int c = m + 1;
    while (!(meta[c - off] & kLeafBit)) ++c;                    // any body of the cell: its path leads through the cell
    const float4 p = posm[meta[c - off] & kLinkMask];
    float o[3] = {root[0], root[1], root[2]};
    float size = root[3];
    for (int lev = 0; lev < l; ++lev) (void)descend_level(p, o, size);
    cx = o[0]; cy = o[1]; cz = o[2];
  }
  return make_float4(cx, cy, cz, M);
}

// Octree::ComputeMass of one cell of the compact tree whose children are done (.h:89-95): node m, its word w, level l.
__device__ __forceinline__ float4 sweep_compact_cell(const float4 *com, const unsigned int *meta, int m, unsigned int w, int l,
                                                     int div_mode, const float4 *__restrict__ posm, const float *root, int off = 0) {
#pragma clang fp contract(off)
  const int end = (int)(w & kLinkMask);
  float M = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
  for (int c = m + 1; c != end;) {
    const float4 ch = com[c - off];
    const unsigned int cw = meta[c - off];
    M = M + ch.w;
    cx = cx + ch.w * ch.x; cy = cy + ch.w * ch.y; cz = cz + ch.w * ch.z;
    c = (cw & kLeafBit) ? c + 1 : (int)(cw & kLinkMask);
  }
  return cell_com_from_sums(M, cx, cy, cz, meta, m, l, div_mode, posm, root, off);
}


// The first (or the next) 21 levels of Octree::Add's descent (.h:50-56, 68-75) of one body: the octant digits, three bits a
// level, and where the descent stands.  plain: every size on the way is 2^-100 or more (a root of 2^-58 and more) — the child
// centre is then the plain fp32 o +- 0.5f * Size (descend_level).  There `p >= o` is read off the sign of the fp32 difference
// p - o (a difference of two floats has the sign of the exact one, and +0 where they are equal — with p's own -0 turned into +0
// first); the child centre is o + copysign(h, p - o), and the inverted signs are gathered ten levels to a 32-bit word: four
// instructions per axis and level.
__device__ __forceinline__ unsigned long long descend_word(const float4 &p, float o[3], float &size, bool plain) {
#pragma clang fp contract(off)
  if (!plain) {
    unsigned long long h = 0;
    for (int lev = 0; lev < kLevelsPerKey; ++lev) h = (h << 3) | (unsigned long long)descend_level(p, o, size);
    return h;
  }
  const float px = p.x + 0.0f, py = p.y + 0.0f, pz = p.z + 0.0f;
  float o0 = o[0], o1 = o[1], o2 = o[2], sz = size;
  auto level = [&](unsigned int acc) {
    const float h = 0.5f * sz;
    const unsigned int dx = __float_as_uint(px - o0), dy = __float_as_uint(py - o1), dz = __float_as_uint(pz - o2);
    o0 = o0 + __uint_as_float((__float_as_uint(h) & 0x7FFFFFFFu) | (dx & 0x80000000u));
    o1 = o1 + __uint_as_float((__float_as_uint(h) & 0x7FFFFFFFu) | (dy & 0x80000000u));
    o2 = o2 + __uint_as_float((__float_as_uint(h) & 0x7FFFFFFFu) | (dz & 0x80000000u));
    sz = h;
    acc = __builtin_amdgcn_alignbit(acc, dx, 31);              // (acc << 1) | sign: 1 where p < o
    acc = __builtin_amdgcn_alignbit(acc, dy, 31);
    return __builtin_amdgcn_alignbit(acc, dz, 31);
  };
  unsigned int a = 0, b = 0, c = 0;
#pragma unroll
  for (int lev = 0; lev < 10; ++lev) a = level(a);
#pragma unroll
  for (int lev = 0; lev < 10; ++lev) b = level(b);
  c = level(c);
  o[0] = o0; o[1] = o1; o[2] = o2; size = sz;
  a = ~a & 0x3FFFFFFFu; b = ~b & 0x3FFFFFFFu; c = ~c & 7u;
  return ((unsigned long long)a << 33) | ((unsigned long long)b << 3) | (unsigned long long)c;
}


// A context that owns the slice [off, off + count) of the bodies (range partition over GPUs) builds the WHOLE tree — every
// device the same one, from the replicated positions: each step of the build is the reference's arithmetic in a fixed order — and
// walks only its own bodies (a body's walk, OctreeSearch.cpp:83-86, reads the finished tree and writes that body alone).  own[j]: the
// sorted position of the slice's j-th body in key order (neighbours in space share their windows' loads); null on a context that
// owns all bodies (j is the sorted position itself).  vel / acc / stage hold the slice's bodies, posm all.
struct WalkSlice {
  const unsigned int *own;
  int off;
};
__device__ __forceinline__ int walk_place(const WalkSlice &S, int j) { return S.own != nullptr ? (int)S.own[j] : j; }


// Workgroups are dealt round-robin over the eight XCDs, each with an L2 of its own: with the plain numbering every XCD walks bodies
// from all over the key order.  Renumbered so that workgroups b, b + 8, b + 16, ... — one XCD's — take CONSECUTIVE runs of the key
// order: neighbours in space, whose walks read the same nodes (speed only: which workgroup walks which bodies changes no result;
// a bijection of [0, gridDim.x) for every grid size).  The lane-per-body walk's numbering (frames 0.5-2 % shorter from N = 65536 on,
// profiles/r05_ab_walk_xcd_runs.txt; the window walks of the smaller systems gained nothing from it and keep the plain one).
__device__ __forceinline__ int xcd_run_block() {
#ifdef NBODY_BH_NO_XCD_MAP
  return (int)blockIdx.x;
#else
  const int g = (int)gridDim.x, x = (int)blockIdx.x & 7, q = g >> 3, r = g & 7;   // XCD x holds q + (x < r) workgroups
  return x * q + min(x, r) + ((int)blockIdx.x >> 3);
#endif
}

// The kernels of a larger system's frame all number their workgroups this way (BH_BUILD_XCD_RUNS; A/B builds: make variant
// EXTRA=-DNBODY_BH_BUILD_NO_XCD_RUNS): the x-th eighth of the key order is one XCD's from the key kernel to the walk, so what a
// launch reads of the launch before — slots, sorted keys, shared digits, nodes — was written through the same L2.
#ifdef NBODY_BH_BUILD_NO_XCD_RUNS
#define BH_BUILD_XCD_RUNS 0
#else
#define BH_BUILD_XCD_RUNS 1
#endif

// ---- walks (kernels_bh_walk.hip)
constexpr int kWvT = 512;                  // small systems: eight waves = eight bodies per workgroup next to the LDS tree
constexpr int kWvK = 128;
constexpr int kWvGT = 256;                 // four bodies per workgroup
constexpr int kWvGK = 192;

// ---- the sorts (kernels_bh_sort.hip)
constexpr int kTsT = 1024;                 // threads of a tile-sort workgroup
constexpr int kTs = 4096;                  // bodies per tile
constexpr int kMergeMaxN = 131072;         // tiles + merge up to here (32 tiles), radix above
constexpr int kMergeW = 8;
constexpr int kMergeSmp = 8192;            // sampled keys in LDS (64 KB)
constexpr int kRxT = 256;                  // threads of a pass's workgroup
constexpr int kRxKpt = 16;                 // keys per thread
constexpr int kRxTile = kRxT * kRxKpt;     // 4096 keys per tile
constexpr int kRxBins = 256;
constexpr int kRxPasses = 8;               // 63 key bits
constexpr int kRxHists = 2 * kRxPasses;    // digit histograms kept: the first key word's eight digits, then the second word's (a sort by BOTH words)
constexpr unsigned int kRxAgg = 1u << 30, kRxIncl = 2u << 30, kRxVal = (1u << 30) - 1u;
constexpr int kKhT = 1024;                 // threads: four bodies each
constexpr int kRxSlices = 16;
constexpr int kWarmMu = 224;               // places of the previous order per bucket: a little under 256, so that a bucket's bodies — their number
                                           // wanders by a few dozen — almost always fit a padded bucket of 256 in the second kernel (512 otherwise)
constexpr int kWarmCap = 384;              // slots per bucket
constexpr int kWarmWin = 64;               // boundaries a workgroup keeps in LDS
constexpr int kStatusRetry = 3;            // header word 3: the frame was given up by the warm sort; queue it again with the cold one
constexpr int kStatusUnsorted = 4;         // ... a COLD sort left keys out of order: an internal error, reported (never seen; bh_lcp_scan_kernel's guard)
constexpr int kBsP = 512;                  // the padded bucket at most

struct RadixPass {
  const unsigned long long *kin; const unsigned int *vin;
  unsigned long long *kout; unsigned int *vout;
  const unsigned int *slice_hist;          // [kRxSlices][kRxHists][256]: how many keys carry each value of each digit (bh_hist_reduce_kernel)
  int digit;                               // which histogram: 0 .. 7 the first word's digits, 8 .. 15 the second word's
  unsigned int *desc;                      // [tiles][256] look-back words of this pass, zero before the launch
  int shift, n;
  const int *status;                       // the tree's header word 3: behind a refused frame the key kernel wrote no histograms, and a
                                           // pass must not scatter by counts that belong to other keys
};

// A sorted array's every `stride`-th key in LDS: a lower-bound search does its first steps there and only the last log2(stride)
// on the array itself — each of those is a dependent load from L2.
// s_smp[q] = keys[q * stride] for q < ceil(count / stride); returns #keys in [0, count) that are < h.
__device__ __forceinline__ int lower_bound_sampled(const unsigned long long *__restrict__ keys, int count, const unsigned long long *s_smp,
                                                   int stride_shift, unsigned long long h) {
  const int nsmp = (count + (1 << stride_shift) - 1) >> stride_shift;
  int x = 0, y = nsmp;
  while (x < y) { const int mid = (x + y) >> 1; if (s_smp[mid] < h) x = mid + 1; else y = mid; }
  if (x == 0) return 0;                                        // keys[0] >= h
  // keys[(x - 1) << shift] < h <= keys[x << shift] (or the end): the answer lies in ((x - 1) << shift, x << shift]
  int lo = ((x - 1) << stride_shift) + 1, hi = min(x << stride_shift, count);
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < h) lo = mid + 1; else hi = mid; }
  return lo;
}

// ---- node numbering, node words, ComputeMass (kernels_bh_build.hip)
constexpr int kNodeSmp = 8192;             // sampled sorted keys bh_nodes_kernel keeps in LDS (64 KB; fewer for systems of many workgroups)
constexpr int kScanBlocks = 1024;          // block totals the consumers scan in LDS; a block is kB * bpt bodies (bpt: a power of two)
template <int NT> constexpr int kChunkNodes = NT == 256 ? 1536 : 3584;
constexpr int kTopT = 1024;                   // threads of bh_sweep_top_kernel: one chunk each
constexpr int kChunkSweepMaxN = 1 << 20;      // larger systems sweep with a launch per level (bh_forces)
constexpr int sweep_bpt(int n) { return n <= 98304 ? 1 : 4; }   // chunks of 256 bodies up to N = 98304, of 1024 above (kernels_bh_build.hip has the measurements)
static_assert((kChunkSweepMaxN + 4 * kB - 1) / (4 * kB) <= kTopT, "bh_sweep_top_kernel: one chunk per thread");

// digits two path keys share (0 .. 42; 42: the same path all the way down)
__device__ __forceinline__ int shared_digits(unsigned long long ha, unsigned long long la, unsigned long long hb, unsigned long long lb) {
  const unsigned long long x = ha ^ hb;
  if (x != 0ull) return (__clzll((long long)x) - 1) / 3;
  const unsigned long long y = la ^ lb;
  if (y != 0ull) return kLevelsPerKey + (__clzll((long long)y) - 1) / 3;
  return kMaxLevels;
}

// ---- the kernels (defined in the file named; launched by bh_frame.hip)
// kernels_bh_small.hip
__global__ __launch_bounds__(kSmT) void bh_small_build_kernel(SmallTree T, const float4 *__restrict__ posm, int n, int P,
                                                              int div_mode, int keep_root, float theta);
// kernels_bh_sort.hip
__global__ __launch_bounds__(kB) void bh_keys_kernel(SmallTree T, const float4 *__restrict__ posm, int n,
                                                     const unsigned int *__restrict__ size_bits, unsigned int *__restrict__ next_size,
                                                     float theta, unsigned long long *__restrict__ key_hi,
                                                     unsigned long long *__restrict__ key_lo);
template <int TS>
__global__ __launch_bounds__(kTsT) void bh_tile_sort_kernel(int n, const unsigned long long *__restrict__ key_hi,
                                                            const unsigned long long *__restrict__ key_lo,
                                                            unsigned long long *__restrict__ tile_hi, unsigned int *__restrict__ tile_idx);
__global__ __launch_bounds__(kB) void bh_tile_merge_kernel(int n, int ts, int stride_shift, const unsigned long long *__restrict__ tile_hi,
                                                           const unsigned int *__restrict__ tile_idx,
                                                           const unsigned long long *__restrict__ klo_body,
                                                           unsigned long long *__restrict__ out_hi, unsigned int *__restrict__ out_idx);
__global__ __launch_bounds__(kKhT) void bh_keys_hist_kernel(SmallTree T, const float4 *__restrict__ posm, int n,
                                                            const unsigned int *__restrict__ size_bits,
                                                            unsigned int *__restrict__ next_size, float theta,
                                                            unsigned long long *__restrict__ key_hi,
                                                            unsigned long long *__restrict__ key_lo,
                                                            unsigned int *__restrict__ part_hist, int both);
__global__ __launch_bounds__(kRxBins) void bh_hist_reduce_kernel(const unsigned int *__restrict__ part_hist, int nparts,
                                                                  unsigned int *__restrict__ slice_hist);
__global__ __launch_bounds__(kRxT) void bh_radix_pass_kernel(RadixPass P);
__global__ __launch_bounds__(kB) void bh_gather_words_kernel(int n, const unsigned long long *__restrict__ by_body, const unsigned int *__restrict__ sidx,
                                                             const int *__restrict__ status, unsigned long long *__restrict__ out);
__global__ __launch_bounds__(kB) void bh_ties_gather_kernel(int n, const unsigned long long *__restrict__ khi,
                                                            const unsigned int *__restrict__ sidx,
                                                            const unsigned long long *__restrict__ klo_body,
                                                            unsigned int *__restrict__ tmp_idx, unsigned long long *__restrict__ tmp_lo);
__global__ __launch_bounds__(kB) void bh_ties_place_kernel(int n, const unsigned long long *__restrict__ khi, unsigned int *__restrict__ sidx,
                                                           const unsigned int *__restrict__ tmp_idx,
                                                           const unsigned long long *__restrict__ tmp_lo);
__global__ __launch_bounds__(kB) void bh_keys_bucket_kernel(SmallTree T, const float4 *__restrict__ posm, int n,
                                                            const unsigned int *__restrict__ size_bits,
                                                            unsigned int *__restrict__ next_size, float theta,
                                                            const unsigned long long *__restrict__ bound,   // [2][nb]: first, second key words
                                                            const unsigned int *__restrict__ prev_idx,
                                                            const float4 *__restrict__ prev_pos,
                                                            unsigned long long *__restrict__ slot_lo, unsigned long long *__restrict__ slot_hi,
                                                            unsigned int *__restrict__ slot_idx, unsigned int *__restrict__ gcount, int nb);
__global__ __launch_bounds__(kB) void bh_bound_kernel(const unsigned long long *__restrict__ khi, const unsigned int *__restrict__ sidx,
                                                      const unsigned long long *__restrict__ klo_body, int nb, unsigned long long *__restrict__ bound);
template <int kBsT>
__global__ __launch_bounds__(kBsT) void bh_bucket_sort_kernel(SmallTree T, int n, int nb, const unsigned int *__restrict__ gcount,
                                                              unsigned int *__restrict__ gcount_next,
                                                              const unsigned long long *__restrict__ slot_hi,
                                                              const unsigned int *__restrict__ slot_idx,
                                                              const unsigned long long *__restrict__ slot_lo,
                                                              unsigned long long *__restrict__ out_hi, unsigned int *__restrict__ out_idx,
                                                              unsigned long long *__restrict__ out_lo, unsigned long long *__restrict__ bound);
// kernels_bh_build.hip
__global__ __launch_bounds__(kB) void bh_lcp_scan_kernel(SmallTree T, int n, int bpt, signed char *__restrict__ lcpS,
                                                         int *__restrict__ first_local, int *__restrict__ block_sum);
__global__ __launch_bounds__(kB) void bh_nodes_kernel(SmallTree T, const float4 *__restrict__ posm, int n,
                                                      const int *__restrict__ first_local, const int *__restrict__ block_sum,
                                                      int block_shift, int *__restrict__ first, const signed char *__restrict__ lcpS,
                                                      int smp_shift);
__global__ __launch_bounds__(kB) void bh_sweep_level_kernel(SmallTree T, const float4 *__restrict__ posm, int n,
                                                            const int *__restrict__ first, const signed char *__restrict__ lcpS,
                                                            int l, int div_mode);
__global__ void bh_finish_kernel(SmallTree T, int n, int keep_root);
template <int NT>
__global__ __launch_bounds__(NT) void bh_sweep_chunks_kernel(SmallTree T, const float4 *__restrict__ posm, int n,
                                                             const int *__restrict__ first, const signed char *__restrict__ lcpS,
                                                             int *__restrict__ straddle, int *__restrict__ kids, int nchunks,
                                                             int div_mode);
__global__ __launch_bounds__(kTopT) void bh_sweep_top_kernel(SmallTree T, const float4 *__restrict__ posm, int n,
                                                             const int *__restrict__ straddle, const int *__restrict__ kids,
                                                             int nchunks, int div_mode, int keep_root);
// kernels_bh_walk.hip
__global__ __launch_bounds__(kB) void bh_own_count_kernel(const unsigned int *__restrict__ sidx, int n, unsigned int lo, unsigned int cnt,
                                                          const int *__restrict__ status, unsigned int *__restrict__ blk);
__global__ __launch_bounds__(kB) void bh_own_list_kernel(const unsigned int *__restrict__ sidx, int n, unsigned int lo, unsigned int cnt,
                                                         const int *__restrict__ status, const unsigned int *__restrict__ blk,
                                                         unsigned int *__restrict__ own);
__global__ __launch_bounds__(kWalkT) void bh_walk_compact_kernel(SmallTree T, float4 *__restrict__ posm, float4 *__restrict__ vel,
                                                                 float4 *__restrict__ acc, int n, float theta, double G, float dt,
                                                                 float *__restrict__ stage, WalkSlice S);
__global__ __launch_bounds__(kWvT) void bh_walk_wave_compact_kernel(SmallTree T, float4 *__restrict__ posm, float4 *__restrict__ vel,
                                                                    float4 *__restrict__ acc, int n, double G, float dt,
                                                                    float *__restrict__ stage, WalkSlice S);
__global__ __launch_bounds__(kWvGT) void bh_walk_wave_rows_kernel(SmallTree T, float4 *__restrict__ posm, float4 *__restrict__ vel,
                                                                  float4 *__restrict__ acc, int n, double G, float dt,
                                                                  float *__restrict__ stage, unsigned int *__restrict__ next_size,
                                                                  float4 *__restrict__ pos_sorted, WalkSlice S);
__global__ __launch_bounds__(kWalkT) void bh_walk_rows_kernel(SmallTree T, float4 *__restrict__ posm, float4 *__restrict__ vel,
                                                              float4 *__restrict__ acc, int n, double G, float dt, float *__restrict__ stage,
                                                              unsigned int *__restrict__ next_size, float4 *__restrict__ pos_sorted,
                                                              WalkSlice S);
__global__ __launch_bounds__(kB) void bh_small_leaf_boxes_kernel(SmallTree T, int n, float4 *__restrict__ out);
// kernels_bh_walk.hip (TWO: two steps to a turn of the loop, the two register sets changing places)
template <bool TWO>
__global__ __launch_bounds__(kB) void bh_walk_lane_kernel(SmallTree T, float4 *__restrict__ posm, float4 *__restrict__ vel,
                                                          float4 *__restrict__ acc, int n, double G, float dt, float *__restrict__ stage,
                                                          unsigned int *__restrict__ next_size, float4 *__restrict__ pos_sorted,
                                                          WalkSlice S);

}  // namespace bh
}  // namespace nbody

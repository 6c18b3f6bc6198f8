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
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>

#include <algorithm>

#include "kernels.h"

namespace nbody {

namespace {

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
constexpr int kRowsMaxN = 20480;           // larger systems up to here walk with sixteen lanes per body on the global tree
                                           // (frames: N = 8192 208 us against 275 with a lane per body, 16384 231 / 270, 32768 321 / 272)

struct SmallTree {
  float4 *com;                  // [cap] preorder nodes: centre of mass, total mass
  unsigned int *meta;           // [cap] leaf bit | level << 25 | (internal: the node after the subtree; leaf: the body)
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

// What the structure phases of bh_small_build_kernel leave in LDS for the node phases: sorted first key words and bodies,
// the second key words by body, the first node of every body's group, the shared digits of neighbours (lcpS[i] = lcp(i-1),
// -1 at both ends) and — when the tree fits in LDS — the body that opens each cell (cells numbered in preorder: a cell's
// number is its node number less the bodies before it).
struct SmallScratch {
  const unsigned long long *hi; const unsigned short *idx; const unsigned long long *lo_by_body;
  const int *first; const signed char *lcpS; const unsigned short *cowner;
  const float *root; int *maxl; int *lvl;
};

// The first body behind the cell of level l (> 0) that holds body i (key order).  Most cells hold a handful of bodies: steps of
// 1, 2, 4, ... from body i until one lands outside, then the halving between the last two — about 2 log2(bodies of the cell)
// looks at the keys instead of log2(n).
__device__ __forceinline__ int cell_end(const SmallScratch &sc, int i, int l, int n) {
  const unsigned long long h0 = sc.hi[i], l0 = l > kLevelsPerKey ? sc.lo_by_body[sc.idx[i]] : 0ull;
  auto inside = [&](int j) {
    const unsigned long long hm = sc.hi[j], lm = l > kLevelsPerKey ? sc.lo_by_body[sc.idx[j]] : 0ull;
    return same_prefix(hm, lm, h0, l0, l);
  };
  int x = i, step = 1;                                         // x: a body of the cell
  while (x + step < n && inside(x + step)) { x += step; step <<= 1; }
  int y = min(x + step, n);                                    // the first body behind the cell lies in (x, y]
  while (y - x > 1) { const int mid = (x + y) >> 1; if (inside(mid)) x = mid; else y = mid; }
  return y;
}

// The tree in LDS (nodes <= kSmNodesLds): the cells' words, ComputeMass level by level, and the hand-over to the walk.  The
// leaves' words are written already, and lvl[l] says where level l's list of cells starts (bh_small_build_kernel's scan pass);
// mine: this thread's own bodies t, t + 1024, ...; leaf_of[body]: its leaf — what goes there (CenterOfMass = Position, TotalMass = Mass,
// .h:85-88) is written once the structure data, whose place the CoMs take, is dead.  kids (or null): room for eight 16-bit node
// numbers per cell — every cell's children are then listed once, by walking the top level of its subtree (a chain of dependent
// reads that needs none of the sums: all cells at once), and a level's step is eight loads side by side instead of that chain.
// A cell of fewer than eight children lists node `nodes` for the rest: a node of mass +0 at (+0, +0, +0), whose terms are +0 —
// and adding +0 to a sum that started at +0 (never -0) leaves every bit of it alone: the step needs no conditions.  One wave
// runs a level's step for up to 64 cells and the levels follow one another: what counts is the number of instructions on that path.
__device__ __forceinline__ void small_tree_in_lds(const SmallTree &T, float4 *com, unsigned int *meta, unsigned short *cells,
                                                  unsigned short *kids, const SmallScratch &sc, const float4 mine[4], const unsigned short *leaf_of,
                                                  const float4 *__restrict__ posm, int n, int nodes, int div_mode, int keep_root) {
#pragma clang fp contract(off)
  const int t = threadIdx.x;
  const int ncells = nodes - n;
  // ---- one word per cell, by cell: body i's cells are those of levels lcp(i-1)+1 .. lcp(i), consecutive nodes from first[i] on
  for (int ci = t; ci < ncells; ci += kSmT) {
    const int i = sc.cowner[ci], m0 = sc.first[i];
    const int q = ci - (m0 - i), l = (int)sc.lcpS[i] + 1 + q;
    const int upper = l > 0 ? cell_end(sc, i, l, n) : n;        // first body behind the cell
    meta[m0 + q] = ((unsigned int)l << kLevelShift) | (unsigned int)sc.first[upper];
    cells[atomicAdd(&sc.lvl[l], 1)] = (unsigned short)(m0 + q);   // into its level's list (afterwards lvl[l] is the list's END)
  }
  lds_barrier();                                             // the structure data is dead from here: the CoMs take its place
  BH_CLOCK(5);
  const int maxl = *sc.maxl;
#pragma unroll
  for (int r = 0; r < 4; ++r) { const int i = t + r * kSmT; if (i < n) com[leaf_of[i]] = mine[r]; }
  if (kids != nullptr) {                                       // the children of every cell, listed (node `nodes`: no more)
    if (t == 0) com[nodes] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = t; k < ncells; k += kSmT) {
      const int m = cells[k], end = (int)(meta[m] & kLinkMask);
      unsigned short *kd = kids + 8 * k;
      int c = 0;
      for (int ch = m + 1; ch != end;) {
        const unsigned int cw = meta[ch];
        kd[c++] = (unsigned short)ch;
        ch = (cw & kLeafBit) ? ch + 1 : (int)(cw & kLinkMask);
      }
      for (; c < 8; ++c) kd[c] = (unsigned short)nodes;
    }
  }
  lds_barrier();
  BH_CLOCK(6);
  // ---- ComputeMass (.h:89-95), deepest cells first.  Which cells a level has and which nodes their children are needs none of
  // the sums: a thread fetches its cell of the NEXT level and that cell's children's numbers before this level's barrier, so that
  // a level's step is the eight loads of the sums, the additions and the store.  (Where each level's list ends: lane l of every
  // wave keeps lvl[l] and hands it out by readlane.)
  const int my_end = sc.lvl[t & 63];
  auto list_end = [&](int l) { return l >= 0 ? __builtin_amdgcn_readlane(my_end, l) : 0; };
  auto cell_sums = [&](int m, int k, int l, const uint4 &pk) {
    if (kids != nullptr) {                                     // the children in octant order = preorder, the eight loads side by side
      const unsigned int kd[8] = {pk.x & 0xFFFFu, pk.x >> 16, pk.y & 0xFFFFu, pk.y >> 16, pk.z & 0xFFFFu, pk.z >> 16, pk.w & 0xFFFFu, pk.w >> 16};
      float4 ch[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) ch[c] = com[kd[c]];
      float M = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
#pragma unroll
      for (int c = 0; c < 8; ++c) { M = M + ch[c].w; cx = cx + ch[c].w * ch[c].x; cy = cy + ch[c].w * ch[c].y; cz = cz + ch[c].w * ch[c].z; }
      com[m] = cell_com_from_sums(M, cx, cy, cz, meta, m, l, div_mode, posm, sc.root);
    } else {
      (void)k;
      com[m] = sweep_compact_cell(com, meta, m, meta[m], l, div_mode, posm, sc.root);
    }
  };
  int m_nx = -1;
  uint4 pk_nx = make_uint4(0u, 0u, 0u, 0u);
  auto fetch = [&](int l) {                                    // this thread's (first) cell of level l
    m_nx = -1;
    if (l < 0) return;
    const int k = list_end(l - 1) + t;
    if (k < list_end(l)) { m_nx = cells[k]; if (kids != nullptr) pk_nx = ((const uint4 *)kids)[k]; }
  };
  fetch(maxl);
  for (int l = maxl; l >= 0; --l) {
    const int m = m_nx, lo_ = list_end(l - 1), hi_ = list_end(l);
    const uint4 pk = pk_nx;
    fetch(l - 1);
    if (m >= 0) cell_sums(m, lo_ + t, l, pk);
    for (int k = lo_ + t + kSmT; k < hi_; k += kSmT)            // (a level of more than 1024 cells)
      cell_sums(cells[k], k, l, kids != nullptr ? ((const uint4 *)kids)[k] : make_uint4(0u, 0u, 0u, 0u));
    lds_barrier();
  }
  BH_CLOCK(7);
  // ---- hand the tree to the walk
  for (int m = t; m < nodes; m += kSmT) { T.com[m] = com[m]; T.meta[m] = meta[m]; }
  if (t == 0) {
    if (!keep_root) { const float4 c = com[0]; T.prev_com[0] = c.x; T.prev_com[1] = c.y; T.prev_com[2] = c.z; }   // .cpp:78
    T.hdr[0] = nodes; T.hdr[1] = nodes - n; T.hdr[2] = n >= 2 ? maxl + 1 : 0; T.hdr[4] = T.hdr[4] + 1;
  }
  BH_CLOCK(8);
}

// The same for a tree too large for LDS (deep chains of single-child cells — more than kSmNodesLds nodes from at most 4096
// bodies): it lives in its global arrays from the start, a thread per node, a pass over all nodes per level.  Slow, correct.
__device__ __forceinline__ void small_tree_in_global(const SmallTree &T, const SmallScratch &sc, const float4 *__restrict__ posm, int n,
                                                     int nodes, int div_mode, int keep_root) {
  const int t = threadIdx.x;
  for (int m = t; m < nodes; m += kSmT) {
    int a = 0, b = n - 1;                                      // the body whose group holds node m
    while (a < b) { const int mid = (a + b + 1) >> 1; if (sc.first[mid] <= m) a = mid; else b = mid - 1; }
    const int i = a, q = m - sc.first[i], lp = sc.lcpS[i], ln = sc.lcpS[i + 1];
    const int open = ln > lp ? ln - lp : 0;
    if (q == open) {                                           // the leaf (its level was noted by the scan pass)
      const unsigned int body = sc.idx[i];
      T.meta[m] = kLeafBit | ((unsigned int)((lp > ln ? lp : ln) + 1) << kLevelShift) | body;
      T.com[m] = posm[body];
    } else {
      const int l = lp + 1 + q;
      const int upper = l > 0 ? cell_end(sc, i, l, n) : n;
      T.meta[m] = ((unsigned int)l << kLevelShift) | (unsigned int)sc.first[upper];
    }
  }
  __threadfence();
  __syncthreads();
  const int maxl = *sc.maxl;
  for (int l = maxl; l >= 0; --l) {
    for (int m = t; m < nodes; m += kSmT) {
      const unsigned int w = T.meta[m];
      if ((w & kLeafBit) || (int)((w >> kLevelShift) & 63u) != l) continue;
      T.com[m] = sweep_compact_cell(T.com, T.meta, m, w, l, div_mode, posm, sc.root);
    }
    __threadfence();
    __syncthreads();
  }
  if (t == 0) {
    if (!keep_root) { const float4 c = T.com[0]; T.prev_com[0] = c.x; T.prev_com[1] = c.y; T.prev_com[2] = c.z; }   // .cpp:78
    T.hdr[0] = nodes; T.hdr[1] = nodes - n; T.hdr[2] = n >= 2 ? maxl + 1 : 0; T.hdr[4] = T.hdr[4] + 1;
  }
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

constexpr int kSmSamples = 256;            // the sample sort's splitters: 128 up to 2048 bodies, 256 above
constexpr int kSmBucketMax = 512;          // more bodies than this in one bucket (that many on one 21-level path): merge sort

__global__ __launch_bounds__(kSmT) void bh_small_build_kernel(SmallTree T, const float4 *__restrict__ posm, int n, int P,
                                                              int div_mode, int keep_root, float theta) {
  __shared__ __attribute__((aligned(16))) unsigned char raw[kSmLds];
  __shared__ int s_scan[kSmT / 64];
  __shared__ float s_red[kSmT / 64];
  __shared__ float s_root[4];
  __shared__ int s_lvl[64];                                    // cells per level, then where each level's list starts, then where it ends
  __shared__ int s_maxl, s_err, s_tie, s_bmax;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (T.hdr[3] != 0) return;                                   // an earlier frame of this call was refused: stay there
  BH_CLOCK(0);
  unsigned long long *lo_by_body = (unsigned long long *)(raw + 2 * kSmBuf);   // [n] second key word of body i
  unsigned short *cells = (unsigned short *)(raw + kSmRegionA + kSmNodesLds * 4);   // [cells] the cells by level
  // the sample sort's tables stand where the node words go later
  unsigned long long *smp = (unsigned long long *)(raw + kSmRegionA);          // [samples] sampled first key words
  unsigned long long *spl = smp + kSmSamples;                                  // [samples] ... sorted: the splitters
  int *bcnt = (int *)(spl + kSmSamples);                                       // [samples + 2] bodies per bucket, then where each bucket starts
  static_assert(2 * kSmSamples * 8 + (kSmSamples + 2) * 4 <= kSmNodesLds * 4, "the sample sort's tables fit");

  // ---- ComputeCubeSize (.cpp:47-56) and the root (.cpp:77-79)
  float mx = 0.0f;
  float4 mine[kSmBodies / kSmT];                               // this thread's bodies: t, t + 1024, ...
#pragma unroll
  for (int r = 0; r < kSmBodies / kSmT; ++r) {
    const int i = t + r * kSmT;
    mine[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n) { mine[r] = posm[i]; mx = fmaxf(mx, fmaxf(fmaxf(fabsf(mine[r].x), fabsf(mine[r].y)), fabsf(mine[r].z))); }
  }
  // the sample: every (n / samples)-th body of the PREVIOUS frame's key order — bodies move little in a frame, so these stand close
  // to the quantiles of this frame's order too and the buckets come out even (any bodies would do: the first frame takes
  // every (n / samples)-th body as numbered)
  const int smp_cap = n > 2048 ? kSmSamples : kSmSamples / 2, nsmp = min(smp_cap, n);
  int sample_body = 0;
  if (t < nsmp) { sample_body = (int)(((long long)t * n + n / 2) / nsmp); sample_body = min(sample_body, n - 1);
                  if (T.hdr[4] > 0) sample_body = min((int)T.sidx[sample_body], n - 1); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if (lane == 0) s_red[wave] = mx;
  if (t == 0) { s_maxl = -1; s_err = 0; }
  lds_barrier();
  if (t == 0) {
    float m = s_red[0];
    for (int w = 1; w < kSmT / 64; ++w) m = fmaxf(m, s_red[w]);
    s_root[0] = T.prev_com[0]; s_root[1] = T.prev_com[1]; s_root[2] = T.prev_com[2]; s_root[3] = m;
    T.root[0] = s_root[0]; T.root[1] = s_root[1]; T.root[2] = s_root[2]; T.root[3] = m;
    T.hdr[7] = (int)__float_as_uint(m);                        // Size travels with the verdict (nbody_tick)
  }
  lds_barrier();
  if (t >= kSmT - 64 && t - (kSmT - 64) <= kMaxLevels) {        // the opening rule per level, as a threshold on d2 (the
    const int l = t - (kSmT - 64);                             // last wave: it owns the fewest bodies)
    float sz = s_root[3];
    for (int q = 0; q < l; ++q) sz = (float)(0.5 * (double)sz);   // .h:74
    T.thr[l] = accept_threshold(sz, theta);
  }
  BH_CLOCK(1);
  // ---- path keys, sort, shared digits.  Almost always the first key word (21 levels) decides the order and nobody needs
  // the second: the first go computes 21 levels per body and sorts on them alone; only when two neighbours turn out to agree in
  // the whole word (bodies closer than Size / 2^21) is it all done again with both words.
  // The sort is a sample sort: 128 bodies' first key words, put in order (every sample counts the samples before it), split the
  // key space where the bodies are — however clustered; a body finds its bucket by seven halvings among the splitters,
  // the buckets are counted (LDS atomics, whose answers also number a bucket's bodies), a scan says where each bucket starts,
  // and every body finds its place among its bucket's bodies by comparing with each of them — in which order the atomics
  // answered does not matter.  Only a bucket of more than kSmBucketMax bodies sends the system to the merge sort (below).
  const unsigned long long *hi = nullptr;
  const unsigned short *idx = nullptr;
  int *first = nullptr;
  signed char *lcpS = nullptr;
  static_assert(4 * (kSmBodies + 4) + (kSmBodies + 16) + 2 * kSmNodesLds <= kSmBuf, "scan, lcp and the cells' owners fit in a sort buffer");
  const bool plain = s_root[3] >= 0x1p-58f;
  unsigned long long *hiA = (unsigned long long *)raw, *hiB = (unsigned long long *)(raw + kSmBuf);
  unsigned short *idxA = (unsigned short *)(raw + kSmBodies * 8), *idxB = (unsigned short *)(raw + kSmBuf + kSmBodies * 8);
  for (int both = 0; both < 2; ++both) {
    unsigned long long kh[kSmBodies / kSmT], kl[kSmBodies / kSmT];
#pragma unroll
    for (int r = 0; r < kSmBodies / kSmT; ++r) {
      const int i = t + r * kSmT;
      kh[r] = ~0ull; kl[r] = 0ull;
      if (i < n) {
        float o[3] = {s_root[0], s_root[1], s_root[2]};
        float size = s_root[3];
        kh[r] = descend_word(mine[r], o, size, plain);
        if (both) kl[r] = descend_word(mine[r], o, size, plain);
        lo_by_body[i] = kl[r];
        hiB[i] = kh[r];                                        // by body, for the sample (the sorted keys go here in the end)
      }
    }
    if (t <= kSmSamples + 1) bcnt[t] = 0;
    if (t == 0) { s_tie = 0; s_bmax = 0; }
    lds_barrier();
    if (!both) BH_CLOCK(2);
    if (t < nsmp) smp[t] = hiB[sample_body];
    lds_barrier();
    {                                                          // a sample's place: the samples before it (equal ones in their own order).
      // 1024 / samples neighbouring lanes share a sample, each looks at its part of the samples, a few DPP adds put it together
      const int parts = kSmT / smp_cap, per = smp_cap / parts;
      const int j = t / parts, part = t % parts;
      const unsigned long long mykey = smp[min(j, nsmp - 1)];
      int before = 0;
      for (int u = 0; u < per; ++u) {
        const int k = part * per + u;
        const unsigned long long sk = smp[min(k, nsmp - 1)];
        before += (k < nsmp && (sk < mykey || (sk == mykey && k < j))) ? 1 : 0;
      }
      before += __shfl_xor(before, 1, 64);
      before += __shfl_xor(before, 2, 64);
      if (parts == 8) before += __shfl_xor(before, 4, 64);
      if (part == 0 && j < nsmp) spl[before] = mykey;
    }
    lds_barrier();
    if (!both) BH_CLOCK(9);
    unsigned int slot[kSmBodies / kSmT], bucket[kSmBodies / kSmT];
#pragma unroll
    for (int r = 0; r < kSmBodies / kSmT; ++r) {                // bucket = splitters below the key (equal first words share a bucket)
      const int i = t + r * kSmT;
      slot[r] = 0u; bucket[r] = 0u;
      if (i < n) {
        int x = 0, y = nsmp;
        while (x < y) { const int mid = (x + y) >> 1; if (spl[mid] < kh[r]) x = mid + 1; else y = mid; }
        bucket[r] = (unsigned int)x;
        slot[r] = (unsigned int)atomicAdd(&bcnt[x], 1);
      }
    }
    lds_barrier();
    if (wave < 5) {                                            // where each bucket starts: exclusive scan of the counts (samples + 1 of them)
      const int c = t <= kSmSamples ? bcnt[t] : 0;
      int incl = c, big = c;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) big = max(big, __shfl_xor(big, off, 64));
      if (lane == 63) s_scan[wave] = incl;
      if (lane == 0) atomicMax(&s_bmax, big);
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_s_barrier();                            // (all sixteen waves meet here: see the else branch)
      int run = incl - c;
      for (int w = 0; w < wave; ++w) run += s_scan[w];
      if (t <= kSmSamples + 1) bcnt[t] = run;                   // (from bcnt[samples + 1] on: n)
    } else {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_s_barrier();
    }
    lds_barrier();
    if (!both) BH_CLOCK(10);
    if (s_bmax <= kSmBucketMax) {
      // the bodies by bucket (in the order the atomics answered), then every body's place among its bucket's bodies: those
      // with a smaller (first word, second word, body)
#pragma unroll
      for (int r = 0; r < kSmBodies / kSmT; ++r) {
        const int i = t + r * kSmT;
        if (i < n) { const int pos = bcnt[bucket[r]] + (int)slot[r]; hiA[pos] = kh[r]; idxA[pos] = (unsigned short)i; }
      }
      lds_barrier();
      if (!both) BH_CLOCK(11);
      int place[kSmBodies / kSmT];
#pragma unroll
      for (int r = 0; r < kSmBodies / kSmT; ++r) {
        const int i = t + r * kSmT;
        place[r] = 0;
        if (i < n) {
          const int a = bcnt[bucket[r]], b = bcnt[bucket[r] + 1];
          // the bodies of its bucket that are not above it: one of them is the body itself, so that many less one stand before
          // it — unless two bodies agree in the whole first word (rare): they get the same place, and the check below sees it
          int notabove = 0;
          if (!both) {
            for (int k = a; k < b; k += 8) {                     // eight loads in flight: the loop is a chain of LDS round trips otherwise
              unsigned long long hk[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) hk[u] = hiA[min(k + u, b - 1)];
#pragma unroll
              for (int u = 0; u < 8; ++u) notabove += (k + u < b && hk[u] <= kh[r]) ? 1 : 0;
            }
            notabove -= 1;
          } else {                                             // second word, then body, where the first words agree
            for (int k = a; k < b; ++k) {
              const unsigned long long hk = hiA[k];
              bool less = hk < kh[r];
              if (hk == kh[r]) {
                const int ik = idxA[k];
                if (ik != i) { const unsigned long long lk = lo_by_body[ik]; less = lk < kl[r] || (lk == kl[r] && ik < i); }
              }
              notabove += less ? 1 : 0;
            }
          }
          place[r] = a + notabove;
          hiB[place[r]] = kh[r]; idxB[place[r]] = (unsigned short)i;
        }
      }
      lds_barrier();
      if (!both) {                                             // two bodies with one place: the second go will tell them apart
#pragma unroll
        for (int r = 0; r < kSmBodies / kSmT; ++r) { const int i = t + r * kSmT; if (i < n && idxB[place[r]] != (unsigned short)i) s_tie = 1; }
      }
      hi = hiB; idx = idxB;
      first = (int *)raw;                                        // [n + 1], in the buffer the sort left behind
    } else {
      // ---- merge sort by rank: runs of L become runs of 2L; every element finds its place by a binary search in the partner
      // run (left run: partner elements strictly before it; right run: partner elements not after it — a stable merge).
      // log2(P) rounds, buffers ping-pong; the rounds whose pairs of runs lie inside a wave's own 64 elements need only that
      // wave's order, the others a barrier.  Ties in the first key word look the second one up by body (second go only).
#pragma unroll
      for (int r = 0; r < kSmBodies / kSmT; ++r) {
        const int i = t + r * kSmT;
        if (i < P) { hiA[i] = kh[r]; idxA[i] = (unsigned short)(i < n ? i : 0xFFFF); }
      }
      lds_barrier();
      int cur = 0;
      for (int L = 1, lg = 0; L < P; L <<= 1, ++lg, cur ^= 1) {
        const unsigned long long *shi = (const unsigned long long *)(raw + cur * kSmBuf);
        const unsigned short *sidx = (const unsigned short *)(raw + cur * kSmBuf + kSmBodies * 8);
        unsigned long long *dhi = (unsigned long long *)(raw + (cur ^ 1) * kSmBuf);
        unsigned short *didx = (unsigned short *)(raw + (cur ^ 1) * kSmBuf + kSmBodies * 8);
        for (int e = t; e < P; e += kSmT) {
          const int run = e >> lg, pos = e & (L - 1);
          const bool left = (run & 1) == 0;
          const int pbase = (run ^ 1) * L;
          const unsigned long long h = shi[e];
          const unsigned short b = sidx[e];
          int x = 0, y = L;
          while (x < y) {
            const int mid = (x + y) >> 1;
            const unsigned long long hp = shi[pbase + mid];
            bool before = hp < h;                                // partner element sorts before mine?
            if (hp == h) {
              const unsigned short bp = sidx[pbase + mid];
              const unsigned long long lp = bp == 0xFFFF ? ~0ull : lo_by_body[bp], lm = b == 0xFFFF ? ~0ull : lo_by_body[b];
              before = left ? lp < lm : lp <= lm;
            }
            if (before) x = mid + 1; else y = mid;
          }
          const int dest = (run & ~1) * L + pos + x;
          dhi[dest] = h; didx[dest] = b;
        }
        if (2 * L <= 64 && 4 * L <= 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
        else lds_barrier();
      }
      hi = (const unsigned long long *)(raw + cur * kSmBuf);
      idx = (const unsigned short *)(raw + cur * kSmBuf + kSmBodies * 8);
      first = (int *)(raw + (cur ^ 1) * kSmBuf);                 // [n + 1], in the buffer the sort left behind
    }
    lcpS = (signed char *)(first + kSmBodies + 4);             // [n + 1]
    if (!both) BH_CLOCK(3);
    // ---- shared digits of neighbours, the keys and the draw order for later (leaf boxes, DrawOctreeBoxes' order)
    for (int i = t; i <= n; i += kSmT) {
      int v = -1;
      unsigned long long li = 0ull;
      if (i < n) li = lo_by_body[idx[i]];
      if (i > 0 && i < n) {
        const unsigned long long x = hi[i - 1] ^ hi[i];
        if (x != 0ull) v = (__clzll((long long)x) - 1) / 3;
        else if (!both) { v = kLevelsPerKey; s_tie = 1; }      // agree in the whole first word: the second go will tell
        else {
          const unsigned long long y = lo_by_body[idx[i - 1]] ^ li;
          if (y != 0ull) v = kLevelsPerKey + (__clzll((long long)y) - 1) / 3;
          else { v = kMaxLevels; s_err = 1; }                  // same path for 42 levels: the reference would recurse on
        }
      }
      lcpS[i] = (signed char)v;
      if (i < n) { T.khi[i] = hi[i]; T.klo[i] = li; T.sidx[i] = idx[i]; }
    }
    if (t < 64) s_lvl[t] = 0;
    lds_barrier();
    if (both || s_tie == 0) break;
    lds_barrier();                                           // everybody has seen the tie flag before the next go clears it
  }
  lds_barrier();
#ifdef NBODY_BH_PHASE_CLOCKS
  if (t == 0) T.clocks[14] = s_bmax;                           // the fullest bucket of the sample sort
#endif
  if (s_err != 0) { if (t == 0) T.hdr[3] = 1; return; }
  // ---- number the nodes: exclusive scan of (cells opened at body i) + 1, four bodies per thread.  The same pass writes the
  // leaves' words, counts the cells by level and notes which body opens each cell (numbered node - bodies before it).
  unsigned short *cowner = (unsigned short *)(lcpS + kSmBodies + 16);   // [cells]
  unsigned int *meta = (unsigned int *)(raw + kSmRegionA);
  unsigned short *leaf_of = cells + (kSmNodesLds - n);        // [n] every body's leaf (at most kSmNodesLds - n cells are listed in front)
  int nodes;
  {
    int c[4], lp[4], ln[4], sum = 0, deep = -1;
    unsigned int body[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 4 * t + q;
      c[q] = 0; lp[q] = 0; ln[q] = -1; body[q] = 0u;
      if (i < n) {
        lp[q] = (int)lcpS[i]; ln[q] = (int)lcpS[i + 1]; body[q] = idx[i];
        const int d = ln[q] - lp[q];
        c[q] = (d > 0 ? d : 0) + 1;
        deep = max(deep, ln[q]);
      }
      sum += c[q];
    }
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(incl, off, 64); if (lane >= off) incl += v; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) deep = max(deep, __shfl_xor(deep, off, 64));
    if (lane == 63) s_scan[wave] = incl;
    if (lane == 0 && deep >= 0) atomicMax(&s_maxl, deep);
    lds_barrier();
    int run = incl - sum, total = 0;
    for (int w = 0; w < kSmT / 64; ++w) { const int v = s_scan[w]; if (w < wave) run += v; total += v; }
    nodes = total;
    if (nodes > T.cap) { if (t == 0) T.hdr[3] = 2; return; }
    const bool in_lds = nodes <= kSmNodesLds;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 4 * t + q;
      if (i < n) {
        first[i] = run;
        const int open = c[q] - 1;
        const int level = (lp[q] > ln[q] ? lp[q] : ln[q]) + 1;  // the leaf: one level below the deepest cell the body shares
        T.leaf_level[i] = (unsigned char)level;
        if (in_lds) {
          for (int k = 0; k < open; ++k) { cowner[run - i + k] = (unsigned short)i; atomicAdd(&s_lvl[lp[q] + 1 + k], 1); }
          leaf_of[body[q]] = (unsigned short)(run + open);
          meta[run + open] = kLeafBit | ((unsigned int)level << kLevelShift) | body[q];
        }
      }
      run += c[q];
    }
    if (t == kSmT - 1) first[n] = run;
    lds_barrier();
    if (t < 64) {                                              // where each level's list of cells starts
      const int cnt = s_lvl[t];
      int inc = cnt;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) { const int v = __shfl_up(inc, off, 64); if (lane >= off) inc += v; }
      s_lvl[t] = inc - cnt;
    }
    lds_barrier();
  }
  BH_CLOCK(4);
  const SmallScratch sc = {hi, idx, lo_by_body, first, lcpS, cowner, s_root, &s_maxl, s_lvl};
  if (nodes <= kSmNodesLds) {   // the tree in LDS: CoMs over the sort's space once the structure is known, words and cell lists behind,
    const int ncells = nodes - n;                                // the children's lists behind the CoMs where there is room
    unsigned short *kids = (nodes + 1 + ncells) * 16 <= kSmRegionA ? (unsigned short *)(raw + (nodes + 1) * 16) : nullptr;
    small_tree_in_lds(T, (float4 *)raw, meta, cells, kids, sc, mine, leaf_of, posm, n, nodes, div_mode, keep_root);
  } else {
    small_tree_in_global(T, sc, posm, n, nodes, div_mode, keep_root);
  }
}

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

constexpr int kWvT = 512;                  // small systems: eight waves = eight bodies per workgroup next to the LDS tree
constexpr int kWvK = 128;
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
  hand_verdict(T);
  if (T.hdr[3] != 0) return;                                   // the frame was refused: nothing moves
  BH_WG_STAMP(0);
  const int t = threadIdx.x;
  const int nodes = T.hdr[0];
  const bool in_lds = nodes <= kSmNodesLds;
  if (t <= kMaxLevels) s_thr[t] = T.thr[t];
  __syncthreads();
  if (in_lds) {
    for (int m0 = t; m0 < nodes; m0 += 4 * kWvT) {               // four nodes' loads in flight per thread: the fill is round trips to L2
      float4 c[4];
      unsigned int w[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int m = min(m0 + u * kWvT, nodes - 1); c[u] = T.com[m]; w[u] = T.meta[m]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int m = m0 + u * kWvT;
        if (m < nodes) {
          const bool leaf = (w[u] & kLeafBit) != 0u;
          s_a[m] = make_float4(c[u].x, c[u].y, c[u].z, leaf ? 0.0f : s_thr[(w[u] >> kLevelShift) & 63u]);
          s_m[m] = c[u].w;
          s_past[m] = (unsigned short)(leaf ? m + 1 : (int)(w[u] & kLinkMask));
        }
      }
    }
    __syncthreads();
  }
  const int wave = t >> 6, lane = t & 63;
  const int k = blockIdx.x * kWaves + wave;
  const bool valid = k < n;                                    // (n: the bodies this context walks — all, or its slice's)
  const int place = valid ? walk_place(S, k) : 0;               // the body's sorted position
  const unsigned int body = valid ? T.sidx[place] : 0u;
  const float4 p = posm[body];
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
constexpr int kWvGT = 256;                 // four bodies per workgroup
constexpr int kWvGK = 192;
__global__ __launch_bounds__(kWvGT) void bh_walk_wave_rows_kernel(SmallTree T, float4 *__restrict__ posm, float4 *__restrict__ vel,
                                                                  float4 *__restrict__ acc, int n, double G, float dt,
                                                                  float *__restrict__ stage, unsigned int *__restrict__ next_size,
                                                                  float4 *__restrict__ pos_sorted, WalkSlice S) {
  constexpr int kWaves = kWvGT / 64;
  __shared__ float s_thr[kMaxLevels + 2];
  __shared__ unsigned int s_list[kWaves][kWvGK];
  __shared__ __attribute__((aligned(16))) float s_term[kWaves][3 * kWvGK];
  hand_verdict(T);
  if (T.hdr[3] != 0) return;
  const int t = threadIdx.x;
  if (t <= kMaxLevels) s_thr[t] = T.thr[t];
  __syncthreads();
  const int nodes = T.hdr[0];
  const int wave = t >> 6, lane = t & 63;
  const int k = blockIdx.x * kWaves + wave;
  const bool valid = k < n;                                    // (n: the bodies this context walks — all, or its slice's)
  const int place = valid ? walk_place(S, k) : 0;               // the body's sorted position
  const unsigned int body = valid ? T.sidx[place] : 0u;
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
  hand_verdict(T);
  if (T.hdr[3] != 0) return;                                   // the frame was refused: nothing moves
  const int t = threadIdx.x;
  if (t <= kMaxLevels) s_thr[t] = T.thr[t];
  __syncthreads();
  const int nodes = T.hdr[0];
  const int group = t / kWalkG, g = t % kWalkG;
  const int k = blockIdx.x * kGroups + group;
  const bool valid = k < n;                                    // (n: the bodies this context walks — all, or its slice's)
  const int place = valid ? walk_place(S, k) : 0;               // the body's sorted position
  const unsigned int body = valid ? T.sidx[place] : 0u;
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


// ---------------------------------------------------------------------------------------------------------------------
// Larger systems (n > kSmBodies): the same compact preorder tree, built by the whole chip — path keys and their order (a frame
// that follows a frame: bh_keys_bucket_kernel + bh_bucket_sort_kernel, from the previous frame's order; a first frame: the cold
// sorts below; bodies that agree in the whole first key word are ordered by the second), the neighbours' shared digits and
// the exclusive scan that numbers the nodes (one launch), one pass that writes the node words and the leaves, ComputeMass (two
// launches up to kChunkSweepMaxN bodies, a launch per level above: body i opens the cell of level l iff
// lcp(i-1) < l <= lcp(i)), and a walk — a wave per body on the global arrays up to kWaveMaxN bodies, sixteen lanes per body up
// to kRowsMaxN, one lane per body above (enough bodies to hide the loads: the windows buy latency, not throughput) — with the
// Tick's update and the next frame's Size in it.  Up to kChunkSweepMaxN bodies nothing waits for the host: seven launches a frame.
//
// ---------------------------------------------------------------------------------------------------------------------
// The larger systems' own sort of the path keys (Octree::Add's order, OctreeSearch.h:60-81): 8-byte first key word + 4-byte
// body index.  Two forms, both hand-written for gfx950 — no library kernel on the path:
//   tiles + merge   up to kMergeMaxN bodies, TWO launches: every workgroup computes the keys of kTs bodies and sorts them in
//                   LDS (the small systems' merge by rank); then every element finds its place among ALL tiles by one
//                   binary search per other tile, several searches in flight.  Bodies that agree in the whole first key
//                   word are ordered by the second one on the spot (it is looked up only then).
//   radix           above: least-significant-digit radix sort, 8 bits a pass, ONE launch per pass ("onesweep"): a tile's
//                   keys are ranked inside the workgroup (per-wave match by ballots, LDS counters), where the tile's keys of
//                   a digit go is found by decoupled look-back over the earlier tiles' counters, and the keys leave through
//                   LDS in bin order.  The global digit histograms come from the key kernel (partial histograms per
//                   workgroup, no global atomics: contended device-scope atomics cost ~0.8 us each here), ties in the whole
//                   first key word are put right afterwards (bh_ties_gather_kernel, bh_ties_place_kernel).
constexpr int kTsT = 1024;                 // threads of a tile-sort workgroup
constexpr int kTs = 4096;                  // bodies per tile
constexpr int kMergeMaxN = 131072;         // tiles + merge up to here (32 tiles), radix above

// What the first workgroup of a frame's first kernel sets up: the root (centre = the previous tree's CoM, half-width = Size as
// the bounds kernel left it: ComputeCubeSize), the header words this frame counts in, the 43 acceptance thresholds of the walk.
__device__ __forceinline__ void bh_frame_setup(const SmallTree &T, const float o[3], float sz, float theta, int nthreads,
                                               unsigned int *__restrict__ next_size) {
  const int t = threadIdx.x;
  if (t < kSizeSlots) next_size[t] = 0u;                      // where this frame's walk leaves the next frame's Size
  if (t == 0) {
    T.root[0] = o[0]; T.root[1] = o[1]; T.root[2] = o[2]; T.root[3] = sz;
    T.hdr[6] = 0;                                             // no two neighbours agree in the whole first key word yet
    T.hdr[7] = (int)__float_as_uint(sz);                      // Size travels with the verdict (nbody_tick)
  }
  if (t < 128) T.lvl[t] = 0;                                  // (bh_sweep_chunks_kernel counts there)
  for (int q = t; q < kDeepSlots; q += nthreads) T.hdr[kHdrDeep + q] = -1;   // deepest level with a cell of >= 2 bodies (bh_lcp_scan_kernel)
  if (t <= kMaxLevels) {
    float s_l = sz;
    for (int q = 0; q < t; ++q) s_l = (float)(0.5 * (double)s_l);   // .h:74
    T.thr[t] = accept_threshold(s_l, theta);
  }
}

// Path keys of all bodies, one lane each (both words, body order); the first workgroup also sets the frame up.
__global__ __launch_bounds__(kB) void bh_keys_kernel(SmallTree T, const float4 *__restrict__ posm, int n,
                                                     const unsigned int *__restrict__ size_bits, unsigned int *__restrict__ next_size,
                                                     float theta, unsigned long long *__restrict__ key_hi,
                                                     unsigned long long *__restrict__ key_lo) {
  // a frame before this one was refused: nothing of this one happens — not the root, not Size in the header, not the next frame's
  // Size words (uniform: the cold sorts never raise the word themselves; it was set before this launch)
  if (T.hdr[3] != 0) return;
  const float sz = frame_size(size_bits);
  float o[3] = {T.prev_com[0], T.prev_com[1], T.prev_com[2]};
  if (blockIdx.x == 0) bh_frame_setup(T, o, sz, theta, kB, next_size);
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  const float4 p = posm[i];
  float size = sz;
  const bool plain = sz >= 0x1p-58f;
  const unsigned long long hi = descend_word(p, o, size, plain), lo = descend_word(p, o, size, plain);
  key_hi[i] = hi; key_lo[i] = lo;
}

// The keys of TS consecutive bodies, sorted in LDS by (first word, second word on a tie, position): tile_hi / tile_idx hold the
// tiles one after the other.  TS follows the size of the system (tile_size): small tiles mean more workgroups at work here and
// more tiles for the merge to look through.
template <int TS>
__global__ __launch_bounds__(kTsT) void bh_tile_sort_kernel(int n, const unsigned long long *__restrict__ key_hi,
                                                            const unsigned long long *__restrict__ key_lo,
                                                            unsigned long long *__restrict__ tile_hi, unsigned int *__restrict__ tile_idx) {
  constexpr int kBuf = TS * (8 + 2);                           // one sort buffer: hi[TS], idx[TS] (local index)
  constexpr int kPer = TS / kTsT > 0 ? TS / kTsT : 1;
  __shared__ __attribute__((aligned(16))) unsigned char raw[2 * kBuf + TS * 8];
  unsigned long long *lo_l = (unsigned long long *)(raw + 2 * kBuf);   // second key word by local index
  const int t = threadIdx.x;
  const int base = blockIdx.x * TS, cnt = min(TS, n - base);
  int P = 64;
  while (P < cnt) P <<= 1;
  {
    unsigned long long *hi0 = (unsigned long long *)raw;
    unsigned short *idx0 = (unsigned short *)(raw + TS * 8);
#pragma unroll
    for (int r = 0; r < kPer; ++r) {
      const int i = t + r * kTsT;
      if (i < P) {
        const bool in = i < cnt;
        lo_l[i] = in ? key_lo[base + i] : ~0ull;
        hi0[i] = in ? key_hi[base + i] : ~0ull; idx0[i] = (unsigned short)i;
      }
    }
  }
  __syncthreads();
  // merge sort by rank (bh_small_build_kernel): runs of L become runs of 2L, every element finds its place by a binary search
  // in the partner run — left run: partner elements strictly before it; right run: partner elements not after it (stable)
  int cur = 0;
  for (int L = 1, lg = 0; L < P; L <<= 1, ++lg, cur ^= 1) {
    const unsigned long long *shi = (const unsigned long long *)(raw + cur * kBuf);
    const unsigned short *sidx = (const unsigned short *)(raw + cur * kBuf + TS * 8);
    unsigned long long *dhi = (unsigned long long *)(raw + (cur ^ 1) * kBuf);
    unsigned short *didx = (unsigned short *)(raw + (cur ^ 1) * kBuf + TS * 8);
    for (int e = t; e < P; e += kTsT) {
      const int run = e >> lg, pos = e & (L - 1);
      const bool left = (run & 1) == 0;
      const int pbase = (run ^ 1) * L;
      const unsigned long long h = shi[e];
      const unsigned short b = sidx[e];
      int x = 0, y = L;
      while (x < y) {
        const int mid = (x + y) >> 1;
        const unsigned long long hp = shi[pbase + mid];
        bool before = hp < h;
        if (hp == h) { const unsigned long long lp = lo_l[sidx[pbase + mid]], lm = lo_l[b]; before = left ? lp < lm : lp <= lm; }
        if (before) x = mid + 1; else y = mid;
      }
      const int dest = (run & ~1) * L + pos + x;
      dhi[dest] = h; didx[dest] = b;
    }
    if (2 * L <= 64 && 4 * L <= 64) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
  }
  const unsigned long long *hi = (const unsigned long long *)(raw + cur * kBuf);
  const unsigned short *idx = (const unsigned short *)(raw + cur * kBuf + TS * 8);
  for (int e = t; e < cnt; e += kTsT) { tile_hi[base + e] = hi[e]; tile_idx[base + e] = (unsigned int)(base + idx[e]); }
}

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

// Every element's place among all tiles: its place in its own tile + for every other tile the number of that tile's elements
// that sort before it (an earlier tile's equal keys come first: the order is (key, tile, place) — stable).  The tiles' sampled keys
// sit in LDS (lower_bound_sampled); kMergeW tiles are searched side by side.  Elements that agree with mine in the whole first
// key word (bodies closer than Size / 2^21: rare) are counted by their second words, looked up only then.
constexpr int kMergeW = 8;
constexpr int kMergeSmp = 8192;            // sampled keys in LDS (64 KB)
__global__ __launch_bounds__(kB) void bh_tile_merge_kernel(int n, int ts, int stride_shift, const unsigned long long *__restrict__ tile_hi,
                                                           const unsigned int *__restrict__ tile_idx,
                                                           const unsigned long long *__restrict__ klo_body,
                                                           unsigned long long *__restrict__ out_hi, unsigned int *__restrict__ out_idx) {
  __shared__ unsigned long long s_smp[kMergeSmp];
  const int ntiles = (n + ts - 1) / ts, per_tile = ts >> stride_shift;   // samples per tile (the tiles' sample ranges do not mix)
  for (int q0 = threadIdx.x; q0 < ntiles * per_tile; q0 += 8 * kB) {      // eight loads in flight per thread
    unsigned long long v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + u * kB, e = q << stride_shift;        // (per_tile * 2^stride_shift = ts: sample q is element q << stride_shift)
      v[u] = (q < ntiles * per_tile && e < n) ? tile_hi[e] : ~0ull;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { const int q = q0 + u * kB; if (q < ntiles * per_tile) s_smp[q] = v[u]; }
  }
  __syncthreads();
  const int e = blockIdx.x * kB + threadIdx.x;
  if (e >= n) return;
  const int a = e / ts;
  int coarse_steps = 1;                                        // halvings that close a range of per_tile samples
  while ((1 << (coarse_steps - 1)) < per_tile) ++coarse_steps;
  const unsigned long long h = tile_hi[e];
  const unsigned int b = tile_idx[e];
  int rank = e - a * ts;
  for (int tb = 0; tb < ntiles; tb += kMergeW) {
    int lo[kMergeW], hi[kMergeW], cnt[kMergeW];
    // first steps on the samples in LDS: tile's keys below mine lie in (lo, hi]
#pragma unroll
    for (int q = 0; q < kMergeW; ++q) {
      const int tile = tb + q;
      cnt[q] = (tile < ntiles && tile != a) ? min(ts, n - tile * ts) : 0;
      lo[q] = 0; hi[q] = (cnt[q] + (1 << stride_shift) - 1) >> stride_shift;   // for now: the range of samples
    }
    for (int step = 0; step < coarse_steps; ++step) {            // a fixed number of halvings, the tiles side by side (LDS round trips)
#pragma unroll
      for (int q = 0; q < kMergeW; ++q) {
        const int mid = (lo[q] + hi[q]) >> 1;
        const bool open = lo[q] < hi[q];
        const unsigned long long sv = s_smp[open ? (tb + q) * per_tile + mid : 0];
        if (open) { if (sv < h) lo[q] = mid + 1; else hi[q] = mid; }
      }
    }
#pragma unroll
    for (int q = 0; q < kMergeW; ++q) {
      const int x = lo[q];
      lo[q] = x == 0 ? 0 : ((x - 1) << stride_shift) + 1;
      hi[q] = x == 0 ? 0 : min(x << stride_shift, cnt[q]);
    }
    // the last log2(stride) steps on the tiles themselves, the tiles side by side (each step is a dependent load from L2)
    for (int step = 0; step < stride_shift; ++step) {
#pragma unroll
      for (int q = 0; q < kMergeW; ++q)
        if (lo[q] < hi[q]) {
          const int mid = (lo[q] + hi[q]) >> 1;
          if (tile_hi[(size_t)(tb + q) * ts + mid] < h) lo[q] = mid + 1; else hi[q] = mid;
        }
    }
    unsigned long long at[kMergeW];
#pragma unroll
    for (int q = 0; q < kMergeW; ++q) at[q] = lo[q] < cnt[q] ? tile_hi[(size_t)(tb + q) * ts + lo[q]] : 0ull;
#pragma unroll
    for (int q = 0; q < kMergeW; ++q) {
      int x = lo[q];
      if (x < cnt[q] && at[q] == h) {                           // the whole first key word agrees: the second words decide
        const int tile = tb + q;
        const unsigned long long lm = klo_body[b];
        while (x < cnt[q] && tile_hi[(size_t)tile * ts + x] == h) {
          const unsigned long long lp = klo_body[tile_idx[(size_t)tile * ts + x]];
          if (!(tile < a ? lp <= lm : lp < lm)) break;          // (equal first words stand in the order of their second words)
          ++x;
        }
      }
      rank += x;
    }
  }
  out_hi[rank] = h; out_idx[rank] = b;
}

// ---- radix sort (onesweep), 8 bits a pass
constexpr int kRxT = 256;                  // threads of a pass's workgroup
constexpr int kRxKpt = 16;                 // keys per thread
constexpr int kRxTile = kRxT * kRxKpt;     // 4096 keys per tile
constexpr int kRxBins = 256;
constexpr int kRxPasses = 8;               // 63 key bits
constexpr unsigned int kRxAgg = 1u << 30, kRxIncl = 2u << 30, kRxVal = (1u << 30) - 1u;

// Path keys of all bodies (both words, body order) and, per workgroup of kRxTile bodies, how many of its keys carry each value
// of each of the first word's eight digits: part_hist[workgroup][digit][value].  (No global atomics: contended device-scope
// atomics cost ~0.8 us each on this part.)
constexpr int kKhT = 1024;                 // threads: four bodies each
__global__ __launch_bounds__(kKhT) void bh_keys_hist_kernel(SmallTree T, const float4 *__restrict__ posm, int n,
                                                            const unsigned int *__restrict__ size_bits,
                                                            unsigned int *__restrict__ next_size, float theta,
                                                            unsigned long long *__restrict__ key_hi,
                                                            unsigned long long *__restrict__ key_lo,
                                                            unsigned int *__restrict__ part_hist) {
  __shared__ unsigned int s_h[kRxPasses][kRxBins];
  const int t = threadIdx.x;
  if (T.hdr[3] != 0) return;                                   // behind a refused frame nothing happens (bh_keys_kernel); the passes return too
  const float sz = frame_size(size_bits);
  const float o0[3] = {T.prev_com[0], T.prev_com[1], T.prev_com[2]};
  if (blockIdx.x == 0) bh_frame_setup(T, o0, sz, theta, kKhT, next_size);
  for (int q = t; q < kRxPasses * kRxBins; q += kKhT) (&s_h[0][0])[q] = 0u;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kRxTile / kKhT; ++r) {
    const int i = blockIdx.x * kRxTile + r * kKhT + t;
    if (i < n) {
      const float4 p = posm[i];
      float o[3] = {o0[0], o0[1], o0[2]};
      float size = sz;
      const bool plain = sz >= 0x1p-58f;
      const unsigned long long hi = descend_word(p, o, size, plain), lo = descend_word(p, o, size, plain);
      key_hi[i] = hi; key_lo[i] = lo;
#pragma unroll
      for (int d = 0; d < kRxPasses; ++d) atomicAdd(&s_h[d][(hi >> (8 * d)) & 0xFFull], 1u);
    }
  }
  __syncthreads();
  unsigned int *out = part_hist + (size_t)blockIdx.x * (kRxPasses * kRxBins);
  for (int q = t; q < kRxPasses * kRxBins; q += kKhT) out[q] = (&s_h[0][0])[q];
}

// The workgroups' counts added up in kRxSlices slices: slice_hist[slice][digit][value] = the counts of the workgroups slice,
// slice + kRxSlices, ...  (a pass adds the slices of its digit and scans them itself: bh_radix_pass_kernel).
constexpr int kRxSlices = 16;
__global__ __launch_bounds__(kRxBins) void bh_hist_reduce_kernel(const unsigned int *__restrict__ part_hist, int nparts,
                                                                  unsigned int *__restrict__ slice_hist) {
  const int d = blockIdx.x, sl = blockIdx.y, v = threadIdx.x;
  unsigned int c = 0;
  for (int w = sl; w < nparts; w += kRxSlices) c += part_hist[((size_t)w * kRxPasses + d) * kRxBins + v];
  slice_hist[((size_t)sl * kRxPasses + d) * kRxBins + v] = c;
}

struct RadixPass {
  const unsigned long long *kin; const unsigned int *vin;
  unsigned long long *kout; unsigned int *vout;
  const unsigned int *slice_hist;          // [kRxSlices][8][256]: how many keys carry each value of each digit (bh_hist_reduce_kernel)
  int digit;
  unsigned int *desc;                      // [tiles][256] look-back words of this pass, zero before the launch
  int shift, n;
  const int *status;                       // the tree's header word 3: behind a refused frame the key kernel wrote no histograms, and a
                                           // pass must not scatter by counts that belong to other keys
};

__global__ __launch_bounds__(kRxT) void bh_radix_pass_kernel(RadixPass P) {
  __shared__ unsigned int s_cnt[kRxT / 64][kRxBins];          // per wave: keys of each digit value seen so far, then where the wave's keys of it start in the tile
  __shared__ unsigned int s_start[kRxBins];                   // where a value's keys start in the tile's sorted order
  __shared__ unsigned int s_goes[kRxBins];                    // ... and where they start in the output
  __shared__ unsigned int s_scan[kRxT / 64];
  __shared__ unsigned long long s_k[kRxTile];
  __shared__ unsigned int s_v[kRxTile];
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  if (*P.status != 0) return;                                  // (uniform over the launch: no workgroup is left waiting for another's tiles)
  // Workgroup g takes the tiles g, g + gridDim.x, ... in this order.  The host launches no more workgroups than the device can
  // hold at once, so every tile a look-back waits for belongs to a workgroup that is running (or will be as soon as another
  // process's kernel leaves) and that never waits for a later tile: the wait always ends.  (A ticket counter would do the same
  // for any grid — and serialise the workgroups' starts on one device-scope atomic: 256 of them cost the pass 20 us.)
  for (int tile = blockIdx.x; tile * kRxTile < P.n; tile += gridDim.x) {
  for (int q = t; q < (kRxT / 64) * kRxBins; q += kRxT) (&s_cnt[0][0])[q] = 0u;
  __syncthreads();
  const int tbase = tile * kRxTile, tcount = min(kRxTile, P.n - tbase);
  // wave w owns the tile's keys [1024 w, 1024 (w + 1)), sixty-four consecutive ones a round: a key's place among the keys of its
  // digit value is (keys of the value in earlier waves) + (in earlier rounds of this wave) + (in lower lanes of this round)
  unsigned long long k[kRxKpt];
  unsigned int v[kRxKpt];
  unsigned short rk[kRxKpt];
#pragma unroll
  for (int r = 0; r < kRxKpt; ++r) {
    const int e = wave * (64 * kRxKpt) + r * 64 + lane;
    const bool valid = e < tcount;
    k[r] = valid ? P.kin[tbase + e] : ~0ull;
    v[r] = valid ? (P.vin ? P.vin[tbase + e] : (unsigned int)(tbase + e)) : 0u;   // the first pass's bodies are the positions themselves
  }
#pragma unroll
  for (int r = 0; r < kRxKpt; ++r) {
    const int e = wave * (64 * kRxKpt) + r * 64 + lane;
    const bool valid = e < tcount;
    const unsigned int d = (unsigned int)(k[r] >> P.shift) & 0xFFu;
    unsigned long long same = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool one = (d >> bit) & 1u;
      const unsigned long long vote = __ballot(one);
      same &= one ? vote : ~vote;
    }
    const unsigned int before = (unsigned int)__popcll(same & ((1ull << lane) - 1ull));
    const unsigned int seen = valid ? s_cnt[wave][d] : 0u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (valid && before == 0u) s_cnt[wave][d] = seen + (unsigned int)__popcll(same);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    rk[r] = (unsigned short)(seen + before);
  }
  __syncthreads();
  // thread t owns digit value t: the waves' counts -> where each wave's keys of the value start; the tile's count
  unsigned int total = 0;
#pragma unroll
  for (int w = 0; w < kRxT / 64; ++w) { const unsigned int c = s_cnt[w][t]; s_cnt[w][t] = total; total += c; }
  // where the tile's keys of value t go: decoupled look-back over the earlier tiles' counts of the value
  unsigned int *mine = P.desc + (size_t)tile * kRxBins + t;
  __hip_atomic_store(mine, (tile == 0 ? kRxIncl : kRxAgg) | total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // (eight earlier tiles' words in flight at a time: with every tile of a pass resident at once the wait is a ripple through
  // the tiles, and its length goes with the latency of one look — ~9 us of a 20 us pass at 256 tiles; without it the pass takes
  // 11.3 us.  Adding up ALL earlier tiles' counts instead, sixteen coherent loads in flight, was tried: 28.7 us a pass.)
  unsigned int earlier = 0;
  for (int p = tile - 1; p >= 0;) {
    unsigned int w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      w[j] = p - j >= 0 ? __hip_atomic_load(P.desc + (size_t)(p - j) * kRxBins + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : kRxIncl;
    int used = 0;
    bool done = false;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (!done && used == j && (w[j] >> 30) != 0u) { earlier += w[j] & kRxVal; used = j + 1; done = (w[j] >> 30) == 2u; }
    if (done) break;
    p -= used;
    if (used == 0) __builtin_amdgcn_s_sleep(1);
  }
  if (tile > 0) __hip_atomic_store(mine, kRxIncl | (earlier + total), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // exclusive scan of the tile's counts over the values: where a value's keys start in the tile's sorted order
  unsigned int incl = total;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const unsigned int u = __shfl_up(incl, off, 64); if (lane >= off) incl += u; }
  if (lane == 63) s_scan[wave] = incl;
  __syncthreads();
  unsigned int sbase = 0;
  for (int w = 0; w < wave; ++w) sbase += s_scan[w];
  const unsigned int start = sbase + incl - total;
  s_start[t] = start;
  // where the keys of value t start in the whole output: the slices' counts of this digit added up, scanned over the values
  unsigned int all = 0;
#pragma unroll
  for (int sl = 0; sl < kRxSlices; ++sl) all += P.slice_hist[((size_t)sl * kRxPasses + P.digit) * kRxBins + t];
  unsigned int gincl = all;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const unsigned int u = __shfl_up(gincl, off, 64); if (lane >= off) gincl += u; }
  __syncthreads();                                              // s_scan has been read by everybody
  if (lane == 63) s_scan[wave] = gincl;
  __syncthreads();
  unsigned int gbase = 0;
  for (int w = 0; w < wave; ++w) gbase += s_scan[w];
  s_goes[t] = (gbase + gincl - all) + earlier - start;         // output index = s_goes[value] + place in the tile's sorted order
  __syncthreads();
  // the keys into LDS in sorted order, then out: consecutive threads write consecutive addresses within a value's run
#pragma unroll
  for (int r = 0; r < kRxKpt; ++r) {
    const int e = wave * (64 * kRxKpt) + r * 64 + lane;
    if (e < tcount) {
      const unsigned int d = (unsigned int)(k[r] >> P.shift) & 0xFFu;
      const unsigned int pos = s_start[d] + s_cnt[wave][d] + rk[r];
      s_k[pos] = k[r]; s_v[pos] = v[r];
    }
  }
  __syncthreads();
  for (int pos = t; pos < tcount; pos += kRxT) {
    const unsigned long long key = s_k[pos];
    const unsigned int d = (unsigned int)(key >> P.shift) & 0xFFu;
    const unsigned int dst = s_goes[d] + (unsigned int)pos;
    P.kout[dst] = key; P.vout[dst] = s_v[pos];
  }
  __syncthreads();                                              // the LDS arrays are the next tile's
  }
}

// After the radix sort on the first key word: runs of bodies that agree in that whole word (closer than Size / 2^21) are put in
// the order of their second words.  Usually there are hardly any — but a runaway body can blow Size up until a whole clump, tens
// of thousands of bodies, shares one first key word (round 4's frames fuzz: the run's first thread sorting it by insertion took
// a minute).  So every body of a run finds its own place: the run's ends by two searches in the sorted first words, its rank by
// counting the run's second words below its own — a wave reads the same word at a time, one broadcast load — ties in both words
// (the frame is refused anyway: 42 levels) by position.  Two launches: the first lays the run's bodies and second words out in
// the order the radix passes left (tmp_idx, tmp_lo), the second writes every body of a run to its place.
__global__ __launch_bounds__(kB) void bh_ties_gather_kernel(int n, const unsigned long long *__restrict__ khi,
                                                            const unsigned int *__restrict__ sidx,
                                                            const unsigned long long *__restrict__ klo_body,
                                                            unsigned int *__restrict__ tmp_idx, unsigned long long *__restrict__ tmp_lo) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  const unsigned long long h = khi[i];
  const bool tie = (i > 0 && khi[i - 1] == h) || (i + 1 < n && khi[i + 1] == h);
  if (!tie) return;
  const unsigned int body = sidx[i];
  tmp_idx[i] = body;
  tmp_lo[i] = klo_body[body];
}
__global__ __launch_bounds__(kB) void bh_ties_place_kernel(int n, const unsigned long long *__restrict__ khi, unsigned int *__restrict__ sidx,
                                                           const unsigned int *__restrict__ tmp_idx,
                                                           const unsigned long long *__restrict__ tmp_lo) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  const unsigned long long h = khi[i];
  const bool tie = (i > 0 && khi[i - 1] == h) || (i + 1 < n && khi[i + 1] == h);
  if (!tie) return;
  int lo = 0, hi = i;                                           // the run's first place: the first key word >= h
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (khi[mid] < h) lo = mid + 1; else hi = mid; }
  const int start = lo;
  lo = i + 1; hi = n;                                           // ... and the first place behind it
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (khi[mid] <= h) lo = mid + 1; else hi = mid; }
  const int end = lo;
  const unsigned long long mine = tmp_lo[i];
  int rank = 0;
  for (int j = start; j < end; ++j) {
    const unsigned long long o = tmp_lo[j];
    rank += (o < mine || (o == mine && j < i)) ? 1 : 0;
  }
  sidx[start + rank] = tmp_idx[i];
}

// ---------------------------------------------------------------------------------------------------------------------
// The sort of a frame that FOLLOWS a frame (round 4): bodies move little in a frame, so the previous frame's key order is
// almost this frame's.  Two launches:
//   bh_keys_bucket_kernel   visits the bodies in the previous order, 256 to a workgroup; a body's new first key word is compared
//                           with the previous frame's sorted keys at every 224th place — the boundaries of n / 224 buckets of 224
//                           consecutive places each — and the body goes into the bucket whose range holds it.  A workgroup's
//                           bodies lie next to each other in space, so they fall into a handful of neighbouring buckets: the
//                           boundaries it needs are a window of 64 around its own place (LDS), its bodies are counted per
//                           bucket in LDS and ONE global atomic per touched bucket reserves their slots (a body outside the
//                           window — it crossed a coarse cell boundary, or the root box moved — searches the boundaries in global
//                           memory and takes a slot by itself).  A bucket has room for 384 bodies; one more and the frame is
//                           given up (header word 3 := 3): every later kernel of it, and of the frames queued behind it, returns
//                           at once, the state stays what it was, and bh_collect queues those frames again, the first of them
//                           with the sorts below.
//   bh_bucket_sort_kernel   a workgroup per bucket: where the bucket starts in the order is the sum of the counts before it; its
//                           bodies are sorted in LDS (the merge by rank of bh_tile_sort_kernel; bodies that agree in the whole
//                           first key word look the second one up) and written to their final places.
// The counts live in two arrays that take turns: a frame's second kernel clears the array the next frame counts in.
constexpr int kWarmMu = 224;               // places of the previous order per bucket: a little under 256, so that a bucket's bodies — their number
                                           // wanders by a few dozen — almost always fit a padded bucket of 256 in the second kernel (512 otherwise)
constexpr int kWarmCap = 384;              // slots per bucket
constexpr int kWarmWin = 64;               // boundaries a workgroup keeps in LDS
constexpr int kStatusRetry = 3;            // header word 3: the frame was given up by the warm sort; queue it again with the cold one
constexpr int kStatusUnsorted = 4;         // ... a COLD sort left keys out of order: an internal error, reported (never seen; bh_lcp_scan_kernel's guard)

__global__ __launch_bounds__(kB) void bh_keys_bucket_kernel(SmallTree T, const float4 *__restrict__ posm, int n,
                                                            const unsigned int *__restrict__ size_bits,
                                                            unsigned int *__restrict__ next_size, float theta,
                                                            const unsigned long long *__restrict__ bound,   // [2][nb]: first, second key words
                                                            const unsigned int *__restrict__ prev_idx,
                                                            const float4 *__restrict__ prev_pos,
                                                            unsigned long long *__restrict__ slot_lo, unsigned long long *__restrict__ slot_hi,
                                                            unsigned int *__restrict__ slot_idx, unsigned int *__restrict__ gcount, int nb) {

  __shared__ unsigned long long s_b[kWarmWin], s_bl[kWarmWin];  // boundaries jlo .. jhi: the previous order's keys (both words) at places 224 j
  __shared__ unsigned int s_cnt[kWarmWin + 1], s_base[kWarmWin + 1];
  __shared__ int s_stop;
  const int t = threadIdx.x, w = blockIdx.x;
  if (t == 0) s_stop = T.hdr[3];                               // (one thread asks: other workgroups of this launch may be giving the frame up)
  __syncthreads();
  if (s_stop != 0) return;                                     // a frame before this one was refused: nothing of this one happens
  const float sz = frame_size(size_bits);
  float o[3] = {T.prev_com[0], T.prev_com[1], T.prev_com[2]};
  if (w == 0) bh_frame_setup(T, o, sz, theta, kB, next_size);
  // bucket(h) = the largest j in 1 .. nb - 1 with boundary j <= h, or 0; the window: boundaries jlo .. jhi around this workgroup's own
  const int mid_j = (int)(((long long)w * kB + kB / 2) / kWarmMu);   // the bucket this workgroup's places lie in
  const int jlo = max(1, mid_j - (kWarmWin / 2 - 1)), jhi = min(nb - 1, mid_j + kWarmWin / 2);
  const int nwin = jhi - jlo + 1;
  if (t < nwin) { s_b[t] = bound[jlo + t]; s_bl[t] = bound[nb + jlo + t]; }   // (the previous order's keys at places 224 j, gathered by its sort: bh_bucket_sort_kernel)
  if (t <= kWarmWin) s_cnt[t] = 0u;
  __syncthreads();
  const int i = w * kB + t;
  const bool valid = i < n;
  unsigned int body = 0u, local = 0u;
  unsigned long long hi = 0ull, lo = 0ull;
  int bucket = 0, q = -1;
  if (valid) {
    body = prev_idx[i];
    // (the previous frame's walk left the positions in its key order — this kernel's order — where nothing else has moved a body
    // since: a coalesced read instead of a 16-byte record out of every 64-byte sector)
    const float4 p = prev_pos != nullptr ? prev_pos[i] : posm[body];
    float size = sz;
    const bool plain = sz >= 0x1p-58f;
    hi = descend_word(p, o, size, plain);
    lo = descend_word(p, o, size, plain);                       // (goes with the body into its slot: no scattered store by body)
    // A boundary is a WHOLE key — both words, compared first word first: when a runaway body has blown Size up until every other
    // body sits in one cell of level 21 (the shipped kind of scene does that within a few hundred frames: Size 1e9, the box 1e3), all
    // first words agree and the second words alone tell the buckets apart (round 4 compared first words only: every body went to ONE
    // bucket, and every frame was given up and queued again with the cold sorts).
    auto not_above = [&](unsigned long long bh, unsigned long long bl) { return bh < hi || (bh == hi && bl <= lo); };   // boundary <= (hi, lo)
    int x = 0, y = nwin;                                       // boundaries of the window that are <= the key
    while (x < y) { const int mid = (x + y) >> 1; if (not_above(s_b[mid], s_bl[mid])) x = mid + 1; else y = mid; }
    if (x == 0 && jlo > 1) {                                   // below the window: the boundaries 1 .. jlo - 1, in global memory
      int a = 1, b = jlo;                                      // first boundary in [1, jlo) that is > the key
      while (a < b) { const int mid = (a + b) >> 1; if (not_above(bound[mid], bound[nb + mid])) a = mid + 1; else b = mid; }
      bucket = a - 1;
    } else if (x == nwin && jhi < nb - 1) {                    // above it
      int a = jhi + 1, b = nb;
      while (a < b) { const int mid = (a + b) >> 1; if (not_above(bound[mid], bound[nb + mid])) a = mid + 1; else b = mid; }
      bucket = a - 1;
    } else {
      bucket = jlo - 1 + x;
    }
    q = bucket - (jlo - 1);
    if (q >= 0 && q <= kWarmWin) local = atomicAdd(&s_cnt[q], 1u); else q = -1;
  }
  __syncthreads();
  if (t <= kWarmWin && s_cnt[t] != 0u) s_base[t] = atomicAdd(&gcount[jlo - 1 + t], s_cnt[t]);
  __syncthreads();
  if (!valid) return;
  const unsigned int pos = q >= 0 ? s_base[q] + local : atomicAdd(&gcount[bucket], 1u);
  if (pos >= (unsigned int)kWarmCap) { T.hdr[3] = kStatusRetry; return; }
  slot_hi[(size_t)bucket * kWarmCap + pos] = hi;
  slot_lo[(size_t)bucket * kWarmCap + pos] = lo;
  slot_idx[(size_t)bucket * kWarmCap + pos] = body;
}

constexpr int kBsP = 512;                  // the padded bucket at most
// kBsT threads: 512 — an element each — where the buckets are few and what counts is one bucket's way through the rounds
// (N = 65536: 14.2 us against 19.0 with 256); 256 — two elements each — where there are thousands of them (2^20: 47.5 against 53.1)
// After a cold sort: the sorted keys at every 224th place, side by side, for the frame that follows (bh_keys_bucket_kernel's boundaries;
// a warm frame's bucket sort gathers them itself)
__global__ __launch_bounds__(kB) void bh_bound_kernel(const unsigned long long *__restrict__ khi, const unsigned int *__restrict__ sidx,
                                                      const unsigned long long *__restrict__ klo_body, int nb, unsigned long long *__restrict__ bound) {
  const int j = blockIdx.x * kB + threadIdx.x;
  if (j < nb) { bound[j] = khi[(size_t)j * kWarmMu]; bound[nb + j] = klo_body[sidx[(size_t)j * kWarmMu]]; }   // (a cold frame's second words stand in body order)
}

template <int kBsT>
__global__ __launch_bounds__(kBsT) void bh_bucket_sort_kernel(SmallTree T, int n, int nb, const unsigned int *__restrict__ gcount,
                                                              unsigned int *__restrict__ gcount_next,
                                                              const unsigned long long *__restrict__ slot_hi,
                                                              const unsigned int *__restrict__ slot_idx,
                                                              const unsigned long long *__restrict__ slot_lo,
                                                              unsigned long long *__restrict__ out_hi, unsigned int *__restrict__ out_idx,
                                                              unsigned long long *__restrict__ out_lo, unsigned long long *__restrict__ bound) {
  static_assert(kWarmCap <= kBsP && kBsP % kBsT == 0, "whole rounds of the workgroup");
  __shared__ unsigned long long s_hi[2][kBsP];
  __shared__ unsigned long long s_lo[kBsP];                    // the second key words, by slot (looked at where two first words agree: in a
                                                               // scene whose Size a runaway body owns that is every comparison)
  __shared__ unsigned short s_ix[2][kBsP];
  __shared__ unsigned int s_body[kBsP];
  __shared__ unsigned int s_part[kBsT / 64];
  if (T.hdr[3] != 0) return;                                   // the frame was given up (or an earlier one refused)
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = blockIdx.x;
  // where the bucket starts: the counts of the buckets before it
  unsigned int sum = 0u;
  for (int j = t; j < b; j += kBsT) sum += gcount[j];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) s_part[wave] = sum;
  const int cnt = (int)gcount[b];
  if (t == 0) gcount_next[b] = 0u;                             // the next frame counts there
  int P = 64;
  while (P < cnt) P <<= 1;
  for (int e = t; e < P; e += kBsT) {
    const bool in = e < cnt;
    s_hi[0][e] = in ? slot_hi[(size_t)b * kWarmCap + e] : ~0ull;
    s_lo[e] = in ? slot_lo[(size_t)b * kWarmCap + e] : ~0ull;
    s_ix[0][e] = (unsigned short)e;
    s_body[e] = in ? slot_idx[(size_t)b * kWarmCap + e] : 0xFFFFFFFFu;
  }
  __syncthreads();
  unsigned int start = 0u;
  for (int wv = 0; wv < kBsT / 64; ++wv) start += s_part[wv];
  // The counts must be those of this frame's n bodies: every bucket's range lies inside [0, n) and the last one ends at n.  Counts
  // that do not add up — words that were not cleared, or were cleared under the key kernel's feet (round 4's creation memsets on the
  // null stream could do that to a first warm frame: DESIGN 7d) — would send the stores below past the arrays' ends, or leave places
  // of the order unwritten for the kernels behind this one to chase links through: the frame is given up instead and comes back with
  // the cold sorts, which count for themselves.  (uniform per workgroup; the words behind T.hdr[3] are read by the next launch)
  if ((unsigned int)cnt > (unsigned int)kWarmCap || start + (unsigned int)cnt > (unsigned int)n || (b == nb - 1 && start + (unsigned int)cnt != (unsigned int)n)) {
    if (t == 0) T.hdr[3] = kStatusRetry;
    return;
  }
  // merge sort by rank (bh_tile_sort_kernel): runs of L become runs of 2L, every element finds its place by a binary search in
  // the partner run — left run: partner elements strictly before it; right run: partner elements not after it (stable)
  int cur = 0;
  for (int L = 1, lg = 0; L < P; L <<= 1, ++lg, cur ^= 1) {
    for (int e = t; e < P; e += kBsT) {
      const int run = e >> lg, pos = e & (L - 1);
      const bool left = (run & 1) == 0;
      const int pbase = (run ^ 1) * L;
      const unsigned long long h = s_hi[cur][e];
      const unsigned short ix = s_ix[cur][e];
      int x = 0, y = L;
      while (x < y) {
        const int mid = (x + y) >> 1;
        const unsigned long long hp = s_hi[cur][pbase + mid];
        bool before = hp < h;
        if (hp == h) {                                         // the whole first key word agrees: the second words decide
          const unsigned long long lp = s_lo[s_ix[cur][pbase + mid]], lm = s_lo[ix];
          before = left ? lp < lm : lp <= lm;
        }
        if (before) x = mid + 1; else y = mid;
      }
      const int dest = (run & ~1) * L + pos + x;
      s_hi[cur ^ 1][dest] = h; s_ix[cur ^ 1][dest] = ix;
    }
    if (2 * L <= 64 && 4 * L <= 64 && P <= kBsT) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
  }
  for (int e = t; e < cnt; e += kBsT) {                        // (the second key words follow into key order: SmallTree::klo_by_body == 0)
    const int ix = s_ix[cur][e];
    out_hi[start + e] = s_hi[cur][e]; out_idx[start + e] = s_body[ix]; out_lo[start + e] = s_lo[ix];
    if ((start + e) % kWarmMu == 0u) {                         // the next frame's bucket boundaries (whole keys), side by side
      bound[(start + e) / kWarmMu] = s_hi[cur][e]; bound[nb + (start + e) / kWarmMu] = s_lo[ix];
    }
  }
}

// digits two path keys share (0 .. 42; 42: the same path all the way down)
__device__ __forceinline__ int shared_digits(unsigned long long ha, unsigned long long la, unsigned long long hb, unsigned long long lb) {
  const unsigned long long x = ha ^ hb;
  if (x != 0ull) return (__clzll((long long)x) - 1) / 3;
  const unsigned long long y = la ^ lb;
  if (y != 0ull) return kLevelsPerKey + (__clzll((long long)y) - 1) / 3;
  return kMaxLevels;
}

// lcpS[i] = lcp(i - 1) (-1 at both ends), and the numbering of the nodes: body i (key order) opens max(lcp(i) - lcp(i-1), 0)
// cells and has one leaf; the exclusive scan of these counts numbers all nodes in preorder.  The scan is done HERE, in the same
// launch — no scan library, no second pass over the data: a workgroup scans its block of kB * bpt consecutive bodies
// (first_local[i] = nodes of the block's earlier bodies) and leaves the block's total in block_sum; the few block totals
// (at most kScanBlocks) are scanned again by every workgroup of the next kernel as it starts (bh_nodes_kernel).
// The second key words stay in body order (T.klo, klo_by_body): they are looked up only where two neighbours agree in the whole
// first word (bodies closer than Size / 2^21).
constexpr int kNodeSmp = 8192;             // sampled sorted keys bh_nodes_kernel keeps in LDS (64 KB; fewer for systems of many workgroups)
constexpr int kScanBlocks = 1024;          // block totals the consumers scan in LDS; a block is kB * bpt bodies (bpt: a power of two)
__global__ __launch_bounds__(kB) void bh_lcp_scan_kernel(SmallTree T, int n, int bpt, signed char *__restrict__ lcpS,
                                                         int *__restrict__ first_local, int *__restrict__ block_sum) {
  __shared__ int s_w[kB / 64];
  __shared__ int s_m[kB / 64];
  __shared__ int s_stop;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // a frame the warm sort gave up (or one queued behind a refused frame) has no order to look at — and must keep its verdict: one
  // thread asks (other workgroups of this launch may be refusing the frame right now)
  if (t == 0) s_stop = T.hdr[3];
  __syncthreads();
  if (s_stop != 0) return;
  const int i0 = (blockIdx.x * kB + t) * bpt;                  // this thread's bodies: i0 .. i0 + bpt - 1, in key order
  int sum = 0, deep = -1;
  auto shared_at = [&](unsigned long long ha, int ia, unsigned long long hb, int ib) {   // digits the bodies at sorted positions ia, ib share
    const unsigned long long x = ha ^ hb;
    if (x != 0ull) return (__clzll((long long)x) - 1) / 3;
    return shared_digits(ha, second_word(T, ia), hb, second_word(T, ib));
  };
  if (i0 < n) {
    unsigned long long h = T.khi[i0];
    int lp = i0 > 0 ? shared_at(T.khi[i0 - 1], i0 - 1, h, i0) : -1;
    for (int q = 0; q < bpt; ++q) {
      const int i = i0 + q;
      if (i >= n) break;
      int ln = -1;
      unsigned long long hn = 0;
      if (i + 1 < n) { hn = T.khi[i + 1]; ln = shared_at(h, i, hn, i + 1); }
      // The kernels behind this one follow links made from the ORDER of the keys (a cell's end, "the node after the subtree"): keys
      // out of order would have them run backwards or off the arrays.  One compare on words already here: a frame sorted from the
      // previous order is given up and comes back with the cold sorts; a cold sort that fails it is an error of this library (status 4).
      if (i + 1 < n && (hn < h || (hn == h && second_word(T, i + 1) < second_word(T, i)))) T.hdr[3] = T.klo_by_body ? kStatusUnsorted : kStatusRetry;
      lcpS[i] = (signed char)lp;
      const int c = (ln > lp ? ln - lp : 0) + 1;
      first_local[i] = c;                                      // the count for now; the scan below turns it into the prefix
      sum += c;
      if (i == n - 1) lcpS[n] = (signed char)-1;
      if (ln == kMaxLevels) T.hdr[3] = 1;                      // the reference would recurse on: the frame is refused
      if (ln >= kLevelsPerKey) T.hdr[6] = 1;                   // neighbours that agree in the whole first key word
      deep = max(deep, ln);
      lp = ln; h = hn;
    }
  }
  // exclusive scan of the threads' sums over the block
  int incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(incl, off, 64); if (lane >= off) incl += u; }
  int m = deep;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
  if (lane == 63) s_w[wave] = incl;
  if (lane == 0) s_m[wave] = m;
  __syncthreads();
  int run = incl - sum;
  for (int w = 0; w < wave; ++w) run += s_w[w];
  for (int q = 0; q < bpt; ++q) {
    const int i = i0 + q;
    if (i >= n) break;
    const int c = first_local[i];
    first_local[i] = run;
    run += c;
  }
  if (t == kB - 1) block_sum[blockIdx.x] = run;
  // deepest level: one atomic per workgroup, spread over kDeepSlots words (sixteen thousand waves on ONE address queue for 0.2 ms)
  if (t == 0) {
    for (int w = 1; w < kB / 64; ++w) m = max(m, s_m[w]);
    if (m >= 0) atomicMax(&T.hdr[kHdrDeep + (blockIdx.x % kDeepSlots)], m);
  }
}

// The block totals of bh_lcp_scan_kernel, scanned: s_base[b] = nodes of the blocks before b, s_base[nblocks] = all nodes.
template <int NT>   // threads of the calling workgroup
__device__ __forceinline__ void scan_block_sums(const int *__restrict__ block_sum, int nblocks, int *s_base, int *s_tmp) {
  constexpr int per = (kScanBlocks + NT - 1) / NT;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  int v[per], sum = 0;
#pragma unroll
  for (int q = 0; q < per; ++q) { const int b = t * per + q; v[q] = b < nblocks ? block_sum[b] : 0; sum += v[q]; }
  int incl = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(incl, off, 64); if (lane >= off) incl += u; }
  if (lane == 63) s_tmp[wave] = incl;
  __syncthreads();
  int run = incl - sum;
  for (int w = 0; w < wave; ++w) run += s_tmp[w];
#pragma unroll
  for (int q = 0; q < per; ++q) { const int b = t * per + q; if (b <= nblocks) s_base[b] = run; run += v[q]; }
  if (t == NT - 1 && NT * per <= nblocks) s_base[nblocks] = run;   // (nblocks == kScanBlocks: the total sits one past the last thread's blocks)
  __syncthreads();
}

// body i (key order): the words of the cells it opens, its leaf's word, CoM and level
__global__ __launch_bounds__(kB) void bh_nodes_kernel(SmallTree T, const float4 *__restrict__ posm, int n,
                                                      const int *__restrict__ first_local, const int *__restrict__ block_sum,
                                                      int block_shift, int *__restrict__ first, const signed char *__restrict__ lcpS,
                                                      int smp_shift) {
  __shared__ int s_base[kScanBlocks + 1];
  __shared__ int s_tmp[kB / 64];
  extern __shared__ unsigned long long s_smp[];                // every 2^smp_shift-th sorted first key word (lower_bound_sampled): dynamic LDS
  if (T.hdr[3] != 0) return;                                   // a frame given up or refused: there is no order to number (uniform: set before this launch)
  const int nblocks = (n + (1 << block_shift) - 1) >> block_shift;
  {
    const int nsmp = (n + (1 << smp_shift) - 1) >> smp_shift;   // eight loads in flight per thread: the fill is a chain of L2 round trips otherwise
    for (int q0 = threadIdx.x; q0 < nsmp; q0 += 8 * kB) {
      unsigned long long v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int q = q0 + u * kB; v[u] = q < nsmp ? T.khi[(size_t)q << smp_shift] : 0ull; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { const int q = q0 + u * kB; if (q < nsmp) s_smp[q] = v[u]; }
    }
  }
  scan_block_sums<kB>(block_sum, nblocks, s_base, s_tmp);
  const int total = s_base[nblocks];
  auto first_of = [&](int j) { return j < n ? s_base[j >> block_shift] + first_local[j] : total; };   // first node of body j's group
  const int i = blockIdx.x * kB + threadIdx.x;
  const bool valid = i < n;
  if (total > T.cap) {                                         // (a pool sized for 42 cells per body cannot run out below 2^25 nodes)
    if (i == 0) { T.hdr[0] = 0; T.hdr[3] = 2; }
    return;
  }
  if (i == 0) T.hdr[0] = total;
  const int lp = valid ? (int)lcpS[i] : 0, ln = valid ? (int)lcpS[i + 1] : 0, m0 = valid ? first_of(i) : 0;
  if (valid) first[i] = m0;                                    // absolute node numbers for the kernels that follow
  if (i == n - 1) first[n] = total;
  const int open = ln > lp ? ln - lp : 0;
  const unsigned long long h0 = valid ? T.khi[i] : 0ull;
  // The cells a body opens, levels lp + 1 .. ln: where each ends is a search, and a wave's bodies open anything from none to a
  // ladder of twenty — so the WAVE shares them out: the cells of its 64 bodies are numbered through (a scan of the counts), lane k
  // takes cells k, k + 64, ... and fetches what it needs of the owning lane by shuffles.  (Cells below the first key word's 21
  // levels need both words — rare — and stay with their own lane.)
  const int lane = threadIdx.x & 63;
  const int open_a = max(0, min(ln, kLevelsPerKey) - lp);       // this body's cells of level <= 21
  int incl = open_a;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) { const int u = __shfl_up(incl, off, 64); if (lane >= off) incl += u; }
  const int cells = __shfl(incl, 63, 64), excl = incl - open_a;
  for (int k0 = 0; k0 < cells; k0 += 64) {
    const int k = k0 + lane;
    int src = 0;                                               // the lane whose cells hold number k: the first lane with incl > k
#pragma unroll
    for (int step = 32; step > 0; step >>= 1) { const int v = __shfl(incl, src + step - 1, 64); if (v <= k) src += step; }
    src = min(src, 63);
    const int s_lp = __shfl(lp, src, 64), s_m0 = __shfl(m0, src, 64), s_excl = __shfl(excl, src, 64);
    const unsigned int hl = __shfl((unsigned int)h0, src, 64), hh = __shfl((unsigned int)(h0 >> 32), src, 64);
    if (k < cells) {
      const int q = k - s_excl, l = s_lp + 1 + q;              // cell of level l whose first body is lane src's
      int upper = n;                                           // first body behind the cell
      if (l > 0) {
        // Most cells hold a handful of bodies: steps of 1, 2, 4, ... 64 from the cell's first body until one lands outside, then
        // the halving between the last two — neighbouring keys, a cache line or two (the small systems' cell_end).  A cell of
        // more than 127 bodies: the first key whose first l digits exceed the cell's = the first key >= (those digits + 1, then
        // zeros), a lower bound on the sorted first key words, its first steps on the samples in LDS.
        const unsigned long long hs = ((unsigned long long)hh << 32) | hl;
        const int sh = 3 * (kLevelsPerKey - l);
        const unsigned long long pre = hs >> sh;
        int x = blockIdx.x * kB + (threadIdx.x & ~63) + src, step = 1;   // x: a body of the cell
        bool found = false;
        while (step <= 64) {
          const int j = x + step;
          if (j >= n || (T.khi[j] >> sh) != pre) { found = true; break; }
          x = j; step <<= 1;
        }
        if (found) {
          int y = min(x + step, n);                                // the first body behind the cell lies in (x, y]
          while (y - x > 1) { const int mid = (x + y) >> 1; if ((T.khi[mid] >> sh) == pre) x = mid; else y = mid; }
          upper = y;
        } else {
          upper = lower_bound_sampled(T.khi, n, s_smp, smp_shift, (pre + 1ull) << sh);
        }
      }
      T.meta[s_m0 + q] = ((unsigned int)l << kLevelShift) | (unsigned int)first_of(upper);
    }
  }
  if (!valid) return;
  if (ln > kLevelsPerKey) {                                    // cells below the first key word's 21 levels (rare): both words
    const unsigned long long l0 = second_word(T, i);
    for (int q = open_a; q < open; ++q) {
      const int l = lp + 1 + q;
      int x = i + 1, y = n;
      while (x < y) {
        const int mid = (x + y) >> 1;
        if (same_prefix(T.khi[mid], second_word(T, mid), h0, l0, l)) x = mid + 1; else y = mid;
      }
      T.meta[m0 + q] = ((unsigned int)l << kLevelShift) | (unsigned int)first_of(x);
    }
  }
  const int level = (lp > ln ? lp : ln) + 1;                   // the leaf: one level below the deepest cell the body shares
  const unsigned int body = T.sidx[i];
  T.meta[m0 + open] = kLeafBit | ((unsigned int)level << kLevelShift) | body;
  T.com[m0 + open] = posm[body];                               // CenterOfMass = Position, TotalMass = Mass (.h:85-88)
  T.leaf_level[i] = (unsigned char)level;
}

// ComputeMass (.h:89-95) of the cells of level l: body i opens one iff lcp(i-1) < l <= lcp(i)
__global__ __launch_bounds__(kB) void bh_sweep_level_kernel(SmallTree T, const float4 *__restrict__ posm, int n,
                                                            const int *__restrict__ first, const signed char *__restrict__ lcpS,
                                                            int l, int div_mode) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n || T.hdr[3] != 0) return;
  const int lp = lcpS[i];
  if (!(lp < l && l <= (int)lcpS[i + 1])) return;
  const int m = first[i] + (l - lp - 1);
  T.com[m] = sweep_compact_cell(T.com, T.meta, m, T.meta[m], l, div_mode, posm, T.root);
}

__global__ void bh_finish_kernel(SmallTree T, int n, int keep_root) {
  if (T.hdr[3] != 0) return;
  if (!keep_root) { const float4 c = T.com[0]; T.prev_com[0] = c.x; T.prev_com[1] = c.y; T.prev_com[2] = c.z; }   // .cpp:78
  int deep = -1;
  for (int q = 0; q < kDeepSlots; ++q) deep = max(deep, T.hdr[kHdrDeep + q]);
  T.hdr[1] = T.hdr[0] - n; T.hdr[2] = deep + 1; T.hdr[4] = T.hdr[4] + 1;
}

// deepest level with a cell of >= 2 bodies: the maximum over the header's kDeepSlots words (bh_lcp_scan_kernel); every thread of
// the workgroup gets it (s_tmp: one int of LDS)
__device__ __forceinline__ int deepest_level(const SmallTree &T, int *s_tmp) {
  if (threadIdx.x < 64) {
    int m = -1;
    for (int q = threadIdx.x; q < kDeepSlots; q += 64) m = max(m, T.hdr[kHdrDeep + q]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = max(m, __shfl_xor(m, off, 64));
    if (threadIdx.x == 0) *s_tmp = m;
  }
  __syncthreads();
  return *s_tmp;
}

// ComputeMass (.h:89-95) in two launches instead of one per level (systems up to kChunkSweepMaxN bodies).  Body i (key order) opens the cell of level l iff
// lcp(i-1) < l <= lcp(i), and a cell's descendants are cells opened by bodies of its own range.  So a workgroup that owns
// the kB bodies [a, b) can finish, deepest level first with a workgroup barrier per level, every cell that ENDS inside its
// chunk — the cell's first body is in the chunk anyway.  What is left are the cells that reach beyond their chunk's end:
// at most one per level and chunk (cells of one level are disjoint, and each of these holds body b), noted in
// straddle[level][chunk] ...
// A chunk stages its nodes — the bodies' groups are consecutive in preorder — in LDS (up to kChunkNodes<NT> of them; a chunk of
// deep chains stays in global memory): a cell's children are met by following the skip links, a chain of dependent loads per
// cell and level — from LDS (N = 65536: 16.1 -> 13.8 us).  A thread owns ONE body: chunks of 1024 bodies are workgroups of 1024
// threads (four bodies to each of 256 threads with 80 KB of LDS left one workgroup of four waves per CU: N = 2^20 52 -> 96 us).  And
// only the levels on which the chunk has a cell at all are visited (a mask of its bodies' ladders).
template <int NT> constexpr int kChunkNodes = NT == 256 ? 1536 : 3584;
template <int NT>       // threads = bodies of a chunk
__global__ __launch_bounds__(NT) void bh_sweep_chunks_kernel(SmallTree T, const float4 *__restrict__ posm, int n,
                                                             const int *__restrict__ first, const signed char *__restrict__ lcpS,
                                                             int *__restrict__ straddle, int *__restrict__ kids, int nchunks,
                                                             int div_mode) {
  constexpr int BPT = 1;
  __shared__ float4 s_com[kChunkNodes<NT>];
  __shared__ unsigned int s_meta[kChunkNodes<NT>];
  __shared__ unsigned int s_mask[2];
  __shared__ int s_strad[kMaxLevels + 1];
  const int chunk = blockIdx.x, base = chunk * NT, t = threadIdx.x;
  if (T.hdr[3] != 0) return;                                    // a refused frame (uniform)
  if (t <= kMaxLevels) s_strad[t] = -1;
  if (t < 2) s_mask[t] = 0u;
  int lp[BPT], ln[BPT], m0[BPT];
  unsigned long long mask = 0ull;                              // the levels this thread's bodies open cells on
#pragma unroll
  for (int q = 0; q < BPT; ++q) {
    const int i = base + q * NT + t;
    lp[q] = i < n ? (int)lcpS[i] : 0; ln[q] = i < n ? (int)lcpS[i + 1] : -1; m0[q] = i < n ? first[i] : 0;
    // levels lp + 1 .. ln.  (lp = -1 for the first body of all: the shift counts stay in 0 .. 43 — `2ull << lp` there is a shift by
    // 63 on this hardware, an empty mask, and the levels only that body opens were left out: round 4's frames fuzz found it)
    if (ln[q] > lp[q]) mask |= ((2ull << ln[q]) - 1ull) & ~((1ull << (lp[q] + 1)) - 1ull);
  }
  const int chunk_start = first[base], chunk_end = first[min(base + NT, n)];   // the chunk's nodes: [chunk_start, chunk_end)
  const int nr = chunk_end - chunk_start;
  const bool in_lds = nr <= kChunkNodes<NT>;
  if (in_lds)
    for (int k = t; k < nr; k += NT) { s_com[k] = T.com[chunk_start + k]; s_meta[k] = T.meta[chunk_start + k]; }
  unsigned int mlo = (unsigned int)mask, mhi = (unsigned int)(mask >> 32);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { mlo |= (unsigned int)__shfl_xor((int)mlo, off, 64); mhi |= (unsigned int)__shfl_xor((int)mhi, off, 64); }
  __syncthreads();
  if ((t & 63) == 0) { if (mlo) atomicOr(&s_mask[0], mlo); if (mhi) atomicOr(&s_mask[1], mhi); }
  // The cells this chunk's bodies open that reach beyond the chunk — at most one per level — are noted for the second launch
  // together with their children.  A cell's children are met by following the skip links from node m + 1: a chain of
  // dependent loads that needs none of the sums.  So the chains are walked HERE, by all chunks at once and one lane per
  // level (the first bodies of a chunk open whole ladders of such cells: one thread walking them all would be the
  // kernel's critical path), and the one workgroup of the second launch finds up to eight node numbers per cell and
  // loads their sums side by side.
#pragma unroll
  for (int q = 0; q < BPT; ++q)
    for (int l = lp[q] + 1; l <= ln[q]; ++l) {
      const int m = m0[q] + (l - lp[q] - 1);
      const unsigned int w = in_lds ? s_meta[m - chunk_start] : T.meta[m];
      if ((int)(w & kLinkMask) > chunk_end) s_strad[l] = m;
    }
  __syncthreads();
  if (t <= kMaxLevels) {
    const int l = t, m = s_strad[l];
    straddle[l * nchunks + chunk] = m;
    if (m >= 0) {                                              // (for the second launch: which levels hand a sum from one chunk's thread to another's)
      atomicAdd(&T.lvl[l], 1);
      if (l > 0 && s_strad[l - 1] >= 0) atomicAdd(&T.lvl[64 + l], 1);
    }
    if (m >= 0) {
      const int end = (int)(T.meta[m] & kLinkMask);
      int *k8 = kids + ((size_t)l * nchunks + chunk) * 8;
      int k = 0;
      for (int c = m + 1; c != end;) {
        const unsigned int cw = (in_lds && c < chunk_end) ? s_meta[c - chunk_start] : T.meta[c];
        k8[k++] = c;
        c = (cw & kLeafBit) ? c + 1 : (int)(cw & kLinkMask);
      }
      for (; k < 8; ++k) k8[k] = -1;
    }
  }
  unsigned long long levels = ((unsigned long long)s_mask[1] << 32) | s_mask[0];
  while (levels != 0ull) {                                     // deepest level first
    const int l = 63 - __clzll((long long)levels);
    levels &= ~(1ull << l);
#pragma unroll
    for (int q = 0; q < BPT; ++q)
      if (lp[q] < l && l <= ln[q]) {
        const int m = m0[q] + (l - lp[q] - 1);
        if (in_lds) {
          const unsigned int w = s_meta[m - chunk_start];
          if ((int)(w & kLinkMask) <= chunk_end) {
            const float4 r = sweep_compact_cell(s_com, s_meta, m, w, l, div_mode, posm, T.root, chunk_start);
            s_com[m - chunk_start] = r;
            T.com[m] = r;
          }
        } else {
          const unsigned int w = T.meta[m];
          if ((int)(w & kLinkMask) <= chunk_end) T.com[m] = sweep_compact_cell(T.com, T.meta, m, w, l, div_mode, posm, T.root);
        }
      }
    if (in_lds) lds_barrier();                                   // (the cells' sums go on to global memory without being waited for)
    else { __threadfence_block(); __syncthreads(); }
  }
}

// ... and finished here by ONE workgroup, again deepest level first: a straddling cell's children are cells that ended
// inside a chunk (done) or straddling cells one level down (done in the round before).  Then the hand-over: the next frame's root centre, the header's counts.
constexpr int kTopT = 1024;
constexpr int kChunkSweepMaxN = 1 << 20;      // larger systems sweep with a launch per level (bh_forces)
// bodies per thread of the first launch: as few as keep the chunks within one per thread of the second launch's workgroup
// chunks of 256 bodies up to N = 98304, of 1024 above: fewer cells are left for the second launch (frames, 256 / 1024: N = 65536
// 192.8 / 192.4 us, 131072 213.9 / 209.6, 262144 265.4 / 252.0; beyond kTopT * kB bodies the second launch has no thread per 256-body chunk)
constexpr int sweep_bpt(int n) { return n <= 98304 ? 1 : 4; }
static_assert((kChunkSweepMaxN + 4 * kB - 1) / (4 * kB) <= kTopT, "bh_sweep_top_kernel: one chunk per thread");
// (Round 4 tried to take the levels' hand-over off the way through L2 — the cells computed here entered into an LDS table keyed by
// (level, node), the final children's sums fetched one and two levels ahead; then a thread per cell instead of per chunk with every
// cell's children fetched before the level loop: 33 - 37 us at N = 65536 against 24.8 for this form, 78 - 93 against 56 at 2^20.
// What the loads of this form wait for is memory other XCDs wrote (~1.5 us away), once per level; the table's looks and the
// cells' numbering cost more than they saved.  A last form had the first launch list every level's cells as 80-byte records —
// children, and for each child whether its sum is final in memory or comes from this launch's own level below (then out of an
// LDS slot per chunk) — with a thread per cell, four cells' records and final sums fetched side by side before the level loop and
// nothing but LDS reads, the additions, an LDS write and an LDS barrier per level: bit-exact, the launch itself 11.8 us against
// 14.8 at N = 8192 and 23.6 / 23.4 at 65536, but the frames no faster on one box — 8192 140.5 us against 137.6, 32768 183.0 /
// 178.9, 65536 193.3 / 191.1, 2^18 246.2 / 245.3: the first launch pays for the tags and the records what the second saves.)
__global__ __launch_bounds__(kTopT) void bh_sweep_top_kernel(SmallTree T, const float4 *__restrict__ posm, int n,
                                                             const int *__restrict__ straddle, const int *__restrict__ kids,
                                                             int nchunks, int div_mode, int keep_root) {
#pragma clang fp contract(off)
  __shared__ int s_deep;
  if (T.hdr[3] != 0) return;
  const int deep = deepest_level(T, &s_deep);
  // one chunk per thread (nchunks <= kTopT up to kChunkSweepMaxN bodies); the cell of the NEXT level and its children's
  // node numbers — which depend on none of the sums — are fetched while this level's sums are formed
  const int c = threadIdx.x;
  const bool mine = c < nchunks;
  // (they come from memory other XCDs wrote, ~1 us away — a whole level's step: so they are asked for THREE levels ahead, and the
  // loads do not wait for one another: the children's numbers are read whether or not there is a cell)
  int m_nx = -1, m_n2 = -1, m_n3 = -1;
  int4 ka_nx = make_int4(-1, -1, -1, -1), kb_nx = ka_nx, ka_n2 = ka_nx, kb_n2 = ka_nx, ka_n3 = ka_nx, kb_n3 = ka_nx;
  auto fetch = [&](int l) {                                    // shift the queue by a level and ask for level l - 2's
    m_nx = m_n2; ka_nx = ka_n2; kb_nx = kb_n2;
    m_n2 = m_n3; ka_n2 = ka_n3; kb_n2 = kb_n3;
    const int l3 = l - 2;
    m_n3 = -1;
    if (mine && l3 >= 0) {
      m_n3 = straddle[l3 * nchunks + c];
      const int4 *k8 = (const int4 *)(kids + ((size_t)l3 * nchunks + c) * 8);
      ka_n3 = k8[0]; kb_n3 = k8[1];
    }
  };
  fetch(deep + 2); fetch(deep + 1); fetch(deep);               // levels deep, deep - 1, deep - 2 on their way; level deep's in hand
  // A chunk's cells form a ladder of consecutive levels, each the child of the next one up: that child's sum is this thread's own
  // result of the step before and comes out of a register — a ladder of single-child cells (the levels above a system that fills a
  // corner of its root box: a runaway body sets Size) then loads nothing at all.
  int m_own = -1;
  float4 r_own = make_float4(0.f, 0.f, 0.f, 0.f);
  // A level's sums must be out in memory before the next level reads them (the store's way to L2 and back: ~1 us a level) — unless
  // nobody reads another thread's: level l has no cell here, or level l - 1 has none, or each has one and both are one chunk's (its
  // own register).  The first launch has counted (T.lvl).
  __shared__ int s_lvl[128];
  if (threadIdx.x < 128) s_lvl[threadIdx.x] = T.lvl[threadIdx.x];
  __syncthreads();
  for (int l = deep; l >= 0; --l) {
    const int m = m_nx;
    const int kid[8] = {ka_nx.x, ka_nx.y, ka_nx.z, ka_nx.w, kb_nx.x, kb_nx.y, kb_nx.z, kb_nx.w};
    fetch(l - 1);
    if (m >= 0) {
      float4 ch[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {                              // up to eight loads in flight
        ch[k] = r_own;
        if (kid[k] >= 0 && kid[k] != m_own) ch[k] = T.com[kid[k]];
      }
      float M = 0.f, cx = 0.f, cy = 0.f, cz = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k)                                // the children in octant order, as sweep_compact_cell adds them
        if (kid[k] >= 0) { M = M + ch[k].w; cx = cx + ch[k].w * ch[k].x; cy = cy + ch[k].w * ch[k].y; cz = cz + ch[k].w * ch[k].z; }
      r_own = cell_com_from_sums(M, cx, cy, cz, T.meta, m, l, div_mode, posm, T.root);
      m_own = m;
      T.com[m] = r_own;
    }
    const int here = s_lvl[l], above = l > 0 ? s_lvl[l - 1] : 0;
    if (!(here == 0 || above == 0 || (here == 1 && above == 1 && s_lvl[64 + l] == 1))) {
      __threadfence_block();
      __syncthreads();
    }
  }
  if (threadIdx.x == 0) {
    if (!keep_root) { const float4 c = T.com[0]; T.prev_com[0] = c.x; T.prev_com[1] = c.y; T.prev_com[2] = c.z; }   // .cpp:78
    T.hdr[1] = T.hdr[0] - n; T.hdr[2] = deep + 1; T.hdr[4] = T.hdr[4] + 1;
  }
}

// Octree::ComputeForces (.h:99-108) on the compact tree, one lane per body in key order, node by node: a 16-byte and a
// 4-byte load, the squared distance and a compare per node (accept_threshold); root, double-precision factor and the three
// multiply-adds only where a term is added.
// (Fetching the NEXT node of the preorder while a node is looked at — the walk goes there whenever it descends or the node is a
// leaf, two steps in three — was tried in round 4: slower at every size, N = 32768 200 us a frame against 185, 65536 213 / 197,
// 2^18 316 / 284, 2^20 843 / 710.)
__global__ __launch_bounds__(kB) void bh_walk_lane_kernel(SmallTree T, float4 *__restrict__ posm, float4 *__restrict__ vel,
                                                          float4 *__restrict__ acc, int n, double G, float dt, float *__restrict__ stage,
                                                          unsigned int *__restrict__ next_size, float4 *__restrict__ pos_sorted,
                                                          WalkSlice S) {
#pragma clang fp contract(off)
  __shared__ float s_thr[kMaxLevels + 2];
  hand_verdict(T);
  if (T.hdr[3] != 0) return;                                   // the frame was refused: nothing moves
  if (threadIdx.x <= kMaxLevels) s_thr[threadIdx.x] = T.thr[threadIdx.x];
  __syncthreads();
  const int k = blockIdx.x * kB + threadIdx.x;
  const bool valid = k < n;                                    // (n: the bodies this context walks — all, or its slice's)
  const int nodes = valid ? T.hdr[0] : 0;
  const int place = valid ? walk_place(S, k) : 0;               // the body's sorted position
  const unsigned int body = valid ? T.sidx[place] : 0u;
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
#else
  // Where the walk goes next follows from the node's test alone — a compare —, not from its term: the NEXT node's load is issued
  // before the term (root, double-precision factor: ~100 dependent instructions) is worked out, and is in flight under it.  At
  // the sizes where a SIMD holds one or two of these waves (N up to ~131072: the walk is a chain of ~220 dependent loads per body,
  // DESIGN 4.5) that takes the term off the chain; at 2^20, where the waves queue for the VALU anyway, it changes nothing.
  // (Not round 4's speculative fetch of node + 1 — wrong one step in three, and slower: this is the node the walk does visit.)
  float4 cm = T.com[0];
  unsigned int w = T.meta[0];
  while (node < nodes) {
    const bool leaf = (w & kLeafBit) != 0u;
    const int past = leaf ? node + 1 : (int)(w & kLinkMask);
    const float ex = p.x - cm.x, ey = p.y - cm.y, ez = p.z - cm.z;
    float d2 = ex * ex + ey * ey;
    d2 = d2 + ez * ez;
    const bool take = leaf || d2 >= s_thr[(w >> kLevelShift) & 63u];   // .h:103 (reading the threshold for leaves as well, without
                                                                       // the branch around it, costs more than the branch: 2^20 737 us a frame against 662;
                                                                       // x and y as v_pk_*_f32 pairs: 648 against 638)
    const int next = (take || d2 == 0.f) ? past : node + 1;    // .h:102: d == 0 adds nothing and ends the subtree; children 0..7 otherwise
    const int fetch = min(next, nodes - 1);                    // (the last step fetches a node nobody looks at)
    const float4 cm_next = T.com[fetch];
    const unsigned int w_next = T.meta[fetch];
    if (take && d2 != 0.f) {
      float tx, ty, tz;
      force_term(cm.x, cm.y, cm.z, cm.w, p, G, tx, ty, tz);
      ax = ax + tx; ay = ay + ty; az = az + tz;
    }
    cm = cm_next; w = w_next; node = next;
  }
#endif
  walk_lane_tail(valid, body, p, ax, ay, az, posm, vel, acc, dt, stage, S.off, next_size, pos_sorted, place);
}


}  // namespace

struct BhState {
  int n = 0, node_cap = 0;
  bool small = false;          // n <= kSmBodies: one workgroup builds the tree in LDS (bh_small_build_kernel)
  SmallTree st{};              // the compact tree (either path)
  int frames_seen = 0;         // st.hdr[4] at the last bh_collect
  unsigned int *size_words = nullptr;   // larger systems: two sets of kSizeSlots device words for ComputeCubeSize that take turns (frame_size)
  int size_word = 0;
  bool size_ready = false;              // the previous frame's walk left this frame's Size there, and nothing has moved a body since
  float4 *pos_sorted = nullptr;         // the positions in the last frame's key order, written by its walk
  bool pos_ready = false;               // ... and they are what posm[b->idx[i]] holds (nothing else has moved a body or sorted since)
  bool external = false;                // the caller holds the position buffer: bodies may move behind the library's back
  // path keys (larger systems): klo = the second key words in body order (SmallTree::klo, klo_by_body); khi / idx
  // end up holding the sorted first key words and bodies (SmallTree::khi, ::sidx), khi2 / idx2 are the sorts' other buffers
  unsigned long long *khi = nullptr, *klo = nullptr, *khi2 = nullptr;
  unsigned int *idx = nullptr, *idx2 = nullptr;
  // radix sort (n > kMergeMaxN): the key kernel's partial digit histograms, where each digit value's keys start, the passes'
  // look-back words and tile tickets (cleared by one memset per frame)
  unsigned int *part_hist = nullptr, *slice_hist = nullptr, *rx_desc = nullptr;
  size_t rx_desc_bytes = 0;
  int rx_resident = 1;                     // workgroups of bh_radix_pass_kernel the device holds at once
  // the sort of a frame that follows a frame (bh_keys_bucket_kernel): slots of kWarmCap bodies per bucket, the buckets' counts (two
  // arrays that take turns), whether b->khi / b->idx hold a previous frame's order, and what bh_collect needs to queue frames again
  unsigned long long *bound = nullptr;     // [2][nb] the sorted keys (first words, then second words) at places 224 j: the next frame's bucket boundaries
  unsigned long long *slot_hi = nullptr, *slot_lo = nullptr, *klo_sorted = nullptr;   // (klo_sorted: a warm frame's second key words, in key order)
  unsigned int *slot_idx = nullptr, *gcount = nullptr;
  int nb = 0, gturn = 0;
  bool warm = false;
  long long warm_frames = 0, retries = 0;  // frames queued with the warm sort; times bh_collect queued frames again (tests, tuning)
  // Frames the warm sort gives up cost a warm attempt AND a cold frame.  After two collects in a row that met a given-up frame the
  // library stays with the cold sorts for cold_span frames (8, doubling up to 64 while the giving-up goes on); a collect whose warm
  // frames all went through starts afresh.
  int giveups_in_row = 0, cold_left = 0, cold_span = 8, warm_since_collect = 0;
  int *first = nullptr, *first_local = nullptr, *block_sum = nullptr;   // [n + 1] first node of every body's group (absolute / within its scan block), the blocks' totals
  int tile_size = kTs;                     // bodies per tile of the tiles + merge sort (1024, 2048 or 4096: tile_size)
  int smp_shift = 0;                       // bh_nodes_kernel keeps every 2^smp_shift-th sorted key in LDS
  bool radix = false;                      // sorts by radix passes (n > bh_merge_max_n()) rather than tiles + merge
  int scan_bpt = 4, scan_shift = 10;       // bodies per thread of bh_lcp_scan_kernel, log2 of its block (kB * bpt bodies)
  signed char *lcpS = nullptr;             // [n + 1] shared digits of neighbours
  int *straddle = nullptr;                 // [kMaxLevels + 1][chunks of kB bodies] cells that reach beyond their chunk (bh_sweep_chunks_kernel)
  int *kids = nullptr;                     // ... and the (up to eight) children of each, [kMaxLevels + 1][chunks][8]
  hipEvent_t ev = nullptr;                 // larger systems: "the verdict and the deepest level are on the host"
  int *counters = nullptr;     // device: the tree's header (SmallTree::hdr; [5]: deepest level, larger systems)
  int *h_counters = nullptr;   // pinned
  int *h_verdict = nullptr;    // pinned and mapped: header words 0 .. 7 as the last frame's walk left them (SmallTree::verdict)
  float *root = nullptr;       // ox, oy, oz, size
  float *prev_com = nullptr;   // the previous tree's root CoM (zero before the first frame)
  int last_nodes = 0, last_levels = 0;
  int div_mode = 0;            // reading of `/=` in ComputeMass (sweep_compact_cell)
  // a context that owns a slice of the bodies (range partition over GPUs): the tree is the whole system's, the walk its own bodies'
  int i_begin = 0, i_count = 0;            // the slice; i_count == n: all bodies
  bool sliced = false;
  unsigned int *own = nullptr, *own_blk = nullptr;   // [i_count] sorted positions of the slice's bodies in key order; [blocks of kB] their counts (bh_own_*_kernel)
  bool level_sweeps = false;   // ComputeMass with a launch per level at any size (NBODY_BH_LEVEL_SWEEPS=1 at creation; always above kChunkSweepMaxN)
};

#define BH_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

// bodies up to which the larger systems sort by tiles + merge (radix above); NBODY_BH_MERGE_MAX_N: tests and A/B measurements
int bh_merge_max_n() {
  const char *e = getenv("NBODY_BH_MERGE_MAX_N");               // read at every bh_create: a context keeps what it was created with
  return e && *e ? atoi(e) : kMergeMaxN;
}

// bodies per tile of the tiles + merge sort: small tiles put more workgroups to work on the sort, but the merge looks through
// every tile for every element.  NBODY_BH_TILE: tests and tuning.
int bh_tile_size(int n) {
  const char *e = getenv("NBODY_BH_TILE");
  if (e && *e) { const int v = atoi(e); if (v == 1024 || v == 2048 || v == 4096) return v; }
  return n <= 16384 ? 1024 : (n <= 98304 ? 2048 : 4096);      // frames, tile 1024 / 2048 / 4096: N = 16384 179 / 181 / 203 us, 32768 230 / 226 / 241, 65536 282 / 261 / 265, 131072 404 / 347 / 336 (profiles/r04_bh_tile_size_sweep.txt)
}

static hipError_t bh_create_state(BhState *b, int n, int i_begin, int i_count);

// The hipMemset calls of bh_create_state go to the NULL stream and return before they have run; the frames run on the context's
// stream, which is non-blocking — it does not wait for the null stream.  This state is created by the first theta > 0 call, right
// in front of its first frame: without the wait here a memset could land in the middle of that frame (larger systems: the Size
// words cleared after part of the bounds kernel's maxima were in — a first frame with a root box too small, round 4's frames fuzz,
// one large scene in ten; small systems: the header's frame count or the previous tree's CoM zeroed after the first frame wrote
// them — a second tree rooted at zero).  EVERY way out of the creation passes through this wait, the small systems' included.
hipError_t bh_create(BhState **out, int n, int i_begin, int i_count) {
  BhState *b = new BhState();
  *out = b;                    // the caller destroys it whatever happens below
  const hipError_t e = bh_create_state(b, n, i_begin, i_count);
  const hipError_t w = hipStreamSynchronize(nullptr);
  return e != hipSuccess ? e : w;
}

static hipError_t bh_create_state(BhState *b, int n, int i_begin, int i_count) {
  if (n > (int)kLinkMask) return hipErrorInvalidValue;      // a leaf's word holds its body's index in 25 bits
  if (i_begin < 0 || i_count < 1 || i_begin + i_count > n) return hipErrorInvalidValue;
  b->n = n;
  b->i_begin = i_begin; b->i_count = i_count;
  b->sliced = i_count != n;
  if (b->sliced) {
    b->external = true;        // the other bodies move behind this context's back (the other devices' walks; the all-gather brings them):
                               // every frame looks at the positions itself for its Size and its keys
    BH_TRY(hipMalloc(&b->own, sizeof(unsigned int) * (size_t)i_count));
    BH_TRY(hipMalloc(&b->own_blk, sizeof(unsigned int) * (size_t)((n + kB - 1) / kB)));
  }
  b->small = n <= kSmBodies;
  BH_TRY(hipMalloc(&b->khi, sizeof(unsigned long long) * n));
  BH_TRY(hipMalloc(&b->klo, sizeof(unsigned long long) * n));
  BH_TRY(hipMalloc(&b->idx, sizeof(unsigned int) * n));
  BH_TRY(hipMalloc(&b->counters, sizeof(int) * kHdrWords));
  BH_TRY(hipMemset(b->counters, 0, sizeof(int) * kHdrWords));
  BH_TRY(hipHostMalloc(&b->h_counters, sizeof(int) * kHdrWords, hipHostMallocDefault));
  BH_TRY(hipHostMalloc(&b->h_verdict, sizeof(int) * 8, hipHostMallocMapped));
  memset(b->h_verdict, 0, sizeof(int) * 8);
  BH_TRY(hipMalloc(&b->root, sizeof(float) * 4));
  BH_TRY(hipMemset(b->root, 0, sizeof(float) * 4));
  BH_TRY(hipMalloc(&b->prev_com, sizeof(float) * 3));
  BH_TRY(hipMemset(b->prev_com, 0, sizeof(float) * 3));    // FVector t = ZeroVector, .cpp:77
  // worst case: every body opens a chain of 42 cells of its own (never, but the pool must not be what fails)
  b->node_cap = (int)std::min<long long>(((long long)(kMaxLevels + 1) * n + 64 + 3) / 4 * 4, (long long)kLinkMask);
  SmallTree &t = b->st;
  BH_TRY(hipMalloc(&t.com, sizeof(float4) * (size_t)b->node_cap));
  BH_TRY(hipMalloc(&t.meta, sizeof(unsigned int) * (size_t)b->node_cap));
  BH_TRY(hipMalloc(&t.leaf_level, (size_t)n));
  BH_TRY(hipMalloc(&t.thr, sizeof(float) * (kMaxLevels + 2)));
  BH_TRY(hipMalloc(&t.lvl, sizeof(int) * 128));
  BH_TRY(hipMemset(t.lvl, 0, sizeof(int) * 128));
  BH_TRY(hipMalloc(&t.clocks, sizeof(long long) * kDbgClocks));
  BH_TRY(hipMemset(t.clocks, 0, sizeof(long long) * kDbgClocks));
  t.khi = b->khi; t.klo = b->klo; t.sidx = b->idx; t.hdr = b->counters; t.root = b->root; t.prev_com = b->prev_com;
  BH_TRY(hipHostGetDevicePointer((void **)&t.verdict, b->h_verdict, 0));
  t.cap = b->node_cap;
  if (b->small) return hipSuccess;
  BH_TRY(hipMalloc(&b->size_words, 2 * kSizeSlots * sizeof(unsigned int)));
  BH_TRY(hipMemset(b->size_words, 0, 2 * kSizeSlots * sizeof(unsigned int)));
  BH_TRY(hipMalloc(&b->khi2, sizeof(unsigned long long) * n));
  BH_TRY(hipMalloc(&b->idx2, sizeof(unsigned int) * n));
  t.klo_by_body = 1;           // the second key words stay where the key kernel put them (second_word())
  b->radix = n > bh_merge_max_n();
  { const char *e = getenv("NBODY_BH_LEVEL_SWEEPS"); b->level_sweeps = e && e[0] == '1'; }   // read at every bh_create, like the sorts' switch
  b->tile_size = bh_tile_size(n);
  { const int budget = n <= 131072 ? kNodeSmp / 4 : 512;       // many workgroups: a smaller table each (its fill is traffic; only cells of more than 127 bodies look at it)
    while ((((n - 1) >> b->smp_shift) + 1) > budget) ++b->smp_shift; }
  if (b->radix) {
    const size_t tiles = (size_t)((n + kRxTile - 1) / kRxTile);
    BH_TRY(hipMalloc(&b->part_hist, sizeof(unsigned int) * tiles * kRxPasses * kRxBins));
    BH_TRY(hipMalloc(&b->slice_hist, sizeof(unsigned int) * kRxSlices * kRxPasses * kRxBins));
    b->rx_desc_bytes = sizeof(unsigned int) * tiles * kRxPasses * kRxBins;
    BH_TRY(hipMalloc(&b->rx_desc, b->rx_desc_bytes));
    int per_cu = 0, dev = 0, cus = 0;
    BH_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, bh_radix_pass_kernel, kRxT, 0));
    BH_TRY(hipGetDevice(&dev));
    BH_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    b->rx_resident = std::max(1, per_cu * cus);
  }
  b->nb = (n + kWarmMu - 1) / kWarmMu;
  BH_TRY(hipMalloc(&b->slot_hi, sizeof(unsigned long long) * (size_t)b->nb * kWarmCap));
  BH_TRY(hipMalloc(&b->bound, sizeof(unsigned long long) * 2 * (size_t)b->nb));   // both key words
  BH_TRY(hipMalloc(&b->pos_sorted, sizeof(float4) * (size_t)n));
  BH_TRY(hipMalloc(&b->slot_lo, sizeof(unsigned long long) * (size_t)b->nb * kWarmCap));
  BH_TRY(hipMalloc(&b->klo_sorted, sizeof(unsigned long long) * (size_t)n));
  BH_TRY(hipMalloc(&b->slot_idx, sizeof(unsigned int) * (size_t)b->nb * kWarmCap));
  BH_TRY(hipMalloc(&b->gcount, sizeof(unsigned int) * 2 * (size_t)b->nb));
  BH_TRY(hipMemset(b->gcount, 0, sizeof(unsigned int) * 2 * (size_t)b->nb));
  BH_TRY(hipMalloc(&b->first, sizeof(int) * ((size_t)n + 1)));
  BH_TRY(hipMalloc(&b->first_local, sizeof(int) * ((size_t)n + 1)));
  BH_TRY(hipMalloc(&b->block_sum, sizeof(int) * (kScanBlocks + 1)));
  BH_TRY(hipMalloc(&b->lcpS, (size_t)n + 1));
  while ((long long)kB * b->scan_bpt * kScanBlocks < (long long)n) { b->scan_bpt *= 2; b->scan_shift += 1; }   // at most kScanBlocks block totals
  if (n <= kChunkSweepMaxN) {
    const size_t nchunks = (size_t)((n + kB * sweep_bpt(n) - 1) / (kB * sweep_bpt(n)));
    BH_TRY(hipMalloc(&b->straddle, sizeof(int) * (size_t)(kMaxLevels + 1) * nchunks));
    BH_TRY(hipMalloc(&b->kids, sizeof(int) * 8 * (size_t)(kMaxLevels + 1) * nchunks));
  }
  BH_TRY(hipEventCreateWithFlags(&b->ev, hipEventDisableTiming));
  return hipSuccess;
}

void bh_destroy(BhState *b) {
  if (!b) return;
  void *ptrs[] = {b->khi, b->klo, b->khi2, b->idx, b->idx2, b->bound, b->pos_sorted, b->slot_hi, b->slot_lo, b->klo_sorted, b->slot_idx, b->gcount, b->size_words, b->part_hist, b->slice_hist, b->rx_desc, b->first, b->first_local, b->block_sum, b->lcpS, b->straddle, b->kids, b->own, b->own_blk,
                  b->counters, b->root, b->prev_com, b->st.com, b->st.meta, b->st.leaf_level, b->st.thr, b->st.lvl, b->st.clocks};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  if (b->h_counters) (void)hipHostFree(b->h_counters);
  if (b->h_verdict) (void)hipHostFree(b->h_verdict);
  if (b->ev) (void)hipEventDestroy(b->ev);
  delete b;
}

bool bh_is_small(const BhState *b) { return b->small; }

// -DNBODY_BH_PHASE_CLOCKS builds only: the last frame's wall_clock64 stamps (100 MHz) at the kernels' phase boundaries
hipError_t bh_debug_clocks(BhState *b, long long out[16 + 3 * 512], hipStream_t s) {
  if (!b->small) return hipErrorInvalidValue;
  BH_TRY(hipStreamSynchronize(s));
  return hipMemcpy(out, b->st.clocks, sizeof(long long) * kDbgClocks, hipMemcpyDeviceToHost);
}
const float *bh_root_device(const BhState *b) { return b->root; }
void bh_debug_sort_counts(const BhState *b, long long *warm_frames, long long *retries) { *warm_frames = b->warm_frames; *retries = b->retries; }

hipError_t bh_debug_poison(BhState *b, int kind, hipStream_t s) {
  if (b->small || kind != 1) return hipErrorInvalidValue;
  return hipMemsetD32Async((hipDeviceptr_t)b->gcount, 3, 2 * (size_t)b->nb, s);
}

// NBODY_BH_WALK=rows: the walks with sixteen lanes per body (round 3) instead of a wave per body — A/B measurements
static bool bh_wave_walk() {
  static const bool v = [] { const char *e = getenv("NBODY_BH_WALK"); return !(e && e[0] == 'r'); }();
  return v;
}

// The walk's view of the context's slice (WalkSlice): for a slice, the list of its bodies' sorted positions is made first —
// behind the frame's sort, T.sidx final.
static WalkSlice bh_walk_slice(BhState *b, hipStream_t s) {
  if (!b->sliced) return WalkSlice{nullptr, 0};
  const dim3 grd((b->n + kB - 1) / kB), blk(kB);
  hipLaunchKernelGGL(bh_own_count_kernel, grd, blk, 0, s, b->st.sidx, b->n, (unsigned int)b->i_begin, (unsigned int)b->i_count,
                     b->counters + 3, b->own_blk);
  hipLaunchKernelGGL(bh_own_list_kernel, grd, blk, 0, s, b->st.sidx, b->n, (unsigned int)b->i_begin, (unsigned int)b->i_count,
                     b->counters + 3, b->own_blk, b->own);
  return WalkSlice{b->own, b->i_begin};
}

// One CreateOctree (.cpp:74-89) + walk (+ update) of a larger system, queued on the stream.  Up to kChunkSweepMaxN bodies nothing
// waits for the host; above, the level-by-level ComputeMass needs the deepest level there (one wait inside).
static hipError_t bh_large_frame(BhState *b, void *posm_v, void *vel, void *acc_v, float theta, double G, float dt, int keep_root,
                                 float *stage, hipStream_t s) {
  float4 *posm = (float4 *)posm_v;
  const int n = b->n;
  const dim3 blk(kB), grd((n + kB - 1) / kB);
  // ComputeCubeSize (.cpp:47-56): the previous frame's walk has left it in this frame's slot words when that frame moved the
  // bodies and nothing else has since (size_ready); otherwise a pass over the positions.  The frame's first kernel clears the
  // other set of words, where this frame's walk (dt > 0) leaves the next frame's.
  unsigned int *cur = b->size_words + (size_t)b->size_word * kSizeSlots, *nxt = b->size_words + (size_t)(b->size_word ^ 1) * kSizeSlots;
  b->size_word ^= 1;
  static const bool size_off = [] { const char *e = getenv("NBODY_BH_SIZE_FROM_WALK"); return e && e[0] == '0'; }();   // A/B, tests
  if (!(b->size_ready && !b->external && !size_off)) {
    BH_TRY(hipMemsetAsync(cur, 0, kSizeSlots * sizeof(unsigned int), s));
    BH_TRY(launch_bounds(0 /* NBODY_PREC_F32 */, posm, 0, n, cur, s, nullptr));
  }
  unsigned int *next_size = dt > 0.0f ? nxt : nullptr;         // (a pass that moves nothing leaves nothing)
  b->size_ready = dt > 0.0f && !b->sliced;
  const unsigned int *size_bits = cur;
  // The order of the 126-bit keys (see "the larger systems' own sort" above).  Either way the sorted first key words end up in
  // b->khi, the bodies in b->idx, and the second key words, still in body order, in b->klo.
  SmallTree &T = b->st;
  T.khi = b->khi; T.sidx = b->idx; T.klo = b->klo; T.klo_by_body = 1;
  static const bool warm_off = [] { const char *e = getenv("NBODY_BH_WARM_SORT"); return e && e[0] == '0'; }();   // A/B, tests
  const bool warm_now = b->warm && !warm_off && b->cold_left == 0;
  if (b->cold_left > 0) b->cold_left -= 1;                      // (the warm sort keeps giving frames up: cold for a while — BhState::giveups_in_row)
  if (warm_now) {
    // a frame that follows a frame: the previous order is almost this frame's (bh_keys_bucket_kernel)
    b->warm_since_collect += 1;
    unsigned int *gc = b->gcount + (size_t)b->gturn * b->nb, *gc_next = b->gcount + (size_t)(b->gturn ^ 1) * b->nb;
    b->gturn ^= 1;
    b->warm_frames += 1;
    hipLaunchKernelGGL(bh_keys_bucket_kernel, grd, blk, 0, s, T, posm, n, size_bits, nxt, theta, b->bound, b->idx,
                       (b->pos_ready && !b->external && !size_off) ? b->pos_sorted : nullptr, b->slot_lo, b->slot_hi,
                       b->slot_idx, gc, b->nb);
    if (b->nb <= 1024)
      hipLaunchKernelGGL(bh_bucket_sort_kernel<512>, dim3(b->nb), dim3(512), 0, s, T, n, b->nb, gc, gc_next, b->slot_hi, b->slot_idx, b->slot_lo, b->khi, b->idx,
                         b->klo_sorted, b->bound);
    else
      hipLaunchKernelGGL(bh_bucket_sort_kernel<256>, dim3(b->nb), dim3(256), 0, s, T, n, b->nb, gc, gc_next, b->slot_hi, b->slot_idx, b->slot_lo, b->khi, b->idx,
                         b->klo_sorted, b->bound);
    T.klo = b->klo_sorted; T.klo_by_body = 0;                    // (this frame's second key words stand in key order)
  } else if (!b->radix) {
    hipLaunchKernelGGL(bh_keys_kernel, grd, blk, 0, s, T, posm, n, size_bits, nxt, theta, b->khi, b->klo);
    const int ts = b->tile_size, tiles = (n + ts - 1) / ts;
    if (ts == 1024) hipLaunchKernelGGL(bh_tile_sort_kernel<1024>, dim3(tiles), dim3(kTsT), 0, s, n, b->khi, b->klo, b->khi2, b->idx2);
    else if (ts == 2048) hipLaunchKernelGGL(bh_tile_sort_kernel<2048>, dim3(tiles), dim3(kTsT), 0, s, n, b->khi, b->klo, b->khi2, b->idx2);
    else hipLaunchKernelGGL(bh_tile_sort_kernel<4096>, dim3(tiles), dim3(kTsT), 0, s, n, b->khi, b->klo, b->khi2, b->idx2);
    int shift = 0;                                               // the tiles' samples must fit the merge's LDS table
    while (((tiles * ts) >> shift) > kMergeSmp) ++shift;
    hipLaunchKernelGGL(bh_tile_merge_kernel, grd, blk, 0, s, n, ts, shift, b->khi2, b->idx2, b->klo, b->khi, b->idx);
  } else {
    const int tiles = (n + kRxTile - 1) / kRxTile;
    BH_TRY(hipMemsetAsync(b->rx_desc, 0, b->rx_desc_bytes, s));
    hipLaunchKernelGGL(bh_keys_hist_kernel, dim3(tiles), dim3(kKhT), 0, s, T, posm, n, size_bits, nxt, theta, b->khi, b->klo, b->part_hist);
    hipLaunchKernelGGL(bh_hist_reduce_kernel, dim3(kRxPasses, kRxSlices), dim3(kRxBins), 0, s, b->part_hist, tiles, b->slice_hist);
    const int pass_grid = std::min(tiles, b->rx_resident);        // all workgroups of a pass resident at once (bh_radix_pass_kernel)
    for (int d = 0; d < kRxPasses; ++d) {                        // eight passes: the keys are back in b->khi / b->idx at the end
      RadixPass P;
      P.kin = (d & 1) ? b->khi2 : b->khi; P.vin = d == 0 ? nullptr : ((d & 1) ? b->idx2 : b->idx);
      P.kout = (d & 1) ? b->khi : b->khi2; P.vout = (d & 1) ? b->idx : b->idx2;
      P.slice_hist = b->slice_hist; P.digit = d;
      P.desc = b->rx_desc + (size_t)d * tiles * kRxBins;
      P.shift = 8 * d; P.n = n; P.status = b->counters + 3;
      hipLaunchKernelGGL(bh_radix_pass_kernel, dim3(pass_grid), dim3(kRxT), 0, s, P);
    }
    // (b->idx2 and b->klo_sorted are free here: the passes ended in b->idx, and a cold frame's second words stay in body order)
    hipLaunchKernelGGL(bh_ties_gather_kernel, grd, blk, 0, s, n, b->khi, b->idx, b->klo, b->idx2, b->klo_sorted);
    hipLaunchKernelGGL(bh_ties_place_kernel, grd, blk, 0, s, n, b->khi, b->idx, b->idx2, b->klo_sorted);
  }
  if (T.klo_by_body) hipLaunchKernelGGL(bh_bound_kernel, dim3((b->nb + kB - 1) / kB), blk, 0, s, b->khi, b->idx, b->klo, b->nb, b->bound);   // (a cold frame)
  const int block = kB * b->scan_bpt;
  hipLaunchKernelGGL(bh_lcp_scan_kernel, dim3((n + block - 1) / block), blk, 0, s, T, n, b->scan_bpt, b->lcpS, b->first_local, b->block_sum);
  hipLaunchKernelGGL(bh_nodes_kernel, grd, blk, sizeof(unsigned long long) * (size_t)(((n - 1) >> b->smp_shift) + 1), s, T, posm, n,
                     b->first_local, b->block_sum, b->scan_shift, b->first, b->lcpS, b->smp_shift);
  // ComputeMass, children before parents.  Up to kChunkSweepMaxN bodies in two launches (the cells that end inside their
  // chunk of kB bodies, then the few that do not, by one workgroup); above it a launch per level over all bodies — there the
  // one workgroup of the second launch would have more than a chunk per thread to look at per level (chunks are 256 bodies up to
  // N = 262144, 1024 bodies above) — and for that the host must know the deepest level: the frame's one wait.
  // NBODY_BH_LEVEL_SWEEPS=1: a launch per level at any size (A/B and tests; read at bh_create).
  if (b->level_sweeps || n > kChunkSweepMaxN) {
    BH_TRY(hipMemcpyAsync(b->h_counters, b->counters, sizeof(int) * kHdrWords, hipMemcpyDeviceToHost, s));
    BH_TRY(hipStreamSynchronize(s));
    int maxl = -1;
    for (int q = 0; q < kDeepSlots; ++q) maxl = std::max(maxl, b->h_counters[kHdrDeep + q]);
    for (int l = maxl; l >= 0; --l)                              // (a refused frame: every kernel from here on returns at once)
      hipLaunchKernelGGL(bh_sweep_level_kernel, grd, blk, 0, s, b->st, posm, n, b->first, b->lcpS, l, b->div_mode);
    hipLaunchKernelGGL(bh_finish_kernel, dim3(1), dim3(1), 0, s, b->st, n, keep_root);
  } else {
    const int bpt = sweep_bpt(n), nchunks = (n + kB * bpt - 1) / (kB * bpt);
    if (bpt == 1)
      hipLaunchKernelGGL(bh_sweep_chunks_kernel<kB>, dim3(nchunks), blk, 0, s, b->st, posm, n, b->first, b->lcpS, b->straddle, b->kids, nchunks, b->div_mode);
    else
      hipLaunchKernelGGL(bh_sweep_chunks_kernel<4 * kB>, dim3(nchunks), dim3(4 * kB), 0, s, b->st, posm, n, b->first, b->lcpS, b->straddle, b->kids, nchunks, b->div_mode);
    hipLaunchKernelGGL(bh_sweep_top_kernel, dim3(1), dim3(std::min(kTopT, (nchunks + 63) / 64 * 64)), 0, s, b->st, posm, n, b->straddle, b->kids,
                       nchunks, b->div_mode, keep_root);   // a thread per chunk: few waves, cheap barriers
  }
  // the walk, with the Tick's update of every body behind it (dt > 0).  One lane per body needs enough bodies to hide its loads;
  // below that, sixteen lanes per body (NBODY_BH_ROWS_MAX_N: tuning).  A slice walks its own bodies only — the count that
  // decides — and leaves neither the next frame's Size nor the positions in key order (they would be its own bodies' alone).
  static const int rows_max_n = [] { const char *e = getenv("NBODY_BH_ROWS_MAX_N"); return e && *e ? atoi(e) : kRowsMaxN; }();
  static const int wave_max_n = [] { const char *e = getenv("NBODY_BH_WAVE_MAX_N"); return e && *e ? atoi(e) : kWaveMaxN; }();
  const WalkSlice S = bh_walk_slice(b, s);
  const int nw = b->i_count;
  float4 *const pos_sorted = b->sliced ? nullptr : b->pos_sorted;
  if (b->sliced) next_size = nullptr;
  if (nw <= wave_max_n && nw <= rows_max_n && bh_wave_walk())
    hipLaunchKernelGGL(bh_walk_wave_rows_kernel, dim3((nw + kWvGT / 64 - 1) / (kWvGT / 64)), dim3(kWvGT), 0, s, b->st, posm, (float4 *)vel,
                       (float4 *)acc_v, nw, G, dt, stage, next_size, pos_sorted, S);
  else if (nw <= rows_max_n)
    hipLaunchKernelGGL(bh_walk_rows_kernel, dim3((nw + kWalkT / kWalkG - 1) / (kWalkT / kWalkG)), dim3(kWalkT), 0, s, b->st, posm, (float4 *)vel,
                       (float4 *)acc_v, nw, G, dt, stage, next_size, pos_sorted, S);
  else
    hipLaunchKernelGGL(bh_walk_lane_kernel, dim3((nw + kB - 1) / kB), blk, 0, s, b->st, posm, (float4 *)vel, (float4 *)acc_v, nw, G, dt, stage,
                       next_size, pos_sorted, S);
  b->pos_ready = !b->sliced;                                    // (every walk of all bodies writes them, moving or not)
  b->warm = true;                                               // b->khi / b->idx hold an order the next frame can start from
  return hipGetLastError();
}

// Queue one frame — CreateOctree (.cpp:74-89), the walk and (dt > 0) the Tick's update (.cpp:28-31) — on the stream; nothing waits
// for the host (systems of more than 2^20 bodies: one wait inside).  Small systems: two launches (bh_small_build_kernel,
// bh_walk_compact_kernel); larger ones: bh_large_frame.
// keep_root: the tree is a diagnostic's (nbody_compute_forces), the next frame's root centre stays what it was.
// stage (optional): the walk also writes every body's FParticle record (10 floats, body order) there — the frame's mirror.
hipError_t bh_frame(BhState *b, void *posm, void *vel, void *acc, float theta, double G, float dt, int keep_root, float *stage,
                    hipStream_t s) {
  if (!b->small) return bh_large_frame(b, posm, vel, acc, theta, G, dt, keep_root, stage, s);
  const int n = b->n;
  int P = 1;
  while (P < n) P <<= 1;
  hipLaunchKernelGGL(bh_small_build_kernel, dim3(1), dim3(kSmT), 0, s, b->st, (const float4 *)posm, n, P, b->div_mode, keep_root, theta);
  // from kSmGlobalWalkN bodies on the waves walk the tree in its global arrays (the larger systems' kernel): with a 146 KB copy of the
  // tree a CU holds one workgroup of eight bodies, and more bodies than that need second rounds (frames, LDS / global: N = 2000
  // 50.5 / 52.8 us, 3000 75.3 / 70.6, 4096 106.2 / 94.0).  (A slice walks its own bodies: their number decides.)
  const WalkSlice S = bh_walk_slice(b, s);
  const int nw = b->i_count;
  if (bh_wave_walk() && nw >= kSmGlobalWalkN)
    hipLaunchKernelGGL(bh_walk_wave_rows_kernel, dim3((nw + kWvGT / 64 - 1) / (kWvGT / 64)), dim3(kWvGT), 0, s, b->st, (float4 *)posm, (float4 *)vel,
                       (float4 *)acc, nw, G, dt, stage, (unsigned int *)nullptr, (float4 *)nullptr, S);
  else if (bh_wave_walk())
    hipLaunchKernelGGL(bh_walk_wave_compact_kernel, dim3((nw + kWvT / 64 - 1) / (kWvT / 64)), dim3(kWvT), 0, s, b->st, (float4 *)posm,
                       (float4 *)vel, (float4 *)acc, nw, G, dt, stage, S);
  else
    hipLaunchKernelGGL(bh_walk_compact_kernel, dim3((nw + kWalkT / kWalkG - 1) / (kWalkT / kWalkG)), dim3(kWalkT), 0, s, b->st, (float4 *)posm,
                       (float4 *)vel, (float4 *)acc, nw, theta, G, dt, stage, S);
  return hipGetLastError();
}

// Size (ComputeCubeSize) of the last frame bh_collect has seen
float bh_last_size(const BhState *b) { float f; unsigned int u = (unsigned int)b->h_counters[7]; memcpy(&f, &u, 4); return f; }

// Wait for the stream and read the verdict of the frames queued since the last call: *status 0 ok, 1 depth limit, 2 node pool, 4 a
// cold sort left keys out of order (an internal error); *frames = how many of them were built (a refused frame and everything queued
// behind it leave the state untouched).  A refusal is cleared here, so that the next call starts afresh.
// *status = kStatusRetry (3): the sort from the previous order gave a frame up (a bucket ran over): that frame and the ones queued
// behind it did nothing and are the caller's to queue again — it knows what they were, has their event pairs, and on several devices
// the collectives that go between them (capi.hip bh_finish, multi.hip); the state is ready for the first of them to sort cold.
hipError_t bh_collect(BhState *b, hipStream_t s, int *status, int *frames) {
  BH_TRY(hipStreamSynchronize(s));                              // (the frames' walks have left the verdict in page-locked memory: hand_verdict)
  memcpy(b->h_counters, b->h_verdict, sizeof(int) * 8);
  const int built = b->h_counters[4] - b->frames_seen;
  b->frames_seen = b->h_counters[4];
  if (frames) *frames = built;
  if (b->h_counters[4] > 0 && built > 0) {
    b->last_nodes = b->n >= 2 ? 1 + 8 * b->h_counters[1] : 1;   // the reference's count: every split makes eight children
    b->last_levels = b->h_counters[2];
  }
  if (!b->small && b->h_counters[3] == kStatusRetry) {
    b->retries += 1;
    b->warm_since_collect = 0;
    if (++b->giveups_in_row >= 2) { b->cold_left = b->cold_span; b->cold_span = std::min(2 * b->cold_span, 64); }
    b->size_ready = false; b->pos_ready = false;                 // (the given-up frame's walk left nothing)
    BH_TRY(hipMemsetAsync(b->counters + 3, 0, sizeof(int), s));
    BH_TRY(hipMemsetAsync(b->gcount, 0, sizeof(unsigned int) * 2 * (size_t)b->nb, s));   // the counts start from zero
    b->warm = false;
    *status = kStatusRetry;
    return hipSuccess;
  }
  if (b->warm_since_collect > 0 && b->h_counters[3] == 0) { b->giveups_in_row = 0; b->cold_span = 8; }   // warm frames that all went through
  b->warm_since_collect = 0;
  if (b->h_counters[3] != 0) { b->size_ready = false; b->pos_ready = false; }   // a refused frame's walk left nothing either
  *status = b->h_counters[3];
  if (*status != 0) BH_TRY(hipMemsetAsync(b->counters + 3, 0, sizeof(int), s));
  return hipSuccess;
}

// a body has been moved by something other than a frame's walk (an upload, the two-kernel update): the next frame looks at the positions itself
void bh_positions_changed(BhState *b) { b->size_ready = false; b->pos_ready = false; }
// the caller holds the position buffer from now on (nbody_device_buffer): every frame looks at the positions itself
void bh_positions_external(BhState *b) { b->external = true; }

hipError_t bh_reset_root(BhState *b, hipStream_t s) {
  b->warm = false;
  b->size_ready = false; b->pos_ready = false;                                              // a new scene: the previous order says nothing about it
  return hipMemsetAsync(b->prev_com, 0, sizeof(float) * 3, s);
}

// What DrawOctreeBoxes hands to DrawDebugBox: (Origin, Size) of the leaf holding each body, written at the body's index
hipError_t bh_leaf_boxes(BhState *b, void *out, hipStream_t s) {
  if (b->last_levels <= 0 && b->last_nodes <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bh_small_leaf_boxes_kernel, dim3((b->n + kB - 1) / kB), dim3(kB), 0, s, b->st, b->n, (float4 *)out);
  return hipGetLastError();
}

void bh_set_div_mode(BhState *b, int div_mode) { b->div_mode = div_mode ? 1 : 0; }

// The bodies in the order DrawOctreeBoxes meets their leaves (OctreeSearch.cpp:36-45: depth first, children 0..7): the
// path keys are the octant digits root to leaf, so key order IS that order.
hipError_t bh_leaf_order(BhState *b, int *out_host, hipStream_t s) {
  if (b->last_levels <= 0 && b->last_nodes <= 0) return hipErrorInvalidValue;
  BH_TRY(hipStreamSynchronize(s));
  return hipMemcpy(out_host, b->st.sidx, sizeof(unsigned int) * (size_t)b->n, hipMemcpyDeviceToHost);
}

// nodes: the reference's count (every cell of >= 2 bodies has eight children, empty ones included); levels with such cells
hipError_t bh_stats(BhState *b, hipStream_t s, int *nodes, int *levels) {
  (void)s;                                                     // the counts are those of the last frame bh_collect has seen
  if (nodes) *nodes = b->last_nodes;
  if (levels) *levels = b->last_levels;
  return hipSuccess;
}

// centre of mass of the root of the last tree built
hipError_t bh_get_tree_com(BhState *b, float out[3], hipStream_t s) {
  if (b->last_nodes <= 0) return hipErrorInvalidValue;
  BH_TRY(hipStreamSynchronize(s));
  return hipMemcpy(out, b->st.com, sizeof(float) * 3, hipMemcpyDeviceToHost);
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

// Systems up to kSmBodies bodies (the reference ships 2000): the whole CreateOctree head by ONE workgroup in ONE launch — see bh_common.h
// for the files of the theta > 0 path; reference lines cited relative to /root/reference/Source/NBody/.
#include "bh_common.h"

namespace nbody {
namespace bh {

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
  // A fresh kernel's first look at anything is a trip to memory the previous launch wrote (~1 us): the verdict, the frame count, the
  // previous root centre, this thread's bodies and its sample's body of the previous order all go out together (the arrays are there
  // whatever the verdict; nothing is written before it is known).
  const int status = T.hdr[3], frames_pre = T.hdr[4];
  const float pc0 = T.prev_com[0], pc1 = T.prev_com[1], pc2 = T.prev_com[2];
  float4 mine[kSmBodies / kSmT];                               // this thread's bodies: t, t + 1024, ...
#pragma unroll
  for (int r = 0; r < kSmBodies / kSmT; ++r) { const int i = t + r * kSmT; mine[r] = posm[min(i, n - 1)]; }
  const int smp_cap = n > 2048 ? kSmSamples : kSmSamples / 2, nsmp = min(smp_cap, n);
  int sample_body = min((int)(((long long)min(t, nsmp - 1) * n + n / 2) / nsmp), n - 1);
  const int sample_prev = (int)T.sidx[sample_body];            // (the previous frame's order; the first frame's is not used)
  if (status != 0) return;                                     // an earlier frame of this call was refused: stay there
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
#pragma unroll
  for (int r = 0; r < kSmBodies / kSmT; ++r) {
    const int i = t + r * kSmT;
    if (i < n) mx = fmaxf(mx, fmaxf(fmaxf(fabsf(mine[r].x), fabsf(mine[r].y)), fabsf(mine[r].z)));
    else mine[r] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // the sample: every (n / samples)-th body of the PREVIOUS frame's key order — bodies move little in a frame, so these stand close
  // to the quantiles of this frame's order too and the buckets come out even (any bodies would do: the first frame takes
  // every (n / samples)-th body as numbered)
  if (frames_pre > 0) sample_body = min(sample_prev, n - 1);
  if (t >= nsmp) sample_body = 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  if (lane == 0) s_red[wave] = mx;
  if (t == 0) { s_maxl = -1; s_err = 0; }
  lds_barrier();
  if (t == 0) {
    float m = s_red[0];
    for (int w = 1; w < kSmT / 64; ++w) m = fmaxf(m, s_red[w]);
    s_root[0] = pc0; s_root[1] = pc1; s_root[2] = pc2; s_root[3] = m;
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


}  // namespace bh
}  // namespace nbody

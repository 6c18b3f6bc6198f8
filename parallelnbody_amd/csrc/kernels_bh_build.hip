// From the sorted path keys to the compact preorder tree: shared digits and node numbering, node words and leaves, ComputeMass
// (OctreeSearch.h:83-97) — see bh_common.h.
#include "bh_common.h"

namespace nbody {
namespace bh {


// lcpS[i] = lcp(i - 1) (-1 at both ends), and the numbering of the nodes: body i (key order) opens max(lcp(i) - lcp(i-1), 0)
// cells and has one leaf; the exclusive scan of these counts numbers all nodes in preorder.  The scan is done HERE, in the same
// launch — no scan library, no second pass over the data: a workgroup scans its block of kB * bpt consecutive bodies
// (first_local[i] = nodes of the block's earlier bodies) and leaves the block's total in block_sum; the few block totals
// (at most kScanBlocks) are scanned again by every workgroup of the next kernel as it starts (bh_nodes_kernel).
// The second key words stay in body order (T.klo, klo_by_body): they are looked up only where two neighbours agree in the whole
// first word (bodies closer than Size / 2^21).
__global__ __launch_bounds__(kB) void bh_lcp_scan_kernel(SmallTree T, int n, int bpt, signed char *__restrict__ lcpS,
                                                         int *__restrict__ first_local, int *__restrict__ block_sum) {
  __shared__ int s_w[kB / 64];
  __shared__ int s_m[kB / 64];
  __shared__ int s_stop;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  // a frame the warm sort gave up (or one queued behind a refused frame) has no order to look at — and must keep its verdict: one
  // thread asks (other workgroups of this launch may be refusing the frame right now)
  const int wg = BH_BUILD_XCD_RUNS ? xcd_run_block() : (int)blockIdx.x;   // (one XCD's workgroups take consecutive bodies, as in every kernel of the frame)
  const int i0 = (wg * kB + t) * bpt;                          // this thread's bodies: i0 .. i0 + bpt - 1, in key order
  // (a fresh kernel's first look at anything is a trip to memory other XCDs wrote, ~1 us: this thread's first keys go out together
  // with the verdict, not behind it and its barrier)
  const unsigned long long h_pre = i0 < n ? T.khi[i0] : 0ull, hp_pre = (i0 < n && i0 > 0) ? T.khi[i0 - 1] : 0ull,
                           hn_pre = i0 + 1 < n ? T.khi[i0 + 1] : 0ull;
  if (t == 0) s_stop = T.hdr[3];
  __syncthreads();
  if (s_stop != 0) return;
  int sum = 0, deep = -1, ties = 0;
  auto shared_at = [&](unsigned long long ha, int ia, unsigned long long hb, int ib) {   // digits the bodies at sorted positions ia, ib share
    const unsigned long long x = ha ^ hb;
    if (x != 0ull) return (__clzll((long long)x) - 1) / 3;
    return shared_digits(ha, second_word(T, ia), hb, second_word(T, ib));
  };
  if (i0 < n) {
    unsigned long long h = h_pre;
    int lp = i0 > 0 ? shared_at(hp_pre, i0 - 1, h, i0) : -1;
    for (int q = 0; q < bpt; ++q) {
      const int i = i0 + q;
      if (i >= n) break;
      int ln = -1;
      unsigned long long hn = 0;
      if (i + 1 < n) { hn = q == 0 ? hn_pre : T.khi[i + 1]; ln = shared_at(h, i, hn, i + 1); }
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
      ties += ln >= kLevelsPerKey ? 1 : 0;                     // neighbours that agree in the whole first key word
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
  if (t == kB - 1) block_sum[wg] = run;
  // deepest level: one atomic per workgroup, spread over kDeepSlots words (sixteen thousand waves on ONE address queue for 0.2 ms)
  if (t == 0) {
    for (int w = 1; w < kB / 64; ++w) m = max(m, s_m[w]);
    if (m >= 0) atomicMax(&T.hdr[kHdrDeep + (blockIdx.x % kDeepSlots)], m);
  }
  // header word 6: how many neighbours agree in the whole first key word (none, in most scenes: no atomic then) — where that is a
  // large part of the bodies (a runaway body owns Size: DESIGN 4.5) the next COLD sort goes by both words (bh_large_frame)
  if (__any(ties != 0)) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ties += __shfl_xor(ties, off, 64);
    if (lane == 0) atomicAdd(&T.hdr[6], ties);
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
  // A fresh kernel's first look at anything is a trip to memory another XCD wrote, ~1 us, and the kernel is a handful of such trips:
  // this body's words are asked for together with the frame's verdict — not behind it, nor behind the table's fill and the barriers of
  // the block totals' scan (the arrays are there whatever the verdict; nothing is written before it is known).
  const int wg = BH_BUILD_XCD_RUNS ? xcd_run_block() : (int)blockIdx.x;   // (one XCD's workgroups take consecutive bodies: bh_sweep_chunks_kernel)
  const int i = wg * kB + threadIdx.x;
  const bool valid = i < n;
  const int status = T.hdr[3];
  const int lp_pre = valid ? (int)lcpS[i] : 0, ln_pre = valid ? (int)lcpS[i + 1] : 0, fl_pre = valid ? first_local[i] : 0;
  const unsigned long long h0_pre = valid ? T.khi[i] : 0ull;
  const unsigned int body_pre = valid ? T.sidx[i] : 0u;
  const unsigned int thr_pre = T.hop ? __float_as_uint(T.thr[min((int)threadIdx.x, kMaxLevels)]) : 0u;
  if (status != 0) return;                                     // a frame given up or refused: there is no order to number (uniform: set before this launch)
  const float4 pos_pre = posm[body_pre];                       // (the leaf's CoM, at the kernel's end)
  __shared__ unsigned int s_thr_bits[kMaxLevels + 2];          // the frame's acceptance thresholds by level (the key kernel's frame setup): T.hop carries them
  if (threadIdx.x <= kMaxLevels + 1) s_thr_bits[threadIdx.x] = thr_pre;   // (waited for by the scan's barriers)
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
  if (total > T.cap) {                                         // (a pool sized for 42 cells per body cannot run out below 2^25 nodes)
    if (i == 0) { T.hdr[0] = 0; T.hdr[3] = 2; }
    return;
  }
  if (i == 0) T.hdr[0] = total;
  const int lp = lp_pre, ln = ln_pre, m0 = valid ? s_base[i >> block_shift] + fl_pre : 0;
  if (valid) first[i] = m0;                                    // absolute node numbers for the kernels that follow
  if (i == n - 1) first[n] = total;
  const int open = ln > lp ? ln - lp : 0;
  const unsigned long long h0 = h0_pre;
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
#ifdef NBODY_BH_NODES_STEP_SEARCH                              // round 4's search, a step after the other, for A/B builds (make variant)
        int x = wg * kB + (threadIdx.x & ~63) + src, step = 1;   // x: a body of the cell
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
#else
        // (Seven probes side by side, then seven more — the keys are sorted and a cell's bodies contiguous, so "in the cell" is
        // monotone along them —: three round trips to memory where steps one after the other took up to thirteen, and a wave
        // waits for the longest chain among its lanes' cells.)
        const int x0 = wg * kB + (threadIdx.x & ~63) + src;   // a body of the cell
        unsigned long long probe[7];
#pragma unroll
        for (int u = 0; u < 7; ++u) probe[u] = T.khi[min(x0 + (2 << u) - 1, n - 1)];   // bodies x0 + 1, 3, 7, ... 127
        int inside = 0;                                            // (monotone: the number of probes inside = the first one outside)
#pragma unroll
        for (int u = 0; u < 7; ++u) inside += (x0 + (2 << u) - 1 < n && (probe[u] >> sh) == pre) ? 1 : 0;
        if (inside < 7) {
          int x = x0 + (1 << inside) - 1, y = min(x0 + (2 << inside) - 1, n);   // the first body behind the cell lies in (x, y]
          while (y - x > 1) {                                      // the stretch cut in eight: at most 64 -> 8 -> 1
            const int w = y - x;
#pragma unroll
            for (int u = 0; u < 7; ++u) probe[u] = T.khi[x + ((w * (u + 1)) >> 3)];    // (x <= j < y <= n)
            int lo = x, hi = y;
#pragma unroll
            for (int u = 0; u < 7; ++u) {
              const int j = x + ((w * (u + 1)) >> 3);
              if (j > x) { if ((probe[u] >> sh) == pre) lo = max(lo, j); else hi = min(hi, j); }
            }
            x = lo; y = hi;
          }
          upper = y;
        } else {
          upper = lower_bound_sampled(T.khi, n, s_smp, smp_shift, (pre + 1ull) << sh);
        }
      }
#endif
      const unsigned int past = (unsigned int)first_of(upper);
      T.meta[s_m0 + q] = ((unsigned int)l << kLevelShift) | past;
      if (T.hop) T.hop[s_m0 + q] = make_uint2(past, s_thr_bits[l]);
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
      const unsigned int past = (unsigned int)first_of(x);
      T.meta[m0 + q] = ((unsigned int)l << kLevelShift) | past;
      if (T.hop) T.hop[m0 + q] = make_uint2(past, s_thr_bits[l]);
    }
  }
  const int level = (lp > ln ? lp : ln) + 1;                   // the leaf: one level below the deepest cell the body shares
  const unsigned int body = body_pre;
  T.meta[m0 + open] = kLeafBit | ((unsigned int)level << kLevelShift) | body;
  if (T.hop) T.hop[m0 + open] = make_uint2(kLeafBit | (unsigned int)(m0 + open + 1), 0u);   // a leaf is taken whatever the distance (.h:103)
  T.com[m0 + open] = pos_pre;                                  // CenterOfMass = Position, TotalMass = Mass (.h:85-88)
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
  // One XCD's workgroups take CONSECUTIVE chunks here and consecutive bodies in bh_nodes_kernel (xcd_run_block): what a chunk reads of
  // its neighbours' nodes — the links of its straddling cells' children, chains of dependent loads — another workgroup of the same
  // XCD wrote a launch ago, through the same L2.
  const int chunk = BH_BUILD_XCD_RUNS ? xcd_run_block() : (int)blockIdx.x, base = chunk * NT, t = threadIdx.x;
  const int status = T.hdr[3];                                  // (looked at below, behind the loads that go out with it)
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
  if (status != 0) return;                                      // a refused frame (uniform: set before this launch; nothing was written so far)
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
#ifdef NBODY_BH_SWEEP_EXPERIMENT                               // A/B builds only (wrong sums): what the chains of the children's links cost
    if (m >= 0 && nchunks < 0) {
#else
    if (m >= 0) {
#endif
      const int end = (int)((in_lds ? s_meta[m - chunk_start] : T.meta[m]) & kLinkMask);   // (the cell itself lies in the chunk)
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
// (kTopT, kChunkSweepMaxN, sweep_bpt: bh_common.h)  bodies per thread of the first launch: as few as keep the chunks within one per
// thread of the second launch's workgroup —
// chunks of 256 bodies up to N = 98304, of 1024 above: fewer cells are left for the second launch (frames, 256 / 1024: N = 65536
// 192.8 / 192.4 us, 131072 213.9 / 209.6, 262144 265.4 / 252.0; beyond kTopT * kB bodies the second launch has no thread per 256-body chunk)
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
  __shared__ int s_lvl[128];
  const int status = T.hdr[3];                                 // (the verdict, the levels' counts and the deepest level go out together:
  if (threadIdx.x < 128) s_lvl[threadIdx.x] = T.lvl[threadIdx.x];   //  one trip to memory — a fresh kernel's first looks are ~1 us each)
  const int deep = deepest_level(T, &s_deep);                  // (its barrier covers s_lvl)
  if (status != 0) return;
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
  for (int l = deep; l >= 0; --l) {
    const int m = m_nx;
    const int kid[8] = {ka_nx.x, ka_nx.y, ka_nx.z, ka_nx.w, kb_nx.x, kb_nx.y, kb_nx.z, kb_nx.w};
    fetch(l - 1);
    if (m >= 0) {
      float4 ch[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {                              // up to eight loads in flight
        ch[k] = r_own;
#if !defined(NBODY_BH_TOP_EXPERIMENT) || (NBODY_BH_TOP_EXPERIMENT & 2) == 0   // A/B builds only (wrong sums; tools/top_experiment.sh): the children's loads
        if (kid[k] >= 0 && kid[k] != m_own) ch[k] = T.com[kid[k]];
#endif
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
#if !defined(NBODY_BH_TOP_EXPERIMENT) || (NBODY_BH_TOP_EXPERIMENT & 1) == 0   // ... and the levels' hand-over through memory
    if (!(here == 0 || above == 0 || (here == 1 && above == 1 && s_lvl[64 + l] == 1))) {
      __threadfence_block();
      __syncthreads();
    }
#endif
  }
  if (threadIdx.x == 0) {
    if (!keep_root) { const float4 c = T.com[0]; T.prev_com[0] = c.x; T.prev_com[1] = c.y; T.prev_com[2] = c.z; }   // .cpp:78
    T.hdr[1] = T.hdr[0] - n; T.hdr[2] = deep + 1; T.hdr[4] = T.hdr[4] + 1;
  }
}


// the instantiations bh_frame.hip launches
template __global__ void bh_sweep_chunks_kernel<kB>(SmallTree T, const float4 *__restrict__ posm, int n,
                                                             const int *__restrict__ first, const signed char *__restrict__ lcpS,
                                                             int *__restrict__ straddle, int *__restrict__ kids, int nchunks,
                                                             int div_mode);
template __global__ void bh_sweep_chunks_kernel<4 * kB>(SmallTree T, const float4 *__restrict__ posm, int n,
                                                             const int *__restrict__ first, const signed char *__restrict__ lcpS,
                                                             int *__restrict__ straddle, int *__restrict__ kids, int nchunks,
                                                             int div_mode);

}  // namespace bh
}  // namespace nbody

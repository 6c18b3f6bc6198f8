// The larger systems' path keys (Octree::Add's descent, OctreeSearch.h:50-81) and their order — see bh_common.h.
#include "bh_common.h"

namespace nbody {
namespace bh {

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


// Every element's place among all tiles: its place in its own tile + for every other tile the number of that tile's elements
// that sort before it (an earlier tile's equal keys come first: the order is (key, tile, place) — stable).  The tiles' sampled keys
// sit in LDS (lower_bound_sampled); kMergeW tiles are searched side by side.  Elements that agree with mine in the whole first
// key word (bodies closer than Size / 2^21: rare) are counted by their second words, looked up only then.
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

// Path keys of all bodies (both words, body order) and, per workgroup of kRxTile bodies, how many of its keys carry each value
// of each of the first word's eight digits: part_hist[workgroup][digit][value].  (No global atomics: contended device-scope
// atomics cost ~0.8 us each on this part.)
__global__ __launch_bounds__(kKhT) void bh_keys_hist_kernel(SmallTree T, const float4 *__restrict__ posm, int n,
                                                            const unsigned int *__restrict__ size_bits,
                                                            unsigned int *__restrict__ next_size, float theta,
                                                            unsigned long long *__restrict__ key_hi,
                                                            unsigned long long *__restrict__ key_lo,
                                                            unsigned int *__restrict__ part_hist, int both) {
  // both != 0: the second word's digit histograms as well (bh_large_frame sorts by both words where runs of equal first words are long)
  __shared__ unsigned int s_h[kRxHists][kRxBins];
  const int t = threadIdx.x;
  if (T.hdr[3] != 0) return;                                   // behind a refused frame nothing happens (bh_keys_kernel); the passes return too
  const float sz = frame_size(size_bits);
  const float o0[3] = {T.prev_com[0], T.prev_com[1], T.prev_com[2]};
  if (blockIdx.x == 0) bh_frame_setup(T, o0, sz, theta, kKhT, next_size);
  const int nh = both ? kRxHists : kRxPasses;
  for (int q = t; q < nh * kRxBins; q += kKhT) (&s_h[0][0])[q] = 0u;
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
      if (both) {
#pragma unroll
        for (int d = 0; d < kRxPasses; ++d) atomicAdd(&s_h[kRxPasses + d][(lo >> (8 * d)) & 0xFFull], 1u);
      }
    }
  }
  __syncthreads();
  unsigned int *out = part_hist + (size_t)blockIdx.x * (kRxHists * kRxBins);
  for (int q = t; q < nh * kRxBins; q += kKhT) out[q] = (&s_h[0][0])[q];
}

// The workgroups' counts added up in kRxSlices slices: slice_hist[slice][digit][value] = the counts of the workgroups slice,
// slice + kRxSlices, ...  (a pass adds the slices of its digit and scans them itself: bh_radix_pass_kernel).
__global__ __launch_bounds__(kRxBins) void bh_hist_reduce_kernel(const unsigned int *__restrict__ part_hist, int nparts,
                                                                  unsigned int *__restrict__ slice_hist) {
  const int d = blockIdx.x, sl = blockIdx.y, v = threadIdx.x;
  unsigned int c = 0;
  for (int w = sl; w < nparts; w += kRxSlices) c += part_hist[((size_t)w * kRxHists + d) * kRxBins + v];
  slice_hist[((size_t)sl * kRxHists + d) * kRxBins + v] = c;
}


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
  for (int sl = 0; sl < kRxSlices; ++sl) all += P.slice_hist[((size_t)sl * kRxHists + P.digit) * kRxBins + t];
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

// out[i] = by_body[sidx[i]]: a key word that stands in body order, laid out in the order a sort has reached (the sort by both words:
// the first words behind the eight passes on the second)
__global__ __launch_bounds__(kB) void bh_gather_words_kernel(int n, const unsigned long long *__restrict__ by_body, const unsigned int *__restrict__ sidx,
                                                             const int *__restrict__ status, unsigned long long *__restrict__ out) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n || *status != 0) return;
  out[i] = by_body[sidx[i]];
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
  const int t = threadIdx.x, w = BH_BUILD_XCD_RUNS ? xcd_run_block() : (int)blockIdx.x;   // (bh_common.h: one XCD, one eighth of the key order)
  // A fresh kernel's first look at anything is a trip to memory other XCDs wrote (~1 us): the body at this thread's place of the
  // previous order, its position, Size and the root's centre are asked for together with the verdict, not behind it and its barrier
  // (the arrays are there whatever the verdict; nothing is written before it is known).
  const int i = w * kB + t;
  const bool valid = i < n;
  const unsigned int body_pre = valid ? prev_idx[i] : 0u;
  const float4 pos_pre = (valid && prev_pos != nullptr) ? prev_pos[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  if (t == 0) s_stop = T.hdr[3];                               // (one thread asks: other workgroups of this launch may be giving the frame up)
  const float sz = frame_size(size_bits);
  float o[3] = {T.prev_com[0], T.prev_com[1], T.prev_com[2]};
  __syncthreads();
  if (s_stop != 0) return;                                     // a frame before this one was refused: nothing of this one happens
  if (w == 0) bh_frame_setup(T, o, sz, theta, kB, next_size);
  // bucket(h) = the largest j in 1 .. nb - 1 with boundary j <= h, or 0; the window: boundaries jlo .. jhi around this workgroup's own
  const int mid_j = (int)(((long long)w * kB + kB / 2) / kWarmMu);   // the bucket this workgroup's places lie in
  const int jlo = max(1, mid_j - (kWarmWin / 2 - 1)), jhi = min(nb - 1, mid_j + kWarmWin / 2);
  const int nwin = jhi - jlo + 1;
  if (t < nwin) { s_b[t] = bound[jlo + t]; s_bl[t] = bound[nb + jlo + t]; }   // (the previous order's keys at places 224 j, gathered by its sort: bh_bucket_sort_kernel)
  if (t <= kWarmWin) s_cnt[t] = 0u;
  __syncthreads();
  unsigned int body = 0u, local = 0u;
  unsigned long long hi = 0ull, lo = 0ull;
  int bucket = 0, q = -1;
  if (valid) {
    body = body_pre;
    // (the previous frame's walk left the positions in its key order — this kernel's order — where nothing else has moved a body
    // since: a coalesced read instead of a 16-byte record out of every 64-byte sector)
    const float4 p = prev_pos != nullptr ? pos_pre : posm[body];
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
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6, b = BH_BUILD_XCD_RUNS ? xcd_run_block() : (int)blockIdx.x;
  // A fresh kernel's first look at anything is a trip to memory other XCDs wrote (~1 us): the verdict, the counts and this thread's
  // first slot of the bucket go out together (a slot beyond the bucket's count holds an older frame's words: loaded, not used; nothing
  // is written before the verdict is known).
  const int status = T.hdr[3];
  const size_t slot_pre = (size_t)b * kWarmCap + min(t, kWarmCap - 1);
  const unsigned long long hi_pre = slot_hi[slot_pre], lo_pre = slot_lo[slot_pre];
  const unsigned int body_pre = slot_idx[slot_pre];
  // where the bucket starts: the counts of the buckets before it
  unsigned int sum = 0u;
  for (int j = t; j < b; j += kBsT) sum += gcount[j];
  const int cnt = (int)gcount[b];
  if (status != 0) return;                                     // the frame was given up (or an earlier one refused)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (lane == 0) s_part[wave] = sum;
  if (t == 0) gcount_next[b] = 0u;                             // the next frame counts there
  int P = 64;
  while (P < cnt) P <<= 1;
  for (int e = t; e < P; e += kBsT) {
    const bool in = e < cnt, pre = e == t && t < kWarmCap;
    s_hi[0][e] = in ? (pre ? hi_pre : slot_hi[(size_t)b * kWarmCap + e]) : ~0ull;
    s_lo[e] = in ? (pre ? lo_pre : slot_lo[(size_t)b * kWarmCap + e]) : ~0ull;
    s_ix[0][e] = (unsigned short)e;
    s_body[e] = in ? (pre ? body_pre : slot_idx[(size_t)b * kWarmCap + e]) : 0xFFFFFFFFu;
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


// the instantiations bh_frame.hip launches
template __global__ void bh_tile_sort_kernel<1024>(int n, const unsigned long long *__restrict__ key_hi,
                                                            const unsigned long long *__restrict__ key_lo,
                                                            unsigned long long *__restrict__ tile_hi, unsigned int *__restrict__ tile_idx);
template __global__ void bh_tile_sort_kernel<2048>(int n, const unsigned long long *__restrict__ key_hi,
                                                            const unsigned long long *__restrict__ key_lo,
                                                            unsigned long long *__restrict__ tile_hi, unsigned int *__restrict__ tile_idx);
template __global__ void bh_tile_sort_kernel<4096>(int n, const unsigned long long *__restrict__ key_hi,
                                                            const unsigned long long *__restrict__ key_lo,
                                                            unsigned long long *__restrict__ tile_hi, unsigned int *__restrict__ tile_idx);
template __global__ void bh_bucket_sort_kernel<512>(SmallTree T, int n, int nb, const unsigned int *__restrict__ gcount,
                                                              unsigned int *__restrict__ gcount_next,
                                                              const unsigned long long *__restrict__ slot_hi,
                                                              const unsigned int *__restrict__ slot_idx,
                                                              const unsigned long long *__restrict__ slot_lo,
                                                              unsigned long long *__restrict__ out_hi, unsigned int *__restrict__ out_idx,
                                                              unsigned long long *__restrict__ out_lo, unsigned long long *__restrict__ bound);
template __global__ void bh_bucket_sort_kernel<256>(SmallTree T, int n, int nb, const unsigned int *__restrict__ gcount,
                                                              unsigned int *__restrict__ gcount_next,
                                                              const unsigned long long *__restrict__ slot_hi,
                                                              const unsigned int *__restrict__ slot_idx,
                                                              const unsigned long long *__restrict__ slot_lo,
                                                              unsigned long long *__restrict__ out_hi, unsigned int *__restrict__ out_idx,
                                                              unsigned long long *__restrict__ out_lo, unsigned long long *__restrict__ bound);

}  // namespace bh
}  // namespace nbody

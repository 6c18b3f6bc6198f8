// The host side of the theta > 0 path: the state of a context (BhState), one frame queued on the stream (bh_frame), the verdict
// (bh_collect) — see bh_common.h for the kernels' files.
#include <cstdlib>
#include <cstring>

#include <algorithm>

#include "bh_common.h"

namespace nbody {

using namespace bh;

struct BhState {
  int n = 0, node_cap = 0;
  bool small = false;          // n <= kSmBodies: one workgroup builds the tree in LDS (bh_small_build_kernel)
  SmallTree st{};              // the compact tree (either path)
  int small_global_walk_n = kSmGlobalWalkN;   // small systems: from here on the walk reads the tree's global arrays (bh_frame)
  int rows_max_n = kRowsMaxN, wave_max_n = kWaveMaxN;   // larger systems: the largest walked with windows / with a wave per body (bh_large_frame)
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
  int ties_seen = -1;                      // neighbours that agreed in the whole first key word in the last frame built (header word 6); -1: none built yet
  int sort_both = -1;                      // NBODY_BH_SORT_BOTH at creation: 0 / 1 force the cold radix sort by the first word / by both; -1: by what was seen
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
  bool warm_off = false;       // NBODY_BH_WARM_SORT=0 at creation: the cold sorts every frame (A/B measurements, tests)
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
  { const char *e = getenv("NBODY_BH_SMALL_GLOBAL_WALK_N"); b->small_global_walk_n = e && *e ? atoi(e) : kSmGlobalWalkN; }   // (tuning; read at every bh_create)
  if (b->small) return hipSuccess;
  BH_TRY(hipMalloc(&t.hop, sizeof(uint2) * (size_t)b->node_cap));   // (every word a walk reads is written by the frame's bh_nodes_kernel)
  BH_TRY(hipMalloc(&b->size_words, 2 * kSizeSlots * sizeof(unsigned int)));
  BH_TRY(hipMemset(b->size_words, 0, 2 * kSizeSlots * sizeof(unsigned int)));
  BH_TRY(hipMalloc(&b->khi2, sizeof(unsigned long long) * n));
  BH_TRY(hipMalloc(&b->idx2, sizeof(unsigned int) * n));
  t.klo_by_body = 1;           // the second key words stay where the key kernel put them (second_word())
  b->radix = n > bh_merge_max_n();
  { const char *e = getenv("NBODY_BH_LEVEL_SWEEPS"); b->level_sweeps = e && e[0] == '1'; }
  { const char *e = getenv("NBODY_BH_WARM_SORT"); b->warm_off = e && e[0] == '0'; }
  { const char *e = getenv("NBODY_BH_ROWS_MAX_N"); b->rows_max_n = e && *e ? atoi(e) : kRowsMaxN; }   // which walk for which size: read at every
  { const char *e = getenv("NBODY_BH_WAVE_MAX_N"); b->wave_max_n = e && *e ? atoi(e) : kWaveMaxN; }   // bh_create, like the sorts' switches (tests, tuning)
  { const char *e = getenv("NBODY_BH_SORT_BOTH"); b->sort_both = e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : -1; }   // read at every bh_create, like the sorts' switch
  b->tile_size = bh_tile_size(n);
  { const int budget = n <= 131072 ? kNodeSmp / 4 : 512;       // many workgroups: a smaller table each (its fill is traffic; only cells of more than 127 bodies look at it)
    while ((((n - 1) >> b->smp_shift) + 1) > budget) ++b->smp_shift; }
  if (b->radix) {
    const size_t tiles = (size_t)((n + kRxTile - 1) / kRxTile);
    BH_TRY(hipMalloc(&b->part_hist, sizeof(unsigned int) * tiles * kRxHists * kRxBins));
    BH_TRY(hipMalloc(&b->slice_hist, sizeof(unsigned int) * kRxSlices * kRxHists * kRxBins));
    b->rx_desc_bytes = sizeof(unsigned int) * tiles * kRxHists * kRxBins;
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
                  b->counters, b->root, b->prev_com, b->st.com, b->st.meta, b->st.hop, b->st.leaf_level, b->st.thr, b->st.lvl, b->st.clocks};
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
  const bool warm_now = b->warm && !b->warm_off && b->cold_left == 0;
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
    // Runs of equal FIRST words are put right by bh_ties_place_kernel, whose rank search is quadratic in a run's length — a handful
    // of bodies in any ordinary scene.  Where the last frame built found them all over (its count stands in header word 6: a runaway
    // body owns Size and every other body shares one cell of level 21) the sort goes by BOTH words instead: eight passes on the second
    // words, the first words gathered into that order, eight passes on them — a stable least-significant-digit sort of the whole
    // 126-bit key, no ties left to place (2^20 bodies in one run: ~0.3 ms instead of ~0.1 s).
    // ... and where no frame of this scene has been built yet (a new state: nothing is known) the sixteen passes are the insurance:
    // ~0.1 ms more on a first frame of 2^20 bodies against ~0.5 s if the scene turns out to be such a run.  NBODY_BH_SORT_BOTH=0|1
    // (read at creation): always the one or the other (tests, A/B).
    const bool both = b->sort_both >= 0 ? b->sort_both != 0 : (b->ties_seen < 0 || b->ties_seen > n / 16);
    BH_TRY(hipMemsetAsync(b->rx_desc, 0, both ? b->rx_desc_bytes : b->rx_desc_bytes / 2, s));
    hipLaunchKernelGGL(bh_keys_hist_kernel, dim3(tiles), dim3(kKhT), 0, s, T, posm, n, size_bits, nxt, theta, b->khi, b->klo, b->part_hist, both ? 1 : 0);
    hipLaunchKernelGGL(bh_hist_reduce_kernel, dim3(both ? kRxHists : kRxPasses, kRxSlices), dim3(kRxBins), 0, s, b->part_hist, tiles, b->slice_hist);
    const int pass_grid = std::min(tiles, b->rx_resident);        // all workgroups of a pass resident at once (bh_radix_pass_kernel)
    auto pass = [&](int hist, const unsigned long long *kin, const unsigned int *vin, unsigned long long *kout, unsigned int *vout) {
      RadixPass P;
      P.kin = kin; P.vin = vin; P.kout = kout; P.vout = vout;
      P.slice_hist = b->slice_hist; P.digit = hist;
      P.desc = b->rx_desc + (size_t)hist * tiles * kRxBins;
      P.shift = 8 * (hist % kRxPasses); P.n = n; P.status = b->counters + 3;
      hipLaunchKernelGGL(bh_radix_pass_kernel, dim3(pass_grid), dim3(kRxT), 0, s, P);
    };
    if (!both) {
      for (int d = 0; d < kRxPasses; ++d)                        // eight passes: the keys are back in b->khi / b->idx at the end
        pass(d, (d & 1) ? b->khi2 : b->khi, d == 0 ? nullptr : ((d & 1) ? b->idx2 : b->idx), (d & 1) ? b->khi : b->khi2, (d & 1) ? b->idx : b->idx2);
      // (b->idx2 and b->klo_sorted are free here: the passes ended in b->idx, and a cold frame's second words stay in body order)
      hipLaunchKernelGGL(bh_ties_gather_kernel, grd, blk, 0, s, n, b->khi, b->idx, b->klo, b->idx2, b->klo_sorted);
      hipLaunchKernelGGL(bh_ties_place_kernel, grd, blk, 0, s, n, b->khi, b->idx, b->idx2, b->klo_sorted);
    } else {
      // the second words (b->klo stays what it is: body order): klo -> khi2 / idx2 -> klo_sorted / idx -> ... -> klo_sorted / idx
      for (int d = 0; d < kRxPasses; ++d)
        pass(kRxPasses + d, d == 0 ? b->klo : ((d & 1) ? b->khi2 : b->klo_sorted), d == 0 ? nullptr : ((d & 1) ? b->idx2 : b->idx),
             (d & 1) ? b->klo_sorted : b->khi2, (d & 1) ? b->idx : b->idx2);
      // the first words in that order, then eight stable passes on them: khi2 / idx -> klo_sorted / idx2 -> ... -> khi / idx
      hipLaunchKernelGGL(bh_gather_words_kernel, grd, blk, 0, s, n, b->khi, b->idx, b->counters + 3, b->khi2);
      for (int d = 0; d < kRxPasses; ++d)
        pass(d, (d & 1) ? b->klo_sorted : b->khi2, (d & 1) ? b->idx2 : b->idx, d == kRxPasses - 1 ? b->khi : ((d & 1) ? b->khi2 : b->klo_sorted),
             (d & 1) ? b->idx : b->idx2);
    }
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
  const int rows_max_n = b->rows_max_n, wave_max_n = b->wave_max_n;   // (NBODY_BH_ROWS_MAX_N / NBODY_BH_WAVE_MAX_N as they stood at creation)
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
  else if (nw >= kLaneTwoStepsMinN)
    hipLaunchKernelGGL(bh_walk_lane_kernel<true>, dim3((nw + kB - 1) / kB), blk, 0, s, b->st, posm, (float4 *)vel, (float4 *)acc_v, nw, G, dt,
                       stage, next_size, pos_sorted, S);
  else
    hipLaunchKernelGGL(bh_walk_lane_kernel<false>, dim3((nw + kB - 1) / kB), blk, 0, s, b->st, posm, (float4 *)vel, (float4 *)acc_v, nw, G, dt,
                       stage, next_size, pos_sorted, S);
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
  if (bh_wave_walk() && nw >= b->small_global_walk_n)
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
  if (built > 0) b->ties_seen = b->h_counters[6];
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
  b->ties_seen = -1;
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

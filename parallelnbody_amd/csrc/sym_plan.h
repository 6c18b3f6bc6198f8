// Work plan of the symmetric (each unordered pair once) force pass: which workgroup evaluates which body pairs, where
// its partial sums go, and in which order they are added up.  Host-only C++ (no HIP): built once per context by
// nbody_create, unit-tested on the CPU through nbody_sym_plan_describe (include/nbody.h).
//
// The bodies are cut into T blocks of BI = 256 * (bodies per lane) bodies — one i-set: what a workgroup of four waves
// holds in registers.  Block pair {a, b} belongs to row a if b lies in the forward half of the ring of blocks from a
// (circulant assignment: every row, hence every rank, which owns a run of consecutive rows, gets the same amount of
// work).  Row a is therefore: its own block (every ordered pair, one-sided) and ONE contiguous ring range of about
// n/2 j-bodies (every pair once, both bodies credited).  A work item is a strip of a row: the row's i-set against a
// contiguous run of 64-body subtiles.  Strip lengths follow guided self-scheduling — each is 1/(K * slots) of the work
// still to hand out — and the hardware's in-order workgroup dispatcher is the queue: long strips first (few partial
// rows: small footprint), short ones last (no tail).
//
// Partial sums live in one pool of float4/double4 segments: every item owns a segment of BI elements for its i-side
// sums and (symmetric items) one of 64 * n_sub elements for its j-side sums.  Two CSR lists over 64-body granules say
// which segments cover a granule, in item order: the order in which reduce_j_kernel / update_sym_kernel add them.  No
// atomics anywhere, so a pass is bit-reproducible for a given plan.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace nbody {

struct SymItem {          // 32 bytes, read by the force kernels with scalar loads
  int32_t i0;             // first body of the i-set (multiple of bi)
  int32_t j0;             // first j body of the strip (multiple of 64)
  int32_t n_sub;          // 64-body subtiles in the strip (>= 1)
  int32_t flags;          // kSymOneSided, kSymNoJSide
  uint32_t slot_i;        // pool element where the i-side sums of bodies [i0, i0 + bi) start
  uint32_t slot_j;        // pool element where the j-side sums of bodies [j0, j0 + 64 n_sub) start (symmetric items)
  int32_t k0;             // even-share plans: first step (0..60, multiple of 4) of the item's FIRST subtile; 0 in the guided plans
  int32_t k_skip;         // even-share plans: steps of the item's LAST subtile left to the next item (it ends at step 64 - k_skip)
                          // (both are the guided planner's scratch while it works and zero when it is done)
};
enum { kSymOneSided = 1,      // the strip lies in the i-set's own block (guided plans; an even-share item: its first subtile does)
       kSymNoJSide = 2,       // the item writes no j-side sums (own_mode 0 only)
       kSymEven = 4 };        // item of an even-share plan: its subtiles follow the row's ring order (they may leave the own block
                              // and wrap from the system's last granule to granule 0), the first and last may be partial

struct SymPlan {
  int bi = 0;             // bodies per i-set
  int T = 0;              // blocks in the system = ceil(n_total / bi)
  int n_pad = 0;          // T * bi: size of the prescaled position array (zero-mass padding beyond n_total)
  int n_gran = 0;         // 64-body granules of the system = ceil(n_total / 64)
  int own_block0 = 0, own_blocks = 0;
  int own_gran0 = 0, own_grans = 0;
  int n_src = 1;          // ranks sharing the bodies
  uint64_t pool_elems = 0;
  bool even = false;      // an even-share plan (build_sym_plan_even): the kernels take the items' k0 / k_skip and ring order
  int n_local = 0;        // items [0, n_local): strips inside the own slice (all of them when the context owns all bodies)
  // pool phases (build_sym_plan's j_budget): phase p = items [phase_item0[p], phase_item0[p + 1]) of the launch order; its
  // j-side lists are j_ptr[p * (n_gran + 1) ...] (absolute positions in j_off).  One phase unless a budget was given and exceeded.
  std::vector<int> phase_item0;
  std::vector<SymItem> items;
  std::vector<uint32_t> i_ptr, i_off;   // CSR over OWN granules: i-side segments (+ offset of the granule inside them)
  std::vector<uint32_t> j_ptr, j_off;   // CSR over ALL granules: j-side segments
};

// slots: workgroups the chip holds at a time; k_guided: a strip is 1/(k_guided * slots) of the remaining work;
// min_sub: shortest strip, in subtiles; own_mode: how the kernel treats strips inside the i-set's own block — 1 (fp32):
// register pairs above the subtile's own pair symmetric, that pair one-sided, j-side sums written; 2 (fp64): the same
// slot by slot; 0: one-sided throughout, no j-side sums (no kernel does that any more; kept for the cost model's tests).  Returns false (and says why) when the owned range does not fit the plan.
// j_budget_elems: 0 = one pass whatever the pool's size; otherwise the j-side segments of a phase may take at most that many
// pool elements, and a system whose j-side segments exceed it is run in several phases that share one area (SymPlan).
bool build_sym_plan(int n_total, int i_begin, int i_count, int bi, int slots, double k_guided, int min_sub, int own_mode,
                    SymPlan *out, std::string *err, uint64_t j_budget_elems = 0, int max_sub_arg = 0);

// The even-share plan (round 5; fp32, one context owning all bodies): exactly n_items work items — as many as the chip holds
// workgroups at a time — of EQUAL cost, so that one round of workgroups starts together and ends together.  Mid-size systems
// (N = 16k ... 131k) step in 50 us ... 2 ms: the guided plan's two or three rounds of strips pay an i-side segment, a prologue
// and an epilogue per strip, and a quantum of one subtile (64 steps x bodies per lane: 10 us of a SIMD) is a tenth of what a
// slot gets.  Here a row — the i-set against its own block and then its forward blocks, ONE run of subtiles in ring order —
// is cut at equal cumulative cost, to four steps of a subtile's 64: an item is {first subtile, first step, subtiles touched,
// steps left off the last}; the rows get items in proportion to their cost (largest remainders).  The rest — segments, lists,
// who adds what in which order — is the guided plan's.  cost_sym / cost_one / cost_move: SIMD cycles of one step of one
// symmetric register pair, of the one-sided pair of an own-block subtile, and of the travelling sums' moves (the issue model,
// DESIGN 4.1): an own-block subtile in register pair pc's slots works NP - pc pairs, the first of them one-sided.
// own_pct: what a step of an own-block subtile costs, in per cent of the model's figure — measured: its loops keep fewer
// steps in flight and their one-sided pair is a chain of dependent operations; whole steps are shortest between 120 and 140
// at every size and in both forms of the kernel (profiles/r05_even_share_knobs.txt).
constexpr int kSymEvenOwnPct = 130;
bool build_sym_plan_even(int n_total, int bi, int n_items, SymPlan *out, std::string *err, int cost_sym = 82, int cost_one = 74,
                         int cost_move = 26, int own_pct = kSymEvenOwnPct);

}  // namespace nbody

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
  int32_t reserved0, reserved1;   // zero (the planner's scratch while it works)
};
enum { kSymOneSided = 1,      // the strip lies in the i-set's own block
       kSymNoJSide = 2 };     // the item writes no j-side sums (own_mode 0 only)

struct SymPlan {
  int bi = 0;             // bodies per i-set
  int T = 0;              // blocks in the system = ceil(n_total / bi)
  int n_pad = 0;          // T * bi: size of the prescaled position array (zero-mass padding beyond n_total)
  int n_gran = 0;         // 64-body granules of the system = ceil(n_total / 64)
  int own_block0 = 0, own_blocks = 0;
  int own_gran0 = 0, own_grans = 0;
  int n_src = 1;          // ranks sharing the bodies
  uint64_t pool_elems = 0;
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

}  // namespace nbody

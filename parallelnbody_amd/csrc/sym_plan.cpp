// Host-side planner of the symmetric force pass (see sym_plan.h).  No device code.
#include "sym_plan.h"

#include "../../include/nbody.h"

#include <algorithm>
#include <cstdio>

namespace nbody {

namespace {

// Does block a own the block pair {a, b}?  b within the forward half of the ring of T blocks; the antipodal pair of an
// even ring goes to its smaller index if that is even, to the larger one if it is odd — so that every run of
// consecutive rows gets the same number of pairs.
bool ring_assigned(int a, int b, int T) {
  if (a == b) return false;
  int d = b - a;
  if (d < 0) d += T;
  if (2 * d < T) return true;
  if (2 * d != T) return false;
  const int lo = a < b ? a : b;
  return ((lo & 1) == 0) == (a == lo);
}

struct Range { int row, j_sub0, n_sub, one_sided; };

// Cost of one 64-body subtile against an i-set, in lane-slot steps (a lane holds S = bi / 256 bodies).  A symmetric strip
// works all S slots.  Inside the own block (own_mode 1, the fp32 kernels) the register pairs — two slots — above the
// subtile's own pair work symmetrically and its own pair one-sided, the pairs below are idle: 2 (NP - pc); own_mode 2
// (the fp64 kernel) does the same slot by slot: S - slot; own_mode 0 would run the whole i-set one-sided.
inline int subtile_cost(const Range &r, int sub_in_range, int bi, int own_mode) {
  const int S = bi / 256;
  if (!r.one_sided || own_mode == 0) return S;
  const int off = (r.j_sub0 + sub_in_range) * 64 - r.row * bi;     // body offset inside the block
  if (own_mode == 2) return S - (off >> 8);                        // slot-granular (fp64 kernel)
  return S < 2 ? S : S - 2 * (off >> 9);
}

bool fail(std::string *err, const char *msg) {
  if (err) *err = msg;
  return false;
}

}  // namespace

bool build_sym_plan(int n_total, int i_begin, int i_count, int bi, int slots, double k_guided, int min_sub, int own_mode,
                    SymPlan *out, std::string *err, uint64_t j_budget_elems, int max_sub_arg) {
  if (n_total <= 0 || i_count <= 0 || i_begin < 0 || i_begin + i_count > n_total) return fail(err, "bad body range");
  if (bi < 64 || bi % 64 != 0) return fail(err, "bodies per i-set must be a multiple of 64");
  if (slots < 1 || !(k_guided >= 0.5) || min_sub < 1) return fail(err, "bad scheduling parameters");
  SymPlan P;
  P.bi = bi;
  P.T = (n_total + bi - 1) / bi;
  P.n_pad = P.T * bi;
  P.n_gran = (n_total + 63) / 64;
  const bool all = i_count == n_total;
  if (all) {
    P.own_block0 = 0; P.own_blocks = P.T; P.n_src = 1;
  } else {
    // sharded: equal slices, each a whole number of blocks (every rank must cut the ring the same way)
    if (n_total % i_count != 0 || i_begin % i_count != 0 || i_count % bi != 0)
      return fail(err, "sharded symmetric contexts need equal slices that are a multiple of the i-set size");
    P.own_block0 = i_begin / bi; P.own_blocks = i_count / bi; P.n_src = n_total / i_count;
  }
  P.own_gran0 = i_begin / 64;
  P.own_grans = (i_count + 63) / 64;
  const int sub_per_block = bi / 64;

  // the rows' ranges, in 64-body subtiles, clipped to the bodies that exist
  std::vector<Range> ranges;
  long long total_cost = 0;
  for (int a = P.own_block0; a < P.own_block0 + P.own_blocks; ++a) {
    const int d0 = a * sub_per_block;
    if (d0 >= P.n_gran) continue;                                    // (cannot happen: the last block holds a body)
    const int dn = std::min(sub_per_block, P.n_gran - d0);
    ranges.push_back({a, d0, dn, 1});
    for (int k = 0; k < dn; ++k) total_cost += subtile_cost(ranges.back(), k, bi, own_mode);
    int h = 0;
    while (h + 1 < P.T && ring_assigned(a, (a + h + 1) % P.T, P.T)) ++h;
    for (int d = h + 1; d < P.T; ++d)
      if (ring_assigned(a, (a + d) % P.T, P.T)) return fail(err, "internal: forward blocks not contiguous");
    if (h == 0) continue;
    const long long ring = (long long)P.T * sub_per_block;
    const long long s = (long long)((a + 1) % P.T) * sub_per_block, len = (long long)h * sub_per_block;
    const long long piece[2][2] = {{s, std::min(s + len, ring)}, {0, std::max(0LL, s + len - ring)}};
    for (const auto &pc : piece) {
      const long long lo = pc[0], hi = std::min<long long>(pc[1], P.n_gran);
      if (hi <= lo) continue;
      // a sharded context cuts its ranges where the own slice begins and ends: a strip then lies either wholly inside the
      // own slice — it needs no other rank's positions and can run while they are still being gathered — or outside it
      long long cut[4] = {lo, hi, hi, hi};
      int n_cut = 1;
      if (!all) {
        const long long o0 = P.own_gran0, o1 = (long long)P.own_gran0 + P.own_grans;
        for (long long c : {o0, o1}) if (c > cut[n_cut - 1] && c < hi) cut[n_cut++] = c;
      }
      cut[n_cut] = hi;
      for (int q = 0; q < n_cut; ++q) {
        ranges.push_back({a, (int)cut[q], (int)(cut[q + 1] - cut[q]), 0});
        total_cost += (cut[q + 1] - cut[q]) * (bi / 256);
      }
    }
  }

  // guided self-scheduling: each strip takes 1/(k * slots) of the cost still to hand out, but never more than 1024 subtiles:
  // a lane adds a strip's terms one after the other in fp32, and the first strips of a system of millions of bodies would
  // otherwise be chains of several hundred thousand terms (N = 2^23: sampled error 8.6e-4 -> 4.4e-7 with the cap, and the pass
  // 24 % faster; N = 2^21: 1.6e-6 -> 9e-7, 3.5 % faster — tools/cap_sweep.py).  Up to N = 2^20 no strip is that long.
  const int max_sub = max_sub_arg > 0 ? max_sub_arg : 1024;
  long long remaining = total_cost;
  uint64_t pool = 0;
  const double kp = k_guided * slots;
  for (const Range &r : ranges) {
    int pos = 0;
    while (pos < r.n_sub) {
      const long long target = (long long)((double)remaining / kp);
      long long cost = 0;
      int n = 0;
      while (pos + n < r.n_sub && n < max_sub && (n < min_sub || cost + subtile_cost(r, pos + n, bi, own_mode) <= target)) {
        cost += subtile_cost(r, pos + n, bi, own_mode);
        ++n;
      }
      if (n >= 4 && n < r.n_sub - pos) {                             // whole 256-body tiles except at a range's end
        for (int k = n - n % 4; k < n; ++k) cost -= subtile_cost(r, pos + k, bi, own_mode);
        n -= n % 4;
      }
      if (r.n_sub - pos - n < min_sub)                              // no slivers
        for (; pos + n < r.n_sub; ++n) cost += subtile_cost(r, pos + n, bi, own_mode);
      SymItem it{};
      it.i0 = r.row * bi;
      it.j0 = (r.j_sub0 + pos) * 64;
      it.n_sub = n;
      it.flags = r.one_sided ? kSymOneSided : 0;
      if (r.one_sided && own_mode == 0) it.flags |= kSymNoJSide;      // does the item produce j-side sums
      it.k_skip = (int32_t)P.items.size();                         // place in the canonical order (= the order of summation)
      P.items.push_back(it);
      pos += n;
      remaining -= cost;
    }
  }
  // Launch order: the items whose strip lies inside the own slice first (stable).  The lists below name pool segments and
  // are built in the canonical order, so the order of launch does not touch the order of summation.
  {
    const int o0 = P.own_gran0 * 64, o1 = o0 + P.own_grans * 64;
    auto local = [&](const SymItem &it) { return it.j0 >= o0 && it.j0 + it.n_sub * 64 <= o1; };
    std::stable_partition(P.items.begin(), P.items.end(), local);
    P.n_local = 0;
    while (P.n_local < (int)P.items.size() && local(P.items[(size_t)P.n_local])) ++P.n_local;
  }
  // Pool segments.  One pass (the usual case): every item owns an i-side and a j-side segment, laid out in canonical order.
  // A j_budget (very large systems: the j-side segments grow as N^2 / (2 bi)) cuts the launch order into PHASES whose j-side
  // segments share one area of at most that size: the items of a phase run, their j-side sums are folded into `send`, the
  // next phase reuses the area.  The i-side segments (N / bi per strip... a few GB) stay to the end.
  std::vector<size_t> canon(P.items.size());                         // canonical position -> launch position
  for (size_t k = 0; k < P.items.size(); ++k) canon[(size_t)P.items[k].k_skip] = k;
  uint64_t total_j = 0;
  for (const SymItem &it : P.items) if (!(it.flags & kSymNoJSide)) total_j += (uint64_t)it.n_sub * 64;
  const bool phased = j_budget_elems != 0 && total_j > j_budget_elems;
  if (!phased) {
    for (size_t c = 0; c < canon.size(); ++c) {
      SymItem &it = P.items[canon[c]];
      if (pool + (uint64_t)bi + (uint64_t)it.n_sub * 64 >= (1ull << 32)) return fail(err, "partial-sum pool exceeds 2^32 elements");
      it.slot_i = (uint32_t)pool; pool += (uint64_t)bi;
      if (!(it.flags & kSymNoJSide)) { it.slot_j = (uint32_t)pool; pool += (uint64_t)it.n_sub * 64; }
      it.k0 = 0;
    }
    P.phase_item0.assign({0, (int)P.items.size()});
  } else {
    for (SymItem &it : P.items) { it.slot_i = (uint32_t)pool; pool += (uint64_t)bi; }
    if (pool + j_budget_elems >= (1ull << 32)) return fail(err, "partial-sum pool exceeds 2^32 elements");
    const uint64_t jbase = pool;
    uint64_t used = 0, area = 0;
    int phase = 0;
    P.phase_item0.assign(1, 0);
    for (size_t k = 0; k < P.items.size(); ++k) {
      SymItem &it = P.items[k];
      const uint64_t need = (it.flags & kSymNoJSide) ? 0 : (uint64_t)it.n_sub * 64;
      if (need > j_budget_elems) return fail(err, "a strip's j-side segment exceeds the pool budget");
      if (used + need > j_budget_elems) { ++phase; used = 0; P.phase_item0.push_back((int)k); }
      it.slot_j = (uint32_t)(jbase + used);
      used += need;
      area = std::max(area, used);
      it.k0 = phase;
    }
    P.phase_item0.push_back((int)P.items.size());
    pool = jbase + area;
  }
  P.pool_elems = pool;
  const int n_phases = (int)P.phase_item0.size() - 1;

  // CSR over the own granules: i-side segments, in item order (a row's items are contiguous)
  P.i_ptr.assign((size_t)P.own_grans + 1, 0);
  for (const SymItem &it : P.items) {
    const int g0 = it.i0 / 64 - P.own_gran0;
    for (int g = g0; g < g0 + sub_per_block && g < P.own_grans; ++g) P.i_ptr[(size_t)g + 1] += 1;
  }
  for (int g = 0; g < P.own_grans; ++g) P.i_ptr[(size_t)g + 1] += P.i_ptr[(size_t)g];
  P.i_off.assign(P.i_ptr.back(), 0);
  {
    std::vector<uint32_t> fill(P.i_ptr.begin(), P.i_ptr.end() - 1);
    for (size_t c = 0; c < canon.size(); ++c) {
      const SymItem &it = P.items[canon[c]];
      const int g0 = it.i0 / 64 - P.own_gran0;
      for (int g = g0; g < g0 + sub_per_block && g < P.own_grans; ++g)
        P.i_off[fill[(size_t)g]++] = it.slot_i + (uint32_t)(g - g0) * 64u;
    }
  }
  // CSR over all granules: j-side segments — one list set per phase, (n_gran + 1) pointers each, offsets into one j_off
  {
    const size_t stride = (size_t)P.n_gran + 1;
    P.j_ptr.assign(stride * (size_t)n_phases, 0);
    for (const SymItem &it : P.items)
      if (!(it.flags & kSymNoJSide))
        for (int k = 0; k < it.n_sub; ++k) P.j_ptr[stride * (size_t)it.k0 + (size_t)(it.j0 / 64 + k) + 1] += 1;
    uint64_t run = 0;
    for (int ph = 0; ph < n_phases; ++ph) {                           // counts -> absolute positions in j_off
      uint32_t *ptr = P.j_ptr.data() + stride * (size_t)ph;
      uint64_t prev = run;
      for (int g = 0; g <= P.n_gran; ++g) { const uint64_t c = ptr[g]; prev += c; ptr[g] = (uint32_t)prev; }
      // ptr[g] now holds the END of granule g - 1's list, i.e. the start of granule g's: ptr[0] = start of the phase
      run = prev;
      if (run >= (1ull << 32)) return fail(err, "j-side lists exceed 2^32 entries");
    }
    P.j_off.assign((size_t)run, 0);
    std::vector<uint32_t> fill(P.j_ptr.size());
    for (int ph = 0; ph < n_phases; ++ph)
      for (int g = 0; g < P.n_gran; ++g) fill[stride * (size_t)ph + (size_t)g] = P.j_ptr[stride * (size_t)ph + (size_t)g];
    for (size_t c = 0; c < canon.size(); ++c) {
      const SymItem &it = P.items[canon[c]];
      if (it.flags & kSymNoJSide) continue;
      for (int k = 0; k < it.n_sub; ++k)
        P.j_off[fill[stride * (size_t)it.k0 + (size_t)(it.j0 / 64 + k)]++] = it.slot_j + (uint32_t)k * 64u;
    }
  }
  for (SymItem &it : P.items) { it.k0 = 0; it.k_skip = 0; }
  *out = std::move(P);
  return true;
}

// ---- the even-share plan ------------------------------------------------------------------------------------------------
bool build_sym_plan_even(int n_total, int bi, int n_items, SymPlan *out, std::string *err, int cost_sym, int cost_one,
                         int cost_move, int own_pct) {
  if (n_total <= 0) return fail(err, "bad body range");
  if (bi < 512 || bi % 512 != 0) return fail(err, "bodies per i-set must be a multiple of 512");
  if (n_items < 1 || cost_sym < 1 || cost_one < 1 || cost_move < 0 || own_pct < 1) return fail(err, "bad scheduling parameters");
  SymPlan P;
  P.even = true;
  P.bi = bi;
  P.T = (n_total + bi - 1) / bi;
  P.n_pad = P.T * bi;
  P.n_gran = (n_total + 63) / 64;
  P.own_block0 = 0; P.own_blocks = P.T; P.n_src = 1;
  P.own_gran0 = 0; P.own_grans = P.n_gran;
  const int spb = bi / 64, NP = bi / 512;
  constexpr int kQuant = 4;                                           // a cut falls on a multiple of four steps (the loops' unroll)

  // a row = the subtiles its i-set meets, in ring order: its own block, then the forward blocks (clipped to the granules that
  // exist; past the last one the run goes on at granule 0).  step_cost[q]: SIMD cycles of ONE step of subtile q.
  struct Row { int a, first, n_sub; long long cost; std::vector<int> step_cost; };
  std::vector<Row> rows;
  long long total = 0;
  for (int a = 0; a < P.T; ++a) {
    Row r{a, a * spb, 0, 0, {}};
    if (r.first >= P.n_gran) continue;                                // (cannot happen: the last block holds a body)
    const int dn = std::min(spb, P.n_gran - r.first);
    for (int k = 0; k < dn; ++k) {
      const int na = NP - (k * 64 >> 9);                              // active register pairs; the first of them one-sided
      r.step_cost.push_back(std::max(1, (na == 1 ? cost_one : (na - 1) * cost_sym + cost_one + cost_move) * own_pct / 100));
    }
    int h = 0;
    while (h + 1 < P.T && ring_assigned(a, (a + h + 1) % P.T, P.T)) ++h;
    for (int d = h + 1; d < P.T; ++d)
      if (ring_assigned(a, (a + d) % P.T, P.T)) return fail(err, "internal: forward blocks not contiguous");
    const long long ring = (long long)P.T * spb, s = (long long)((a + 1) % P.T) * spb;
    for (long long x = 0; x < (long long)h * spb; ++x)
      if ((s + x) % ring < P.n_gran) r.step_cost.push_back(NP * cost_sym + cost_move);
    r.n_sub = (int)r.step_cost.size();
    for (int c : r.step_cost) r.cost += 64LL * c;
    total += r.cost;
    rows.push_back(std::move(r));
  }
  // items per row: in proportion to the rows' cost, largest remainders first; every row at least one, none more than its
  // steps allow
  const int R = (int)rows.size();
  if (n_items < R) n_items = R;
  std::vector<int> m((size_t)R, 1);
  {
    std::vector<std::pair<double, int>> rem;
    int given = 0;
    for (int r = 0; r < R; ++r) {
      const double share = (double)n_items * (double)rows[(size_t)r].cost / (double)total;
      const int cap = rows[(size_t)r].n_sub * (64 / kQuant);
      m[(size_t)r] = std::max(1, std::min(cap, (int)share));
      given += m[(size_t)r];
      rem.push_back({share - (double)(int)share, r});
    }
    std::sort(rem.begin(), rem.end(), [](const std::pair<double, int> &x, const std::pair<double, int> &y) {
      return x.first != y.first ? x.first > y.first : x.second < y.second; });
    for (size_t k = 0; given < n_items && k < rem.size(); ++k) {
      const int r = rem[k].second;
      if (m[(size_t)r] < rows[(size_t)r].n_sub * (64 / kQuant)) { ++m[(size_t)r]; ++given; }
    }
  }
  uint64_t pool = 0;
  for (int r = 0; r < R; ++r) {
    const Row &row = rows[(size_t)r];
    const int mr = m[(size_t)r];
    // cut t (0 < t < mr) = the step position (64 q + k, k a multiple of kQuant) whose cumulative cost is nearest t / mr of the row's
    std::vector<long long> cut((size_t)mr + 1, 0);
    cut[(size_t)mr] = 64LL * row.n_sub;
    {
      int q = 0;
      long long before = 0;                                           // cost of the subtiles in front of q
      for (int t = 1; t < mr; ++t) {
        const double want = (double)row.cost * (double)t / (double)mr;
        while (q + 1 < row.n_sub && (double)(before + 64LL * row.step_cost[(size_t)q]) <= want) { before += 64LL * row.step_cost[(size_t)q]; ++q; }
        long long k = (long long)(((want - (double)before) / (double)row.step_cost[(size_t)q]) / kQuant + 0.5) * kQuant;
        k = std::max(0LL, std::min(64LL, k));
        long long pos = 64LL * q + k;
        pos = std::max(pos, cut[(size_t)t - 1] + kQuant);             // every item works at least one quantum
        pos = std::min(pos, 64LL * row.n_sub - (long long)(mr - t) * kQuant);
        cut[(size_t)t] = pos;
      }
    }
    for (int t = 0; t < mr; ++t) {
      const long long p0 = cut[(size_t)t], p1 = cut[(size_t)t + 1];
      const int q0 = (int)(p0 / 64), q1 = (int)((p1 - 1) / 64);
      SymItem it{};
      it.i0 = row.a * bi;
      int g = row.first + q0;
      if (g >= P.n_gran) g -= P.n_gran;
      it.j0 = g * 64;
      it.n_sub = q1 - q0 + 1;
      it.flags = kSymEven | ((g * 64 >= it.i0 && g * 64 < it.i0 + bi) ? kSymOneSided : 0);
      it.k0 = (int32_t)(p0 - 64LL * q0);
      it.k_skip = (int32_t)(64LL * (q1 + 1) - p1);
      if (pool + (uint64_t)bi + (uint64_t)it.n_sub * 64 >= (1ull << 32)) return fail(err, "partial-sum pool exceeds 2^32 elements");
      it.slot_i = (uint32_t)pool; pool += (uint64_t)bi;
      it.slot_j = (uint32_t)pool; pool += (uint64_t)it.n_sub * 64;
      P.items.push_back(it);
    }
  }
  P.pool_elems = pool;
  P.n_local = (int)P.items.size();
  P.phase_item0.assign({0, (int)P.items.size()});
  // the lists: item order is the order of summation, as in the guided plans
  auto gran_of = [&](const SymItem &it, int k) { int g = it.j0 / 64 + k; return g >= P.n_gran ? g - P.n_gran : g; };
  P.i_ptr.assign((size_t)P.own_grans + 1, 0);
  for (const SymItem &it : P.items)
    for (int g = it.i0 / 64; g < it.i0 / 64 + spb && g < P.own_grans; ++g) P.i_ptr[(size_t)g + 1] += 1;
  for (int g = 0; g < P.own_grans; ++g) P.i_ptr[(size_t)g + 1] += P.i_ptr[(size_t)g];
  P.i_off.assign(P.i_ptr.back(), 0);
  {
    std::vector<uint32_t> fill(P.i_ptr.begin(), P.i_ptr.end() - 1);
    for (const SymItem &it : P.items)
      for (int g = it.i0 / 64; g < it.i0 / 64 + spb && g < P.own_grans; ++g)
        P.i_off[fill[(size_t)g]++] = it.slot_i + (uint32_t)(g - it.i0 / 64) * 64u;
  }
  P.j_ptr.assign((size_t)P.n_gran + 1, 0);
  for (const SymItem &it : P.items)
    for (int k = 0; k < it.n_sub; ++k) P.j_ptr[(size_t)gran_of(it, k) + 1] += 1;
  for (int g = 0; g < P.n_gran; ++g) P.j_ptr[(size_t)g + 1] += P.j_ptr[(size_t)g];
  P.j_off.assign(P.j_ptr.back(), 0);
  {
    std::vector<uint32_t> fill(P.j_ptr.begin(), P.j_ptr.end() - 1);
    for (const SymItem &it : P.items)
      for (int k = 0; k < it.n_sub; ++k) P.j_off[fill[(size_t)gran_of(it, k)]++] = it.slot_j + (uint32_t)k * 64u;
  }
  *out = std::move(P);
  return true;
}

}  // namespace nbody

// C-ABI view of the even-share planner: items as eight words each — i0, j0, n_sub, flags, slot_i, slot_j, k0, k_skip.
extern "C" int nbody_sym_plan_describe_even(int32_t n_total, int32_t bodies_per_iset, int32_t n_items_wanted, int32_t *n_items,
                                            uint64_t *pool_elems, int32_t *items, int32_t items_cap) {
  nbody::SymPlan P;
  std::string why;
  if (!nbody::build_sym_plan_even(n_total, bodies_per_iset, n_items_wanted, &P, &why)) return NBODY_ERR_UNSUPPORTED;
  if (n_items) *n_items = (int32_t)P.items.size();
  if (pool_elems) *pool_elems = P.pool_elems;
  if (items) {
    if (items_cap < (int32_t)P.items.size()) return NBODY_ERR_INVALID;
    for (size_t k = 0; k < P.items.size(); ++k) {
      const nbody::SymItem &it = P.items[k];
      int32_t *o = items + 8 * k;
      o[0] = it.i0; o[1] = it.j0; o[2] = it.n_sub; o[3] = it.flags;
      o[4] = (int32_t)it.slot_i; o[5] = (int32_t)it.slot_j; o[6] = it.k0; o[7] = it.k_skip;
    }
  }
  return NBODY_OK;
}

// C-ABI view of the planner (host only, no device): lets the CPU test suite check that a plan covers every body pair
// exactly once and that its segments do not overlap.
extern "C" int nbody_sym_plan_describe_tenths(int32_t n_total, int32_t i_begin, int32_t i_count, int32_t bodies_per_iset,
                                              int32_t slots, int32_t k_guided_x10, int32_t min_sub, int32_t own_mode,
                                              int32_t *n_items, uint64_t *pool_elems, int32_t *items, int32_t items_cap) {
  nbody::SymPlan P;
  std::string why;
  if (i_count == 0) i_count = n_total - i_begin;
  if (k_guided_x10 < 1) return NBODY_ERR_INVALID;
  if (!nbody::build_sym_plan(n_total, i_begin, i_count, bodies_per_iset, slots, k_guided_x10 / 10.0, min_sub, own_mode, &P, &why, 0))
    return NBODY_ERR_UNSUPPORTED;
  if (n_items) *n_items = (int32_t)P.items.size();
  if (pool_elems) *pool_elems = P.pool_elems;
  if (items) {
    if (items_cap < (int32_t)P.items.size()) return NBODY_ERR_INVALID;
    static_assert(sizeof(nbody::SymItem) == 32, "SymItem is eight 32-bit words");
    for (size_t k = 0; k < P.items.size(); ++k) {
      const nbody::SymItem &it = P.items[k];
      int32_t *o = items + 8 * k;
      o[0] = it.i0; o[1] = it.j0; o[2] = it.n_sub; o[3] = it.flags;
      o[4] = (int32_t)it.slot_i; o[5] = (int32_t)it.slot_j; o[6] = 0; o[7] = 0;
    }
  }
  return NBODY_OK;
}

// The same for a plan whose j-side segments must share an area of at most j_budget_elems pool elements (phases); phases[]
// receives the first item of every phase followed by the item count (n_phases + 1 values).
extern "C" int nbody_sym_plan_describe_phased(int32_t n_total, int32_t i_begin, int32_t i_count, int32_t bodies_per_iset,
                                              int32_t slots, int32_t k_guided_x10, int32_t min_sub, uint64_t j_budget_elems,
                                              int32_t *n_items, uint64_t *pool_elems, int32_t *items, int32_t items_cap,
                                              int32_t *n_phases, int32_t *phases, int32_t phases_cap) {
  nbody::SymPlan P;
  std::string why;
  if (i_count == 0) i_count = n_total - i_begin;
  if (k_guided_x10 < 1) return NBODY_ERR_INVALID;
  if (!nbody::build_sym_plan(n_total, i_begin, i_count, bodies_per_iset, slots, k_guided_x10 / 10.0, min_sub, 1, &P, &why, j_budget_elems))
    return NBODY_ERR_UNSUPPORTED;
  if (n_items) *n_items = (int32_t)P.items.size();
  if (pool_elems) *pool_elems = P.pool_elems;
  if (n_phases) *n_phases = (int32_t)P.phase_item0.size() - 1;
  if (phases) {
    if (phases_cap < (int32_t)P.phase_item0.size()) return NBODY_ERR_INVALID;
    for (size_t k = 0; k < P.phase_item0.size(); ++k) phases[k] = P.phase_item0[k];
  }
  if (items) {
    if (items_cap < (int32_t)P.items.size()) return NBODY_ERR_INVALID;
    for (size_t k = 0; k < P.items.size(); ++k) {
      const nbody::SymItem &it = P.items[k];
      int32_t *o = items + 8 * k;
      o[0] = it.i0; o[1] = it.j0; o[2] = it.n_sub; o[3] = it.flags;
      o[4] = (int32_t)it.slot_i; o[5] = (int32_t)it.slot_j; o[6] = 0; o[7] = 0;
    }
  }
  return NBODY_OK;
}

extern "C" int nbody_sym_plan_describe(int32_t n_total, int32_t i_begin, int32_t i_count, int32_t bodies_per_iset,
                                       int32_t slots, int32_t k_guided, int32_t min_sub, int32_t own_mode, int32_t *n_items,
                                       uint64_t *pool_elems, int32_t *items, int32_t items_cap) {
  if (k_guided < 1 || k_guided > 100000) return NBODY_ERR_INVALID;
  return nbody_sym_plan_describe_tenths(n_total, i_begin, i_count, bodies_per_iset, slots, 10 * k_guided, min_sub, own_mode,
                                        n_items, pool_elems, items, items_cap);
}

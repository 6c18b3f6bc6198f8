"""Host logic of the symmetric force pass: the work plan (parallelnbody_amd/csrc/sym_plan.{h,cpp}) through its C-ABI
view nbody_sym_plan_describe.  CPU only — no kernel runs here.

A plan is correct when, over all ranks of a job, every unordered pair of different bodies is evaluated exactly once
(symmetric strips) or, inside an i-set's own block, every ordered pair once (one-sided strips), and when no two partial-sum
segments overlap.  The GPU parity tests then check the arithmetic; this file checks the bookkeeping for the sizes
BASELINE.json names and for ragged / sharded ones."""
import numpy as np
import pytest


def _coverage(nb, n, ranks, bi, slots, k, min_sub, own_mode=1):
    """sym[a, g] / one[a, g]: how many items pair block a's i-set with 64-body subtile g, over all ranks."""
    T, G = -(-n // bi), -(-n // 64)
    sym = np.zeros((T, G), np.int32)
    one = np.zeros((T, G), np.int32)
    pools = []
    for r in range(ranks):
        ib, ic = (r * (n // ranks), n // ranks) if ranks > 1 else (0, 0)
        items, pool = nb.sym_plan(n, ib, ic, bi, slots, k, min_sub, own_mode)
        assert len(items) > 0
        segs = []
        for i0, j0, n_sub, flags, slot_i, slot_j, _, _ in items:
            assert i0 % bi == 0 and j0 % 64 == 0 and n_sub >= 1 and j0 // 64 + n_sub <= G
            if ranks > 1:
                assert ib <= i0 < ib + ic
            tgt = one if flags & 1 else sym
            tgt[i0 // bi, j0 // 64:j0 // 64 + n_sub] += 1
            segs.append((np.uint32(slot_i), bi))
            assert bool(flags & 2) == bool(flags & 1 and own_mode == 0)      # only the fp64 own-block strips lack j-side sums
            if not flags & 2:
                segs.append((np.uint32(slot_j), 64 * n_sub))
        segs.sort()
        end = 0
        for s, ln in segs:                      # segments are disjoint and inside the pool
            assert int(s) >= end
            end = int(s) + ln
        assert end <= pool
        pools.append(pool)
    return sym, one, pools


@pytest.mark.parametrize("n,ranks,bi,slots", [
    (1024, 1, 512, 1024), (2000, 1, 512, 1024), (5000, 1, 1024, 1024), (40000, 1, 2048, 768), (65536, 1, 4096, 512),
    (65536, 1, 1024, 1024), (100003, 1, 2048, 768), (65536, 2, 2048, 768), (65536, 8, 4096, 512), (49152, 3, 1024, 1024),
    (32768, 4, 512, 1024)])
@pytest.mark.parametrize("own_mode", [0, 1, 2])
def test_every_pair_exactly_once(nb, n, ranks, bi, slots, own_mode):
    sym, one, _ = _coverage(nb, n, ranks, bi, slots, 3, 4, own_mode)
    T, G = sym.shape
    blk = np.arange(G) * 64 // bi                       # block of each subtile
    # inside its own block an i-set meets every subtile exactly once through an own-block strip (the kernel then works the
    # register pairs above the subtile's own pair symmetrically and that pair one-sided — or, fp64, all of it one-sided),
    # and never through a symmetric strip
    own = blk[None, :] == np.arange(T)[:, None]
    assert np.array_equal(one, own.astype(np.int32))
    assert not sym[own].any()
    # two subtiles of different blocks: the pair is evaluated by exactly one of the two rows
    E = sym[blk, :]                                      # E[gi, gj] = does gi's row take subtile gj
    off = blk[:, None] != blk[None, :]
    assert np.array_equal((E + E.T)[off], np.ones(off.sum(), np.int32))


@pytest.mark.parametrize("n,ranks,bi,slots,k,min_sub", [
    # what capi's choose_algorithm arrives at on a 256-CU device with the guided plan (the default outside 16385 <= N < 139264,
    # and for Kahan, fp64 and sharded contexts everywhere): (bodies per i-set, workgroup slots, K, shortest strip)
    (12288, 1, 512, 1024, 1, 2), (16384, 1, 512, 1024, 1, 2), (20480, 1, 1024, 1024, 1, 2), (32768, 1, 2048, 768, 1, 2),
    (65536, 1, 4096, 512, 1.5, 2), (131072, 1, 4096, 512, 1.5, 4), (262144, 1, 4096, 512, 3, 4), (65536, 2, 4096, 512, 1, 2),
    (131072, 8, 4096, 512, 1, 4), (100003, 1, 4096, 512, 1.5, 2)])
def test_the_librarys_own_plans_cover_every_pair_once(nb, n, ranks, bi, slots, k, min_sub):
    sym, one, _ = _coverage(nb, n, ranks, bi, slots, k, min_sub, 1)
    T, G = sym.shape
    blk = np.arange(G) * 64 // bi
    own = blk[None, :] == np.arange(T)[:, None]
    assert np.array_equal(one, own.astype(np.int32)) and not sym[own].any()
    E = sym[blk, :]
    off = blk[:, None] != blk[None, :]
    assert np.array_equal((E + E.T)[off], np.ones(off.sum(), np.int32))


def test_headline_plan_shape(nb):
    # N = 2^20, sixteen bodies per lane, 512 resident workgroups: long strips first, 256-body strips last, a pool of a
    # few GB (the round-1 layout needed 8.6 GB of private rows here)
    items, pool = nb.sym_plan(1 << 20, 0, 0, 4096, 512, 3, 4)
    sym = items[items[:, 3] == 0]
    assert sym[0, 2] > 1000 and sym[-1, 2] == 4
    assert sym[0, 2] == sym[:, 2].max() and sym[-len(sym) // 10:, 2].max() <= 16   # long strips first, short ones last
    assert 8000 < len(items) < 20000
    assert pool * 16 < 3.5e9
    # every rank of an 8-GPU job gets the same amount of work
    work = []
    for r in range(8):
        it, _ = nb.sym_plan(1 << 20, r << 17, 1 << 17, 4096, 512, 3, 4)
        work.append(int(it[it[:, 3] == 0][:, 2].sum()))
    assert max(work) - min(work) <= 4096 // 64


def test_unplannable_ranges_are_refused(nb):
    with pytest.raises(nb.NBodyError):
        nb.sym_plan(65536, 1000, 3000, 1024, 512, 3, 4)          # slices must be equal multiples of the i-set
    with pytest.raises(nb.NBodyError):
        nb.sym_plan(65536, 0, 32768, 1000, 512, 3, 4)            # i-sets are multiples of 64 bodies


@pytest.mark.parametrize("n,ranks,bi", [(65536, 2, 2048), (65536, 8, 4096), (131072, 8, 4096), (49152, 3, 1024), (1 << 20, 8, 4096)])
def test_sharded_plans_launch_the_strips_inside_the_own_slice_first(nb, n, ranks, bi):
    # the strips whose j range lies inside the rank's own slice need no other rank's positions; the plan cuts
    # its ranges at the slice's ends (no strip straddles them) and puts those strips first in launch order, so that they
    # can run while the all-gather is still in flight (nbody_step_begin_local).  About 1/ranks of a rank's work.
    ic = n // ranks
    for r in range(ranks):
        items, _ = nb.sym_plan(n, r * ic, ic, bi, 512, 3, 4)
        j0, j1 = items[:, 1], items[:, 1] + 64 * items[:, 2]
        inside = (j0 >= r * ic) & (j1 <= (r + 1) * ic)
        outside = (j1 <= r * ic) | (j0 >= (r + 1) * ic)
        assert np.all(inside | outside)                                   # nobody straddles a slice end
        k = int(inside.sum())
        assert k > 0 and inside[:k].all() and not inside[k:].any()        # local strips first
        share = items[inside, 2].sum() / items[:, 2].sum()
        assert 0.5 / ranks < share < 2.0 / ranks
    items, _ = nb.sym_plan(n, 0, 0, bi, 512, 3, 4)                         # one context: everything is local
    assert np.all(items[:, 1] + 64 * items[:, 2] <= -(-n // 64) * 64)


@pytest.mark.parametrize("n,ranks,bi,budget_frac", [(65536, 1, 4096, 0.3), (131072, 1, 4096, 0.11), (65536, 2, 2048, 0.4)])
def test_pool_phases_share_one_j_side_area(nb, n, ranks, bi, budget_frac):
    # very large systems: the j-side segments (N^2 / (2 bi) elements) are run in phases that share one pool area.  Same
    # items, same pairs; inside a phase no two segments overlap, the area is reused from phase to phase, the i-side
    # segments stay apart to the end, and the pool is what the budget promises.
    ic = n // ranks
    for r in range(ranks):
        ib, icc = (r * ic, ic) if ranks > 1 else (0, 0)
        base, pool1 = nb.sym_plan(n, ib, icc, bi, 512, 3, 4)
        total_j = int(64 * base[base[:, 3] & 2 == 0][:, 2].sum())
        budget = int(total_j * budget_frac)
        items, pool, ph = nb.sym_plan_phased(n, budget, ib, icc, bi, 512, 3, 4)
        assert len(ph) - 1 >= int(1 / budget_frac) and ph[0] == 0 and ph[-1] == len(items) and np.all(np.diff(ph) > 0)
        # the same strips as the one-pass plan, in the same launch order
        np.testing.assert_array_equal(items[:, :4], base[:, :4])
        i_end = len(items) * bi                                  # i-side segments first, one per item
        assert sorted(items[:, 4].astype(np.int64).tolist()) == list(range(0, i_end, bi))
        area = 0
        for p in range(len(ph) - 1):
            seg = items[ph[p]:ph[p + 1]]
            s0 = seg[:, 5].astype(np.int64)
            ln = 64 * seg[:, 2].astype(np.int64)
            order = np.argsort(s0)
            assert s0[order][0] == i_end                         # every phase starts at the area's first element
            assert np.all(s0[order][1:] >= (s0 + ln)[order][:-1])   # disjoint inside the phase
            assert (s0 + ln).max() - i_end <= budget
            area = max(area, int((s0 + ln).max()) - i_end)
        assert pool == i_end + area and pool < pool1
    # a budget that everything fits into: one phase, the one-pass plan exactly (slots included)
    items, pool, ph = nb.sym_plan_phased(n, 1 << 40, 0, 0, bi, 512, 3, 4)
    base, pool1 = nb.sym_plan(n, 0, 0, bi, 512, 3, 4)
    assert len(ph) == 2 and pool == pool1
    np.testing.assert_array_equal(items, base)


def test_block_kernel_pairs_per_workgroup_rule():
    """forces_block_pk_kernel: how many register pairs of bodies a workgroup owns (csrc/capi.hip block_pairs).  A CU works
    through ceil(workgroups / CUs) workgroups of `pairs` pairs each: the rule takes the smallest product, larger workgroups
    on a tie.  Checked against the choices measured fastest (or within 6 %) in profiles/r03_block_kernel_np_by_n.txt."""
    import math
    import parallelnbody_amd as nb
    L = nb.lib()
    measured = {2000: 4, 3000: 6, 4096: 8, 5000: 5, 6000: 6, 7000: 7, 8192: 8, 9216: 6, 10240: 5, 12288: 8, 14336: 7,
                16384: 8, 18432: 6, 20480: 8, 24576: 8}
    for n, want in measured.items():
        assert L.nbody_block_pairs_describe(n, 256) == want, n
    for cus in (64, 120, 256, 304):
        for n in list(range(1, 300)) + [1000, 2047, 2048, 2049, 8191, 8193, 16383, 16384, 100000]:
            got = L.nbody_block_pairs_describe(n, cus)
            assert 2 <= got <= 8
            cost = lambda k: math.ceil(math.ceil(n / (2 * k)) / cus) * k
            best = min(cost(k) for k in range(2, 9))
            assert cost(got) == best and all(cost(k) > best for k in range(got + 1, 9)), (n, cus, got)
    assert L.nbody_block_pairs_describe(0, 256) == 0


# ---- the even-share plan (round 5; csrc/sym_plan.h build_sym_plan_even) ------------------------------------------------

def _even_coverage(nb, n, bi, n_items):
    """cov[a, g, s]: how many items work steps [4 s, 4 s + 4) of 64-body subtile g against block a's i-set; plus the items'
    costs by the planner's model and the segment check."""
    items, pool = nb.sym_plan_even(n, bi, n_items)
    T, G, NP = -(-n // bi), -(-n // 64), bi // 512
    cov = np.zeros((T, G, 16), np.int32)
    cost = np.zeros(len(items))
    segs = []
    for idx, (i0, j0, n_sub, flags, slot_i, slot_j, k0, k_skip) in enumerate(items):
        assert flags & 4 and i0 % bi == 0 and j0 % 64 == 0 and 0 <= j0 < G * 64 and n_sub >= 1
        assert k0 % 4 == 0 and k_skip % 4 == 0 and 0 <= k0 < 64 and 0 <= k_skip < 64
        assert n_sub > 1 or k0 < 64 - k_skip                       # a single subtile: a non-empty range of steps
        a = i0 // bi
        for q in range(n_sub):
            g = (j0 // 64 + q) % G                                  # ring order: past the last granule comes granule 0
            s0 = k0 // 4 if q == 0 else 0
            s1 = (64 - k_skip) // 4 if q == n_sub - 1 else 16
            cov[a, g, s0:s1] += 1
            off = g * 64 - i0
            if 0 <= off < bi:                                       # own block: NP - pc pairs, the first one-sided
                na = NP - (off >> 9)
                per_step = (74 if na == 1 else (na - 1) * 82 + 74 + 26) * 130 // 100     # kSymEvenOwnPct
            else:
                per_step = NP * 82 + 26
            cost[idx] += 4 * (s1 - s0) * per_step
        assert bool(flags & 1) == (0 <= j0 - i0 < bi)
        segs.append((np.uint32(slot_i), bi))
        segs.append((np.uint32(slot_j), 64 * n_sub))
    segs.sort()
    end = 0
    for s, ln in segs:
        assert int(s) >= end
        end = int(s) + ln
    assert end <= pool
    return items, cov, cost


@pytest.mark.parametrize("n,bi,n_items", [
    (16384, 1024, 1024), (16384, 2048, 768), (20480, 2048, 768), (32768, 2048, 768), (32768, 4096, 512), (65536, 4096, 512),
    (131072, 4096, 512), (17408, 2048, 768), (40000, 2048, 768), (100003, 4096, 512), (20001, 1024, 1024), (5000, 1024, 64),
    (2000, 512, 16), (65536, 4096, 1024)])
def test_even_share_plans_cover_every_pair_once_in_equal_shares(nb, n, bi, n_items):
    items, cov, cost = _even_coverage(nb, n, bi, n_items)
    T, G, _ = cov.shape
    assert len(items) == max(n_items, T)
    blk = np.arange(G) * 64 // bi
    own = blk[None, :] == np.arange(T)[:, None]
    # inside its own block an i-set works every step of every subtile exactly once
    assert np.all(cov[own] == 1)
    # elsewhere a (row, subtile) is worked in all of its 64 steps or in none ...
    full = cov.min(axis=2)
    assert np.array_equal(full, cov.max(axis=2)) and full.max() == 1
    # ... and of two subtiles of different blocks exactly one row takes the pair
    E = full[blk, :]
    off = blk[:, None] != blk[None, :]
    assert np.array_equal((E + E.T)[off], np.ones(off.sum(), np.int32))
    # equal shares: within a row to a quantum of four steps, between rows to the rounding of items per row
    rows = items[:, 0] // bi
    for a in range(T):
        c = cost[rows == a]
        quantum = 4 * ((bi // 512) * 82 + 26) * 130 // 100           # four steps of the dearest kind (own block, every pair at work)
        assert c.max() - c.min() <= 2 * quantum, (a, c.min(), c.max())
    if len(items) >= 8 * T:
        assert cost.max() <= 1.08 * cost.mean(), (cost.max(), cost.mean())


def test_even_share_plan_of_the_mid_sizes_is_even_to_a_percent(nb):
    # what the library runs at N = 32768 (eight bodies per lane, 768 slots) and 65536 (sixteen, 512)
    for n, bi, slots in ((32768, 2048, 768), (65536, 4096, 512)):
        items, cov, cost = _even_coverage(nb, n, bi, slots)
        assert len(items) == slots and cost.max() <= 1.015 * cost.mean()
        # an i-side segment per item: 16 B x bodies per i-set x items — 25 MB and 34 MB (the guided plans: 56 MB and 278 MB)
        assert len(items) * bi * 16 < 35e6

"""The arithmetic the wave kernel's integer pre-filter rests on (root_wave.hip), checked on the CPU
with the product's own density tables and fixed-point weights: a candidate whose summed SATURATED byte distance over the
pre-filter terms exceeds the cut-off derived from ANY lower bound of the maximum has fixed-point weight exactly 0 — so
dropping it changes neither the log-sum-exp nor a draw; guess-and-refine (start at 4 edits, widen, re-scan at the
cut-off the best survivor implies) ends with a superset of the candidates that carry weight."""
import ctypes as C

import numpy as np

import helpers

PRE_CLAMP, CUT_ALL, FIX_CUTOFF = 42, 126, 28.5


def _tables(oracle, max_len=80):
    mr, md, ml, nb, logl = helpers.density_tables_cpu(oracle, max_len)
    atd = np.empty((ml + 1, md + 1))
    for L in range(ml + 1):
        for d in range(md + 1):
            l = nb[(L + 4) // 5, d]          # the three fp64 operations of add_typos.jl:61-63, in the kernels' order
            l -= logl[L] * d
            l -= 1.629048269010741 * d
            atd[L, d] = l
    with np.errstate(invalid="ignore", divide="ignore"):
        per_edit = -atd[1:, 1:] / np.arange(1, md + 1)[None, :]
    cmin = np.nanmin(per_edit[np.isfinite(per_edit)])
    return atd, cmin, ml


def _fixw(oracle, x):
    f = oracle.lib().pco_fixw
    f.restype, f.argtypes = C.c_uint64, [C.c_double]
    return np.array([f(float(v)) for v in x], dtype=np.uint64)


def test_prefilter_never_drops_a_candidate_that_carries_weight(oracle):
    atd, cmin, ml = _tables(oracle)
    inv_c = 1.0 / (cmin * (1.0 - 1e-9))
    rng = np.random.default_rng(7)
    n_checked = n_dropped = 0
    for trial in range(60):
        K, n_terms = int(rng.integers(50, 600)), 5
        L = rng.integers(1, ml - 10, size=(n_terms, K))
        # mostly far candidates, a few near ones (typos), some beyond the saturation point
        d = np.minimum(rng.integers(0, 75, size=(n_terms, K)), np.maximum(L, 1) + 5)
        near = rng.choice(K, size=max(1, K // 40), replace=False)
        d[:, near] = rng.integers(0, 4, size=(n_terms, len(near)))
        if trial % 3 == 0:
            d[3:, near[0]] = rng.integers(45, 70, size=(n_terms - 3,))  # near on the pre-filter terms, far on the others
        prior = np.log(rng.integers(1, 50, size=K)) - np.log(5000.0)
        pmax = prior.max()
        s = prior + sum(atd[L[f], d[f]] for f in range(n_terms))
        m = s.max()
        w = _fixw(oracle, s - m)
        Dp = np.minimum(d[:3], PRE_CLAMP).sum(axis=0)          # what the packed byte sums see
        assert Dp.max() <= CUT_ALL
        for bi, bound in enumerate((m - 1.0, s[rng.integers(K)] - 1.0, np.partition(s, K // 2)[K // 2])):
            x = (pmax - bound + FIX_CUTOFF) * inv_c
            cut = int(x) + 2 if 0 <= x < CUT_ALL - 2 else CUT_ALL
            dropped = Dp > cut
            assert not w[dropped].any(), (trial, cut, s[dropped].max() - m)
            if bi == 0:  # the bound a well-explained row has: the score of its current referent
                n_dropped += int(dropped.sum())
                n_checked += K
        # guess-and-refine from a useless bound
        cut, cut_max = 4, CUT_ALL
        while True:
            surv = np.flatnonzero(Dp <= cut)
            if len(surv) == 0:
                assert cut < cut_max
                cut = min(2 * cut + 2, cut_max)
                continue
            best = s[surv].max()
            x = (pmax - best + FIX_CUTOFF) * inv_c
            need = min(int(x) + 2 if 0 <= x < CUT_ALL - 2 else CUT_ALL, cut_max)
            if need > cut:
                surv = np.flatnonzero(Dp <= need)
            break
        assert set(np.flatnonzero(w)) <= set(surv)
        # and the log-sum-exp / prefix of the survivors alone are those of the full enumeration
        assert w[surv].sum() == w.sum()
    assert n_dropped > 0.9 * n_checked  # with a tight bound the filter removes almost everything


def test_block_minimum_level_keeps_every_candidate_within_the_cut(oracle):
    """The coarse level of the scan (root_wave.hip: compact_min_kernel + the block list): a block of 64 candidates is
    read only if the sum of its three per-row minima is within the cut-off.  Every candidate within the cut-off lies in
    such a block (its own bytes bound the minima from above), so the two-level scan finds exactly the survivors of the
    streamed rows — for any tables, any cut-off, padding and dead rows included."""
    rng = np.random.default_rng(11)
    for trial in range(200):
        n = int(rng.integers(1, 700))
        kpad = (n + 15) & ~15
        rows = np.minimum(rng.integers(0, 60, size=(3, kpad)), PRE_CLAMP).astype(np.int64)
        rows[:, n:] = 0                                   # padding bytes are 0 (compact_pair_kernel)
        near = rng.integers(0, n, size=rng.integers(0, 4))
        rows[:, near] = rng.integers(0, 4, size=(3, len(near)))  # a few genuinely close candidates
        alive = np.zeros(kpad, dtype=bool)
        alive[:n] = rng.random(n) < 0.9
        cut = int(rng.integers(0, CUT_ALL + 1))
        want = np.nonzero((rows.sum(0) <= cut) & alive)[0]
        kblk = (kpad + 63) >> 6
        got = []
        for kb in range(kblk):
            blk = rows[:, 64 * kb:64 * kb + 64]
            if blk.min(1).sum() > cut:                     # the three minima of the block
                continue
            k = np.arange(64 * kb, min(64 * kb + 64, kpad))
            got.extend(k[(blk.sum(0) <= cut) & alive[k]].tolist())
        assert np.array_equal(np.array(got, dtype=want.dtype), want)  # same set, ascending order


def test_a_total_of_one_unit_decides_every_draw(oracle):
    """The decided-group shortcut (root_wave.hip): when the fixed-point total of a list is exactly one unit, all entries
    but one have weight 0 (the maximum alone weighs one unit), so min{k : prefix_k > x} is the same k for every
    x in [0, U) — whatever Philox would have produced."""
    rng = np.random.default_rng(3)
    one = 1 << 40
    hits = 0
    for trial in range(2000):
        n = int(rng.integers(1, 6))
        s = rng.normal(0, 25, size=n)
        if rng.random() < 0.5:
            s[rng.integers(0, n)] += 60.0
        w = _fixw(oracle, s - s.max())
        U = int(w.sum())
        if U != one:
            continue
        hits += 1
        assert (w > 0).sum() == 1 and int(w.max()) == one
        prefix = np.cumsum(w.astype(object))
        first = int(np.argmax(w > 0))
        for x in (0, 1, one // 2, one - 1):
            assert int(np.argmax(prefix > x)) == first
    assert hits > 500

"""CPU tests: the oracle against formula-derived known-answer values and
brute-force properties (the reference ships no tests or goldens — SURVEY §4)."""
import itertools
import json
import math
import os
import random

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = json.load(open(os.path.join(ROOT, "tests", "golden", "kat.json")))


def py_osa(a, b):
    la, lb = len(a), len(b)
    d = [[0] * (lb + 1) for _ in range(la + 1)]
    for i in range(la + 1):
        d[i][0] = i
    for j in range(lb + 1):
        d[0][j] = j
    for i in range(1, la + 1):
        for j in range(1, lb + 1):
            c = 0 if a[i - 1] == b[j - 1] else 1
            d[i][j] = min(d[i - 1][j] + 1, d[i][j - 1] + 1, d[i - 1][j - 1] + c)
            if i > 1 and j > 1 and a[i - 1] == b[j - 2] and a[i - 2] == b[j - 1]:
                d[i][j] = min(d[i][j], d[i - 2][j - 2] + 1)
    return d[la][lb]


def bfs_true_dl(src, alphabet, maxlen, maxdist):
    """Unrestricted Damerau-Levenshtein by breadth-first search over edit operations."""
    dist = {src: 0}
    frontier = [src]
    for d in range(1, maxdist + 1):
        nxt = []
        for s in frontier:
            cands = []
            for i in range(len(s) + 1):
                for c in alphabet:
                    cands.append(s[:i] + c + s[i:])
            for i in range(len(s)):
                cands.append(s[:i] + s[i + 1:])
                for c in alphabet:
                    cands.append(s[:i] + c + s[i + 1:])
            for i in range(len(s) - 1):
                cands.append(s[:i] + s[i + 1] + s[i] + s[i + 2:])
            for t in cands:
                if len(t) <= maxlen and t not in dist:
                    dist[t] = d
                    nxt.append(t)
        frontier = nxt
    return dist


def test_add_typos_kat(oracle):
    for e in KAT["add_typos"]:
        got = oracle.add_typos_strings(e["obs"], e["word"], e["max_typos"])
        assert got == pytest.approx(e["value"], rel=1e-12, abs=1e-12), e
    assert oracle.add_typos_strings(None, "abc") == 0.0  # add_typos.jl:51-53


def test_string_prior_kat(oracle):
    from pclean_amd.encode import StringPool, load_lm_params
    init, trans = load_lm_params()
    for e in KAT["string_prior"]:
        pool = StringPool()
        pool.add(e["s"])
        _, off, lm, _ = pool.arrays()
        got = oracle.string_prior(lm[off[0]:off[1]], e["min"], e["max"], init, trans)
        if math.isinf(e["value"]):
            assert got == e["value"]
        else:
            assert got == pytest.approx(e["value"], rel=1e-12)


def test_scalar_densities_kat(oracle):
    L = oracle.lib()
    for e in KAT["negbin"]:
        assert L.pco_negbin_logpdf(e["r"], e["p"], e["k"]) == pytest.approx(e["value"], rel=1e-11, abs=1e-11)
    for e in KAT["normal"]:
        assert L.pco_normal_logpdf(e["x"], e["mu"], e["sigma"]) == pytest.approx(e["value"], rel=1e-13)
    for e in KAT["transformed_gaussian"]:
        assert L.pco_transformed_gaussian(e["backward"], e["abs_deriv"], e["mu"], e["sigma"]) == pytest.approx(
            e["value"], rel=1e-12)
    for e in KAT["maybe_swap"]:
        assert L.pco_maybe_swap(0, 1, e["same"], e["n"], e["p"]) == pytest.approx(e["value"], rel=1e-14)
    assert L.pco_maybe_swap(1, 1, 0, 4, 0.1) == 0.0 and L.pco_maybe_swap(1, 0, 0, 4, 0.1) == -1000.0
    assert L.pco_time_prior() == pytest.approx(KAT["time_prior"], rel=1e-15)
    for s, want in KAT["time_regex"].items():
        assert oracle.time_regex(s) == want, s
    assert L.pco_choose_uniformly(54) == pytest.approx(-math.log(54))
    # duplicates are log-sum-exp'ed; logprobs() does not normalise (utils.jl:33-36)
    assert oracle.choose_proportionally(2, [1, 2, 2, 3], [0.1, 0.2, 0.3, 0.4]) == pytest.approx(math.log(0.5))
    assert oracle.choose_proportionally(9, [1, 2], [0.5, 0.5]) == -math.inf


def test_crp_normalises(oracle):
    # sum_k exp(prior_k) + exp(prior_new) == 1
    L = oracle.lib()
    counts = [5, 1, 9, 2]
    for (s, d) in [(1.0, 0.0), (0.37, 0.42)]:
        tot = sum(counts)
        p = sum(math.exp(L.pco_py_existing(c, tot, s, d)) for c in counts) + math.exp(L.pco_py_new(len(counts), tot, s, d))
        assert p == pytest.approx(1.0, rel=1e-12)
    # pitman_yor_score (trace.jl:65-78) by hand for two clusters of sizes 2 and 1
    s, d = 1.3, 0.2
    want = (math.log(1 * d + s) - math.log(0 + s)) + (math.log(1 - d) - math.log(0 + 1 + s)) + (
        math.log(2 * d + s) - math.log(2 + s))
    assert oracle.pitman_yor_score(s, d, [2, 1]) == pytest.approx(want, rel=1e-13)


def test_lse_and_ess(oracle):
    x = [-3.0, -1.0, -2.5, -math.inf]
    assert oracle.logsumexp(x) == pytest.approx(math.log(sum(math.exp(v) for v in x)), rel=1e-14)
    assert oracle.logsumexp([-math.inf, -math.inf]) == -math.inf
    w = np.array([0.5, 0.25, 0.125, 0.125])
    assert oracle.ess(np.log(w) + 3.0) == pytest.approx(1.0 / np.sum(w ** 2), rel=1e-12)


def test_osa_matches_python_dp(oracle):
    rnd = random.Random(7)
    for _ in range(400):
        a = "".join(rnd.choice("abcx ") for _ in range(rnd.randint(0, 9)))
        b = "".join(rnd.choice("abcx ") for _ in range(rnd.randint(0, 9)))
        assert oracle.osa(a, b) == py_osa(a, b), (a, b)


def test_dl_matches_bfs(oracle):
    assert oracle.dl("ca", "abc") == 2 and oracle.osa("ca", "abc") == 3  # the classic OSA != DL case
    for src in ["", "ab", "abc", "abca", "ca"]:
        dist = bfs_true_dl(src, "abc", maxlen=5, maxdist=3)
        for t, d in dist.items():
            assert oracle.dl(src, t) == d, (src, t)
            assert oracle.osa(src, t) >= d


def test_osa_vs_dl_on_hospital_pairs(oracle, hospital_columns):
    """the two StringDistances flavours on every unique pair of the dataset.
    Measured here: they differ on a few hundred far-apart pairs (e.g. zip '35640' vs '36854':
    OSA 4, DL 3) but never where the OSA distance is <= 2, and DL <= OSA always."""
    from pclean_amd.encode import StringPool
    n_diff = 0
    for col, vals in hospital_columns.items():
        pool = StringPool()
        ids = pool.add_all(vals)
        sym, off, _, _ = pool.arrays()
        a = oracle.pair_table(sym, off, ids, ids, 0)
        b = oracle.pair_table(sym, off, ids, ids, 1)
        assert np.all(b <= a), col
        assert np.array_equal(a[a <= 2], b[a <= 2]), col
        assert np.all(np.diag(a) == 0) and np.array_equal(b, b.T)
        n_diff += int(np.sum(a != b))
    assert 0 < n_diff < 2000


def test_detmath_close_to_libm(oracle):
    L = oracle.lib()
    rnd = np.random.default_rng(3)
    xs = np.concatenate([rnd.uniform(-745, 5, 4000), rnd.uniform(-1, 1, 2000), [0.0, -0.0, 1e-300, -30.0, -28.5]])
    for x in xs:
        e = L.pco_det_exp(float(x))
        assert e == pytest.approx(math.exp(x), rel=4e-16, abs=5e-324)
    ys = np.concatenate([np.exp(rnd.uniform(-700, 700, 4000)), rnd.uniform(0.5, 2.0, 3000), [1.0, 5e-324, 2.0 ** 60]])
    for y in ys:
        assert L.pco_det_log(float(y)) == pytest.approx(math.log(y), rel=4e-16, abs=3e-16)
    assert L.pco_det_exp(-math.inf) == 0.0 and L.pco_det_log(0.0) == -math.inf
    assert L.pco_fixw(0.0) == 1 << 40 and L.pco_fixw(-math.inf) == 0 and L.pco_fixw(-29.0) == 0
    assert L.pco_fixw(math.log(0.5)) in (1 << 39, (1 << 39) - 1)


def test_philox_known_answers(oracle):
    # Random123 KAT vectors for philox4x32-10
    assert list(oracle.philox((0, 0, 0, 0), (0, 0))) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert list(oracle.philox((0xffffffff,) * 4, (0xffffffff,) * 2)) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert list(oracle.philox((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0))) == [
        0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]

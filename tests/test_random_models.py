"""`random(dist, args...)` of the noise models: the oracle restatement (oracle/random.h) is pinned
on CPU through distributional properties derived from the reference formulas; on the GPU the
HIP samplers must reproduce it draw for draw (integer / index / code-point outputs bit-exact,
Normal draws bit-exact as well since both sides share pclean_log and IEEE sqrt/div)."""
import numpy as np
import pytest

from pclean_amd import sampling
from pclean_amd.encode import ALPHABET, load_lm_params

WORDS = ["birmingham", "a", "", "st vincents east", "35235", "heart attack patients given aspirin at arrival",
         "ab", "zzzzzzzzzzzzzzzzzzzzzzzzzzzzzz"]


def _osa(a, b):
    d = [[0] * (len(b) + 1) for _ in range(len(a) + 1)]
    for i in range(len(a) + 1):
        d[i][0] = i
    for j in range(len(b) + 1):
        d[0][j] = j
    for i in range(1, len(a) + 1):
        for j in range(1, len(b) + 1):
            d[i][j] = min(d[i - 1][j] + 1, d[i][j - 1] + 1, d[i - 1][j - 1] + (a[i - 1] != b[j - 1]))
            if i > 1 and j > 1 and a[i - 1] == b[j - 2] and a[i - 2] == b[j - 1]:
                d[i][j] = min(d[i][j], d[i - 2][j - 2] + 1)
    return d[len(a)][len(b)]


def test_add_typos_random_distribution(oracle):
    ro = oracle.RandomOracle()
    n = 20000
    words = ["birmingham"] * n  # r = ceil(10/5) = 2 -> NegativeBinomial(2, 0.9): mean 2*0.1/0.9
    out = sampling.random_add_typos(ro, words, None, seed=7, stream=1)
    changed = np.array([w != "birmingham" for w in out])
    # P(no typo) = 0.9^2 = 0.81; a typo can also be a no-op (substituting the same letter, swapping equal letters)
    assert abs((~changed).mean() - 0.81) < 0.02
    lens = np.array([len(w) for w in out])
    assert lens.min() >= 5 and lens.max() <= 15
    # with max_typos = 1 every sample is within one restricted-DL edit (add_typos.jl:38)
    out1 = sampling.random_add_typos(ro, words[:3000], 1, seed=7, stream=2)
    assert max(_osa(w, "birmingham") for w in set(out1)) <= 1
    # inserted / substituted letters are lowercase a-z
    assert set("".join(out)) <= set("birmingham") | set("abcdefghijklmnopqrstuvwxyz")
    # all four typo kinds occur: longer, shorter, transposed, substituted
    one = [w for w in set(out1) if w != "birmingham"]
    assert any(len(w) == 11 for w in one) and any(len(w) == 9 for w in one)
    assert any(len(w) == 10 and sorted(w) == sorted("birmingham") for w in one)
    assert any(len(w) == 10 and sorted(w) != sorted("birmingham") for w in one)
    # edge cases: empty word (r = 0 -> never a typo), one-letter word (transpose is a no-op)
    e = sampling.random_add_typos(ro, WORDS, None, seed=3, stream=0)
    assert e[2] == "" and len(e) == len(WORDS)
    # determinism and independence of the batch composition: element i only depends on (seed, stream, i)
    again = sampling.random_add_typos(ro, WORDS[:4], None, seed=3, stream=0)
    assert again == e[:4]


def test_string_prior_random_distribution(oracle):
    ro = oracle.RandomOracle()
    init, trans = load_lm_params()
    n = 30000
    out = sampling.random_string_prior(ro, n, 3, 30, seed=11, stream=0)
    lens = np.array([len(s) for s in out])
    assert lens.min() == 3 and lens.max() == 30
    assert abs(lens.mean() - 16.5) < 0.2  # DiscreteUniform(3, 30)
    first = np.bincount([ALPHABET.index(s[0]) for s in out], minlength=28) / n
    assert np.abs(first - init / init.sum()).max() < 0.01
    # bigram frequencies after 't' follow column 't' of the transition matrix (string_prior.jl:32)
    t = ALPHABET.index("t")
    nxt = np.zeros(28)
    for s in out:
        for a, b in zip(s[:-1], s[1:]):
            if a == "t":
                nxt[ALPHABET.index(b)] += 1
    assert nxt.sum() > 5000
    assert np.abs(nxt / nxt.sum() - trans[t] / trans[t].sum()).max() < 0.02
    # impossible transitions (exact zeros of the table) never occur
    zero = {(ALPHABET[p], ALPHABET[q]) for p in range(28) for q in range(28) if trans[p, q] == 0.0}
    assert len(zero) == 25
    seen = {(a, b) for s in out[:5000] for a, b in zip(s[:-1], s[1:])}
    assert not (seen & zero)


def test_categorical_normal_swap_time(oracle):
    ro = oracle.RandomOracle()
    probs = np.array([0.5, 0.0, 0.3, 0.2])
    got = sampling.random_choose_proportionally(ro, 40000, list("abcd"), probs, seed=1, stream=0)
    freq = np.array([got.count(c) for c in "abcd"]) / 40000
    assert freq[1] == 0 and np.abs(freq - probs).max() < 0.01
    got = sampling.random_choose_uniformly(ro, 30000, [10, 20, 30], seed=1, stream=1)
    assert abs(np.mean(got) - 20) < 0.2
    x = sampling.random_add_noise(ro, np.full(50000, 1500.0), 150.0, seed=2, stream=0)
    assert abs(x.mean() - 1500) < 3 and abs(x.std() - 150) < 2
    assert abs(np.mean(np.abs(x - 1500) < 150) - 0.6827) < 0.01
    y = sampling.random_transformed_gaussian(ro, np.full(50000, 1500.0), 150.0, 1 / 1000.0, seed=2, stream=0)
    assert np.array_equal(y, x * (1 / 1000.0))  # t.forward(rand(Normal)) with the same draws
    vals = ["7:10 a.m."] * 20000
    opts = [["7:10 a.m.", "7:16 a.m.", "9:40 a.m."]] * 20000
    sw = sampling.random_maybe_swap(ro, vals, opts, np.full(20000, 0.3), seed=5, stream=0)
    # swapped w.p. 0.3, and a swap picks each of the 3 options (incl. the same value) uniformly
    assert abs(np.mean([s != "7:10 a.m." for s in sw]) - 0.2) < 0.01
    assert sampling.random_maybe_swap(ro, vals[:100], opts[:100], np.zeros(100), seed=5) == vals[:100]
    assert all(s in opts[0] for s in sampling.random_maybe_swap(ro, vals[:100], opts[:100], np.ones(100), seed=5))
    tm = sampling.random_time_prior(ro, 20000, seed=9)
    hours = np.array([int(t.split(":")[0]) for t in tm])
    mins = np.array([int(t.split(":")[1].split()[0]) for t in tm])
    assert hours.min() == 1 and hours.max() == 12 and mins.min() == 1 and mins.max() == 60
    assert abs(np.mean([t.endswith("a.m.") for t in tm]) - 0.5) < 0.02


@pytest.mark.gpu
def test_gpu_samplers_match_oracle(oracle):
    from pclean_amd._lib import HipContext
    ro = oracle.RandomOracle()
    hip = HipContext(0)
    try:
        init, trans = load_lm_params()
        words = (WORDS * 700)[:5000]
        for mt in (None, 0, 2):
            assert sampling.random_add_typos(hip, words, mt, seed=42, stream=3) == \
                sampling.random_add_typos(ro, words, mt, seed=42, stream=3)
        assert sampling.random_add_typos(hip, [], None) == []
        a, la = hip.random_string_prior(20000, 3, 30, init, trans, 5, 1)
        b, lb = ro.random_string_prior(20000, 3, 30, init, trans, 5, 1)
        assert np.array_equal(a, b) and np.array_equal(la, lb)
        a, la = hip.random_string_prior(100, 0, 1, init, trans, 5, 2)
        b, lb = ro.random_string_prior(100, 0, 1, init, trans, 5, 2)
        assert np.array_equal(a, b) and np.array_equal(la, lb) and la.min() == 0
        logp = np.log(np.array([0.5, 1e-300, 0.3, 0.2, 0.0]) + 0.0)
        assert np.array_equal(hip.random_categorical(50000, logp, 8, 0), ro.random_categorical(50000, logp, 8, 0))
        mean = np.linspace(-5, 3000, 50000)
        assert np.array_equal(hip.random_normal(mean, 150.0, 1.0, 9, 0), ro.random_normal(mean, 150.0, 1.0, 9, 0))
        assert np.array_equal(hip.random_normal(mean, 0.25, 1e-3, 9, 1), ro.random_normal(mean, 0.25, 1e-3, 9, 1))
        prob = np.tile([0.0, 1e-5, 0.3, 0.999, 1.0], 4000)
        nopt = np.tile([1, 2, 3, 9, 4], 4000).astype(np.int32)
        assert np.array_equal(hip.random_maybe_swap(prob, nopt, 3, 0), ro.random_maybe_swap(prob, nopt, 3, 0))
        assert np.array_equal(hip.random_time_prior(10000, 4, 0), ro.random_time_prior(10000, 4, 0))
    finally:
        hip.close()

"""GPU parity (through the C ABI): pair tables, StringPrior scores and the
numeric contract (detmath / Philox) against the CPU oracle. Integer results
are bit-exact; doubles are bit-exact by construction and asserted as such."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pool(vals):
    from pclean_amd.encode import StringPool
    pool = StringPool()
    ids = pool.add_all(vals)
    return pool, ids


@pytest.mark.parametrize("mode", [0, 1])
def test_pair_tables_hospital_bit_exact(hip, oracle, hospital_columns, mode):
    """Every unique (observed, latent) string pair of hospital_dirty.csv, 15 columns (SURVEY §7.2 iii)."""
    allvals = []
    for vals in hospital_columns.values():
        allvals += vals
    pool, _ = _pool(allvals)
    sym, off, _, _ = pool.arrays()
    hip.load_strings(sym, off)
    for t, (col, vals) in enumerate(hospital_columns.items()):
        ids = np.array([pool.index[v] for v in vals], dtype=np.int32)
        hip.build_pair_table(t, ids, ids, mode)
        got = hip.get_pair_table(t, len(ids), len(ids))
        want = oracle.pair_table(sym, off, ids, ids, mode)
        assert np.array_equal(got, want), (col, mode, int(np.sum(got != want)))


def test_pair_table_edge_cases(hip, oracle):
    """Ragged / empty / unicode / transposition cases, rectangular table."""
    obs = ["", "a", "ab", "ba", "abc", "ca", "xbirmingham", "birmingahm", "münchen", "日本語テキスト", "a" * 70]
    lat = ["", "b", "ba", "abc", "birmingham", "munchen", "日本語", "a" * 64 + "b" * 6, "acb"]
    pool, _ = _pool(obs + lat)
    sym, off, _, _ = pool.arrays()
    hip.load_strings(sym, off)
    oi = np.array([pool.index[v] for v in obs], dtype=np.int32)
    li = np.array([pool.index[v] for v in lat], dtype=np.int32)
    for mode in (0, 1):
        hip.build_pair_table(20 + mode, oi, li, mode)
        got = hip.get_pair_table(20 + mode, len(oi), len(li))
        want = oracle.pair_table(sym, off, oi, li, mode)
        assert np.array_equal(got, want), mode
    assert got[obs.index("ca"), lat.index("abc")] == 2  # unrestricted DL


def test_pair_table_random_large(hip, oracle):
    rnd = np.random.default_rng(5)
    words = ["".join(rnd.choice(list("abcdex "), size=rnd.integers(1, 40))) for _ in range(300)]
    pool, ids = _pool(words)
    sym, off, _, _ = pool.arrays()
    hip.load_strings(sym, off)
    ids = np.unique(ids)
    hip.build_pair_table(30, ids[:150], ids, 0)
    got = hip.get_pair_table(30, 150, len(ids))
    assert np.array_equal(got, oracle.pair_table(sym, off, ids[:150], ids, 0))


def test_density_tables_match_oracle(hip, oracle):
    mr, md, ml, nb, logl = hip.get_density_tables()
    L = oracle.lib()
    for r in range(1, mr + 1):
        for d in range(0, md + 1, 3):
            assert nb[r, d] == pytest.approx(L.pco_negbin_logpdf(float(r), 0.9, d), rel=1e-13, abs=1e-13)
    # full AddTypos density assembled the kernels' way == oracle's add_typos.jl restatement
    for Lw in (2, 5, 10, 11, 36, 64):
        for d in (0, 1, 2, 7):
            r = (Lw + 4) // 5
            l = nb[r, d]
            l -= logl[Lw] * d
            l -= 1.629048269010741 * d
            assert l == pytest.approx(oracle.add_typos(d, Lw), rel=1e-14, abs=1e-14)


def test_string_prior_scores(hip, oracle, hospital_columns):
    from pclean_amd.encode import load_lm_params, lm_log_tables
    init, trans = load_lm_params()
    init_l, trans_l = lm_log_tables()
    vals = hospital_columns["City"] + hospital_columns["HospitalName"] + ["", "ab", "Zürich", "x" * 40]
    pool, _ = _pool(vals)
    _, off, lm, _ = pool.arrays()
    got = hip.string_prior_scores(lm, off, 3, 30, init_l, trans_l)
    for s in range(len(pool)):
        want = oracle.string_prior(lm[off[s]:off[s + 1]], 3, 30, init, trans)
        assert got[s] == want, (pool.strings[s], got[s], want)


def test_detmath_and_philox_bit_exact_on_device(hip, oracle):
    L = oracle.lib()
    rnd = np.random.default_rng(11)
    x = np.concatenate([rnd.uniform(-60, 0, 20000), rnd.uniform(-745, 709, 5000), np.exp(rnd.uniform(-300, 300, 20000)),
                        [0.0, 1.0, -28.5, -28.4999, 5e-324, 1e-310, -np.inf, np.inf]])
    e, l, f = hip.debug_detmath(x)
    for i in range(len(x)):
        he, hl, hf = L.pco_det_exp(float(x[i])), L.pco_det_log(float(x[i])), L.pco_fixw(float(x[i]))
        assert (e[i] == he) or (np.isnan(e[i]) and np.isnan(he)), x[i]
        assert (l[i] == hl) or (np.isnan(l[i]) and np.isnan(hl)), x[i]
        assert int(f[i]) == hf, x[i]
    rows = np.arange(0, 5000, dtype=np.uint32)
    got = hip.debug_rand64(0x1234567890abcdef, rows, 0x10002, 7, 3)
    for r in (0, 1, 77, 4999):
        assert int(got[r]) == L.pco_rand64(0x1234567890abcdef, int(r), 0x10002, 7, 3)

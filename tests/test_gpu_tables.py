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


def test_pair_table_long_patterns_and_dl_random(hip, oracle):
    """Multi-word bit-parallel OSA (observed strings of 65..250 symbols incl. transpositions across the 64-bit
    word boundaries) and the LDS-matrix unrestricted DL kernel on random short strings, against the oracle's DP."""
    rnd = np.random.default_rng(11)
    alpha = list("abcdefgh ")
    long_words = []
    for n in (63, 64, 65, 66, 100, 127, 128, 129, 130, 184, 192, 193, 250):
        w = rnd.choice(alpha, size=n)
        long_words.append("".join(w))
        sw = w.copy()
        for pos in (n // 2, min(63, n - 2), min(64, n - 2), min(127, n - 2), n - 2):  # adjacent swaps around word boundaries
            if 0 <= pos < n - 1:
                sw[pos], sw[pos + 1] = sw[pos + 1], sw[pos]
        long_words.append("".join(sw))
        long_words.append("".join(np.delete(w, rnd.integers(0, n, size=3))))
    pool, ids = _pool(long_words)
    sym, off, _, _ = pool.arrays()
    hip.load_strings(sym, off)
    ids = np.unique(ids)
    hip.build_pair_table(31, ids, ids, 0)
    assert np.array_equal(hip.get_pair_table(31, len(ids), len(ids)), oracle.pair_table(sym, off, ids, ids, 0))
    # the same long strings through the unrestricted-DL kernel (dl_wave_kernel: 64 lanes per pair, up to 4 column chunks
    # per lane; the 9-letter alphabet makes gapped transpositions common)
    hip.build_pair_table(31, ids, ids, 1)
    assert np.array_equal(hip.get_pair_table(31, len(ids), len(ids)), oracle.pair_table(sym, off, ids, ids, 1))
    # every lane-group shape of that kernel: strings of at most 16 / 32 / 64 / 128 / 192 symbols, empty strings included
    for top in (16, 32, 33, 64, 100, 160):
        words = ["".join(rnd.choice(list("abcx"), size=int(rnd.integers(0, top + 1)))) for _ in range(90)] + ["", "a" * top]
        pool2, ids2 = _pool(words)
        sym2, off2, _, _ = pool2.arrays()
        hip.load_strings(sym2, off2)
        ids2 = np.unique(ids2)
        hip.build_pair_table(31, ids2, ids2, 1)
        assert np.array_equal(hip.get_pair_table(31, len(ids2), len(ids2)), oracle.pair_table(sym2, off2, ids2, ids2, 1)), top
    short = ["".join(rnd.choice(list("abcx"), size=rnd.integers(0, 30))) for _ in range(400)]
    pool, ids = _pool(short)
    sym, off, _, _ = pool.arrays()
    hip.load_strings(sym, off)
    ids = np.unique(ids)
    for mode in (0, 1):
        hip.build_pair_table(32 + mode, ids[:200], ids, mode)
        got = hip.get_pair_table(32 + mode, 200, len(ids))
        assert np.array_equal(got, oracle.pair_table(sym, off, ids[:200], ids, mode)), mode


def test_dl_seg_kernel_every_length_class_and_symbol_width(hip, oracle, monkeypatch):
    """The linear-space unrestricted-DL kernel (dl_seg_kernel: latent strings sorted by length, one launch per length
    class with 1 / 2 / 4 / 8 lanes per pair, results un-permuted) on tables that hold every class at once, with 8-bit and
    16-bit symbols (> 255 distinct characters), rectangular shapes whose row length is not a multiple of four, and strings
    beyond its 254 symbols (the table then takes the matrix kernels): against the oracle's DP, and against the LDS-matrix
    kernel it replaces (PCLEAN_DL_KERNEL=wave) on every pair."""
    rnd = np.random.default_rng(23)
    for wide in (False, True):
        alpha = [chr(0x4e00 + k) for k in range(300)] if wide else list("abcdx ")
        lens = [0, 1, 2, 5, 31, 32, 33, 63, 64, 65, 100, 127, 128, 129, 200, 253, 254] + [int(x) for x in rnd.integers(0, 255, size=60)]
        base = ["".join(rnd.choice(alpha, size=n)) for n in lens]
        muts = []
        for w in base:  # near copies: substitutions, adjacent and gapped transpositions
            c = list(w)
            for _ in range(int(rnd.integers(0, 4))):
                if len(c) > 3:
                    q = int(rnd.integers(0, len(c) - 2))
                    c[q], c[q + 1] = c[q + 1], c[q]
                    if rnd.random() < 0.5:
                        c.insert(q + 1, alpha[int(rnd.integers(len(alpha)))])
                    if rnd.random() < 0.5:
                        c[int(rnd.integers(len(c)))] = alpha[int(rnd.integers(len(alpha)))]
            muts.append("".join(c)[:254])
        pool, _ = _pool(base + muts)
        sym, off, _, _ = pool.arrays()
        hip.load_strings(sym, off)
        li = np.unique(np.array([pool.index[v] for v in base + muts], dtype=np.int32))
        oi = li[::2][:61]
        assert len(li) % 4 != 0 or len(li[:-1]) % 4 != 0
        for lat in (li, li[:-1], li[:-2], li[:-3]):
            monkeypatch.setenv("PCLEAN_DL_KERNEL", "seg")
            hip.build_pair_table(42, oi, lat, 1)
            got = hip.get_pair_table(42, len(oi), len(lat))
            monkeypatch.setenv("PCLEAN_DL_KERNEL", "wave")
            hip.build_pair_table(43, oi, lat, 1)
            old = hip.get_pair_table(43, len(oi), len(lat))
            assert np.array_equal(got, old), (wide, len(lat), int((got != old).sum()))
        monkeypatch.delenv("PCLEAN_DL_KERNEL")
        assert np.array_equal(got, oracle.pair_table(sym, off, oi, lat, 1)), wide
    # strings beyond 254 symbols: the table falls back to the matrix kernels and is still right
    words = ["".join(rnd.choice(list("abc"), size=n)) for n in (3, 40, 255, 256, 200)]
    pool, ids = _pool(words)
    sym, off, _, _ = pool.arrays()
    hip.load_strings(sym, off)
    ids = np.unique(ids)
    hip.build_pair_table(42, ids, ids, 1)
    assert np.array_equal(hip.get_pair_table(42, len(ids), len(ids)), oracle.pair_table(sym, off, ids, ids, 1))


def test_synthetic_table_osa_vs_dl(hip, capsys):
    """The 1M-row bench builds its pair tables with the (bit-parallel) restricted distance (OSA; the semantics of
    StringDistances < 0.11) while the real datasets default to unrestricted Damerau-Levenshtein.  The two flavours
    are NOT identical on the synthetic generator's strings: random digit strings (provider numbers, zip codes,
    phone numbers) hold gapped transpositions.  Measured here at 30k rows / 600 hospitals: they differ on well
    under 1 % of the (observed, latent) pairs of a column, never by more than one edit, and only where the
    distance is >= 2 (an observed value and its own clean value are never affected)."""
    from pclean_amd import experiments as ex
    from pclean_amd.model import LoweredModel
    from pclean_amd.synth import synth_hospital
    dirty, clean, _ = synth_hospital(30_000, 600, 3)
    m = ex.hospital_model(ex.possibilities_of(dirty))
    lw = LoweredModel(m, ex.hospital_query(m), dirty)
    sym, off, _, _ = lw.pool.arrays()
    hip.load_strings(sym, off)
    n_pairs = n_diff = 0
    report = []
    for key, (pid, odom, ldom) in lw.pair_id.items():
        oi, li = odom.id_array(), ldom.id_array()
        if lw.pool.lens[oi].max() > 60 or lw.pool.lens[li].max() > 60:
            oi, li = oi[:400], li[:4000]  # the 90-character measure names: a sample keeps the exact DL kernel quick
        hip.build_pair_table(40, oi, li, 0)
        a = hip.get_pair_table(40, len(oi), len(li)).astype(np.int32)
        hip.build_pair_table(41, oi, li, 1)
        b = hip.get_pair_table(41, len(oi), len(li)).astype(np.int32)
        diff = a != b
        assert (b <= a).all(), key                      # unrestricted DL never exceeds the restricted distance
        assert (b[diff] >= 2).all(), key                # ... and only differs between strings at least 2 edits apart
        assert diff.mean() < 0.05, (key, float(diff.mean()))
        report.append((key[0], a.size, int(diff.sum()), int((a - b).max())))
        n_pairs += a.size
        n_diff += int(diff.sum())
    assert n_pairs > 10_000_000
    with capsys.disabled():
        print("\n[osa vs dl] column pairs differing: " + ", ".join(f"{c} {d}/{n} (max {mx})" for c, n, d, mx in report)
              + f"; total {n_diff}/{n_pairs} = {100.0 * n_diff / n_pairs:.3f} %")

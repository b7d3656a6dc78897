"""CPU tests of the host-side mirror: DSL lowering, trace commit / GC, accuracy, config."""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

import helpers


@pytest.fixture(scope="module")
def S():
    return helpers.hospital_setup()


def test_lowering_matches_survey_appendix_b(S):
    lw, tr = S["lw"], S["trace"]
    # derived model shapes: 2 blocks in Record, 11 + 4 observation terms at the roots
    assert len(lw.blocks) == 2
    b0, b1 = lw.blocks
    assert b0["root_class"] == "Hospital" and b1["root_class"] == "Measure"
    assert b0["nodes"][0][3] == 11 and b1["nodes"][0][3] == 4
    assert b1["ctx_src_block"] == [0] and b1["ctx_src_col"] == [lw.colidx["Hospital"]["loc.county.state"]]
    # flattened layouts (dependency_tracking.jl:88-96): a Hospital row stores Place/County/Type values too
    assert [c.name for c in lw.layout["Hospital"]] == ["loc", "loc.county", "loc.county.state", "loc.county.county",
                                                       "loc.city", "type", "type.desc", "provider", "name", "addr",
                                                       "phone", "owner", "zip", "service"]
    # dataset facts of BASELINE.md §2
    assert tr.tables["Hospital"].n == 45 and tr.tables["Measure"].n == 28 and tr.tables["Condition"].n == 5
    assert S["obs"].shape == (15, 1000) and (S["obs"] >= 0).all()
    # the stateavg JuliaNode is tabulated over state x code
    fn = lw.fn_tables[0]
    assert fn.shape == (len(lw.latent_dom[("County", "state")]), len(lw.latent_dom[("Measure", "code")]))
    dom = [v for k, v in lw.pair_id.items() if k[0] == "stateavg_obs"][0][2]
    assert dom.string(fn[0, 0]) == f"{lw.latent_dom[('County', 'state')].string(0)}_{lw.latent_dom[('Measure', 'code')].string(0)}"


def test_inference_config_defaults():
    from pclean_amd.engine import InferenceConfig
    c = InferenceConfig(1, 20, use_mh_instead_of_pg=True)
    assert c.num_particles == 2  # infer_config.jl:11-13
    c = InferenceConfig(3, 20)
    assert (c.use_dd_proposals, c.use_lo_sweeps, c.use_mh_instead_of_pg, c.rejuv_frequency, c.reporting_frequency) == (
        True, True, False, 50, 100)


def test_initial_trace_counts_and_accuracy(S):
    from pclean_amd.analysis import evaluate_accuracy
    lw, tr = S["lw"], S["trace"]
    for bi, blk in enumerate(lw.blocks):
        t = tr.tables[blk["root_class"]]
        assert np.array_equal(np.bincount(tr.cur[bi], minlength=t.n), t.counts[:t.n])
    assert tr.tables["Place"].counts[:tr.tables["Place"].n].sum() == tr.tables["Hospital"].n
    acc = evaluate_accuracy(lw, tr, S["dirty"], S["clean"])
    assert acc["errors"] == 509  # dirty cells over all common columns
    assert acc["precision"] > 0.99 and acc["recall"] > 0.8 and acc["imputed"] == 0


def test_oracle_sweeps_commit_and_gc(S, oracle):
    """Three batched oracle sweeps + product commit: new latent rows appear, emptied rows are
    garbage-collected recursively, counts stay consistent, Dirichlet counts follow."""
    from pclean_amd._lib import InferConfig
    S2 = helpers.hospital_setup(n_rows=500, seed=1)
    lw, tr, obs = S2["lw"], S2["trace"], S2["obs"]
    created = 0
    for sweep in range(3):
        logp = helpers.option_logp_cpu(oracle, lw, tr)
        w = helpers.mirror_world(oracle, lw, obs, tr, None, 1, logp)
        nb, n = tr.cur.shape
        choice = np.empty((nb, n), dtype=np.int32)
        c = InferConfig(1, 4, 1, 1, 0, 50, 100)
        oracle.lib().pco_sweep_batched(w.h, C.byref(c), C.c_uint64(5), C.c_uint32(sweep), nb, C.c_int64(0),
                                       oracle._p(np.ascontiguousarray(tr.cur), C.c_int32),
                                       oracle._p(choice, C.c_int32), None, None)
        new_rows = {}
        for b, blk in enumerate(lw.blocks):
            k = oracle.lib().pco_new_rows_count(b)
            if k:
                rows = np.empty(k, dtype=np.int32)
                vals = np.empty((k, len(blk["nodes"])), dtype=np.int32)
                oracle.lib().pco_new_rows_get(b, len(blk["nodes"]), oracle._p(rows, C.c_int32), oracle._p(vals, C.c_int32))
                new_rows[b] = (rows, vals)
                created += k
        tr.commit(choice, new_rows)
        for bi, blk in enumerate(lw.blocks):
            t = tr.tables[blk["root_class"]]
            assert np.array_equal(np.bincount(tr.cur[bi], minlength=t.n), t.counts[:t.n])
        for cname, t in tr.tables.items():
            assert np.all((t.counts[:t.n] > 0) == t.live[:t.n])
        # Place counts = number of live hospitals pointing at them, etc.
        H, P = tr.tables["Hospital"], tr.tables["Place"]
        live_h = np.nonzero(H.live[:H.n])[0]
        assert np.array_equal(np.bincount(H.cols[lw.colidx["Hospital"]["loc"], live_h], minlength=P.n), P.counts[:P.n])
        p = tr.params[("Hospital", "owner_dist")]
        assert p.counts.sum() == len(live_h) and np.all(p.counts >= 0)
    assert created > 0


def test_missing_observation_encoding():
    from pclean_amd.model import AddTypos, ChooseUniformly, LoweredModel, Model, Query
    m = Model()
    a = m.add_class("A")
    a.choice("x", ChooseUniformly(["aa", "bb"]))
    o = m.add_class("Obs")
    o.fk("a", "A")
    o.choice("y", AddTypos("a.x"))
    lw = LoweredModel(m, Query(m, "Obs", {"Y": ("a.x", "y")}), {"Y": ["aa", None, "bx"]})
    obs = lw.encode_observations({"Y": ["aa", None, "bx"]})
    assert obs.tolist() == [[0, -1, 1]]


def test_save_results_and_bulk_commit_equivalence(S, tmp_path):
    """save_results (analysis.jl:15-33) writes the reconstructed table and one file per latent class; the
    array-based row creation / deletion of trace.py equals the row-by-row path."""
    import os

    import pandas as pd

    from pclean_amd.analysis import reconstructed_table, save_results
    lw, tr, dirty, clean = S["lw"], S["trace"], S["dirty"], S["clean"]
    d = save_results(str(tmp_path), "hospital", lw, tr, dirty, timestamp=False)
    files = sorted(os.listdir(d))
    assert "reconstructed_Record.csv" in files and "inferred_Hospital.csv" in files and len(files) == 1 + len(tr.tables)
    rec = pd.read_csv(os.path.join(d, "reconstructed_Record.csv"), dtype=str, keep_default_na=False)
    assert len(rec) == tr.cur.shape[1] and list(rec.columns) == list(dirty.keys())
    # the initial trace holds the clean values wherever they are possible latent values
    same = sum(a == b for a, b in zip(rec["City"], clean["City"]))
    assert same > 0.95 * len(rec)
    assert reconstructed_table(lw, tr, dirty)["ProviderNumber"][0] == rec["ProviderNumber"][0]
    hosp = pd.read_csv(os.path.join(d, "inferred_Hospital.csv"), dtype=str, keep_default_na=False)
    assert len(hosp) == tr.tables["Hospital"].n_live and {"id", "loc", "type", "name", "zip"} <= set(hosp.columns)

    # bulk vs sequential creation of new Measure rows (block 1), then deletion: identical tables
    import copy
    S2 = helpers.hospital_setup(n_rows=120, seed=2)
    lw = S2["lw"]
    t1, t2 = S2["trace"], copy.deepcopy(S2["trace"])
    blk = lw.blocks[1]
    rng = np.random.default_rng(0)
    nn = len(blk["nodes"])
    vals = np.zeros((40, nn), dtype=np.int32)
    for cn, node in enumerate(blk["nodes"]):
        if cn == 0:
            vals[:, cn] = -1
        elif node[0] == 1:
            info = blk["node_info"][cn]
            vals[:, cn] = rng.integers(0, len(lw.option_values[(info["cls"], info["attr"])]), 40)
        else:
            vals[:, cn] = rng.integers(0, t1.tables[blk["node_info"][cn]["cls"]].n, 40)
    vals[::7, [cn for cn, node in enumerate(blk["nodes"]) if cn and node[0] == 0][0]] = -1  # some nested NEW referents
    a = t1.materialise_bulk(1, vals)
    b = np.array([t2._materialise(1, 0, v) for v in vals])
    m1, m2 = t1.tables["Measure"], t2.tables["Measure"]
    # row ids differ only by the order in which simple and nested proposals are created; contents must agree
    assert sorted(map(tuple, m1.cols[:, a].T.tolist())) == sorted(map(tuple, m2.cols[:, b].T.tolist()))
    for c in t1.tables:
        assert t1.tables[c].n_live == t2.tables[c].n_live or c == "Measure"
        assert t1.tables[c].counts[:t1.tables[c].n].sum() == t2.tables[c].counts[:t2.tables[c].n].sum()
    t1.delete_rows_bulk("Measure", a)
    for r in b:
        t2.delete_row("Measure", int(r))
    t1.check_consistency()
    t2.check_consistency()
    for c in t1.tables:
        assert t1.tables[c].n_live == t2.tables[c].n_live
        assert np.array_equal(np.sort(t1.tables[c].counts[:t1.tables[c].n][t1.tables[c].live[:t1.tables[c].n]]),
                              np.sort(t2.tables[c].counts[:t2.tables[c].n][t2.tables[c].live[:t2.tables[c].n]]))


def test_shuffled_rows_keep_the_batched_init_from_spawning_duplicates(oracle):
    """The hospital table ships sorted by entity — the worst case for the batched initialize_trace WITHOUT its
    in-batch merge pass (a batch only sees latent rows of earlier batches).  In random order
    (experiments.shuffle_rows) far fewer duplicate hospitals are created; with the merge pass (the default,
    inference.jl:20-37 emulated) the file order ends within a third of the random order's entity count."""
    from oracle_engine import OracleEngine
    from pclean_amd import experiments as ex
    from pclean_amd.engine import InferenceConfig
    from pclean_amd.inference import initialize_trace
    from pclean_amd.model import LoweredModel
    from pclean_amd.trace import Trace
    dirty, clean = ex.hospital_data()
    dirty = {c: v[:400] for c, v in dirty.items()}
    n_hosp = {}
    for shuffled, rounds in ((False, 0), (True, 0), (False, 2), (True, 2)):
        d = dirty
        if shuffled:
            (d,), perm = ex.shuffle_rows([dirty], 0)
            assert sorted(perm.tolist()) == list(range(400)) and d["City"][0] == dirty["City"][perm[0]]
        m = ex.hospital_model(ex.possibilities_of(d))
        lw = LoweredModel(m, ex.hospital_query(m), d)
        obs = lw.encode_observations(d)
        tr = Trace(lw, obs.shape[1], 0)
        initialize_trace(OracleEngine(oracle, lw, obs, cached=True), tr, InferenceConfig(1, 2, use_mh_instead_of_pg=True), 1,
                         max_batch=128, merge_rounds=rounds)
        tr.check_consistency()
        n_hosp[(shuffled, rounds)] = tr.tables["Hospital"].n_live
    true_hospitals = len(set(dirty["ProviderNumber"]))  # incl. a few typo'd provider numbers
    assert n_hosp[(True, 0)] < 0.5 * n_hosp[(False, 0)] and n_hosp[(True, 0)] <= 2 * true_hospitals
    assert n_hosp[(False, 2)] <= 1.34 * n_hosp[(True, 2)] and n_hosp[(False, 2)] < 0.35 * n_hosp[(False, 0)], n_hosp


def test_rents_pipeline_on_cpu(oracle, tmp_path):
    """rents (numeric Gaussian column, keyed atoms, missing cells) through the product's host code with the
    oracle as engine: evaluate_accuracy's numeric comparison and save_results."""
    import os

    from oracle_engine import OracleEngine
    from pclean_amd import experiments as ex
    from pclean_amd.analysis import evaluate_accuracy, save_results
    from pclean_amd.engine import InferenceConfig
    from pclean_amd.inference import initialize_trace, run_inference
    from pclean_amd.model import LoweredModel
    from pclean_amd.trace import Trace
    dirty, clean = ex.rents_data()
    (dirty, clean), _ = ex.shuffle_rows([dirty, clean], 0)
    dirty = {c: v[:800] for c, v in dirty.items()}
    clean = {c: v[:800] for c, v in clean.items()}
    m = ex.rents_model(dirty)
    lw = LoweredModel(m, ex.rents_query(m), dirty)
    obs = lw.encode_observations(dirty)
    tr = Trace(lw, obs.shape[1], 0)
    cfg = InferenceConfig(1, 2, use_mh_instead_of_pg=True, rejuv_frequency=500)
    eng = OracleEngine(oracle, lw, obs)
    initialize_trace(eng, tr, cfg, 0, max_batch=256)
    run_inference(eng, tr, cfg, 0)
    tr.check_consistency()
    acc = evaluate_accuracy(lw, tr, dirty, clean)
    n_missing = sum(v is None for c in ("State", "Room Type") for v in dirty[c])
    assert acc["imputed"] == n_missing > 50 and acc["correctly_imputed"] > 0.3 * acc["imputed"]
    assert acc["errors"] > 10 and 0.2 < acc["f1"] < 1.0
    d = save_results(str(tmp_path), "rents", lw, tr, dirty, timestamp=False)
    assert sorted(os.listdir(d)) == ["inferred_County.csv", "reconstructed_Obs.csv"]


def test_pitman_yor_score_lgamma_form_matches_oracle_direct_form(oracle):
    """trace.pitman_yor_score (O(K), lgamma form) against the oracle's literal loop over every customer
    (trace.jl:65-78) on random count vectors and hyper-parameters."""
    import numpy as np

    from pclean_amd.trace import Trace
    rng = np.random.default_rng(5)
    for _ in range(60):
        k = int(rng.integers(1, 40))
        counts = rng.integers(1, 200, size=k).astype(np.int64)
        if rng.random() < 0.3:
            counts[rng.integers(0, k)] = 1
        s, d = float(rng.gamma(1.0, 1.0)) + 1e-3, float(rng.random()) * 0.999
        if rng.random() < 0.2:
            d = 0.0
        got = Trace.pitman_yor_score(s, d, counts)
        want = oracle.pitman_yor_score(s, d, counts)
        assert abs(got - want) <= 1e-9 * max(1.0, abs(want)), (s, d, counts, got, want)


def test_bulk_commit_helpers_match_their_row_by_row_definitions():
    """trace.unique_rows (hashed grouping of new-row proposals) against np.unique(axis=0) + first-occurrence order,
    LatentTable.alloc_many against repeated alloc() (free list consumed from its end, then fresh ids)."""
    from pclean_amd.trace import LatentTable, unique_rows
    rng = np.random.default_rng(1)
    for k in (0, 1, 5, 1000):
        v = rng.integers(-2, 4, size=(k, 5)).astype(np.int32)
        first, grp = unique_rows(v)
        if k:
            u, f, inv = np.unique(v, axis=0, return_index=True, return_inverse=True)
            o = np.argsort(f, kind="stable")
            rank = np.empty(len(o), np.int64)
            rank[o] = np.arange(len(o))
            assert np.array_equal(first, f[o]) and np.array_equal(grp, rank[np.asarray(inv).reshape(-1)])
    t1, t2 = LatentTable(3), LatentTable(3)
    for t in (t1, t2):
        for _ in range(20):
            t.alloc()
        t.free = [3, 7, 11, 15]
    for k in (9, 2, 0):
        assert [t1.alloc() for _ in range(k)] == list(t2.alloc_many(k))
        assert t1.n == t2.n and t1.free == t2.free and t2.cols.shape[1] >= t2.n
    t1.free, t2.free = [1, 2, 3], [1, 2, 3]
    assert [t1.alloc() for _ in range(2)] == list(t2.alloc_many(2)) and t1.free == t2.free


def test_chosen_dummy_values_get_their_prior_draw(oracle):
    """block_proposal.jl:58-60 through inference.resample_dummies: every placeholder of a TimePrior attribute is
    replaced by a random(TimePrior) string that joins the latent domain (ids of the atoms and the dummy keep their
    place, the lowering is rebuilt in place, the engine reloads); deterministic in (seed, stamp); sweeps go on."""
    from oracle_engine import OracleEngine
    from pclean_amd.engine import InferenceConfig
    from pclean_amd.inference import resample_dummies
    got = []
    for rep in range(2):
        S = helpers.flights_setup()
        lw, tr, obs = S["lw"], S["trace"], S["obs"]
        eng = OracleEngine(oracle, lw, obs, cached=True)
        t = tr.tables["Flight"]
        before = t.cols[:, :t.n].copy()
        sizes = {k: len(d) for k, d in lw.latent_dom.items()}
        atoms = {k: [d.string(i) for i in range(len(d))] for k, d in lw.latent_dom.items()}
        dummies = {}
        for a in ("sdt", "sat", "adt", "aat"):
            j = lw.colidx["Flight"][a]
            dummies[a] = (j, lw.latent_dom[("Flight", a)].get("**:** p.m."))
        n_dummy = sum(int(((t.cols[j, :t.n] == dv) & t.live[:t.n]).sum()) for j, dv in dummies.values())
        assert n_dummy > 5
        assert resample_dummies(eng, tr, 7, 1) == n_dummy
        assert resample_dummies(eng, tr, 7, 2) == 0  # nothing left to replace
        for k, d in lw.latent_dom.items():  # domains only grew at their end
            assert len(d) >= sizes[k] and [d.string(i) for i in range(sizes[k])] == atoms[k]
        new_strings = []
        for a, (j, dv) in dummies.items():
            col = t.cols[j, :t.n]
            assert not ((col == dv) & t.live[:t.n]).any()
            moved = np.flatnonzero(col != before[j])
            assert (before[j, moved] == dv).all()
            new_strings += [lw.latent_dom[("Flight", a)].string(int(v)) for v in col[moved]]
        assert all(re.match(r"^[0-9]?[0-9]:[0-9]?[0-9] [ap]\.m\.$", s_) for s_ in new_strings)
        tr.check_consistency()
        choice, chosen, logml, new_rows = eng.sweep(tr, InferenceConfig(1, 2, use_mh_instead_of_pg=True), 1, 0)
        assert np.isfinite(logml).all()
        got.append(new_strings)
        if rep == 0:
            # the drawn strings are values OUTSIDE the options (maybe_swap.jl:18 `in(val, options)`) — also the ~20 %
            # of them that spell an atom of some other flight: the literal interpreter (strings) and the oracle (ids)
            # must agree on the one-particle log marginal likelihood of every row of such a flight
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
            sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
            import literal as lit
            import make_literal_fixtures_flights as mk
            foreign = sum(s_ in [lw.latent_dom[("Flight", a)].string(i) for i in range(sizes[("Flight", a)])]
                          for a in dummies for s_ in new_strings)
            assert foreign > 0  # the interesting case occurs
            choice, chosen, logml, new_rows = eng.sweep(tr, InferenceConfig(1, 1), 3, 0)
            assert np.array_equal(choice, tr.cur)
            lt0 = lit.lit_trace_from(lw, tr)
            changed = np.flatnonzero((t.cols[:, :t.n] != before).any(axis=0))
            rows = np.flatnonzero(np.isin(tr.cur[0], changed))[::3]
            assert len(rows) > 40
            for i in rows:
                want = mk.row_fixture(dict(S, trace=tr), lt0, int(i))["logml"]
                assert abs(logml[i] - want) <= 1e-9 * max(1.0, abs(want)), (i, logml[i], want)
    assert got[0] == got[1]


def test_unobserved_string_attributes_get_bigram_draws(oracle):
    """Hospitals whose records lost their name and phone cells: the new Hospital row's StringPrior attributes have no
    observation, the enumeration chooses the ProposalDummyValue (99.99 % of the prior mass) and random(StringPrior)
    supplies the value (block_proposal.jl:58-60).  The drawn strings join the latent domain AFTER the dummy and are
    never options of a later proposal; inference goes on (unkeyed domains: option tables keep their size)."""
    from oracle_engine import OracleEngine
    from pclean_amd import experiments as ex
    from pclean_amd.engine import InferenceConfig
    from pclean_amd.inference import initialize_trace, run_inference
    from pclean_amd.model import LoweredModel
    from pclean_amd.trace import Trace
    dirty, clean = ex.hospital_data()
    dirty = {c: list(v[:240]) for c, v in dirty.items()}
    prov = list(dict.fromkeys(dirty["ProviderNumber"]))[:3]
    for i in range(240):
        if dirty["ProviderNumber"][i] in prov:
            dirty["HospitalName"][i] = None
            dirty["PhoneNumber"][i] = None
    (d,), _ = ex.shuffle_rows([dirty], 0)
    m = ex.hospital_model(ex.possibilities_of(d))
    lw = LoweredModel(m, ex.hospital_query(m), d)
    obs = lw.encode_observations(d)
    n_options = len(lw.option_values[("Hospital", "name")])
    tr = Trace(lw, obs.shape[1], 0)
    eng = OracleEngine(oracle, lw, obs, cached=True)
    cfg = InferenceConfig(2, 2, use_mh_instead_of_pg=True)
    initialize_trace(eng, tr, cfg, 1, max_batch=64)
    run_inference(eng, tr, cfg, 1)
    tr.check_consistency()
    for attr, lo, hi in (("name", 3, 50), ("phone", 10, 10)):
        dom = lw.latent_dom[("Hospital", attr)]
        extras = list(getattr(dom, "extra", {}))
        assert extras and all(lo <= len(s_) <= hi and set(s_) <= set("abcdefghijklmnopqrstuvwxyz .") for s_ in extras)
        assert len(lw.option_values[("Hospital", attr)]) == dom.n_base() == len(dom) - len(extras)
        t = tr.tables["Hospital"]
        col = t.cols[lw.colidx["Hospital"][attr], :t.n][t.live[:t.n]]
        assert not (col == dom.get(m.classes["Hospital"].attr(attr).dist.dummy_value())).any()  # no placeholder left
    assert len(lw.option_values[("Hospital", "name")]) == n_options
    # every one of these dummies had no observation below it: the sweeps' particle weights were the reference's
    assert tr.dummy_cases["unobserved"] >= 6 and tr.dummy_cases["observed"] == 0


def test_stable_argsort_ids_equals_numpy_stable_sort():
    """inference.stable_argsort_ids (16-bit radix digits) against np.argsort(kind="stable") on id-like keys."""
    from pclean_amd.inference import stable_argsort_ids
    rng = np.random.default_rng(1)
    for n, lo, hi in ((0, 0, 1), (10, -1, 5), (100000, -1, 10500), (200000, 0, 300000), (1000, 5, 6), (5000, 70000, 70010)):
        k = rng.integers(lo, hi, n).astype(np.int32) if n else np.zeros(0, np.int32)
        assert np.array_equal(stable_argsort_ids(k), np.argsort(k, kind="stable")), (n, lo, hi)


def test_prior_proposals_end_to_end(oracle):
    """use_dd_proposals = false through the product's host code (initialize_trace + run_inference over every class) with
    the oracle engine: the trace stays consistent; the data-driven proposals do better on the same budget."""
    import helpers
    from oracle_engine import OracleEngine
    from pclean_amd.analysis import evaluate_accuracy
    from pclean_amd.engine import InferenceConfig
    from pclean_amd.inference import initialize_trace, run_inference
    from pclean_amd.trace import Trace
    S = helpers.hospital_setup(n_rows=150)
    lw, obs = S["lw"], S["obs"]
    f1 = {}
    for dd in (False, True):
        eng = OracleEngine(oracle, lw, obs)
        tr = Trace(lw, obs.shape[1], 1)
        cfg = InferenceConfig(1, 4, use_dd_proposals=dd)
        initialize_trace(eng, tr, cfg, 5, max_batch=32)
        run_inference(eng, tr, cfg, 5)
        tr.check_consistency()
        f1[dd] = evaluate_accuracy(lw, tr, S["dirty"], S["clean"])["f1"]
    assert 0.0 <= f1[False] <= f1[True]


def test_prior_proposals_end_to_end_flights(oracle):
    """use_dd_proposals = false on a plan with equality-constrained slots, a MaybeSwap scoring block and MaybeSwap evidence
    in the latent sweeps (flights): the product's host code runs through with the oracle engine and leaves a consistent
    trace.  (Prior draws almost never satisfy a noise-free observation: nearly every weight is -inf and the result is as
    poor as the reference's would be — the flag is there for completeness, block_proposal.jl:168.)"""
    from oracle_engine import OracleEngine
    from test_flights_cpu import flights_setup
    from pclean_amd.analysis import evaluate_accuracy
    from pclean_amd.engine import InferenceConfig
    from pclean_amd.inference import initialize_trace, run_inference
    from pclean_amd.trace import Trace
    dirty, clean, lw, obs = flights_setup()
    eng = OracleEngine(oracle, lw, obs)
    tr = Trace(lw, obs.shape[1], 1)
    cfg = InferenceConfig(1, 3, use_dd_proposals=False, rejuv_frequency=500)
    initialize_trace(eng, tr, cfg, 5, max_batch=256)
    run_inference(eng, tr, cfg, 5)
    tr.check_consistency()
    assert 0.0 <= evaluate_accuracy(lw, tr, dirty, clean)["f1"] < 0.5


def test_prior_proposals_end_to_end_rents(oracle):
    """use_dd_proposals = false on rents (Gaussian term with own enumerated choices; Gaussian evidence in the County sweeps):
    the product's host code runs through with the oracle engine, every row ends with own choices, the trace is consistent,
    and the data-driven proposals do better on the same budget."""
    import helpers
    from oracle_engine import OracleEngine
    from pclean_amd.analysis import evaluate_accuracy
    from pclean_amd.engine import InferenceConfig
    from pclean_amd.inference import initialize_trace, run_inference
    from pclean_amd.trace import Trace
    R = helpers.rents_setup(n_rows=300)
    lw, obs = R["lw"], R["obs"]
    f1 = {}
    for dd in (False, True):
        eng = OracleEngine(oracle, lw, obs)
        tr = Trace(lw, obs.shape[1], 1)
        cfg = InferenceConfig(2, 4, use_dd_proposals=dd)
        initialize_trace(eng, tr, cfg, 5, max_batch=64)
        run_inference(eng, tr, cfg, 5)
        tr.check_consistency()
        assert (tr.locals[0] >= 0).all()
        f1[dd] = evaluate_accuracy(lw, tr, R["dirty"], R["clean"])["f1"]
    assert 0.0 <= f1[False] < f1[True]


def test_py_moves_commute_with_the_sweep_of_the_class_own_rows(oracle):
    """A latent class without learned parameters is swept in ONE batch and its table's Pitman-Yor moves are made after
    the batch (inference.latent_sweep).  The claim behind it: those moves' conditional (the table's reference counts,
    its rows) is untouched by updates of the class's own rows, and the row updates do not consume the trace's RNG — so
    the hyper-parameters after the sweep are THE SAME NUMBERS as under the cut schedule with interleaved moves (what
    differs between the two schedules is only how stale the frozen tables are for the later rows)."""
    from oracle_engine import OracleEngine
    from pclean_amd.engine import InferenceConfig
    from pclean_amd.inference import initialize_trace, latent_sweep
    from pclean_amd.trace import Trace
    S = helpers.hospital_setup(n_rows=400)
    lw, obs = S["lw"], S["obs"]
    cfg = InferenceConfig(1, 2, use_mh_instead_of_pg=True, rejuv_frequency=10)
    out = {}
    for mode in ("one_batch", "cut"):
        tr = Trace(lw, obs.shape[1], 7)
        eng = OracleEngine(oracle, lw, obs)
        initialize_trace(eng, tr, cfg, 7)
        cname = next(c for c in lw.model.class_order if c in lw.latent_plans and not tr.has_learned_parameters(c)
                     and tr.tables[c].n_live > 40)
        if mode == "cut":  # the round-2 schedule: sub-batches with the moves in between
            tr.has_learned_parameters = lambda c: True
        t = tr.tables[cname]
        counts_before = t.counts[:t.n].copy()
        latent_sweep(eng, tr, cname, cfg, 7, 0)
        assert np.array_equal(tr.tables[cname].counts[:len(counts_before)], counts_before)  # own sweep: counts untouched
        out[mode] = (cname, tr.tables[cname].strength, tr.tables[cname].discount)
    assert out["one_batch"] == out["cut"], out

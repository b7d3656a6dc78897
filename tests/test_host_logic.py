"""CPU tests of the host-side mirror: DSL lowering, trace commit / GC, accuracy, config."""
import ctypes as C

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

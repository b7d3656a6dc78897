"""The `clinic` program (tests/clinic_program.py): a block with three reference slots, a single-argument JuliaNode,
JuliaNodes across the slots of one block (two context values in the third slot's plan), latent classes with several
per-evidence-row context sources.  CPU: the C++ oracle reproduces the literal interpreter's per-candidate scores
(tests/golden/literal_scores_clinic.json) through the product's lowering, and the whole inference runs through the host
code.  GPU: HIP == literal fixture, HIP == oracle bit for bit on observed and latent sweeps."""
import ctypes as C
import json
import os
import sys

import numpy as np
import pytest

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def _check_fixture(S, score_node, rtol=1e-12):
    import literal_check
    lw, tr = S["lw"], S["trace"]
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "literal_scores_clinic.json")))
    n = 0
    for r in fx["rows"]:
        i = r["row"]
        for bi, fb in enumerate(r["blocks"]):
            blk = lw.blocks[bi]
            cname = blk["root_class"]
            t = tr.tables[cname]
            ctxv = np.zeros((1, 4), dtype=np.int32)
            for c, (sb, col) in enumerate(zip(blk["ctx_src_block"], blk["ctx_src_col"])):
                ctxv[0, c] = tr.tables[lw.blocks[sb]["root_class"]].cols[col, tr.cur[sb, i]]
            lse, scores = score_node(bi, np.array([i], np.int32), ctxv, np.array([tr.cur[bi, i]], np.int32), t.n)
            scores = np.asarray(scores).reshape(-1)
            seen = 0
            for k in range(t.n):
                key = literal_check._content_key(lw, tr, cname, k)
                if key in fb["cands"]:
                    want = fb["cands"][key]
                    assert abs(scores[k] - want) <= rtol * max(1.0, abs(want)), (i, bi, key, scores[k], want)
                    seen += 1
                else:
                    assert scores[k] == -np.inf
            assert seen == len(fb["cands"])
            assert abs(scores[t.n] - fb["new"]) <= rtol * max(1.0, abs(fb["new"])), (i, bi, scores[t.n], fb["new"])
            assert abs(lse[0] - fb["lse"]) <= 1e-9 * max(1.0, abs(fb["lse"]))
            n += seen + 1
    return n


def test_oracle_reproduces_the_literal_clinic_scores(oracle):
    import make_clinic_fixture as mk
    S = mk.clinic_setup()
    lw, tr, obs = S["lw"], S["trace"], S["obs"]
    assert lw.block_group == [0, 0, 0] and lw.blocks[2]["ctx_src_block"] == [0, 1]
    w = helpers.mirror_world(oracle, lw, obs, tr, None, 1, helpers.option_logp_cpu(oracle, lw, tr))

    def score_node(block, rows, ctxv, excl, n_rows):
        lse, scores = w.eval_tree(block, 0, rows[0], ctxv[0], excl[0], n_rows + 1)
        return np.array([lse]), scores

    assert _check_fixture(S, score_node) >= 300


def test_clinic_inference_runs_through_the_host_code(oracle):
    """initialize_trace + run_inference (every class: latent sweeps with two context sources) with the oracle engine;
    gloo world-size-1 semantics; the trace stays consistent and most cells are repaired."""
    import clinic_program as cp
    from oracle_engine import OracleEngine
    from pclean_amd.analysis import evaluate_accuracy
    from pclean_amd.engine import InferenceConfig
    from pclean_amd.inference import initialize_trace, run_inference
    from pclean_amd.trace import Trace
    S = cp.clinic_program()
    lw, obs = S["lw"], S["obs"]
    eng = OracleEngine(oracle, lw, obs)
    tr = Trace(lw, obs.shape[1], 2)
    cfg = InferenceConfig(2, 6)
    initialize_trace(eng, tr, cfg, 8, max_batch=16)
    run_inference(eng, tr, cfg, 8)
    tr.check_consistency()
    acc = evaluate_accuracy(lw, tr, S["dirty"], S["clean"])
    assert acc["f1"] > 0.7, acc


@pytest.mark.gpu
def test_hip_clinic_program(oracle):
    import clinic_program as cp
    import make_clinic_fixture as mk
    from pclean_amd._lib import InferConfig
    from pclean_amd.engine import Engine, InferenceConfig
    from pclean_amd.inference import build_evidence, initialize_trace, latent_current_choices, run_inference
    from pclean_amd.trace import Trace
    S = mk.clinic_setup()
    lw, tr, obs = S["lw"], S["trace"], S["obs"]
    n = obs.shape[1]
    eng = Engine(lw, obs, dist_mode=1)
    try:
        eng.upload_trace(tr)

        def score_node(block, rows, ctxv, excl, n_rows):
            lse, scores, _ = eng.hip.score_node(block, 0, rows, ctxv=ctxv, excl=excl, n_cand=n_rows + 1, want_scores=True)
            return lse, scores

        assert _check_fixture(S, score_node) >= 300
        # observed sweeps: HIP == oracle (PG 6 particles: no resampling between the three slots of the block; MH)
        world = helpers.mirror_world(oracle, lw, obs, tr, eng)
        for P, mh in ((6, 0), (2, 1)):
            choice, chosen, logml, new_rows = eng.sweep(tr, InferenceConfig(1, P, use_mh_instead_of_pg=bool(mh)), 21, 2)
            och = np.empty((3, n), dtype=np.int32)
            ocp = np.empty(n, dtype=np.int32)
            oml = np.empty(n)
            c = InferConfig(1, P, 1, 1, mh, 50, 100)
            oracle.lib().pco_sweep_batched(world.h, C.byref(c), C.c_uint64(21), C.c_uint32(2), 3, C.c_int64(0),
                                           oracle._p(np.ascontiguousarray(tr.cur), C.c_int32), oracle._p(och, C.c_int32),
                                           oracle._p(ocp, C.c_int32), oracle._p(oml, C.c_double))
            assert np.array_equal(choice, och) and np.array_equal(chosen, ocp) and np.array_equal(logml, oml), (P, mh)
        # latent sweeps (two per-evidence-row context sources): HIP == oracle
        cfg = InferenceConfig(1, 4)
        for cname in lw.model.class_order:
            if cname not in lw.latent_plans:
                continue
            pl = lw.latent_plans[cname]
            live, ev_off, ev_rows, ev_ctx = build_evidence(lw, tr, cname)
            assert ev_ctx is not None and len(pl["ctx_sources"]) == 2
            excl = latent_current_choices(lw, tr, cname, live, cfg)
            eng.hip.set_active_rows(0, -1)
            got = eng.hip.sweep_latent(cfg.as_c(), 9, 1, pl["block_id"], pl["roots"], live, ev_off, ev_rows, ev_ctx, excl,
                                       len(pl["nodes"]))
            want = world.sweep_latent(InferConfig(1, 4, 1, 1, 0, 50, 100), 9, 1, pl["block_id"], pl["roots"], live, ev_off,
                                      ev_rows, ev_ctx, excl, len(pl["nodes"]))
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), cname
        # end to end on the GPU
        tr2 = Trace(lw, n, 2)
        cfg2 = InferenceConfig(2, 6)
        initialize_trace(eng, tr2, cfg2, 8, max_batch=16)
        run_inference(eng, tr2, cfg2, 8)
        tr2.check_consistency()
    finally:
        eng.close()

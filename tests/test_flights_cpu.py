"""Host logic of the flights program (BASELINE.json configs[3]) driven by the oracle on CPU:
lowering of the scoring block / MaybeSwap evidence terms / keyed TimePrior atoms, commits of a
two-class latent schema, Beta-Bernoulli error-probability updates, evaluate_accuracy."""
import ctypes as C

import numpy as np

import helpers
from pclean_amd import experiments as ex
from pclean_amd._lib import DENS_MAYBE_SWAP, InferConfig
from pclean_amd.analysis import evaluate_accuracy
from pclean_amd.inference import build_evidence, commit_latent
from pclean_amd.model import LoweredModel
from pclean_amd.trace import Trace


def flights_setup():
    dirty, clean = ex.flights_data()
    m = ex.flights_model(dirty)
    lw = LoweredModel(m, ex.flights_query(m), dirty)
    return dirty, clean, lw, lw.encode_observations(dirty)


def oracle_sweep(oracle, lw, w, cfg, seed, sweep, cur):
    nb, n = cur.shape
    choice = np.empty((nb, n), dtype=np.int32)
    chosen = np.empty(n, dtype=np.int32)
    logml = np.empty(n)
    oracle.lib().pco_sweep_batched(w.h, C.byref(cfg), C.c_uint64(seed), C.c_uint32(sweep), nb, C.c_int64(0),
                                   oracle._p(np.ascontiguousarray(cur), C.c_int32), oracle._p(choice, C.c_int32),
                                   oracle._p(chosen, C.c_int32), oracle._p(logml, C.c_double))
    new_rows = {}
    for b, blk in enumerate(lw.blocks):
        if blk.get("score"):
            continue
        k = oracle.lib().pco_new_rows_count(b)
        if k:
            rows = np.empty(k, dtype=np.int32)
            vals = np.empty((k, len(blk["nodes"])), dtype=np.int32)
            oracle.lib().pco_new_rows_get(b, len(blk["nodes"]), oracle._p(rows, C.c_int32), oracle._p(vals, C.c_int32))
            new_rows[b] = (rows, vals)
    return choice, chosen, logml, new_rows


def test_flights_lowering():
    dirty, clean, lw, obs = flights_setup()
    assert obs.shape == (6, 2376)
    assert [bool(b.get("score")) for b in lw.blocks] == [False, False, True]  # run.jl:24-34: three blocks
    assert [b["root_class"] for b in lw.blocks[:2]] == ["Flight", "TrackingWebsite"]
    sb = lw.score_blocks[2]
    assert len(sb["terms"]) == 4 and sb["prob"]["consts"] == [1e-5] and len(sb["prob"]["keys"]) in (38, 39)
    pf = lw.fn_tables[sb["prob"]["fn"]]
    # error_prob (run.jl:28): 1e-5 when the website is the airline's own (first two letters of the flight id)
    sdom, fdom = lw.latent_dom[("TrackingWebsite", "name")], lw.latent_dom[("Flight", "flight_id")]
    for x in range(pf.shape[0]):
        for y in range(0, pf.shape[1], 7):
            own = sdom.string(x).lower() == fdom.string(y)[:2].lower()
            assert (pf[x, y] == 0) == own
    # Flight's latent plan: every time leaf carries its key constraint and the MaybeSwap evidence term
    pl = lw.latent_plans["Flight"]
    kinds = [t[3] for t in pl["terms"]]
    assert kinds.count(DENS_MAYBE_SWAP) == 4 and lw.latent_ev_prob == {"Flight": 2}


def test_flights_oracle_inference(oracle):
    dirty, clean, lw, obs = flights_setup()
    tr = Trace(lw, obs.shape[1], 0)
    cfg = InferConfig(1, 4, 1, 1, 0, 50, 100)

    def world():
        return helpers.mirror_world(oracle, lw, obs, tr, None, option_logp=helpers.option_logp_cpu(oracle, lw, tr))

    choice, chosen, logml, new_rows = oracle_sweep(oracle, lw, world(), cfg, 3, 0x7fffffff, tr.cur)
    assert np.isfinite(logml).all()
    tr.commit_batch(0, obs.shape[1], choice, new_rows, dedup=True)
    tr.check_consistency()
    assert tr.tables["TrackingWebsite"].n_live == 38
    f1 = []
    for it in range(3):
        tr.resample_parameters()
        assert np.all((tr.prob_param.value > 0) & (tr.prob_param.value < 1))
        for cname in ["TrackingWebsite", "Flight"]:
            pl = lw.latent_plans[cname]
            live, ev_off, ev_rows, ev_ctx = build_evidence(lw, tr, cname)
            if cname == "Flight":
                assert ev_ctx.shape == (len(ev_rows), 2) and ev_ctx[:, 0].max() < len(tr.prob_table())
            excl = np.full((len(pl["roots"]), len(live)), -1, dtype=np.int32)
            got = world().sweep_latent(cfg, 5, it, pl["block_id"], pl["roots"], live, ev_off, ev_rows, ev_ctx, excl,
                                       len(pl["nodes"]))
            commit_latent(lw, tr, cname, live, got[0], got[1])
            tr.check_consistency()
        choice, chosen, logml, new_rows = oracle_sweep(oracle, lw, world(), cfg, 3, it, tr.cur)
        assert np.isfinite(logml).all()
        tr.commit(choice, new_rows)
        tr.check_consistency()
        f1.append(evaluate_accuracy(lw, tr, dirty, clean)["f1"])
    acc = evaluate_accuracy(lw, tr, dirty, clean)
    assert acc["errors"] == 2608 and acc["imputed"] == 2312
    assert f1[-1] > 0.8 and f1[-1] > f1[0]
    # reliable websites end with smaller error probabilities than the prior mean 10/60 suggests for noisy ones
    assert tr.prob_param.value.min() < 0.15 < tr.prob_param.value.max()


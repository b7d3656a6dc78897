"""oracle/pruned.h (the CPU baseline's second leg: grouping + exact pruning on one thread) gives the results of the plain
batched sweep bit for bit."""
import ctypes as C

import numpy as np
import pytest

import helpers
from pclean_amd._lib import InferConfig


def _new_rows(oracle, lw):
    out = {}
    for b, blk in enumerate(lw.blocks):
        if blk.get("score"):
            continue
        k = oracle.lib().pco_new_rows_count(b)
        nn = len(blk["nodes"])
        rows, vals = np.empty(k, dtype=np.int32), np.empty((k, nn), dtype=np.int32)
        if k:
            oracle.lib().pco_new_rows_get(b, nn, oracle._p(rows, C.c_int32), oracle._p(vals, C.c_int32))
        out[b] = (rows, vals)
    return out


@pytest.mark.parametrize("program", ["hospital", "synthetic", "rents", "flights"])
def test_pruned_sweep_equals_the_plain_batched_sweep(oracle, program):
    """Choices, chosen particles, log marginal likelihoods and new-row records of pco_sweep_batched_pruned ==
    pco_sweep_batched for PG (20 particles) and MH, on states where rows move and new rows are proposed; the statistics say
    that the savers really engaged (memo hits, pruned candidates, skipped new-row branches) where the plan allows them —
    rents (Gaussian term) and flights (equality-only slots, scoring block) take the grouped full enumeration."""
    if program == "hospital":
        S = helpers.hospital_setup(n_rows=400)
    elif program == "synthetic":
        from pclean_amd import synth
        from pclean_amd import experiments as ex
        from pclean_amd.engine import InferenceConfig
        from pclean_amd.inference import initialize_trace
        from pclean_amd.model import LoweredModel
        from pclean_amd.trace import Trace
        from oracle_engine import OracleEngine
        dirty, clean = synth.synth_hospital(1500, 40, seed=3)[:2]
        m = ex.hospital_model(ex.possibilities_of(dirty))
        lw = LoweredModel(m, ex.hospital_query(m), dirty)
        obs = lw.encode_observations(dirty)
        tr = Trace(lw, obs.shape[1], 3)
        initialize_trace(OracleEngine(oracle, lw, obs), tr, InferenceConfig(1, 4), 3, max_batch=256)
        S = dict(lw=lw, obs=obs, trace=tr)
    elif program == "rents":
        S = helpers.rents_setup(n_rows=300)
    else:
        S = helpers.flights_setup()
    lw, tr, obs = S["lw"], S["trace"], S["obs"]
    w = helpers.mirror_world(oracle, lw, obs, tr, None, 1, helpers.option_logp_cpu(oracle, lw, tr))
    n = min(obs.shape[1], 600)
    engaged = False
    for P, mh in ((20, 0), (2, 1)):
        c = InferConfig(1, P, 1, 1, mh, 50, 100)
        plain = w.sweep_batched(c, 11, 1, tr.cur)
        plain_new = _new_rows(oracle, lw)
        got = w.sweep_batched(c, 11, 1, tr.cur, n_rows=n, pruned=True)
        got_new = _new_rows(oracle, lw)
        assert np.array_equal(got[0], plain[0][:, :n]) and np.array_equal(got[1], plain[1][:n]), (program, P, mh)
        assert np.array_equal(got[2], plain[2][:n]), (program, P, mh, np.abs(got[2] - plain[2][:n]).max())
        for b in plain_new:
            keep = plain_new[b][0] < n
            assert np.array_equal(got_new[b][0], plain_new[b][0][keep]) and np.array_equal(got_new[b][1], plain_new[b][1][keep]), (program, b)
        st = got[3]
        assert st["root_evaluations"] >= n
        if program in ("hospital", "synthetic"):
            assert st["candidates_pruned"] > 10 * st["candidates_scored_exactly"] and st["new_row_branches_skipped"] > 0, st
            assert st["served_by_the_memo"] > 0 and st["child_memo_hits"] > st["child_memo_misses"], st
            engaged = True
    assert engaged or program in ("rents", "flights")

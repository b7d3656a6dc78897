"""Weight of a particle that chose a ProposalDummyValue with an observation below the node (block_proposal.jl:58-60):
block marginal - log(dummy mass) + logdensity(obs | drawn string) - logdensity(obs | placeholder).
tests/golden/literal_dummy_weight.json holds the literal interpreter's numbers (strings, its own densities) for the
`people` program (tests/dummy_program.py; generator scripts/make_dummy_weight_fixture.py); the C++ oracle (CPU) and
the HIP path (GPU) must reproduce them, and the committed latent rows must hold exactly the strings that were weighed."""
import json
import os

import numpy as np
import pytest

import dummy_program as dp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fixture():
    return json.load(open(os.path.join(ROOT, "tests", "golden", "literal_dummy_weight.json")))


def _check_against_fixture(fx, logml, new_rows):
    rows_new, vals_new = new_rows[0]
    assert np.array_equal(rows_new, np.arange(len(fx["rows"])))  # an empty table: every row creates its referent
    for r in fx["rows"]:
        i = r["row"]
        assert int(vals_new[i, 1]) == r["drawn_option"]
        assert int(vals_new[i, 0]) == -1  # one particle: -1 - 0
        assert abs(float(logml[i]) - r["logml"]) <= 1e-9 * max(1.0, abs(r["logml"])), (i, r["obs"], float(logml[i]), r["logml"])
    return sum(1 for r in fx["rows"] if r["drew_dummy"] and r["obs"] is not None)


def _commit_and_strings(engine, lw, tr, seed, sweep, choice, new_rows):
    from pclean_amd.inference import resample_dummies
    tr.commit_batch(0, tr.cur.shape[1], choice, new_rows, dedup=False, sweep_idx=sweep)
    resample_dummies(engine, tr, seed, 1)
    dom = lw.latent_dom[("Person", "name")]
    t = tr.tables["Person"]
    return [dom.string(int(t.cols[lw.colidx["Person"]["name"], int(tr.cur[0, i])])) for i in range(tr.cur.shape[1])]


def test_oracle_reproduces_the_literal_dummy_weights(oracle):
    from oracle_engine import OracleEngine
    from pclean_amd.engine import InferenceConfig
    from pclean_amd.trace import Trace
    fx = _fixture()
    m, q, dirty, lw, obs = dp.people_program()
    eng = OracleEngine(oracle, lw, obs)
    tr = Trace(lw, obs.shape[1], 0)
    choice, chosen, logml, new_rows = eng.sweep(tr, InferenceConfig(1, 1), fx["seed"], fx["sweep"])
    assert _check_against_fixture(fx, logml, new_rows) >= 8
    # the committed rows hold the strings whose likelihood entered the weights
    strings = _commit_and_strings(eng, lw, tr, fx["seed"], fx["sweep"], choice, new_rows)
    for r in fx["rows"]:
        want = r["drawn_string"] if r["drew_dummy"] else fx["atoms"][r["drawn_option"]]
        assert strings[r["row"]] == want, (r["row"], strings[r["row"]], want)


@pytest.mark.gpu
def test_hip_reproduces_the_dummy_weights(oracle):
    """HIP == oracle bit for bit (1 and 6 particles, unrestricted and restricted distances) and == the literal fixture"""
    import ctypes as C

    import helpers
    from pclean_amd import _lib
    from pclean_amd._lib import InferConfig
    from pclean_amd.engine import Engine, InferenceConfig
    from pclean_amd.trace import Trace
    fx = _fixture()
    m, q, dirty, lw, obs = dp.people_program()
    n = obs.shape[1]
    for dist_mode in (_lib.DIST_DL, _lib.DIST_OSA):
        eng = Engine(lw, obs, dist_mode=dist_mode)
        try:
            tr = Trace(lw, n, 0)
            eng.upload_trace(tr)
            for P in (1, 6):
                choice, chosen, logml, new_rows = eng.sweep(tr, InferenceConfig(1, P), fx["seed"], fx["sweep"])
                world = helpers.mirror_world(oracle, lw, obs, tr, eng)
                och = np.empty((1, n), dtype=np.int32)
                ocp = np.empty(n, dtype=np.int32)
                oml = np.empty(n)
                c = InferConfig(1, P, 1, 1, 0, 50, 100)
                oracle.lib().pco_sweep_batched(world.h, C.byref(c), C.c_uint64(fx["seed"]), C.c_uint32(fx["sweep"]), 1,
                                               C.c_int64(0), oracle._p(np.ascontiguousarray(tr.cur), C.c_int32),
                                               oracle._p(och, C.c_int32), oracle._p(ocp, C.c_int32), oracle._p(oml, C.c_double))
                assert np.array_equal(choice, och) and np.array_equal(chosen, ocp) and np.array_equal(logml, oml), (dist_mode, P)
                k = oracle.lib().pco_new_rows_count(0)
                orows, ovals = np.empty(k, dtype=np.int32), np.empty((k, 2), dtype=np.int32)
                oracle.lib().pco_new_rows_get(0, 2, oracle._p(orows, C.c_int32), oracle._p(ovals, C.c_int32))
                assert np.array_equal(new_rows[0][0], orows) and np.array_equal(new_rows[0][1], ovals)
                if P == 1 and dist_mode == _lib.DIST_DL:
                    assert _check_against_fixture(fx, logml, new_rows) >= 8
                    saved = (choice.copy(), {0: (new_rows[0][0].copy(), new_rows[0][1].copy())})
        finally:
            eng.close()
    # the committed rows hold the strings whose likelihood entered the weights (the commit grows the latent domain and
    # reloads the engine: last)
    eng = Engine(lw, obs, dist_mode=_lib.DIST_DL)
    try:
        strings = _commit_and_strings(eng, lw, Trace(lw, n, 0), fx["seed"], fx["sweep"], *saved)
        for r in fx["rows"]:
            want = r["drawn_string"] if r["drew_dummy"] else fx["atoms"][r["drawn_option"]]
            assert strings[r["row"]] == want
    finally:
        eng.close()

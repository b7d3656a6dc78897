"""GPU parity on the rents program (BASELINE.json configs[2]: numeric Gaussian + categorical
classes): TransformedGaussian / MeanParameter terms with enumerated own choices (br, unit),
equality constraints of directly observed values, keyed StringPrior atoms, missing observations."""
import ctypes as C

import numpy as np
import pytest

import helpers
from pclean_amd import experiments as ex
from pclean_amd._lib import InferConfig
from pclean_amd.analysis import evaluate_accuracy
from pclean_amd.engine import Engine, InferenceConfig
from pclean_amd.inference import build_evidence, commit_latent, initialize_trace, observed_sweep, run_inference
from pclean_amd.model import LoweredModel
from pclean_amd.trace import Trace

pytestmark = pytest.mark.gpu


def rents_setup(n_rows, units=None):
    dirty, clean = ex.rents_data()
    dirty = {c: v[:n_rows] for c, v in dirty.items()}
    clean = {c: v[:n_rows] for c, v in clean.items()}
    m = ex.rents_model(dirty, units)
    q = ex.rents_query(m)
    lw = LoweredModel(m, q, dirty)
    return dirty, clean, lw, lw.encode_observations(dirty)


def oracle_sweep(oracle, world, cfg, seed, sweep, cur, n_nodes):
    nb, n = cur.shape
    choice = np.empty((nb, n), dtype=np.int32)
    chosen = np.empty(n, dtype=np.int32)
    logml = np.empty(n)
    c = InferConfig(1, cfg.num_particles, int(cfg.use_dd_proposals), 1, int(cfg.use_mh_instead_of_pg), 50, 100)
    oracle.lib().pco_sweep_batched(world.h, C.byref(c), C.c_uint64(seed), C.c_uint32(sweep), nb, C.c_int64(0),
                                   oracle._p(np.ascontiguousarray(cur), C.c_int32), oracle._p(choice, C.c_int32),
                                   oracle._p(chosen, C.c_int32), oracle._p(logml, C.c_double))
    new_rows = {}
    for b in range(nb):
        k = oracle.lib().pco_new_rows_count(b)
        if k:
            rows = np.empty(k, dtype=np.int32)
            vals = np.empty((k, n_nodes[b]), dtype=np.int32)
            oracle.lib().pco_new_rows_get(b, n_nodes[b], oracle._p(rows, C.c_int32), oracle._p(vals, C.c_int32))
            new_rows[b] = (rows, vals)
    return choice, chosen, logml, new_rows, world.get_locals(0, n)


@pytest.mark.parametrize("particles,mh,dd,nonlinear", [(2, True, True, False), (20, False, True, False), (2, True, False, False),
                                                       (6, False, False, False), (6, False, True, True), (4, False, False, True)])
def test_rents_sweep_and_latent_parity(oracle, particles, mh, dd, nonlinear):
    """dd = False: prior proposals (use_dd_proposals = false) — County rows' attributes from their priors under the Gaussian
    evidence of the referring rows (their own choices given), the observed class with its own choices sampled per particle
    (gauss_prior_kernel), the retained particle keeping the row's current ones.
    nonlinear: `unit` chooses among dollars, square-root dollars and log-dollars — Transformations that are not linear
    (transformed_gaussian.jl:5-9): backward(x) and log|deriv| come from the lowering's derived numeric columns
    (tests/test_nonlinear_transformation.py holds the C++ oracle's reading of them against the literal interpreter)."""
    from pclean_amd.inference import latent_current_choices
    units = None
    if nonlinear:
        from test_nonlinear_transformation import nonlinear_units
        units = nonlinear_units()
    dirty, clean, lw, obs = rents_setup(3000, units)
    assert lw.xnum.shape == (5 if nonlinear else 1, 3000)
    assert (obs[2] < 0).sum() > 100 and (obs[3] < 0).sum() > 100  # missing State / Room Type
    eng = Engine(lw, obs, dist_mode=1)
    try:
        cfg0 = InferenceConfig(1, particles, use_mh_instead_of_pg=mh, rejuv_frequency=500)
        cfg = InferenceConfig(1, particles, use_mh_instead_of_pg=mh, rejuv_frequency=500, use_dd_proposals=dd)
        tr = Trace(lw, obs.shape[1], 2)
        initialize_trace(eng, tr, cfg0, 2, max_batch=512)
        tr.check_consistency()
        assert (tr.locals[0] >= 0).all()
        n_nodes = [len(b["nodes"]) for b in lw.blocks]
        for sweep in range(2):
            # latent class County: external likelihood incl. the Gaussian observations of referring rows
            pl = lw.latent_plans["County"]
            live, ev_off, ev_rows, ev_ctx = build_evidence(lw, tr, "County")
            assert ev_ctx is not None and ev_ctx.shape == (len(ev_rows), 2)
            excl = (np.full((len(pl["roots"]), len(live)), -1, dtype=np.int32) if dd
                    else latent_current_choices(lw, tr, "County", live, cfg))
            eng.upload_trace(tr)
            eng.hip.set_active_rows(0, -1)
            world = helpers.mirror_world(oracle, lw, obs, tr, eng)
            got = eng.hip.sweep_latent(cfg.as_c(), 5, sweep, pl["block_id"], pl["roots"], live, ev_off, ev_rows, ev_ctx,
                                       excl, len(pl["nodes"]))
            c = InferConfig(1, particles, int(dd), 1, int(mh), 50, 100)
            want = world.sweep_latent(c, 5, sweep, pl["block_id"], pl["roots"], live, ev_off, ev_rows, ev_ctx, excl,
                                      len(pl["nodes"]))
            assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
            commit_latent(lw, tr, "County", live, got[0], got[1])
            tr.check_consistency()
            # observed class
            eng.upload_trace(tr)
            world = helpers.mirror_world(oracle, lw, obs, tr, eng)
            world.set_cur_locals(0, tr.locals[0])
            choice, chosen, logml, new_rows = eng.sweep(tr, cfg, 5, sweep)
            locals_gpu = tr.pending_locals[0].copy()
            o = oracle_sweep(oracle, world, cfg, 5, sweep, tr.cur, n_nodes)
            assert np.array_equal(choice, o[0]) and np.array_equal(chosen, o[1])
            assert np.array_equal(logml, o[2])
            np.testing.assert_allclose(logml, o[2], rtol=1e-12)
            assert set(new_rows) == set(o[3])
            for b in new_rows:
                assert np.array_equal(new_rows[b][0], o[3][b][0]) and np.array_equal(new_rows[b][1], o[3][b][1])
            assert np.array_equal(locals_gpu, o[4]), "own choices (br, unit) differ"
            tr.commit_locals()
            tr.commit(choice, new_rows)
            tr.resample_parameters()
            tr.check_consistency()
    finally:
        eng.close()


def test_rents_end_to_end():
    dirty, clean, lw, obs = rents_setup(8000)
    eng = Engine(lw, obs, dist_mode=1)
    try:
        cfg = InferenceConfig(2, 2, use_mh_instead_of_pg=True, rejuv_frequency=500)
        tr = Trace(lw, obs.shape[1], 0)
        initialize_trace(eng, tr, cfg, 0, max_batch=1024)
        run_inference(eng, tr, cfg, 0)
        tr.check_consistency()
        acc = evaluate_accuracy(lw, tr, dirty, clean)
        assert acc["imputed"] > 1000 and acc["correctly_imputed"] > 0.4 * acc["imputed"]
        assert acc["f1"] > 0.60  # 8000 of the 50 000 rows; the full table (F1 0.689 vs sequential 0.687) is tests/test_gpu_f1_vs_sequential.py
        # the /1000 unit errors get repaired: corrected rents equal the clean rent for most damaged cells
        dn = np.array([float(v) for v in dirty["Monthly Rent"]])
        cn = np.array([float(v) for v in clean["Monthly Rent"]])
        bad = np.nonzero(dn != cn)[0]
        ours = np.round(dn * np.array([1.0, 1000.0])[tr.locals[0][:, 1]])
        assert len(bad) > 20 and (ours[bad] == cn[bad]).mean() > 0.9
    finally:
        eng.close()

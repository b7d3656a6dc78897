"""GPU parity on the flights program (BASELINE.json configs[3]): a pure scoring block (MaybeSwap with
per-website Beta error probabilities -> unequal particle weights at the final choice), keyed TimePrior
atoms, and MaybeSwap external-likelihood terms in the Flight class sweep."""
import ctypes as C

import numpy as np
import pytest

import helpers
from pclean_amd._lib import InferConfig
from pclean_amd.analysis import evaluate_accuracy
from pclean_amd.engine import Engine, InferenceConfig
from pclean_amd.inference import build_evidence, commit_latent, initialize_trace, run_inference
from pclean_amd.trace import Trace
from test_flights_cpu import flights_setup, oracle_sweep

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("particles,mh,dd", [(2, True, True), (10, False, True), (2, True, False), (6, False, False)])
def test_flights_sweep_and_latent_parity(oracle, particles, mh, dd):
    """dd = False: prior proposals (use_dd_proposals = false, block_proposal.jl:42-84,168) on a plan with equality-constrained
    slots, a MaybeSwap scoring block and MaybeSwap evidence terms in the latent sweeps — referents from the CRP prior, new
    rows' choices from their (keyed) prior proposals, weights = likelihood of the sampled values; particle 0 retained."""
    from pclean_amd.inference import latent_current_choices
    dirty, clean, lw, obs = flights_setup()
    eng = Engine(lw, obs, dist_mode=1)
    try:
        cfg0 = InferenceConfig(1, particles, use_mh_instead_of_pg=mh, rejuv_frequency=500)
        cfg = InferenceConfig(1, particles, use_mh_instead_of_pg=mh, rejuv_frequency=500, use_dd_proposals=dd)
        c = InferConfig(1, cfg.num_particles, int(dd), 1, int(mh), 50, 100)
        tr = Trace(lw, obs.shape[1], 2)
        initialize_trace(eng, tr, cfg0, 2, max_batch=512)
        tr.check_consistency()
        for sweep in range(2):
            tr.resample_parameters()
            for cname in ["TrackingWebsite", "Flight"]:
                pl = lw.latent_plans[cname]
                live, ev_off, ev_rows, ev_ctx = build_evidence(lw, tr, cname)
                excl = (np.full((len(pl["roots"]), len(live)), -1, dtype=np.int32) if dd
                        else latent_current_choices(lw, tr, cname, live, cfg))
                eng.upload_trace(tr)
                eng.hip.set_active_rows(0, -1)
                world = helpers.mirror_world(oracle, lw, obs, tr, eng)
                got = eng.hip.sweep_latent(cfg.as_c(), 5, sweep, pl["block_id"], pl["roots"], live, ev_off, ev_rows,
                                           ev_ctx, excl, len(pl["nodes"]))
                want = world.sweep_latent(c, 5, sweep, pl["block_id"], pl["roots"], live, ev_off, ev_rows, ev_ctx, excl,
                                          len(pl["nodes"]))
                assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), cname
                commit_latent(lw, tr, cname, live, got[0], got[1])
                tr.check_consistency()
            eng.upload_trace(tr)
            world = helpers.mirror_world(oracle, lw, obs, tr, eng)
            choice, chosen, logml, new_rows = eng.sweep(tr, cfg, 5, sweep)
            o = oracle_sweep(oracle, lw, world, c, 5, sweep, tr.cur)
            assert np.array_equal(choice, o[0]) and np.array_equal(chosen, o[1])
            assert np.array_equal(logml, o[2])
            assert set(new_rows) == set(o[3])
            for b in new_rows:
                assert np.array_equal(new_rows[b][0], o[3][b][0]) and np.array_equal(new_rows[b][1], o[3][b][1])
            tr.commit(choice, new_rows)
            tr.check_consistency()
    finally:
        eng.close()


def test_flights_end_to_end():
    dirty, clean, lw, obs = flights_setup()
    eng = Engine(lw, obs, dist_mode=1)
    try:
        cfg = InferenceConfig(4, 2, use_mh_instead_of_pg=True, rejuv_frequency=500)
        tr = Trace(lw, obs.shape[1], 0)
        initialize_trace(eng, tr, cfg, 0, max_batch=512)
        run_inference(eng, tr, cfg, 0)
        tr.check_consistency()
        acc = evaluate_accuracy(lw, tr, dirty, clean)
        assert acc["errors"] == 2608 and acc["imputed"] == 2312
        assert acc["f1"] > 0.875  # 0.89; sequential reference 0.890 (tests/golden/sequential_f1.json)
    finally:
        eng.close()

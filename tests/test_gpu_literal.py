"""The HIP path (pclean_score_node through the C ABI) against the LITERAL interpreter's fixtures
(tests/golden/literal_scores.json): per-candidate scores of hospital rows computed on the GPU from the product's
lowering, pair tables and option tables must equal what the model description + strings give, to 1e-12."""
import numpy as np
import pytest

import helpers
import literal_check
from pclean_amd.engine import Engine

pytestmark = pytest.mark.gpu


def test_hip_path_reproduces_literal_scores():
    S = helpers.hospital_setup()
    eng = Engine(S["lw"], S["obs"], dist_mode=1)
    try:
        eng.upload_trace(S["trace"])

        def score_node(block, rows, ctxv, excl, n_rows):
            lse, scores, _ = eng.hip.score_node(block, 0, rows, ctxv=ctxv, excl=excl, n_cand=n_rows + 1, want_scores=True)
            return lse, scores

        assert literal_check.check(S, score_node) > 1000
    finally:
        eng.close()


def test_hip_path_reproduces_literal_scores_rents():
    """rents fixtures (tests/golden/literal_scores_rents.json): equality constraints of noise-free observations, keyed
    StringPrior, ChooseProportionally, own choices enumerated inside the candidate branch, TransformedGaussian."""
    S = helpers.rents_setup()
    eng = Engine(S["lw"], S["obs"], dist_mode=1)
    try:
        eng.upload_trace(S["trace"])

        def score_node(block, rows, ctxv, excl, n_rows):
            lse, scores, _ = eng.hip.score_node(block, 0, rows, ctxv=ctxv, excl=excl, n_cand=n_rows + 1, want_scores=True)
            return lse, scores

        assert literal_check.check_rents(S, score_node) >= 48
    finally:
        eng.close()


def test_hip_path_reproduces_literal_scores_flights():
    """flights fixtures (tests/golden/literal_scores_flights.json): prior-only slots with equality constraints, keyed
    TimePrior / StringPrior proposals of the new row, and the MaybeSwap scoring block — through pclean_score_node and
    the logml of a one-particle sweep."""
    from pclean_amd.engine import InferenceConfig
    S = helpers.flights_setup()
    eng = Engine(S["lw"], S["obs"], dist_mode=1)
    try:
        eng.upload_trace(S["trace"])

        def score_node(block, rows, ctxv, excl, n_rows):
            lse, scores, _ = eng.hip.score_node(block, 0, rows, ctxv=ctxv, excl=excl, n_cand=n_rows + 1, want_scores=True)
            return lse, scores

        choice, chosen, logml, new_rows = eng.sweep(S["trace"], InferenceConfig(1, 1), 3, 0)
        assert (choice == S["trace"].cur).all() and not new_rows
        assert literal_check.check_flights(S, score_node, logml) > 300
    finally:
        eng.close()


def _hip_eval_ev(eng):
    """eval_ev of literal_check.check_latent*: one latent row against its evidence set through pclean_score_node_ev."""
    def eval_ev(block_id, node_id, ev_rows, ev_ctx, excl, n_scores):
        ev_rows = np.ascontiguousarray(ev_rows, dtype=np.int32)
        lse, scores, _ = eng.hip.score_node_ev(block_id, node_id, [0], [0, len(ev_rows)], ev_rows, ev_ctx=ev_ctx,
                                               excl=None if excl is None or excl < 0 else [excl], n_cand=n_scores,
                                               want_scores=True)
        return float(lse[0]), scores[0]
    return eval_ev


def test_hip_path_reproduces_literal_scores_of_latent_rows():
    """The latent-class path (proposal_compiler.jl:306-350) against the literal interpreter DIRECTLY: every root of the
    latent plans of 26 hospital latent rows scored against its evidence set by the HIP library (aggregated evidence,
    candidate_score_ev), incl. the cross-block JuliaNode term with per-evidence-row ctx."""
    S = helpers.hospital_setup()
    eng = Engine(S["lw"], S["obs"], dist_mode=1)
    try:
        eng.upload_trace(S["trace"])
        assert literal_check.check_latent(S, _hip_eval_ev(eng)) > 3500
    finally:
        eng.close()


def test_hip_path_reproduces_literal_scores_of_latent_flights():
    """Latent Flight rows: keyed TimePrior options under the MaybeSwap observations of every referring row with that
    row's own error probability (per-evidence-row ctx)."""
    S = helpers.flights_setup()
    eng = Engine(S["lw"], S["obs"], dist_mode=1)
    try:
        eng.upload_trace(S["trace"])
        assert literal_check.check_latent_flights(S, _hip_eval_ev(eng)) > 150
    finally:
        eng.close()


def test_hip_path_reproduces_literal_scores_of_latent_counties():
    """Latent County rows of rents: keyed StringPrior options under AddTypos evidence, ChooseProportionally options under
    equality + TransformedGaussian evidence with every referring row's current own choices."""
    S = helpers.rents_setup()
    tr = S["trace"]
    n = len(tr.locals[0])
    tr.locals[0][:] = np.stack([np.arange(n) % 5, np.arange(n) % 2], axis=1)  # the fixture's own choices
    eng = Engine(S["lw"], S["obs"], dist_mode=1)
    try:
        eng.upload_trace(tr)
        assert literal_check.check_latent_rents(S, _hip_eval_ev(eng)) > 700
    finally:
        eng.close()

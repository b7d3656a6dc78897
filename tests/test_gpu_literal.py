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

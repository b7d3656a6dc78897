"""A TransformedGaussian whose Transformation is NOT linear (transformed_gaussian.jl:5-9 takes any forward / backward /
deriv; 15-16 evaluates backward(x) and |deriv(backward(x))| per observation).  The lowering turns such a unit into two
derived numeric columns — backward(x) and log|deriv(backward(x))| of every row, evaluated once on the host — that the
kernels and the C++ oracle read per row (pclean_gauss::t_x_col / t_lad_col).  Held here against the LITERAL interpreter,
which calls the Transformation's own functions on the observed number (oracle/literal.py: GaussBlockProposal._gauss)."""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")]

import helpers
from pclean_amd.model import Transformation


def nonlinear_units():
    """dollars (linear, the rents program's own); square-root dollars and log-dollars (neither linear) — three units x five
    room types = 15 of the 16 combinations a candidate branch enumerates"""
    return [Transformation(lambda x: x, lambda x: x, lambda x: 1.0),
            Transformation(lambda v: math.sqrt(v), lambda x: x * x, lambda v: 0.5 / math.sqrt(v)),
            Transformation(lambda v: math.log(v), lambda x: math.exp(min(x, 20.0)), lambda v: 1.0 / v)]


def test_lowering_of_nonlinear_units_builds_the_derived_columns():
    S = helpers.rents_setup(300, units=nonlinear_units())
    lw = S["lw"]
    spec = lw.gauss_spec
    assert spec["t_linear"] == [True, False, False]
    assert spec["t_x_col"] == [-1, 1, 3] and spec["t_lad_col"] == [-1, 2, 4]
    assert lw.xnum.shape == (5, 300)
    x = lw.xnum[0]
    ok = ~np.isnan(x)
    assert ok.sum() > 250
    assert np.array_equal(np.isnan(lw.xnum[1]), ~ok) and np.array_equal(np.isnan(lw.xnum[4]), ~ok)
    np.testing.assert_array_equal(lw.xnum[1][ok], x[ok] * x[ok])
    np.testing.assert_allclose(lw.xnum[2][ok], np.log(0.5 / np.abs(x[ok])), rtol=1e-15)  # deriv(backward(x)) = 0.5 / sqrt(x^2)
    # host-side uses of backward(x): the mean parameter's sufficient statistics and the query's output column
    rows = np.flatnonzero(ok)[:50]
    for ui in range(3):
        got = lw.gauss_backward(rows, np.full(len(rows), ui))
        want = np.array([nonlinear_units()[ui].backward(float(v)) for v in x[rows]])
        np.testing.assert_allclose(got, want, rtol=1e-15)


def test_cpp_oracle_reproduces_literal_scores_with_nonlinear_units(oracle):
    """every candidate score of 24 rents rows (all missingness patterns) under three Transformations, two of them non-linear:
    the literal interpreter calls unit.backward / unit.deriv on the observed number, the C++ oracle reads the lowering's
    derived columns"""
    import literal as lit
    import make_literal_fixtures_rents as gen
    import literal_check
    S = helpers.rents_setup(600, units=nonlinear_units())
    lw, tr, obs = S["lw"], S["trace"], S["obs"]
    rows = gen.pick_rows(S["dirty"], obs.shape[1])
    fx = [gen.row_fixture(S, i) for i in rows]
    w = helpers.mirror_world(oracle, lw, obs, tr, None, 1, helpers.option_logp_cpu(oracle, lw, tr))
    t = tr.tables["County"]
    n_checked = 0
    for r in fx:
        i = r["row"]
        lse, scores = w.eval_tree(0, 0, i, np.zeros(2, np.int32), int(tr.cur[0, i]), t.n + 1)
        scores = np.asarray(scores).reshape(-1)
        seen = 0
        for k in range(t.n):
            key = literal_check._content_key(lw, tr, "County", k)
            if key in r["cands"]:
                want = r["cands"][key]
                assert abs(scores[k] - want) <= 1e-10 * max(1.0, abs(want)), (i, key, scores[k], want)
                seen += 1
            else:
                assert scores[k] == -np.inf, (i, key, scores[k])
        assert seen == len(r["cands"])
        assert abs(scores[t.n] - r["new"]) <= 1e-10 * max(1.0, abs(r["new"])), (i, "new", scores[t.n], r["new"])
        assert abs(lse - r["lse"]) <= 1e-9 * max(1.0, abs(r["lse"])), (i, "lse", lse, r["lse"])
        n_checked += seen + 1
    assert n_checked >= 48
    # ... and the units really differ on these rows (the non-linear columns are read)
    S2 = helpers.rents_setup(600)
    w2 = helpers.mirror_world(oracle, S2["lw"], S2["obs"], S2["trace"], None, 1, helpers.option_logp_cpu(oracle, S2["lw"], S2["trace"]))
    i = fx[0]["row"]
    a = w.eval_tree(0, 0, i, np.zeros(2, np.int32), int(tr.cur[0, i]), t.n + 1)[0]
    b = w2.eval_tree(0, 0, i, np.zeros(2, np.int32), int(S2["trace"].cur[0, i]), S2["trace"].tables["County"].n + 1)[0]
    assert a != b

"""The C++ oracle (which consumes the product's lowering) against the LITERAL interpreter's fixtures (model
description + strings only, oracle/literal.py): every candidate score of 14 hospital rows x 2 blocks to 1e-12 —
a wrong plan, term-to-node assignment, ctx wiring, column map, pair table, fn table or option table in
LoweredModel would show up here (VERDICT r1 weak #1b)."""
import numpy as np

import helpers
import literal_check


def test_cpp_oracle_reproduces_literal_scores(oracle):
    S = helpers.hospital_setup()
    lw, tr, obs = S["lw"], S["trace"], S["obs"]
    w = helpers.mirror_world(oracle, lw, obs, tr, None, 1, helpers.option_logp_cpu(oracle, lw, tr))

    def score_node(block, rows, ctxv, excl, n_rows):
        lse, scores = w.eval_tree(block, 0, rows[0], ctxv[0], excl[0], n_rows + 1)
        return [lse], scores

    assert literal_check.check(S, score_node) > 1000


def test_cpp_oracle_reproduces_literal_scores_rents(oracle):
    """rents: noise-free observations of latent attributes, keyed StringPrior atoms + dummy, ChooseProportionally with
    a learned parameter, own uniform choices enumerated inside the candidate branch, TransformedGaussian with an
    IndexedLookup mean — 24 rows covering every missingness pattern."""
    S = helpers.rents_setup()
    lw, tr, obs = S["lw"], S["trace"], S["obs"]
    w = helpers.mirror_world(oracle, lw, obs, tr, None, 1, helpers.option_logp_cpu(oracle, lw, tr))

    def score_node(block, rows, ctxv, excl, n_rows):
        lse, scores = w.eval_tree(block, 0, rows[0], ctxv[0], excl[0], n_rows + 1)
        return [lse], scores

    assert literal_check.check_rents(S, score_node) >= 48


def test_cpp_oracle_reproduces_literal_scores_flights(oracle):
    """flights: slots whose only observations are noise-free (CRP prior + equality), new rows with StringPrior and
    keyed TimePrior proposals, the MaybeSwap scoring block (missing observations included) through the learned error
    probabilities — per-candidate scores and the one-particle log marginal likelihood of 62 rows."""
    import ctypes as C
    from pclean_amd._lib import InferConfig
    S = helpers.flights_setup()
    lw, tr, obs = S["lw"], S["trace"], S["obs"]
    w = helpers.mirror_world(oracle, lw, obs, tr, None, 1, helpers.option_logp_cpu(oracle, lw, tr))

    def score_node(block, rows, ctxv, excl, n_rows):
        lse, scores = w.eval_tree(block, 0, rows[0], ctxv[0], excl[0], n_rows + 1)
        return [lse], scores

    nb, n = tr.cur.shape
    choice, chosen, logml = np.empty((nb, n), np.int32), np.empty(n, np.int32), np.empty(n)
    cfg = InferConfig(1, 1, 1, 1, 0, 50, 100)  # one particle: the retained one
    oracle.lib().pco_sweep_batched(w.h, C.byref(cfg), C.c_uint64(3), C.c_uint32(0), nb, C.c_int64(0),
                                   oracle._p(np.ascontiguousarray(tr.cur), C.c_int32), oracle._p(choice, C.c_int32),
                                   oracle._p(chosen, C.c_int32), oracle._p(logml, C.c_double))
    assert np.array_equal(choice, tr.cur)
    assert literal_check.check_flights(S, score_node, logml) > 300


def test_cpp_oracle_reproduces_literal_scores_of_latent_rows(oracle):
    """Latent-class rejuvenation: 26 latent rows of the six hospital classes, every root of their latent
    plans — own choices and reference slots — scored against the evidence set, incl. the cross-block JuliaNode term
    whose other argument comes from each evidence row's OTHER referent (the per-evidence-row ctx of build_evidence)."""
    S = helpers.hospital_setup()
    lw, tr, obs = S["lw"], S["trace"], S["obs"]
    w = helpers.mirror_world(oracle, lw, obs, tr, None, 1, helpers.option_logp_cpu(oracle, lw, tr))
    assert literal_check.check_latent(S, w.eval_tree_ev) > 3500


def test_cpp_oracle_reproduces_literal_scores_of_latent_flights(oracle):
    """Latent Flight rows: keyed TimePrior options scored by the MaybeSwap observations of every referring row with
    that row's own error probability (the per-evidence-row ctx of build_evidence), missing observations included."""
    S = helpers.flights_setup()
    lw, tr, obs = S["lw"], S["trace"], S["obs"]
    w = helpers.mirror_world(oracle, lw, obs, tr, None, 1, helpers.option_logp_cpu(oracle, lw, tr))
    assert literal_check.check_latent_flights(S, w.eval_tree_ev) > 150


def test_cpp_oracle_reproduces_literal_scores_of_latent_counties(oracle):
    """Latent County rows of rents: keyed StringPrior options under AddTypos evidence, ChooseProportionally options under
    equality + TransformedGaussian evidence with every referring row's CURRENT own choices (per-evidence-row locals)."""
    S = helpers.rents_setup()
    lw, tr, obs = S["lw"], S["trace"], S["obs"]
    n = len(tr.locals[0])
    tr.locals[0][:] = np.stack([np.arange(n) % 5, np.arange(n) % 2], axis=1)  # the fixture's own choices
    w = helpers.mirror_world(oracle, lw, obs, tr, None, 1, helpers.option_logp_cpu(oracle, lw, tr))
    assert literal_check.check_latent_rents(S, w.eval_tree_ev) > 700


def test_oracle_row_by_row_evidence_matches_aggregated_and_literal(oracle):
    """The oracle's evidence-set scores in the REFERENCE's order of operations (one ExternalLikelihoodNode per referring
    row, proposal_compiler.jl:306-350: row by row, every term of a row) against (a) the literal fixtures and (b) its
    aggregated order (per term, distinct (ctx, observed value) pairs x multiplicity — the order the HIP path uses):
    the two orders differ only by fp64 rounding, < 1e-9 relative on every candidate of every fixture row."""
    for setup, check, floor in ((helpers.hospital_setup, literal_check.check_latent, 3500),
                                (helpers.flights_setup, literal_check.check_latent_flights, 150),
                                (helpers.rents_setup, literal_check.check_latent_rents, 700)):
        S = setup()
        lw, tr, obs = S["lw"], S["trace"], S["obs"]
        if setup is helpers.rents_setup:
            n = len(tr.locals[0])
            tr.locals[0][:] = np.stack([np.arange(n) % 5, np.arange(n) % 2], axis=1)
        w = helpers.mirror_world(oracle, lw, obs, tr, None, 1, helpers.option_logp_cpu(oracle, lw, tr))
        worst = [0.0]

        def both(block_id, node_id, ev_rows, ev_ctx, excl, n_scores):
            w.set_ev_row_by_row(False)
            lse_a, sc_a = w.eval_tree_ev(block_id, node_id, ev_rows, ev_ctx, excl, n_scores)
            w.set_ev_row_by_row(True)
            lse_r, sc_r = w.eval_tree_ev(block_id, node_id, ev_rows, ev_ctx, excl, n_scores)
            sc_a, sc_r = np.asarray(sc_a), np.asarray(sc_r)
            fin = np.isfinite(sc_a)
            assert np.array_equal(fin, np.isfinite(sc_r))
            if fin.any():
                worst[0] = max(worst[0], float(np.max(np.abs(sc_a[fin] - sc_r[fin]) / np.maximum(1.0, np.abs(sc_r[fin])))))
            assert abs(lse_a - lse_r) <= 1e-9 * max(1.0, abs(lse_r))
            return lse_r, sc_r

        assert check(S, both) > floor
        assert worst[0] < 1e-9, worst[0]


def test_literal_densities_match_kats():
    """The literal interpreter's own densities against SURVEY Appendix D's formula-derived values."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import literal as lit
    assert abs(lit.add_typos_logpdf("abc", "abc") - (-0.105360515658)) < 1e-11
    assert abs(lit.add_typos_logpdf("birminghxm", "birmingham") - (-5.751792305755)) < 1e-11
    assert abs(lit.add_typos_logpdf("bxrmxngham", "birmingham") - (-11.580545652645)) < 1e-11
    assert abs(lit.string_prior_logpdf("birmingham", 3, 30) - (-31.989617526969)) < 1e-10
    assert lit.string_prior_logpdf("al", 3, 30) == -np.inf
    assert abs(lit.string_prior_logpdf("2053258100", 10, 10) - (-33.322045101752)) < 1e-10
    assert lit.damerau_levenshtein("ca", "abc") == 2 and lit.damerau_levenshtein("ca", "abc", restricted=True) == 3


def test_prior_proposal_weights_equal_the_literal_likelihood(oracle):
    """use_dd_proposals = false (block_proposal.jl:42-56,68-84,168): a particle draws its referents from the CRP prior and
    its weight is the likelihood of the row's observations given the drawn values.  One particle, no current referents
    (initialisation mode): the sweep's log marginal likelihood estimate of a row is that particle's weight — the sum,
    over the blocks, of the AddTypos densities of the observed strings given the CHOSEN referents' strings, which the
    literal interpreter evaluates from the model description and the strings alone (its own Damerau-Levenshtein and
    negative-binomial).  Checked for every row whose particle chose existing referents in both blocks (a drawn NEW
    referent's sampled values are covered by the bit-exact HIP-vs-oracle tests)."""
    import literal as lit
    from oracle_engine import OracleEngine
    from pclean_amd.engine import InferenceConfig
    S = helpers.hospital_setup()
    lw, tr, dirty, m, q = S["lw"], S["trace"], S["dirty"], S["model"], S["query"]
    ocls = m.classes[q.cls]
    lt = lit.lit_trace_from(lw, tr)
    eng = OracleEngine(oracle, lw, S["obs"])
    cur = tr.cur.copy()
    tr.cur[:] = -1  # no retained particle: every particle is a prior draw
    try:
        choice, chosen, logml, new_rows = eng.sweep(tr, InferenceConfig(1, 1, use_dd_proposals=False), 21, 0, 0, 300)
    finally:
        tr.cur[:] = cur
    blocks = [b for b in ocls.blocks]
    checked = 0
    for i in range(300):
        if (choice[:, i] < 0).any():
            continue
        obs = {q.obsmap[c]: dirty[c][i] for c in q.obsmap}
        vals_of = {}
        for bi, battrs in enumerate(blocks):
            fk = [a for a in battrs if ocls.attr(a).kind == "fk"][0]
            bp0 = lit.BlockProposal.__new__(lit.BlockProposal)
            bp0.trace, bp0.model = lt, m
            for p, v in bp0._flat(ocls.attr(fk).target, int(choice[bi, i])).items():
                vals_of[fk + "." + p] = v
        want = 0.0
        for bi, battrs in enumerate(blocks):
            fk = [a for a in battrs if ocls.attr(a).kind == "fk"][0]
            bp = lit.BlockProposal(lt, q, battrs, obs, vals_of, restricted=False)
            own = {p[len(fk) + 1:]: v for p, v in vals_of.items() if p.startswith(fk + ".")}
            want += sum(bp._lik(t, own) for t in bp.terms if t["obs"] is not None)
        assert abs(logml[i] - want) <= 1e-10 * max(1.0, abs(want)), (i, logml[i], want)
        checked += 1
    assert checked > 200


def test_prior_proposal_weights_flights_equal_the_literal_likelihood(oracle):
    """use_dd_proposals = false on flights (block_proposal.jl:62-64,168): equality-constrained slots and the MaybeSwap
    scoring block.  One particle that is the retained one (every row keeps its current referents): the log marginal
    likelihood estimate of a row is the likelihood of its observations given those referents — 0 for the noise-free
    observations that hold, plus the four MaybeSwap densities through the learned error probabilities, which the literal
    interpreter evaluates from the model description (score_block).  The data-driven sweep of the same state gives the
    enumeration's marginal instead: the flag really switches the weight."""
    import literal as lit
    from oracle_engine import OracleEngine
    from pclean_amd.engine import InferenceConfig
    S = helpers.flights_setup()
    lw, tr, dirty, m, q = S["lw"], S["trace"], S["dirty"], S["model"], S["query"]
    ocls = m.classes[q.cls]
    lt = lit.lit_trace_from(lw, tr)
    eng = OracleEngine(oracle, lw, S["obs"])
    n = 400
    choice, chosen, logml, new_rows = eng.sweep(tr, InferenceConfig(1, 1, use_dd_proposals=False), 5, 0, 0, n)
    assert np.array_equal(choice[:, :n], tr.cur[:, :n]) and not new_rows
    dd = eng.sweep(tr, InferenceConfig(1, 1), 5, 0, 0, n)
    blocks = [b for b in ocls.blocks]
    slot_blocks = [b for b in blocks if any(ocls.attr(a).kind == "fk" for a in b)]
    score_blocks = [b for b in blocks if not any(ocls.attr(a).kind == "fk" for a in b)]
    assert len(score_blocks) == 1
    checked = differs = 0
    for i in range(n):
        row = {c: dirty[c][i] for c in q.obsmap}
        referents = {}
        for bi, battrs in enumerate(slot_blocks):
            fk = [a for a in battrs if ocls.attr(a).kind == "fk"][0]
            referents[fk] = int(tr.cur[bi, i])  # (lit_trace_from keeps the product's row ids as keys)
        want = lit.score_block(lt, q, score_blocks[0], row, referents)
        assert abs(logml[i] - want) <= 1e-10 * max(1.0, abs(want)), (i, logml[i], want)
        checked += 1
        differs += abs(dd[2][i] - logml[i]) > 1e-9
    assert checked == n and differs > 0


def test_prior_proposal_weights_rents_equal_the_literal_likelihood(oracle):
    """use_dd_proposals = false on rents (a Gaussian term whose own choices the data-driven proposal enumerates inside the
    candidate branch, experiments/rents/run.jl:19-25).  One particle, the retained one: it keeps the row's current county AND
    its current own choices (room type, unit), and the log marginal likelihood estimate of the row is the likelihood of
    its observations given them — the typo density of the county name, the ChooseUniformly density of an OBSERVED room
    type (a sampled choice's density cancels against its proposal, block_proposal.jl:42-56), and the TransformedGaussian
    density of the rent at the current unit — which the literal interpreter's pieces give from strings and the model
    description.  The chosen particle reports the own choices it kept."""
    import importlib.util
    import literal as lit
    from oracle_engine import OracleEngine
    from pclean_amd.engine import InferenceConfig
    import os
    spec = importlib.util.spec_from_file_location("mk_rents", os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                                                       "make_literal_fixtures_rents.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    S = helpers.rents_setup()
    lw, tr, dirty, m, q = S["lw"], S["trace"], S["dirty"], S["model"], S["query"]
    ocls = m.classes[q.cls]
    n = 300
    gattr = ocls.attr(lw.gauss_spec["gauss_attr"])
    units = ocls.attr(gattr.dist.unit).dist.options
    rooms = ocls.attr("br").dist.options
    for i in range(n):  # a current state of the own choices: the observed room type (else one by row), units alternating
        rt = dirty["Room Type"][i]
        tr.locals[0][i] = (rooms.index(rt) if rt is not None else i % len(rooms), i % len(units))
    lt = lit.lit_trace_from(lw, tr)
    mean_of = mk.mean_lookup(lw, tr)
    eng = OracleEngine(oracle, lw, S["obs"])
    choice, chosen, logml, new_rows = eng.sweep(tr, InferenceConfig(1, 1, use_dd_proposals=False), 8, 0, 0, n)
    assert np.array_equal(choice[:, :n], tr.cur[:, :n]) and not new_rows
    assert np.array_equal(tr.pending_locals[0][:n], tr.locals[0][:n])
    battrs = ocls.blocks[0]
    checked = with_number = 0
    for i in range(n):
        row = {c: dirty[c][i] for c in q.obsmap}
        gb = lit.GaussBlockProposal(lt, q, battrs, row, mean_of)
        vals = lt.tables["County"][int(tr.cur[0, i])]
        want = 0.0
        for path, v in gb.direct.items():
            assert v is None or vals[path] == v
        for path, v, mt in gb.typos:
            if v is not None:
                want += lit.add_typos_logpdf(v, vals[path], mt)
        own = {"br": rooms[tr.locals[0][i, 0]], gattr.dist.unit: units[tr.locals[0][i, 1]]}
        for name in gb.own:
            if gb.own_obs.get(name) is not None:
                assert own[name] == gb.own_obs[name]
                want += -np.log(len(ocls.attr(name).dist.options))
        if gb.x is not None:
            unit = own[gattr.dist.unit]
            args = {a: (vals[a.split(".", 1)[1]] if "." in a else own[a]) for a in gb.look_args}
            xb = unit.backward(gb.x)
            want += lit.normal_logpdf(xb, mean_of(args), gattr.dist.std) - np.log(abs(unit.deriv(xb)))
            with_number += 1
        assert abs(logml[i] - want) <= 1e-10 * max(1.0, abs(want)), (i, logml[i], want)
        checked += 1
    assert checked == n and with_number > 200
    # more particles: the others sample their own choices; the product's host code takes them from the chosen particle
    eng.sweep(tr, InferenceConfig(1, 6, use_dd_proposals=False), 8, 1, 0, n)
    loc = tr.pending_locals[0][:n]
    assert ((loc[:, 0] >= 0) & (loc[:, 0] < len(rooms)) & (loc[:, 1] >= 0) & (loc[:, 1] < len(units))).all()
    seen = np.array([dirty["Room Type"][i] is not None for i in range(n)])
    assert (loc[seen, 0] == tr.locals[0][:n][seen, 0]).all()  # an observed own choice is never sampled

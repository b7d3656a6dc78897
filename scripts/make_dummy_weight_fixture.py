"""tests/golden/literal_dummy_weight.json (TEST INFRASTRUCTURE): the weight of a fresh particle that chose a
ProposalDummyValue with an observation below the node (block_proposal.jl:58-60), computed from STRINGS by the literal
interpreter's own densities (oracle/literal.py: its Damerau-Levenshtein, scipy's nbinom, the lmparams CSVs) for the
`people` program (tests/dummy_program.py): per row the enumerated option scores (atoms + the dummy with its placeholder),
the log marginal, and — where the one-particle sweep (seed 11, sweep 5) drew the dummy — the string random(StringPrior)
returned for that particle's private stream and the weight
    log marginal - log(dummy mass) + logdensity(obs | drawn) - logdensity(obs | placeholder).
The C++ oracle (CPU suite) and the HIP path (-m gpu) must reproduce `logml` and the drawn strings."""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np

import dummy_program as dp
import literal as lit
import oracle as orc
from oracle_engine import OracleEngine
from pclean_amd import sampling
from pclean_amd.engine import InferenceConfig
from pclean_amd.trace import Trace

SEED, SWEEP = 11, 5

if __name__ == "__main__":
    orc.build()
    m, q, dirty, lw, obs = dp.people_program()
    eng = OracleEngine(orc, lw, obs)
    tr = Trace(lw, obs.shape[1], 0)
    choice, chosen, logml, new_rows = eng.sweep(tr, InferenceConfig(1, 1), SEED, SWEEP)
    rows_new, vals_new = new_rows[0]
    d = m.classes["Person"].attr("name").dist
    ph = d.dummy_value()
    prior = [lit.string_prior_logpdf(a, d.min_len, d.max_len) for a in dp.ATOMS]
    log_dm = math.log1p(-math.exp(lit.logsumexp(prior)))
    out = []
    for i, o in enumerate(dirty["Name"]):
        scores = [p + lit.add_typos_logpdf(o, a) for p, a in zip(prior, dp.ATOMS)] + [log_dm + lit.add_typos_logpdf(o, ph)]
        lse = lit.logsumexp(scores)
        k = int(vals_new[list(rows_new).index(i), 1])
        rec = dict(row=i, obs=o, option_scores=scores, log_marginal=lse, drawn_option=k, drew_dummy=k == len(dp.ATOMS))
        corr = 0.0
        if rec["drew_dummy"]:
            key = sampling.dummy_seed(SEED, (0 << 16) | 1, 0, SWEEP)
            s = sampling.random_string_prior_at(orc.RandomOracle(), [key], [i], d.min_len, d.max_len)[0]
            rec["drawn_string"] = s
            corr = -log_dm + lit.add_typos_logpdf(o, s) - lit.add_typos_logpdf(o, ph)
        rec["correction"] = corr
        rec["logml"] = lse + corr
        out.append(rec)
    n_obs_dummy = sum(1 for r in out if r["drew_dummy"] and r["obs"] is not None)
    assert n_obs_dummy >= 8, n_obs_dummy
    path = os.path.join(ROOT, "tests", "golden", "literal_dummy_weight.json")
    json.dump(dict(seed=SEED, sweep=SWEEP, atoms=dp.ATOMS, placeholder=ph, log_dummy_mass=log_dm, rows=out),
              open(path, "w"), indent=1)
    err = max(abs(r["logml"] - float(logml[r["row"]])) for r in out)
    print(f"{len(out)} rows, {sum(r['drew_dummy'] for r in out)} drew the dummy ({n_obs_dummy} with an observation below), "
          f"max |literal - oracle| = {err:.3e}")

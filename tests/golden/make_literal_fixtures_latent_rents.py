"""Generates tests/golden/literal_scores_latent_rents.json: rejuvenation of latent County rows of the rents program
against their evidence sets — `name` (keyed StringPrior atoms + dummy) scored by the AddTypos observation of every
referring row, `state` (ChooseProportionally) by the noise-free State observations (equality) and by the
TransformedGaussian rent of every referring row with THAT row's current own choices (room type, unit:
transformed_gaussian.jl:15-16 through avg_rent[state, countykey, br]).  Literal interpreter, strings only; the C++
oracle must reproduce the scores through the product's latent plan, build_evidence and per-evidence-row locals.

usage: python tests/golden/make_literal_fixtures_latent_rents.py"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
import numpy as np

import helpers
import literal as lit
from make_literal_fixtures_rents import mean_lookup


def fixtures(S):
    lw, tr, dirty, m, q = S["lw"], S["trace"], S["dirty"], S["model"], S["query"]
    ocls = m.classes[q.cls]
    lt = lit.lit_trace_from(lw, tr)
    mean_of = mean_lookup(lw, tr)
    n = tr.cur.shape[1]
    locs = lw.locals[0]                                  # own choices of the block, e.g. ['br', 'unit']
    own = {a: ocls.attr(a).dist.options for a in locs}
    g = ocls.attr(lw.gauss_spec["gauss_attr"])
    look = ocls.attr(g.dist.mean)
    members = {}
    for i in range(n):
        members.setdefault(int(tr.cur[0, i]), []).append(i)
    keys = sorted(members)
    out = []
    for key in [keys[j] for j in sorted(set(np.linspace(0, len(keys) - 1, 14).astype(int)))]:
        row = lt.tables["County"][key]
        rec = dict(cls="County", content="|".join(f"{p}={v}" for p, v in sorted(row.items())), n_evidence=len(members[key]), roots={})
        # name: the key's atoms + dummy, AddTypos(observed County | name, 2) over the referring rows
        a = m.classes["County"].attr("name")
        options, lps = lit.own_choice_proposal(lt, "County", a, {"countykey": row["countykey"]})
        mt = ocls.attr("county_name").dist.max_typos
        sc = {}
        for o, lp in zip(options, lps):
            sc[o] = lp + sum(lit.add_typos_logpdf(dirty["County"][i], o, mt) for i in members[key] if dirty["County"][i] is not None)
        rec["roots"]["name"] = dict(scores=sc, lse=lit.logsumexp(list(sc.values())))
        # state: prior, equality with the observed State cells, Gaussian rents with the rows' current own choices
        a = m.classes["County"].attr("state")
        probs = lt.params[("County", a.dist.param)]
        sc = {}
        for o, p in zip(a.dist.options, probs):
            s_ = math.log(p) if p > 0 else -math.inf
            for i in members[key]:
                if dirty["State"][i] is not None and dirty["State"][i] != o:
                    s_ = -math.inf
                    break
                vals = {name: own[name][int(tr.locals[0][i, li])] for li, name in enumerate(locs)}
                unit = vals[g.dist.unit]
                args = {arg: ({"state": o, "countykey": row["countykey"]}[arg.split(".", 1)[1]] if "." in arg else vals[arg])
                        for arg in look.args}
                xb = unit.backward(float(dirty["Monthly Rent"][i]))
                s_ += lit.normal_logpdf(xb, mean_of(args), g.dist.std) - math.log(abs(unit.deriv(xb)))
            sc[o] = s_
        rec["roots"]["state"] = dict(scores=sc, lse=lit.logsumexp(list(sc.values())))
        out.append(rec)
    return out


def main():
    S = helpers.rents_setup()
    S["trace"].locals[0][:] = np.stack([np.arange(len(S["trace"].locals[0])) % 5, np.arange(len(S["trace"].locals[0])) % 2], axis=1)
    fx = dict(program="rents (experiments/rents/run.jl): latent County rows against their evidence sets, state = "
                      "helpers.rents_setup() with own choices br = row % 5, unit = row % 2",
              rows=fixtures(S))
    path = os.path.join(ROOT, "tests", "golden", "literal_scores_latent_rents.json")
    json.dump(fx, open(path, "w"), indent=0, sort_keys=True)
    n_sc = sum(len(r["scores"]) for rec in fx["rows"] for r in rec["roots"].values())
    print(f"wrote {path}: {len(fx['rows'])} counties, {n_sc} scores, {os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()

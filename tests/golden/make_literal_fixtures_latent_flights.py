"""Generates tests/golden/literal_scores_latent_flights.json: rejuvenation of latent Flight / TrackingWebsite rows of the
flights program against their evidence sets — for a time attribute the keyed TimePrior proposal (the flight's atoms +
dummy) scored by the MaybeSwap observation of EVERY referring row with that row's own error probability
(run.jl:28-34: 1e-5 for the airline's own website, the learned per-website probability otherwise), missing observations
included (maybe_swap.jl:18-22).  Computed by the LITERAL interpreter from the trace's strings; the C++ oracle must
reproduce the scores through the product's latent plans, build_evidence and per-evidence-row probability index.

usage: python tests/golden/make_literal_fixtures_latent_flights.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np

import helpers
import literal as lit

TIME_ATTRS = {"sdt": "sched_dep_time", "sat": "sched_arr_time", "adt": "act_dep_time", "aat": "act_arr_time"}


def fixtures(S):
    lw, tr, dirty, m, q = S["lw"], S["trace"], S["dirty"], S["model"], S["query"]
    ocls = m.classes[q.cls]
    lt = lit.lit_trace_from(lw, tr)
    n = tr.cur.shape[1]
    members = {}
    for i in range(n):
        members.setdefault(int(tr.cur[0, i]), []).append(i)
    keys = sorted(members)
    out = []
    for key in [keys[j] for j in sorted(set(np.linspace(0, len(keys) - 1, 12).astype(int)))]:
        row = lt.tables["Flight"][key]
        fid = row["flight_id"]
        rec = dict(cls="Flight", content="|".join(f"{p}={v}" for p, v in sorted(row.items())), n_evidence=len(members[key]), roots={})
        for attr, col in TIME_ATTRS.items():
            a = m.classes["Flight"].attr(attr)
            options, lps = lit.own_choice_proposal(lt, "Flight", a, {"flight_id": fid})
            ms = ocls.attr(attr).dist  # the observed class's MaybeSwap of the same name
            j = ocls.attr(ms.prob)
            sc = {}
            for o, lp in zip(options, lps):
                s_ = lp
                for i in members[key]:
                    src = lt.tables["TrackingWebsite"][int(tr.cur[1, i])]["name"]
                    r = j.fn.fn(src, fid)
                    prob = r if isinstance(r, float) else lt.params[(q.cls, j.fn.param)][r]
                    s_ += lit.maybe_swap_logpdf(dirty[col][i], o, ms.options[fid], prob)
                sc[o] = s_
            rec["roots"][attr] = dict(scores=sc, lse=lit.logsumexp(list(sc.values())))
        out.append(rec)
    return out


def main():
    S = helpers.flights_setup()
    fx = dict(program="flights (experiments/flights/run.jl): latent Flight rows against their evidence sets, state = helpers.flights_setup()",
              rows=fixtures(S))
    path = os.path.join(ROOT, "tests", "golden", "literal_scores_latent_flights.json")
    json.dump(fx, open(path, "w"), indent=0, sort_keys=True)
    n_sc = sum(len(r["scores"]) for rec in fx["rows"] for r in rec["roots"].values())
    print(f"wrote {path}: {len(fx['rows'])} flights, {n_sc} scores, {os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()

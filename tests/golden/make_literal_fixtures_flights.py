"""Generates tests/golden/literal_scores_flights.json: for rows of flights_dirty.csv the per-candidate scores of the
two reference-slot blocks (noise-free observations only: CRP prior + equality, new row with StringPrior / TimePrior
proposals) and the value of the scoring block (four MaybeSwap observations through the row's current referents),
computed by the LITERAL interpreter (oracle/literal.py: PriorSlotProposal, score_block).  Their sum is the row's log
marginal likelihood estimate of a one-particle conditional SMC sweep (the retained particle: nothing moves), which
the C++ oracle (CPU) and the HIP path (-m gpu) must reproduce.

usage: python tests/golden/make_literal_fixtures_flights.py"""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import helpers
import literal as lit

ROWS = list(range(0, 2376, 41)) + [33, 500, 1234, 2375]


def row_fixture(S, lt0, i):
    lw, tr, dirty, m, q = S["lw"], S["trace"], S["dirty"], S["model"], S["query"]
    ocls = m.classes[q.cls]
    row = {c: dirty[c][i] for c in q.obsmap}
    slots = [(b[0], ocls.attr(b[0]).target) for b in ocls.blocks[:2]]
    refs = {slot: int(tr.cur[bi, i]) for bi, (slot, _) in enumerate(slots)}
    lt = copy.deepcopy(lt0)
    for slot, cls in slots:  # unincorporate the row (row_inference.jl:115-126)
        lt.unrefer(cls, refs[slot])
    out = dict(row=i, blocks=[])
    total = 0.0
    for slot, cls in slots:
        direct = {da.split(".", 1)[1]: row[c] for c, da in q.obsmap.items() if da.startswith(slot + ".")}
        sc = lit.PriorSlotProposal(lt, cls, direct).scores()
        cands = {"|".join(f"{p}={v}" for p, v in sorted(lt.tables[cls][k].items())): v_ for k, v_ in sc.items()
                 if k != "NEW" and v_ != float("-inf")}
        lse = lit.logsumexp(list(sc.values()))
        total += lse
        out["blocks"].append(dict(cls=cls, cands=cands, new=sc["NEW"], lse=lse))
    out["score_block"] = lit.score_block(lt0, q, ocls.blocks[2], row, refs)
    out["logml"] = total + out["score_block"]
    return out


def main():
    S = helpers.flights_setup()
    lt0 = lit.lit_trace_from(S["lw"], S["trace"])
    fx = dict(program="flights (experiments/flights/run.jl), flights_dirty.csv, latent state = helpers.flights_setup()",
              rows=[row_fixture(S, lt0, i) for i in sorted(set(ROWS))])
    path = os.path.join(ROOT, "tests", "golden", "literal_scores_flights.json")
    json.dump(fx, open(path, "w"), indent=0, sort_keys=True)
    print(f"wrote {path}: {len(fx['rows'])} rows, {os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()

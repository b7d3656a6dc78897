"""Generates tests/golden/literal_scores_rents.json: per-candidate scores of the enumerated block proposal of rows of
rents_dirty.csv (reference slot + noise-free observations of latent attributes + keyed StringPrior + own uniform
choices + TransformedGaussian), computed by the LITERAL interpreter (oracle/literal.py: GaussBlockProposal — model
description + strings, nothing of the product's plan arrays).  Both the C++ oracle (tests/test_literal_fixtures.py,
CPU) and the HIP path (tests/test_gpu_literal.py, -m gpu) must reproduce them.

usage: python tests/golden/make_literal_fixtures_rents.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import helpers
import literal as lit

N_PICK = 24


def mean_lookup(lw, tr):
    """{lookup argument: string} -> current value of the IndexedParameter entry (decoding of the dense table only)."""
    spec = lw.gauss_spec
    ocls = lw.model.classes[lw.query.cls]
    look = ocls.attr(ocls.attr(spec["gauss_attr"]).dist.mean)

    def mean_of(args):
        idx = 0
        for arg, d, st in zip(look.args, spec["dims"], spec["strides"]):
            if d[0] == "cand":
                idx += st * lw.latent_dom[d[3]].get(args[arg])
            else:
                idx += st * ocls.attr(arg).dist.options.index(args[arg])
        return float(tr.mean_param.value[idx])

    return mean_of


def pick_rows(dirty, n):
    """six rows of each kind: everything observed / missing State / missing Room Type / both missing"""
    per = {}
    for i in range(n):
        kind = (dirty["State"][i] is None, dirty["Room Type"][i] is None)
        if len(per.setdefault(kind, [])) < N_PICK // 4:
            per[kind].append(i)
    return sorted(i for rows in per.values() for i in rows)


def row_fixture(S, i):
    lw, tr, dirty, m, q = S["lw"], S["trace"], S["dirty"], S["model"], S["query"]
    ocls = m.classes[q.cls]
    lt = lit.lit_trace_from(lw, tr)
    battrs = ocls.blocks[0]
    fk = [a for a in battrs if ocls.attr(a).kind == "fk"][0]
    lt.unrefer(ocls.attr(fk).target, int(tr.cur[0, i]))  # unincorporate the row (row_inference.jl:115-126)
    row = {c: dirty[c][i] for c in q.obsmap}
    bp = lit.GaussBlockProposal(lt, q, battrs, row, mean_lookup(lw, tr))
    sc = bp.scores()
    cands = {"|".join(f"{p}={v}" for p, v in sorted(lt.tables["County"][k].items())): v_ for k, v_ in sc.items() if k != "NEW"}
    finite = {k: v for k, v in cands.items() if v != float("-inf")}
    return dict(row=i, cls="County", cands=finite, n_impossible=len(cands) - len(finite), new=sc["NEW"],
                lse=lit.logsumexp(list(sc.values())))


def main():
    S = helpers.rents_setup()
    rows = pick_rows(S["dirty"], S["obs"].shape[1])
    fx = dict(program="rents (experiments/rents/run.jl), rents_dirty.csv first 600 rows, latent state = helpers.rents_setup()",
              rows=[row_fixture(S, i) for i in rows])
    path = os.path.join(ROOT, "tests", "golden", "literal_scores_rents.json")
    json.dump(fx, open(path, "w"), indent=0, sort_keys=True)
    n = sum(len(r["cands"]) + 1 for r in fx["rows"])
    print(f"wrote {path}: {len(fx['rows'])} rows, {n} finite candidate scores, {os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()

"""Generates tests/golden/literal_scores.json: per-candidate scores of the enumerated block proposals of rows of
hospital_dirty.csv, computed by the LITERAL interpreter (oracle/literal.py: model description + strings, nothing of
the product's lowering).  Both the C++ oracle (tests/test_literal_fixtures.py, CPU) and the HIP path
(tests/test_gpu_literal.py, -m gpu) must reproduce them to 1e-12 relative.

usage: python tests/golden/make_literal_fixtures.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np

import helpers
import literal as lit

ROWS = [0, 1, 5, 17, 60, 123, 250, 251, 400, 517, 640, 777, 901, 999]


def content_key(lt, cls, key):
    """The candidate's identity independent of row ids: its flattened values, path-sorted."""
    bp = lit.BlockProposal.__new__(lit.BlockProposal)
    bp.trace, bp.model = lt, lt.model
    flat = bp._flat(cls, key)
    return "|".join(f"{p}={flat[p]}" for p in sorted(flat))


def row_fixture(S, i):
    lw, tr, dirty, m, q = S["lw"], S["trace"], S["dirty"], S["model"], S["query"]
    ocls = m.classes[q.cls]
    lt = lit.lit_trace_from(lw, tr)
    obs = {q.obsmap[c]: dirty[c][i] for c in q.obsmap}
    blocks = [b for b in ocls.blocks]
    # values of the row's CURRENT referents (the retained particle) — the context of later blocks
    cur_vals = {}
    for bi, battrs in enumerate(blocks):
        fk = [a for a in battrs if ocls.attr(a).kind == "fk"][0]
        key = int(tr.cur[bi, i])
        bp = lit.BlockProposal.__new__(lit.BlockProposal)
        bp.trace, bp.model = lt, m
        for p, v in bp._flat(ocls.attr(fk).target, key).items():
            cur_vals[fk + "." + p] = v
    # unincorporate the row (run_smc!, row_inference.jl:115-126): both referents lose one reference
    for bi, battrs in enumerate(blocks):
        fk = [a for a in battrs if ocls.attr(a).kind == "fk"][0]
        lt.unrefer(ocls.attr(fk).target, int(tr.cur[bi, i]))
    out = dict(row=i, blocks=[])
    for bi, battrs in enumerate(blocks):
        fk = [a for a in battrs if ocls.attr(a).kind == "fk"][0]
        tcls = ocls.attr(fk).target
        bp = lit.BlockProposal(lt, q, battrs, obs, cur_vals, restricted=False)
        sc = bp.scores()
        cands = {content_key(lt, tcls, k): v for k, v in sc.items() if k != "NEW"}
        out["blocks"].append(dict(cls=tcls, cands=cands, new=sc["NEW"], lse=lit.logsumexp(list(sc.values())),
                                  context={a: cur_vals[a] for j in ocls.attrs if j.kind == "julia" for a in j.args
                                           if not a.startswith(fk + ".")} if bi else {}))
    return out


def main():
    S = helpers.hospital_setup()
    fx = dict(program="hospital (experiments/hospital/run.jl), hospital_dirty.csv, latent state = helpers.hospital_setup()",
              distance="unrestricted Damerau-Levenshtein", rows=[row_fixture(S, i) for i in ROWS])
    path = os.path.join(ROOT, "tests", "golden", "literal_scores.json")
    json.dump(fx, open(path, "w"), indent=0, sort_keys=True)
    n = sum(len(b["cands"]) + 1 for r in fx["rows"] for b in r["blocks"])
    print(f"wrote {path}: {len(fx['rows'])} rows, {n} candidate scores, {os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()

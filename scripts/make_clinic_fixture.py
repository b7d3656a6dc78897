"""tests/golden/literal_scores_clinic.json (TEST INFRASTRUCTURE): per-candidate scores of the `clinic` program
(tests/clinic_program.py: three reference slots in one block, a single-argument JuliaNode, JuliaNodes across the slots,
two context values read by the third slot) computed by the LITERAL interpreter (oracle/literal.py) from strings: every
slot of the block enumerated given the row's current values of the earlier slots — the lowering's one-plan-per-slot
decomposition of a multi-slot block (pclean_amd/model.py: _build_blocks).  The C++ oracle (CPU) and the HIP path (GPU)
must reproduce them through the product's lowering."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np

import clinic_program as cp
import literal as lit
from pclean_amd.trace import Trace

ROWS = list(range(0, 120, 5))


def clinic_setup():
    S = cp.clinic_program()
    lw, clean, dirty = S["lw"], S["clean"], S["dirty"]
    n = S["obs"].shape[1]
    by = {0: {"name": clean["DName"], "spec": clean["DSpec"]},
          1: {"city": clean["City"], "code": [t[2:] for t in clean["Tag"]]},
          2: {"kind": clean["Kind"]}}
    S["trace"] = Trace.from_clean_values(lw, by, n, 3)
    return S


def content_key(lt, cls, key):
    bp = lit.BlockProposal.__new__(lit.BlockProposal)
    bp.trace, bp.model = lt, lt.model
    flat = bp._flat(cls, key)
    return "|".join(f"{p}={flat[p]}" for p in sorted(flat))


def row_fixture(S, i):
    lw, tr, dirty, m, q = S["lw"], S["trace"], S["dirty"], S["model"], S["query"]
    ocls = m.classes[q.cls]
    lt = lit.lit_trace_from(lw, tr)
    obs = {q.obsmap[c]: dirty[c][i] for c in q.obsmap}
    blocks = lw.engine_blocks
    cur_vals = {}
    for bi, battrs in enumerate(blocks):
        fk = [a for a in battrs if ocls.attr(a).kind == "fk"][0]
        bp = lit.BlockProposal.__new__(lit.BlockProposal)
        bp.trace, bp.model = lt, m
        for p, v in bp._flat(ocls.attr(fk).target, int(tr.cur[bi, i])).items():
            cur_vals[fk + "." + p] = v
    for bi, battrs in enumerate(blocks):
        fk = [a for a in battrs if ocls.attr(a).kind == "fk"][0]
        lt.unrefer(ocls.attr(fk).target, int(tr.cur[bi, i]))
    out = dict(row=i, blocks=[])
    for bi, battrs in enumerate(blocks):
        fk = [a for a in battrs if ocls.attr(a).kind == "fk"][0]
        tcls = ocls.attr(fk).target
        sc = lit.BlockProposal(lt, q, battrs, obs, cur_vals, restricted=False).scores()
        out["blocks"].append(dict(cls=tcls, cands={content_key(lt, tcls, k): v for k, v in sc.items() if k != "NEW"},
                                  new=sc["NEW"], lse=lit.logsumexp(list(sc.values()))))
    return out


if __name__ == "__main__":
    S = clinic_setup()
    fx = dict(program="clinic (tests/clinic_program.py)", distance="unrestricted Damerau-Levenshtein",
              rows=[row_fixture(S, i) for i in ROWS])
    path = os.path.join(ROOT, "tests", "golden", "literal_scores_clinic.json")
    json.dump(fx, open(path, "w"), indent=0, sort_keys=True)
    print(f"{len(fx['rows'])} rows, {sum(len(b['cands']) + 1 for r in fx['rows'] for b in r['blocks'])} candidate scores")

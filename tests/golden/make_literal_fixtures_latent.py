"""Generates tests/golden/literal_scores_latent.json: rejuvenation of LATENT rows of the hospital program against their
evidence sets (every likelihood term of every observed row that refers to the latent row, ExternalLikelihoodNodes of
proposal_compiler.jl:306-350) — per-option scores of own choices and per-candidate scores of reference slots, computed
by the LITERAL interpreter (oracle/literal.py: LatentProposal; evidence rows and contexts derived here from the trace's
strings, not from the product's build_evidence).  The C++ oracle must reproduce them from the product's latent plans,
build_evidence and ctx wiring (tests/test_literal_fixtures.py).

usage: python tests/golden/make_literal_fixtures_latent.py"""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np

import helpers
import literal as lit

PER_CLASS = 5
# latent class -> (index of the observed class's block it hangs below, path below that block's slot)
WHERE = {"Hospital": (0, ""), "Place": (0, "loc."), "County": (0, "loc.county."), "HospitalType": (0, "type."),
         "Measure": (1, ""), "Condition": (1, "condition.")}


def follow(lt, model, cls, key, sub):
    for part in [p for p in sub.split(".") if p]:
        a = model.classes[cls].attr(part)
        key = lt.tables[cls][key][part]
        cls = a.target
    return cls, key


def flat_key(lt, cls, key):
    bp = lit.BlockProposal.__new__(lit.BlockProposal)
    bp.trace, bp.model = lt, lt.model
    flat = bp._flat(cls, key)
    return "|".join(f"{p}={flat[p]}" for p in sorted(flat))


def fixtures(S):
    lw, tr, dirty, m, q = S["lw"], S["trace"], S["dirty"], S["model"], S["query"]
    ocls = m.classes[q.cls]
    lt0 = lit.lit_trace_from(lw, tr)
    n = tr.cur.shape[1]
    slots = [[a for a in b if ocls.attr(a).kind == "fk"][0] for b in ocls.blocks]
    # values of every row's current referents (JuliaNode arguments that live in the other block)
    bp0 = lit.BlockProposal.__new__(lit.BlockProposal)
    bp0.trace, bp0.model = lt0, m
    flat_of = {}
    out = []
    for cname, (bi, sub) in WHERE.items():
        top_cls = ocls.attr(slots[bi]).target
        members = {}
        for i in range(n):
            c, k = follow(lt0, m, top_cls, int(tr.cur[bi, i]), sub)
            assert c == cname
            members.setdefault(k, []).append(i)
        keys = sorted(members)
        picks = [keys[j] for j in sorted(set(np.linspace(0, len(keys) - 1, PER_CLASS).astype(int)))]
        for key in picks:
            evidence = []
            for i in members[key]:
                ctx = {}
                for ob, slot in enumerate(slots):
                    if ob == bi:
                        continue
                    k2 = int(tr.cur[ob, i])
                    fk2 = (ocls.attr(slot).target, k2)
                    if fk2 not in flat_of:
                        flat_of[fk2] = bp0._flat(*fk2)
                    ctx.update({slot + "." + p: v for p, v in flat_of[fk2].items()})
                evidence.append(({q.obsmap[c]: dirty[c][i] for c in q.obsmap}, ctx))
            rec = dict(cls=cname, content=flat_key(lt0, cname, key), n_evidence=len(evidence), roots={})
            for a in m.classes[cname].attrs:
                if a.kind == "choice":
                    lp = lit.LatentProposal(lt0, q, ocls.blocks[bi], sub, evidence)
                    sc = lp.leaf_scores(cname, a.name)
                    rec["roots"][a.name] = dict(kind="leaf", scores=sc, lse=lit.logsumexp(list(sc.values())))
                elif a.kind == "fk":
                    lt = copy.deepcopy(lt0)
                    lt.unrefer(a.target, lt.tables[cname][key][a.name])  # the row's own reference is taken back first
                    lp = lit.LatentProposal(lt, q, ocls.blocks[bi], sub, evidence)
                    sc = lp.slot_scores(a.name)
                    cands = {flat_key(lt, a.target, k): v for k, v in sc.items() if k != "NEW"}
                    rec["roots"][a.name] = dict(kind="fk", cands=cands, new=sc["NEW"], lse=lit.logsumexp(list(sc.values())))
            out.append(rec)
    return out


def main():
    S = helpers.hospital_setup()
    fx = dict(program="hospital (experiments/hospital/run.jl): latent rows against their evidence sets, state = helpers.hospital_setup()",
              rows=fixtures(S))
    path = os.path.join(ROOT, "tests", "golden", "literal_scores_latent.json")
    json.dump(fx, open(path, "w"), indent=0, sort_keys=True)
    n = sum(len(r.get("scores", r.get("cands", {}))) + 1 for rec in fx["rows"] for r in rec["roots"].values())
    print(f"wrote {path}: {len(fx['rows'])} latent rows, {n} scores, {os.path.getsize(path) / 1e3:.0f} kB")


if __name__ == "__main__":
    main()

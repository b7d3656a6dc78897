"""tests/golden/plans_<program>.json: the static plan IR (include/pclean_hip.h: pclean_node / pclean_term / children /
colmap / ctx sources, latent-class plans, scoring blocks, table layouts, option tables) that pclean_amd.model.LoweredModel
produces for the three experiment programs on the first rows of this image's datasets, as plain JSON — the reference
output a Julia maintainer's lowering (julia/PCleanHIP.jl: lower_model) can be diffed against without running Python.
tests/test_plan_goldens.py regenerates and compares them.  usage: python scripts/make_plan_goldens.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

from pclean_amd import experiments as ex
from pclean_amd.model import LoweredModel

N_ROWS = {"hospital": 200, "flights": 300, "rents": 400}


def lowered(name):
    n = N_ROWS[name]
    if name == "hospital":
        dirty, _ = ex.hospital_data()
        dirty = {c: v[:n] for c, v in dirty.items()}
        m = ex.hospital_model(ex.possibilities_of(dirty))
        return LoweredModel(m, ex.hospital_query(m), dirty), dirty
    if name == "flights":
        dirty, _ = ex.flights_data()
        dirty = {c: v[:n] for c, v in dirty.items()}
        m = ex.flights_model(dirty)
        return LoweredModel(m, ex.flights_query(m), dirty), dirty
    dirty, _ = ex.rents_data()
    dirty = {c: v[:n] for c, v in dirty.items()}
    m = ex.rents_model(dirty)
    return LoweredModel(m, ex.rents_query(m), dirty), dirty


def _l(a):
    return np.asarray(a).tolist()


def describe(lw):
    out = dict(classes=list(lw.model.class_order), observed_class=lw.query.cls,
               layout={c: [dict(name=col.name, kind=col.kind, cls=col.cls, attr=col.attr, target=col.target) for col in cols]
                       for c, cols in lw.layout.items()},
               table_id=dict(lw.table_id), option_id={f"{c}.{a}": i for (c, a), i in lw.option_id.items()},
               obs_cols=list(lw.obs_cols),
               latent_domain_sizes={f"{c}.{a}": len(d) for (c, a), d in lw.latent_dom.items()},
               option_values={f"{c}.{a}": _l(v) for (c, a), v in lw.option_values.items() if len(v) <= 64},
               pair_tables={str(pid): dict(observed=key[0], latent=list(key[1]) if isinstance(key[1], tuple) else key[1],
                                           n_obs=len(od), n_lat=len(ld)) for key, (pid, od, ld) in lw.pair_id.items()},
               fn_tables={str(fid): list(fn.shape) for fid, fn in lw.fn_tables.items()}, blocks=[], latent_plans={})
    for bi, blk in enumerate(lw.blocks):
        if blk.get("score"):
            out["blocks"].append(dict(score=True, args=[_l(x) if not isinstance(x, int) else x for x in lw.score_block_args(bi)[1:]]))
            continue
        nodes, terms, children, colmap, csb, csc = lw.block_arrays(bi)
        out["blocks"].append(dict(root_class=blk["root_class"], nodes=[_l(list(n)) for n in nodes], terms=[_l(list(t)) for t in terms],
                                  children=_l(children), colmap=_l(colmap), ctx_src_block=_l(csb), ctx_src_col=_l(csc),
                                  node_info=[dict(kind=i["kind"], cls=i["cls"], attr=i["attr"], path=i["path"]) for i in blk["node_info"]]))
    for cname, pl in lw.latent_plans.items():
        nodes, terms, children, colmap, _, _ = lw.latent_block_arrays(cname)
        out["latent_plans"][cname] = dict(block_id=pl["block_id"], src_block=pl["src_block"], path=pl["path"], roots=_l(pl["roots"]),
                                          root_attr=list(pl["root_attr"]), nodes=[_l(list(n)) for n in nodes],
                                          terms=[_l(list(t)) for t in terms], children=_l(children), colmap=_l(colmap))
    return out


if __name__ == "__main__":
    for name in N_ROWS:
        lw, _ = lowered(name)
        path = os.path.join(ROOT, "tests", "golden", f"plans_{name}.json")
        json.dump(dict(program=name, rows=N_ROWS[name], node_fields=["kind", "table", "term_begin", "n_terms", "child_begin",
                       "n_children", "parent", "parent_fk_col", "cacheable", "colmap_begin", "dummy_value", "dummy_spec"],
                       term_fields=["obs_col", "cand_col", "pair_table", "dens_kind", "max_typos", "ctx_slot", "fn_table", "ctx_mode"],
                       **describe(lw)), open(path, "w"), indent=1, sort_keys=True)
        print(name, os.path.getsize(path), "bytes")

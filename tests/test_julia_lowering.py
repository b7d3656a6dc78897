"""The lowering julia/PCleanHIP.jl states (PCleanModel + Query + data -> plan IR), walked in Python on the reference's own
model structures (tests/julia_lowering.py over tests/julia_refmodel.py), must produce the plans of the product's lowering
(pclean_amd/model.py) for all three experiment programs: the committed goldens tests/golden/plans_*.json array by array, and
the live LoweredModel's table contents (option tables, value-function tables, pair-table strings, Gaussian specs)."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import julia_lowering as L  # noqa: E402
import julia_refmodel as R  # noqa: E402
import make_plan_goldens as G  # noqa: E402

PROGRAMS = ["hospital", "flights", "rents"]


def name_of(model, cls, v):
    """dotted DSL name of vertex v of class cls (None: an argument node without a name)"""
    cm = model.classes[cls]
    n = cm.node(v)
    if isinstance(n, R.SubmodelNode):
        fk = cm.node(n.foreign_key_node_id)
        while isinstance(fk, R.SubmodelNode):
            fk = fk.subnode
        head = name_of(model, cls, n.foreign_key_node_id)
        return head + "." + name_of(model, fk.target_class, n.subnode_id)
    inv = {vv: k for k, vv in cm.names.items()}
    return inv.get(v)


@pytest.fixture(scope="module", params=PROGRAMS)
def both(request):
    name = request.param
    plw, dirty = G.lowered(name)
    rm, rq = R.build_reference_model(plw.model, plw.query)
    data = {c: list(dirty[c]) for c in rq.columns}
    lw = L.lower(rm, rq, data)
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", f"plans_{name}.json")))
    return name, plw, lw, rm, rq, golden


def key_name(rm, key):
    return f"{key[0]}.{name_of(rm, key[0], key[1])}"


def test_reference_structures_number_vertices_like_the_builder():
    """spot checks of the structure mimic against the reference's own numbering rules (builder.jl): argument nodes precede
    their choice, a reference slot is followed by one copy per target node, copies shift their arguments"""
    plw, _ = G.lowered("hospital")
    rm, rq = R.build_reference_model(plw.model, plw.query)
    county = rm.classes["County"]
    assert isinstance(county.node(1), R.ParameterNode) and county.names["state"] == 3  # const options at 2, the choice at 3
    assert county.node(3).arg_node_ids == [2, 1]
    place = rm.classes["Place"]
    assert isinstance(place.node(1), R.ForeignKeyNode) and place.node(1).vmap == {i: i + 1 for i in range(1, 8)}
    assert isinstance(place.node(4), R.SubmodelNode) and place.node(4).subnode.arg_node_ids == [3, 2]
    rec = rm.classes["Record"]
    assert rec.names["hosp"] == 1 and len(rec.nodes) == 67 and len(rec.blocks) == 2
    assert L.resolve(rm, "Record", rq.cleanmap["State"]) == ("County", 3)


def test_transliteration_reproduces_the_plan_goldens(both):
    name, plw, lw, rm, rq, g = both
    assert g["classes"] == rm.class_order and g["observed_class"] == rq.cls
    assert g["table_id"] == lw.table_id
    assert g["option_id"] == {key_name(rm, k): i for k, i in lw.option_id.items()}
    assert g["latent_domain_sizes"] == {key_name(rm, k): len(lw.latent_dom[k]) for k in lw.dom_keys}
    assert list(g["latent_domain_sizes"]) == sorted(g["latent_domain_sizes"])  # (json sort_keys: order is pinned by option_id)
    assert g["obs_cols"] == [name_of(rm, rq.cls, v) for v in lw.obs_vertices]
    for c, cols in g["layout"].items():
        assert [col["name"] for col in cols] == [name_of(rm, c, v) for v in lw.layout[c]]
    for k, vals in g["option_values"].items():
        key = next(kk for kk in lw.dom_keys if key_name(rm, kk) == k)
        assert vals == lw.option_values[key]
    assert len(g["blocks"]) == len(lw.blocks)
    for bi, (gb, blk) in enumerate(zip(g["blocks"], lw.blocks)):
        if gb.get("score"):
            assert blk.score
            assert gb["args"] == [x for x in L.score_block_args(lw, bi)]
            continue
        assert gb["root_class"] == blk.root_class
        assert gb["nodes"] == [list(n) for n in blk.nodes]
        assert gb["terms"] == [list(t) for t in blk.terms]
        assert gb["children"] == blk.children and gb["colmap"] == blk.colmap
        assert gb["ctx_src_block"] == blk.ctx_block and gb["ctx_src_col"] == blk.ctx_col
        for nid, info in enumerate(gb["node_info"]):
            assert info["cls"] == blk.node_class[nid]
            assert info["kind"] == ("fk" if blk.nodes[nid][0] == L.NODE_FK else "leaf")
    assert list(g["latent_plans"]) == sorted(lw.plan_keys)
    for cname, gp in g["latent_plans"].items():
        pl = lw.latent_plans[cname]
        assert gp["block_id"] == pl["block_id"] and gp["src_block"] == pl["src_block"] and gp["roots"] == pl["roots"]
        assert gp["root_attr"] == [name_of(rm, cname, v) for v in pl["root_vertex"]]
        assert gp["nodes"] == [list(n) for n in pl["nodes"]]
        assert gp["terms"] == [list(t) for t in pl["terms"]]
        assert gp["children"] == pl["children"] and gp["colmap"] == pl["colmap"]
    assert {int(k): v for k, v in g["fn_tables"].items()} == {i: [len(f), len(f[0])] for i, f in enumerate(lw.fn_tables)}
    assert len(g["pair_tables"]) == len(lw.pair_keys)
    for (ov, key), (pid, od, ld) in lw.pair_id.items():
        gp = g["pair_tables"][str(pid)]
        assert gp["observed"] == name_of(rm, rq.cls, ov) and gp["n_obs"] == len(od) and gp["n_lat"] == len(ld)
        assert gp["latent"] == ([key[0], name_of(rm, *key)] if key[0] != "julia" else ["julia", name_of(rm, rq.cls, key[1])])


def test_transliteration_builds_the_same_tables_as_the_product_lowering(both):
    """contents, not only shapes: domains' strings in order, option tables (values / key column / count column), the
    value-function tables, pair-table string lists, numeric columns, plan order, the Gaussian specs"""
    name, plw, lw, rm, rq, g = both
    assert [key_name(rm, k) for k in lw.dom_keys] == [f"{c}.{a}" for (c, a) in plw.latent_dom]  # creation order = numbering
    for k in lw.dom_keys:
        pd = plw.latent_dom[(k[0], name_of(rm, *k))]
        assert [lw.pool.strings[i] for i in lw.latent_dom[k].ids] == [plw.pool.strings[i] for i in pd.ids]
        assert lw.latent_dom[k].n_base == pd.n_base()
        pk = (k[0], name_of(rm, *k))
        assert lw.option_values[k] == plw.option_values[pk].tolist()
        assert (k in lw.option_keycol) == (pk in plw.option_keycol)
        if k in lw.option_keycol:
            assert lw.option_keycol[k] == plw.option_keycol[pk].tolist() and lw.option_ncol[k] == plw.option_ncol[pk].tolist()
    for v in lw.obs_vertices:
        od = plw.obs_dom[name_of(rm, rq.cls, v)]
        assert [lw.pool.strings[i] for i in lw.obs_dom[v].ids] == [plw.pool.strings[i] for i in od.ids]
    assert lw.num_cols == plw.num_cols
    assert len(lw.fn_tables) == len(plw.fn_tables)
    for i, f in enumerate(lw.fn_tables):
        assert np.array_equal(np.asarray(f), plw.fn_tables[i])
    for (ov, key), (pid, od, ld) in lw.pair_id.items():
        pkey = (name_of(rm, rq.cls, ov), (key[0], name_of(rm, *key)) if key[0] != "julia" else ("julia", name_of(rm, rq.cls, key[1])))
        ppid, pod, pld = plw.pair_id[pkey]
        assert ppid == pid
        plat = pld.ids if hasattr(pld, "ids") else pld
        assert [lw.pool.strings[i] for i in ld] == [plw.pool.strings[i] for i in plat]
    assert {key_name(rm, k): v for k, v in lw.eq_pairs.items()} == {f"{k[1][0]}.{k[1][1]}": v for k, v in plw.eq_pairs.items()}
    assert sorted(p for p, _ in lw.eq_pairs.values()) == sorted(p for p, _ in plw.eq_pairs.values())
    assert sorted(lw.same_pairs) == sorted(plw.same_pairs)
    assert lw.plan_keys == list(plw.latent_plans)
    assert lw.block_group == list(plw.block_group)
    assert lw.latent_ev_locals == plw.latent_ev_locals and lw.latent_ev_prob == plw.latent_ev_prob
    assert sorted(lw.gauss) == sorted(plw.gauss)
    for k, gs in lw.gauss.items():
        ps = plw.gauss[k]
        assert [tuple(x) for x in gs["kinds"]] == [tuple(x) for x in ps["kinds"]]
        for f in ("x_col", "n_mean", "strides", "n_locals", "local_n", "local_obs", "t_local", "sigma", "t_scale", "t_lad"):
            assert gs[f] == ps[f], (k, f)
        assert tuple(gs["transform"]) == tuple(ps["transform"])
        assert name_of(rm, rq.cls, gs["param"]) == ps["param"][1]
    if lw.prob_spec is not None:
        pp = plw.prob_spec
        assert lw.prob_spec["fn"] == pp["fn"] and list(lw.prob_spec["a"]) == list(pp["a"]) and list(lw.prob_spec["b"]) == list(pp["b"])
        assert lw.prob_spec["consts"] == list(pp["consts"]) and [tuple(k) if isinstance(k, (list, tuple)) else k for k in lw.prob_spec["keys"]] == \
            [tuple(k) if isinstance(k, (list, tuple)) else k for k in pp["keys"]]


def test_lowering_refuses_what_it_cannot_state():
    """guards of the Julia text, walked here: a JuliaNode under AddTypos with two values of its own slot, a Gaussian with a
    non-linear unit"""
    plw, dirty = G.lowered("hospital")
    rm, rq = R.build_reference_model(plw.model, plw.query)
    rec = rm.classes["Record"]
    j = rec.node(rec.node(rec.names["stateavg_obs"]).arg_node_ids[0])
    keep = list(j.arg_node_ids)
    j.arg_node_ids = [keep[1], keep[1]]
    with pytest.raises(NotImplementedError):
        L.lower(rm, rq, {c: list(dirty[c]) for c in rq.columns})
    j.arg_node_ids = keep
    plw, dirty = G.lowered("rents")
    rm, rq = R.build_reference_model(plw.model, plw.query)
    obs = rm.classes["Obs"]
    units = L.const_args(obs, obs.node(obs.names["unit"]))[0]

    class Cubic:
        def backward(self, x):
            return x ** 3

        def deriv(self, x):
            return 1.0
    units[0] = Cubic()
    with pytest.raises(NotImplementedError):
        L.lower(rm, rq, {c: list(dirty[c]) for c in rq.columns})


# ---- the Julia text itself (never executed here: no Julia in the image) -------------------------------------------------------
JL = os.path.join(ROOT, "julia", "PCleanHIP.jl")


def _julia_functions(text):
    import re
    names = set(re.findall(r"^\s*function\s+([A-Za-z_][A-Za-z_0-9]*!?)", text, flags=re.M))
    names |= set(re.findall(r"^([A-Za-z_][A-Za-z_0-9]*!?)\([^=\n]*\)\s*=", text, flags=re.M))
    return names


def test_julia_section_states_every_function_of_the_transliteration():
    """tests/julia_lowering.py and section 3 of PCleanHIP.jl are the same algorithm function by function: every function of the
    Python walk exists under the same name (with Julia's `!` where it mutates) in the Julia file"""
    import inspect
    text = open(JL).read()
    jl = _julia_functions(text)
    mine = [n for n, f in inspect.getmembers(L, inspect.isfunction) if f.__module__ == L.__name__]
    assert len(mine) >= 25
    missing = [n for n in mine if n not in jl and n + "!" not in jl]
    assert not missing, missing
    # the guards the Python walk raises exist as error(...) texts on the Julia side
    for msg in ("one value of this slot, at most one of an earlier slot", "context must come from an earlier slot",
                "keyed atoms need their key attribute observed directly", "only linear Transformations",
                "a block without a reference slot may only hold MaybeSwap observations", "exactly one candidate-side index value may be unobserved",
                "at most two enumerated own choices", "one Gaussian observation per block"):
        assert msg in text and msg in open(L.__file__).read(), msg


def test_julia_ccalls_name_exported_symbols_only():
    """every `ccall((:pclean_..., lib)` of the Julia file names a function include/pclean_hip.h declares"""
    import re
    text = open(JL).read()
    header = open(os.path.join(ROOT, "include", "pclean_hip.h")).read()
    syms = set(re.findall(r"ccall\(\(:([a-z_0-9]+),\s*lib\)", text))
    assert len(syms) >= 30
    declared = set(re.findall(r"\b(pclean_[a-z_0-9]+)\s*\(", header))
    assert syms <= declared, sorted(syms - declared)


def test_julia_text_is_balanced():
    """cheap syntax hygiene for a file that cannot be parsed here: brackets balance outside strings and comments, and every
    block opener at the start of a statement has its `end`"""
    import re
    text = open(JL).read()
    text = re.sub(r'"(?:\\.|[^"\\])*"', '""', text, flags=re.S)   # strings (docstrings may span lines)
    text = re.sub(r"'(?:\\.|[^'\\])'", "''", text)              # chars
    depth = {"(": 0, "[": 0, "{": 0}
    close = {")": "(", "]": "[", "}": "{"}
    for ln, line in enumerate(text.split("\n"), start=1):
        if re.match(r"^(function |end$|struct |mutable struct )", line):   # a top-level statement starts with everything closed
            assert all(v == 0 for v in depth.values()), (ln, line, depth)
        s = line.split("#")[0]
        for ch in s:
            if ch in depth:
                depth[ch] += 1
            elif ch in close:
                depth[close[ch]] -= 1
                assert depth[close[ch]] >= 0, (ln, line)
    assert depth == {"(": 0, "[": 0, "{": 0}, depth


def test_julia_block_keywords_balance():
    """function / for / if / while / begin / struct / let / try / do / module at bracket depth 0 (comprehension `for`s and index
    `end`s live inside brackets) pair up with `end`, and every top-level definition starts with all blocks closed"""
    import re
    text = open(JL).read()
    text = re.sub(r'"(?:\\.|[^"\\])*"', '""', text, flags=re.S)
    text = re.sub(r"'(?:\\.|[^'\\])'", "''", text)
    tok = re.compile(r"[A-Za-z_][A-Za-z_0-9!]*|[\(\)\[\]\{\}]")
    depth, stack = 0, []
    for ln, line in enumerate(text.split("\n"), start=1):
        line = line.split("#")[0]
        if re.match(r"^(function |struct |mutable struct )", line):
            assert [t for t, _ in stack] == ["module"], (ln, stack)
        for m in tok.finditer(line):
            t = m.group(0)
            if t in "([{":
                depth += 1
            elif t in ")]}":
                depth -= 1
            elif depth == 0 and t in ("function", "for", "if", "while", "begin", "struct", "let", "try", "do", "module", "quote"):
                stack.append((t, ln))
            elif depth == 0 and t == "end":
                assert stack, ("end without opener", ln)
                stack.pop()
    assert not stack, stack

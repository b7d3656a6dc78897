"""TEST INFRASTRUCTURE: section 3 of julia/PCleanHIP.jl (`lower`: PCleanModel + Query + data -> the static plan IR) walked in
Python — same functions, same names, same order of steps, on the reference's own model structures (tests/julia_refmodel.py
builds them the way src/dsl/builder.jl does).  There is no Julia in this image; what CAN be pinned is the algorithm the Julia
text states: tests/test_julia_lowering.py runs this transliteration on the three experiment programs and holds its arrays
against tests/golden/plans_*.json, the output of the product's own lowering (pclean_amd/model.py works on the DSL's names and
dotted paths, this one on vertex ids, argument nodes and SubmodelNode copies: two routes to the same plans).

Keep the two texts in step: a change to the Julia lowering is made here too, function by function (the Julia file cites this
one).  Vertex ids are 1-based as in Julia; table / column / node / option indices are 0-based as the C ABI wants them."""
import math

from julia_refmodel import (ForeignKeyNode, JuliaNode, KeyProbe, KeyRef, ParameterNode, RandomChoiceNode, SubmodelNode,
                            strip_subnodes)
from pclean_amd import model as M

NODE_FK, NODE_LEAF = 0, 1
DENS_ADD_TYPOS, DENS_EQUAL, DENS_MAYBE_SWAP = 0, 1, 2
MAX_CTX = 4
DUMMY_STRING_PRIOR, DUMMY_TIME_PRIOR = 1, 2
GSRC_CAND, GSRC_OBS, GSRC_LOCAL, GSRC_EVCTX = "cand", "obs", "local", "evctx"


class Pool:
    def __init__(self):
        self.index, self.strings = {}, []

    def id(self, s):
        if s not in self.index:
            self.index[s] = len(self.strings)
            self.strings.append(s)
        return self.index[s]


class Domain:  # value index (0-based) <-> pool id; extras (strings drawn for chosen dummies) are never options
    def __init__(self):
        self.ids, self.pos, self.n_base = [], {}, 0

    def add(self, pool, s, extra=False):
        pid = pool.id(s)
        if not extra and pid in self.pos:
            return self.pos[pid]
        self.ids.append(pid)
        j = len(self.ids) - 1
        if not extra:
            self.pos[pid] = j
            self.n_base = len(self.ids)
        return j

    def value_of(self, pool, s):
        return self.pos[pool.index[s]]

    def string(self, pool, j):
        return pool.strings[self.ids[j]]

    def __len__(self):
        return len(self.ids)


class LBlock:
    def __init__(self, root_class, root_vertex, group):
        self.root_class, self.root_vertex, self.group = root_class, root_vertex, group
        self.score = False
        self.nodes, self.terms, self.children, self.colmap = [], [], [], []
        self.ctx_block, self.ctx_col = [], []
        self.node_class, self.node_vertex, self.node_path = [], [], []  # per node: class, own vertex of the choice / slot, slot chain below the root


class Lowered:
    def __init__(self, model, query, extra_latent):
        self.model, self.query, self.pool = model, query, Pool()
        self.layout, self.col_of, self.table_id = {}, {}, {}
        self.dom_keys, self.latent_dom = [], {}          # latent domains in creation order = numbering of the option tables
        self.option_id, self.option_values, self.option_keycol, self.option_ncol = {}, {}, {}, {}
        self.keyed = {}                                   # (class, vertex) of a choice whose atoms depend on a key => key vertex
        self.obs_dom, self.obs_col, self.obs_vertices = {}, {}, []
        self.direct_obs = {}                              # noise-free observations: observed vertex => (class, own vertex) it shows
        self.numeric_obs, self.num_cols = {}, []
        self.never_missing = set()
        self.pair_keys, self.pair_id = [], {}             # (observed vertex, latent key) => (id, observed domain, latent pool ids)
        self.eq_pairs, self.same_pairs = {}, {}
        self.next_pair = 0
        self.fn_tables = []
        self.blocks, self.score_blocks, self.prob_spec = [], {}, None
        self.gauss, self.locals, self.gauss_spec = {}, {}, None
        self.cross_terms = []
        self.plan_keys, self.latent_plans = [], {}
        self.latent_ev_locals, self.latent_ev_prob = {}, {}
        self.extra_latent = extra_latent


# ---- helpers over the reference's structures (PCleanHIP.jl, same names) -------------------------------------------------
def value_vertices(cm):
    return [v for v, n in enumerate(cm.nodes, start=1) if isinstance(strip_subnodes(n), (RandomChoiceNode, ForeignKeyNode))]


def const_args(cm, node):
    return [cm.node(a).f() for a in node.arg_node_ids if isinstance(cm.node(a), JuliaNode) and not cm.node(a).arg_node_ids]


def resolve(model, cls, v):
    """the class and own vertex a (possibly nested) SubmodelNode vertex v of class cls stands for"""
    n = model.classes[cls].node(v)
    while isinstance(n, SubmodelNode):
        fk = strip_subnodes(model.classes[cls].node(n.foreign_key_node_id))
        v, cls = n.subnode_id, fk.target_class
        n = model.classes[cls].node(v)
    return cls, v


def slot_of_vertex(ocm, v):
    """the reference slot of the observed class whose (possibly nested) copy vertex v is: add_foreign_key! files EVERY node of
    the target class — its own nested copies included — under the one slot (builder.jl:140-150)"""
    n = ocm.node(v)
    return n.foreign_key_node_id if isinstance(n, SubmodelNode) else v


def path_below(model, ocm, fk, v):
    """vertex ids from slot fk down to the value copy vertex v holds: the chain of nested slot vertices (each in ITS class),
    then the value's own vertex in the class that declares it"""
    cls = ocm.node(fk).target_class
    u = ocm.node(v).subnode_id
    chain = []
    while True:
        n = model.classes[cls].node(u)
        if not isinstance(n, SubmodelNode):
            chain.append(u)
            return chain
        chain.append(n.foreign_key_node_id)
        cls = model.classes[cls].node(n.foreign_key_node_id).target_class
        u = n.subnode_id


def flat_vertex(model, cls, path):
    """vertex of class cls that holds the value reached by following path (nested slot vertices, then the own vertex)"""
    if len(path) == 1:
        return path[0]
    fk = model.classes[cls].node(path[0])
    return fk.vmap[flat_vertex(model, fk.target_class, path[1:])]


def leaf_args(ocm, v):
    """slot-copy vertices an observed-class vertex ultimately depends on (through AddTypos / JuliaNode arguments)"""
    n = ocm.node(v)
    if isinstance(n, SubmodelNode):
        return [v]
    if not isinstance(n, (RandomChoiceNode, JuliaNode)):
        return []
    return [x for a in n.arg_node_ids for x in leaf_args(ocm, a)]


def keyed_atoms_node(cm, n):
    """the argument node of a StringPrior / TimePrior choice that computes its atoms from ANOTHER vertex of the class
    (`possibilities[countykey]`, `times_for_flight["$flight_id-..."]`), or None when the atoms are a constant"""
    a = cm.node(n.arg_node_ids[2 if isinstance(n.dist, M.StringPrior) else 0])
    return a if isinstance(a, JuliaNode) and a.arg_node_ids else None


def dummy_value(n, cm):
    if isinstance(n.dist, M.TimePrior):
        return "**:** p.m."                                  # time_prior.jl:17-19
    lo, hi = const_args(cm, n)[:2]
    return "*" * ((lo + hi) // 2)                            # string_prior.jl:24-26


def atoms_of_key(anode, k):
    try:
        return list(anode.f(k))
    except (KeyError, IndexError):                            # a key value nothing is listed under (the key's own dummy)
        return None


# ---- domains ----------------------------------------------------------------------------------------------------------------
def new_domain(lw, key):
    lw.latent_dom[key] = Domain()
    lw.dom_keys.append(key)
    return lw.latent_dom[key]


def build_domains(lw, data):
    m = lw.model
    for cls in m.class_order:
        cm = m.classes[cls]
        for v, n in enumerate(cm.nodes, start=1):
            if not isinstance(n, RandomChoiceNode):
                continue
            d = n.dist
            if isinstance(d, (M.StringPrior, M.TimePrior)):
                dom = new_domain(lw, (cls, v))
                anode = keyed_atoms_node(cm, n)
                if anode is not None:
                    lw.keyed[(cls, v)] = anode.arg_node_ids[0]      # filled below, once the key's domain exists
                else:
                    for s in const_args(cm, n)[2]:
                        dom.add(lw.pool, s)
                    dom.add(lw.pool, dummy_value(n, cm))
            elif isinstance(d, M.ChooseProportionally) or (isinstance(d, M.ChooseUniformly) and
                                                            all(isinstance(o, str) for o in const_args(cm, n)[0])):
                dom = new_domain(lw, (cls, v))
                for s in const_args(cm, n)[0]:
                    dom.add(lw.pool, s)
    ocm = m.classes[lw.query.cls]
    for col in lw.query.columns:
        v = lw.query.obsmap[col]
        n = ocm.node(v)
        vals = [x for x in data[col] if x is not None]
        if isinstance(n, RandomChoiceNode) and isinstance(n.dist, (M.AddTypos, M.MaybeSwap)):
            dom = Domain()
            for x in vals:
                dom.add(lw.pool, x)
            lw.obs_dom[v] = dom
        elif isinstance(n, RandomChoiceNode) and isinstance(n.dist, M.TransformedGaussian):
            lw.numeric_obs[v] = len(lw.num_cols)
            lw.num_cols.append(col)
            continue
        else:  # a latent value (or an own discrete choice) observed without noise: the observed domain IS the latent domain
            key = (lw.query.cls, v) if isinstance(n, RandomChoiceNode) else resolve(m, lw.query.cls, v)
            if key not in lw.latent_dom:                         # Unmodeled: its values are whatever is observed
                dom = new_domain(lw, key)
                for x in vals:
                    dom.add(lw.pool, x)
            lw.obs_dom[v] = lw.latent_dom[key]
            lw.direct_obs[v] = key
        lw.obs_col[v] = len(lw.obs_vertices)
        lw.obs_vertices.append(v)
        if all(x is not None for x in data[col]):
            lw.never_missing.add(v)
    for (cls, v), kv in lw.keyed.items():                        # keyed atoms: every key's atoms in key order, the dummy last
        cm = m.classes[cls]
        n = cm.node(v)
        dom, kdom, anode = lw.latent_dom[(cls, v)], lw.latent_dom[(cls, kv)], keyed_atoms_node(cm, n)
        for j in range(len(kdom)):
            atoms = atoms_of_key(anode, kdom.string(lw.pool, j))
            for s in atoms or []:
                dom.add(lw.pool, s)
        dom.add(lw.pool, dummy_value(n, cm))
    for key in lw.dom_keys:
        for s in lw.extra_latent.get(key, []):
            lw.latent_dom[key].add(lw.pool, s, extra=True)


def build_layouts(lw):
    m = lw.model
    nxt = 0
    for cls in m.class_order:
        if cls == lw.query.cls:
            continue
        lw.layout[cls] = value_vertices(m.classes[cls])
        lw.col_of[cls] = {v: j for j, v in enumerate(lw.layout[cls])}
        lw.table_id[cls] = nxt
        nxt += 1
    for key in lw.dom_keys:
        dom = lw.latent_dom[key]
        lw.option_id[key] = nxt
        nxt += 1
        lw.option_values[key] = list(range(dom.n_base))          # option k of discrete_proposal = value k; the dummy last
        if key in lw.keyed:  # options = for every key: its atoms, then one dummy option; column 1 = the key, column 2 = its atom count
            cls, v = key
            cm = m.classes[cls]
            n = cm.node(v)
            kdom, anode = lw.latent_dom[(cls, lw.keyed[key])], keyed_atoms_node(cm, n)
            dummy = dom.value_of(lw.pool, dummy_value(n, cm))
            vals, keys, ncol = [], [], []
            for j in range(len(kdom)):
                atoms = atoms_of_key(anode, kdom.string(lw.pool, j))
                if atoms is None:
                    continue
                for s in atoms:
                    vals.append(dom.value_of(lw.pool, s))
                    keys.append(j)
                vals.append(dummy)
                keys.append(j)
                ncol.extend([len(atoms)] * (len(atoms) + 1))
            lw.option_values[key], lw.option_keycol[key], lw.option_ncol[key] = vals, keys, ncol
    lw.next_table = nxt


# ---- blocks -----------------------------------------------------------------------------------------------------------------
def pair_for(lw, obs_v, key, lat_ids):
    if (obs_v, key) not in lw.pair_id:
        lw.pair_id[(obs_v, key)] = (lw.next_pair, lw.obs_dom[obs_v], lat_ids)
        lw.pair_keys.append((obs_v, key))
        lw.next_pair += 1
    return lw.pair_id[(obs_v, key)][0]


def eq_pair_for(lw, key):
    """0/1 identity table over a shared domain (the observed value must equal the latent value)"""
    if key not in lw.eq_pairs:
        lw.eq_pairs[key] = (lw.next_pair, len(lw.latent_dom[key]))
        lw.next_pair += 1
    return lw.eq_pairs[key][0]


def cterm(lw, t, cand_col):
    return (lw.obs_col[t["obs"]], cand_col, t["pair"], t["dens"], t["max_typos"], -1 if t["ctx"] is None else t["ctx"][0],
            -1 if t["ctx"] is None else t["ctx"][1], 0)


def block_terms(lw, ocm, bi, fk, names, fk_block, blk):
    """observation terms of one engine block: every AddTypos observation whose latent argument lies below slot fk, then the
    noise-free observations of values below it (equality constraints)"""
    m = lw.model
    terms = []

    def below(v):
        return isinstance(ocm.node(v), SubmodelNode) and slot_of_vertex(ocm, v) == fk

    for v in names:
        n = ocm.node(v)
        if not (isinstance(n, RandomChoiceNode) and isinstance(n.dist, M.AddTypos) and v in lw.obs_col):
            continue
        word = n.arg_node_ids[0]                                   # AddTypos(word[, max_typos]) (add_typos.jl:50)
        mt = int(ocm.node(n.arg_node_ids[1]).f()) if len(n.arg_node_ids) > 1 else -1
        if below(word):                                            # obs ~ AddTypos(slot.path)
            cls, own = resolve(m, lw.query.cls, word)
            pid = pair_for(lw, v, (cls, own), lw.latent_dom[(cls, own)].ids)
            terms.append(dict(obs=v, path=path_below(m, ocm, fk, word), pair=pid, dens=DENS_ADD_TYPOS, max_typos=mt, ctx=None))
            continue
        j = ocm.node(word)                                         # obs ~ AddTypos(f(args...)): a JuliaNode
        assert isinstance(j, JuliaNode)
        local = [a for a in j.arg_node_ids if below(a)]
        others = [a for a in j.arg_node_ids if not below(a)]
        if len(local) != 1 or len(others) > 1:
            raise NotImplementedError("JuliaNode under AddTypos: one value of this slot, at most one of an earlier slot")
        lc, lown = resolve(m, lw.query.cls, local[0])
        ldom = lw.latent_dom[(lc, lown)]
        if not others:                                             # f(value): a pair table over the strings f(v)
            ids = [lw.pool.id(str(j.f(lw.pool.strings[pid]))) for pid in ldom.ids]
            pid = pair_for(lw, v, ("julia", word), ids)
            terms.append(dict(obs=v, path=path_below(m, ocm, fk, local[0]), pair=pid, dens=DENS_ADD_TYPOS, max_typos=mt, ctx=None))
            continue
        oc, oown = resolve(m, lw.query.cls, others[0])             # f(an earlier slot's value, value): context + fn table
        odom = lw.latent_dom[(oc, oown)]
        ofk = slot_of_vertex(ocm, others[0])
        sb = fk_block[ofk]
        if sb >= bi:
            raise NotImplementedError("context must come from an earlier slot")
        src = (sb, lw.col_of[lw.blocks[sb].root_class][ocm.node(others[0]).subnode_id])
        have = list(zip(blk.ctx_block, blk.ctx_col))
        if src in have:                                            # two JuliaNodes reading the same earlier value share its slot
            slot = have.index(src)
        else:
            slot = len(have)
            if slot >= MAX_CTX:
                raise NotImplementedError("more than MAX_CTX context values in one block")
            blk.ctx_block.append(src[0])
            blk.ctx_col.append(src[1])
        order = (j.arg_node_ids.index(others[0]), j.arg_node_ids.index(local[0]))
        jdom = Domain()
        fn = [[0] * len(ldom) for _ in range(len(odom))]           # fn[other][local]
        for x in range(len(odom)):
            for y in range(len(ldom)):
                argv = [None, None]
                argv[order[0]] = odom.string(lw.pool, x)
                argv[order[1]] = ldom.string(lw.pool, y)
                fn[x][y] = jdom.add(lw.pool, str(j.f(*argv)))
        lw.fn_tables.append(fn)
        fid = len(lw.fn_tables) - 1
        pid = pair_for(lw, v, ("julia", word), jdom.ids)
        lpath = path_below(m, ocm, fk, local[0])
        terms.append(dict(obs=v, path=lpath, pair=pid, dens=DENS_ADD_TYPOS, max_typos=mt, ctx=(slot, fid)))
        # the same observation also constrains the OTHER argument's class (external likelihood of e.g. County.state through
        # Record.stateavg_obs): the latent plans of the classes below the earlier slot pick it up (copy_subtree)
        lw.cross_terms.append(dict(obs=v, pair=pid, max_typos=mt, fn=fid, ctx_block=sb, ctx_path=path_below(m, ocm, ofk, others[0]),
                                   local_block=bi, local_path=lpath))
    for v in lw.obs_vertices:                                      # noise-free observations of values below the root slot
        if v in lw.direct_obs and below(v):
            terms.append(dict(obs=v, path=path_below(m, ocm, fk, v), pair=eq_pair_for(lw, lw.direct_obs[v]), dens=DENS_EQUAL,
                              max_typos=-1, ctx=None))
    return terms


def emit_fk_node(lw, blk, cls, slot_vertex, prefix, terms, parent, parent_fk_col):
    """the node of class cls (a reference slot) with the terms whose value lives in its sub-tree; returns its node id"""
    cm = lw.model.classes[cls]
    nid = len(blk.nodes)
    blk.nodes.append(None)
    blk.node_class.append(cls)
    blk.node_vertex.append(slot_vertex)
    blk.node_path.append(list(prefix))
    tb = len(blk.terms)
    for t in terms:                                                # candidate column = the flattened column of the value
        blk.terms.append(cterm(lw, t, lw.col_of[cls][flat_vertex(lw.model, cls, t["path"])]))
    nt = len(blk.terms) - tb
    kids, colsrc = [], {}
    for v, n in enumerate(cm.nodes, start=1):                      # own attributes in declaration (vertex) order
        if isinstance(n, ForeignKeyNode):
            sub = [dict(t, path=t["path"][1:]) for t in terms if len(t["path"]) > 1 and t["path"][0] == v]
            cid = emit_fk_node(lw, blk, n.target_class, v, prefix + [v], sub, nid, lw.col_of[cls][v])
            kids.append(cid)
            colsrc[v] = (-1, -1)
            for i, vv in n.vmap.items():                           # flattened copies come from the child's columns
                if vv in lw.col_of[cls] and i in lw.col_of[n.target_class]:
                    colsrc[vv] = (cid, lw.col_of[n.target_class][i])
        elif isinstance(n, RandomChoiceNode) and (cls, v) in lw.latent_dom:
            sub = [t for t in terms if t["path"] == [v]]
            cid = len(blk.nodes)
            ltb = len(blk.terms)
            for t in sub:
                blk.terms.append(cterm(lw, t, 0))
            n_leaf_terms = len(sub)
            if (cls, v) in lw.keyed:  # atoms belong to the key they were listed under: the option's key must equal the row's
                kt = [t for t in terms if t["path"] == [lw.keyed[(cls, v)]] and t["dens"] == DENS_EQUAL]
                if len(kt) != 1:
                    raise NotImplementedError("keyed atoms need their key attribute observed directly")
                blk.terms.append(cterm(lw, kt[0], 1))
                n_leaf_terms += 1
            cacheable = int(n_leaf_terms == 1 and len(sub) == 1 and sub[0]["ctx"] is None)
            dval, dspec = 0, 0
            if isinstance(n.dist, (M.StringPrior, M.TimePrior)):
                dval = lw.latent_dom[(cls, v)].value_of(lw.pool, dummy_value(n, cm)) + 1
                if isinstance(n.dist, M.TimePrior):
                    dspec = DUMMY_TIME_PRIOR
                else:
                    lo, hi = const_args(cm, n)[:2]
                    dspec = DUMMY_STRING_PRIOR | (lo << 8) | (hi << 16)
            blk.nodes.append((NODE_LEAF, lw.option_id[(cls, v)], ltb, n_leaf_terms, 0, 0, nid, -1, cacheable, 0, dval, dspec))
            blk.node_class.append(cls)
            blk.node_vertex.append(v)
            blk.node_path.append(list(prefix) + [v])
            kids.append(cid)
            colsrc[v] = (cid, 0)
    cb = len(blk.children)
    blk.children.extend(kids)
    cmb = len(blk.colmap) // 2
    for v in lw.layout[cls]:
        blk.colmap.extend(colsrc.get(v, (-1, -1)))
    blk.nodes[nid] = (NODE_FK, lw.table_id[cls], tb, nt, cb, len(kids), parent, parent_fk_col, 0, cmb, 0, 0)
    return nid


def value_source(lw, ocm, fk_block, v):
    """(engine block, column of that block's root table) of the latent value a slot-copy vertex v of the observed class holds"""
    fk = slot_of_vertex(ocm, v)
    return fk_block[fk], lw.col_of[ocm.node(fk).target_class][ocm.node(v).subnode_id]


def lower_score_block(lw, bi, ocm, names, fk_block):
    """a block without a reference slot: MaybeSwap observations of values chosen in earlier blocks (flights: block 3)"""
    m = lw.model
    terms, prob_spec = [], None
    for v in names:
        n = ocm.node(v)
        if not isinstance(n, RandomChoiceNode):
            continue
        if not isinstance(n.dist, M.MaybeSwap):
            raise NotImplementedError("a block without a reference slot may only hold MaybeSwap observations")
        val_v, opt_v, prob_v = n.arg_node_ids                      # MaybeSwap(val, options, prob) (maybe_swap.jl:13)
        onode = ocm.node(opt_v)                                    # options = f(key): a JuliaNode of one slot-copy vertex
        key_v = onode.arg_node_ids[0]
        vcls, vown = resolve(m, lw.query.cls, val_v)
        kcls, kown = resolve(m, lw.query.cls, key_v)
        vdom, kdom = lw.latent_dom[(vcls, vown)], lw.latent_dom[(kcls, kown)]
        pid = lw.next_pair                                         # 0/1 "same string" table: observed values x latent domain
        lw.next_pair += 1
        lw.same_pairs[pid] = (lw.obs_dom[v], vdom)
        nopt = [[1] for _ in range(len(kdom))]                     # number of options under every key (MaybeSwap's length(options))
        for j in range(len(kdom)):
            opts = atoms_of_key(onode, kdom.string(lw.pool, j))
            if opts is not None:
                nopt[j][0] = len(opts)
        lw.fn_tables.append(nopt)
        fid = len(lw.fn_tables) - 1
        vn = m.classes[vcls].node(vown)
        terms.append(dict(obs=lw.obs_col[v], pair=pid, val=value_source(lw, ocm, fk_block, val_v),
                          key=value_source(lw, ocm, fk_block, key_v), nopt_fn=fid,
                          other=vdom.value_of(lw.pool, dummy_value(vn, m.classes[vcls])), vertex=v, val_vertex=val_v))
        if prob_spec is None:                                      # the error probability: a JuliaNode of two values and the parameter
            pn = ocm.node(prob_v)
            vals = [a for a in pn.arg_node_ids if not isinstance(ocm.node(a), ParameterNode)]
            par = [a for a in pn.arg_node_ids if isinstance(ocm.node(a), ParameterNode)]
            if not (isinstance(pn, JuliaNode) and len(vals) == 2 and len(par) == 1):
                raise NotImplementedError("MaybeSwap's probability: a JuliaNode of two latent values and an indexed parameter")
            (acls, aown), (bcls, bown) = resolve(m, lw.query.cls, vals[0]), resolve(m, lw.query.cls, vals[1])
            adom, bdom = lw.latent_dom[(acls, aown)], lw.latent_dom[(bcls, bown)]
            keys, consts = [], []
            pf = [[0] * len(bdom) for _ in range(len(adom))]
            for x in range(len(adom)):
                for y in range(len(bdom)):
                    argv = {vals[0]: adom.string(lw.pool, x), vals[1]: bdom.string(lw.pool, y), par[0]: KeyProbe()}
                    r = pn.f(*[argv[a] for a in pn.arg_node_ids])
                    if isinstance(r, KeyRef):
                        if r.key not in keys:
                            keys.append(r.key)
                        pf[x][y] = -1 - keys.index(r.key)
                    else:
                        if r not in consts:
                            consts.append(r)
                        pf[x][y] = consts.index(r)
            pf = [[len(consts) + (-1 - e) if e < 0 else e for e in row] for row in pf]   # constants first, then one entry per key
            lw.fn_tables.append(pf)
            prob_spec = dict(fn=len(lw.fn_tables) - 1, a=value_source(lw, ocm, fk_block, vals[0]),
                             b=value_source(lw, ocm, fk_block, vals[1]), consts=consts, keys=keys, param=par[0])
    lw.score_blocks[bi] = dict(terms=terms, prob=prob_spec)
    lw.prob_spec = prob_spec


def lower_gaussian(lw, bi, blk, ocm, names, fk):
    """x ~ TransformedGaussian(param[f(root values, own choices)], std, unit) with own ChooseUniformly choices
    (experiments/rents/run.jl:19-25) -> the pclean_gauss specs of the block's root and of the open leaf of its new-row branch"""
    m = lw.model
    ga = [v for v in names if isinstance(ocm.node(v), RandomChoiceNode) and isinstance(ocm.node(v).dist, M.TransformedGaussian)]
    if not ga:
        return
    if len(ga) > 1:
        raise NotImplementedError("one Gaussian observation per block")
    g = ocm.node(ga[0])
    mean_v, std_v, unit_v = g.arg_node_ids                         # TransformedGaussian(mean, std, t) (transformed_gaussian.jl:11)
    look = ocm.node(mean_v)
    par = [a for a in look.arg_node_ids if isinstance(ocm.node(a), ParameterNode)]
    if not (isinstance(look, JuliaNode) and len(par) == 1):
        raise NotImplementedError("TransformedGaussian's mean: a JuliaNode indexing ONE learned parameter")
    units = const_args(ocm, ocm.node(unit_v))[0]
    t_scale, t_lad = [], []
    for u in units:  # the kernels evaluate backward(x) as x * backward(1) and log|g'| as a constant: linear units only
        b1 = float(u.backward(1.0))
        d1 = float(u.deriv(b1))
        lin = abs(float(u.backward(0.0))) <= 1e-12 and all(
            abs(float(u.backward(x)) - x * b1) <= 1e-9 * max(1.0, abs(x * b1)) and
            abs(float(u.deriv(u.backward(x))) - d1) <= 1e-9 * max(1.0, abs(d1)) for x in (0.5, 2.0, -3.0, 1267.0))
        if not lin:
            raise NotImplementedError("TransformedGaussian: only linear Transformations (backward(x) = c x)")
        t_scale.append(b1)
        t_lad.append(math.log(abs(d1)))
    locs, dims = [], []                                             # own enumerated choices; (kind, payload, n values) per index argument
    for a in look.arg_node_ids:
        if a in par:
            continue
        n = ocm.node(a)
        if isinstance(n, SubmodelNode):
            if slot_of_vertex(ocm, a) != fk:
                raise NotImplementedError("index values come from the block's own slot")
            cn, own = resolve(m, lw.query.cls, a)
            dims.append(("cand", a, len(lw.latent_dom[(cn, own)])))
        else:
            if not (isinstance(n, RandomChoiceNode) and isinstance(n.dist, M.ChooseUniformly)):
                raise NotImplementedError("own index arguments must be ChooseUniformly choices")
            if a not in locs:
                locs.append(a)
            dims.append(("local", locs.index(a), len(const_args(ocm, n)[0])))
    if unit_v not in locs:
        locs.append(unit_v)
    if len(locs) > 2:
        raise NotImplementedError("at most two enumerated own choices")
    strides, acc = [], 1
    for d in reversed(dims):
        strides.insert(0, acc)
        acc *= d[2]
    lw.locals[bi] = locs
    spec = dict(x_col=lw.numeric_obs[ga[0]], param=par[0], n_mean=acc, strides=strides, n_locals=len(locs),
                local_n=[len(const_args(ocm, ocm.node(l))[0]) for l in locs],
                local_obs=[lw.obs_col[l] if l in lw.direct_obs else -1 for l in locs], t_local=locs.index(unit_v),
                sigma=float(ocm.node(std_v).f()), t_scale=t_scale, t_lad=t_lad)
    lw.gauss_spec = spec
    rc = ocm.node(fk).target_class
    # (a) the block's root: candidate-side index values come from the candidate's columns
    lw.gauss[(bi, 0)] = dict(spec, kinds=[(GSRC_CAND, lw.col_of[rc][ocm.node(d[1]).subnode_id]) if d[0] == "cand" else (GSRC_LOCAL, d[1])
                                          for d in dims], transform=(GSRC_LOCAL, spec["t_local"]))
    # (b) the new-row branch: the leaf of the ONE candidate-side value that is not always observed carries the term, the
    #     others are read from their direct observations
    open_dims = [d for d in dims if d[0] == "cand" and not (d[1] in lw.direct_obs and d[1] in lw.never_missing)]
    if len(open_dims) != 1:
        raise NotImplementedError("exactly one candidate-side index value may be unobserved")
    open_path = path_below(m, ocm, fk, open_dims[0][1])
    for nid in range(len(blk.nodes)):
        if blk.nodes[nid][0] == NODE_LEAF and blk.node_path[nid] == open_path:
            kinds = [(GSRC_LOCAL, d[1]) if d[0] == "local" else (GSRC_CAND, 0) if d is open_dims[0] else (GSRC_OBS, lw.obs_col[d[1]])
                     for d in dims]
            lw.gauss[(bi, nid)] = dict(spec, kinds=kinds, transform=(GSRC_LOCAL, spec["t_local"]))
            node = list(blk.nodes[nid])
            node[8] = 0                                            # not cacheable any more
            blk.nodes[nid] = tuple(node)


def build_blocks(lw):
    ocm = lw.model.classes[lw.query.cls]
    fk_block = {}
    eblocks = []                                                   # (model block, vertices) per ENGINE block
    for ub, names in enumerate(ocm.blocks):
        fks = [v for v in names if isinstance(ocm.node(v), ForeignKeyNode)]
        if len(fks) <= 1:
            eblocks.append((ub, list(names)))
            continue
        # several slots in one block: one engine block per slot, an observation goes to the LAST slot it mentions
        groups = [[f] for f in fks]
        for v in names:
            n = ocm.node(v)
            if isinstance(n, (ForeignKeyNode, SubmodelNode)):
                continue
            hs = [fks.index(slot_of_vertex(ocm, a)) for a in leaf_args(ocm, v) if slot_of_vertex(ocm, a) in fks]
            groups[max(hs) if hs else len(fks) - 1].append(v)
        for g in groups:
            eblocks.append((ub, g))
    lw.block_group = [ub for ub, _ in eblocks]
    for bi, (ub, names) in enumerate(eblocks):
        fks = [v for v in names if isinstance(ocm.node(v), ForeignKeyNode)]
        if fks:
            fk_block[fks[0]] = bi
    for bi, (ub, names) in enumerate(eblocks):
        fks = [v for v in names if isinstance(ocm.node(v), ForeignKeyNode)]
        if not fks:
            lower_score_block(lw, bi, ocm, names, fk_block)
            blk = LBlock(None, None, ub)
            blk.score = True
            lw.blocks.append(blk)
            continue
        fk = fks[0]
        blk = LBlock(ocm.node(fk).target_class, fk, ub)
        lw.blocks.append(blk)
        terms = block_terms(lw, ocm, bi, fk, names, fk_block, blk)
        emit_fk_node(lw, blk, blk.root_class, fk, [], terms, -1, -1)
        lower_gaussian(lw, bi, blk, ocm, names, fk)


# ---- latent-class plans -----------------------------------------------------------------------------------------------------
def build_latent_plans(lw):
    """for every latent class T: the sub-plans of its own attributes, scored against all observed rows that (transitively)
    refer to a row of T — the children of T's node in the observed plan, re-rooted (model.py: _build_latent_plans)"""
    nxt = len(lw.blocks)
    for bi, blk in enumerate(lw.blocks):
        if blk.score:
            continue
        for nid in range(len(blk.nodes)):
            node = blk.nodes[nid]
            cls = blk.node_class[nid]
            if node[0] != NODE_FK or cls in lw.latent_plans:
                continue
            plan = dict(block_id=nxt, src_block=bi, src_node=nid, cls=cls, path=list(blk.node_path[nid]), nodes=[], terms=[],
                        children=[], colmap=[], node_class=[], node_vertex=[], roots=[], root_vertex=[],
                        ctx_sources=list(zip(blk.ctx_block, blk.ctx_col)))
            for k in range(node[4], node[4] + node[5]):
                child = blk.children[k]
                plan["roots"].append(copy_subtree(lw, blk, child, plan, -1, bi))
                plan["root_vertex"].append(blk.node_vertex[child])
            lw.latent_plans[cls] = plan
            lw.plan_keys.append(cls)
            nxt += 1


def copy_subtree(lw, blk, nid, plan, parent, bi):
    node = blk.nodes[nid]
    new_id = len(plan["nodes"])
    plan["nodes"].append(None)
    plan["node_class"].append(blk.node_class[nid])
    plan["node_vertex"].append(blk.node_vertex[nid])
    tb = len(plan["terms"])
    for t in blk.terms[node[2]:node[2] + node[3]]:
        t = list(t)
        if t[5] >= 0:
            t[7] = 1                                               # the context now comes from the evidence row (fn[ctx][candidate])
        plan["terms"].append(tuple(t))
    # cross-block JuliaNode observations whose OTHER argument lives in this sub-tree: fn[candidate][ctx of the evidence row]
    p = blk.node_path[nid]
    for ct in lw.cross_terms:
        if ct["ctx_block"] != bi:
            continue
        q = ct["ctx_path"]
        if node[0] == NODE_LEAF and p == q:
            col = 0
        elif node[0] == NODE_FK and len(q) > len(p) and q[:len(p)] == p:
            col = lw.col_of[blk.node_class[nid]][flat_vertex(lw.model, blk.node_class[nid], q[len(p):])]
        else:
            continue
        lb = ct["local_block"]
        src = (lb, lw.col_of[lw.blocks[lb].root_class][flat_vertex(lw.model, lw.blocks[lb].root_class, ct["local_path"])])
        if src not in plan["ctx_sources"]:
            if len(plan["ctx_sources"]) >= MAX_CTX:
                raise NotImplementedError("more than MAX_CTX per-evidence-row context values")
            plan["ctx_sources"].append(src)
        plan["terms"].append((lw.obs_col[ct["obs"]], col, ct["pair"], DENS_ADD_TYPOS, ct["max_typos"], plan["ctx_sources"].index(src),
                              ct["fn"], 2))
    # MaybeSwap observations (scoring blocks) of this value: the external likelihood of the referring rows, each with its own
    # error probability (evidence ctx slot 0 = index into the prob table)
    ocm = lw.model.classes[lw.query.cls]
    for sbi in sorted(lw.score_blocks):
        for t in lw.score_blocks[sbi]["terms"]:
            if node[0] == NODE_LEAF and t["val"][0] == bi and path_below(lw.model, ocm, lw.blocks[bi].root_vertex, t["val_vertex"]) == p:
                plan["terms"].append((t["obs"], 0, t["pair"], DENS_MAYBE_SWAP, 2, 0, t["other"], 1))
                lw.latent_ev_prob[plan["cls"]] = sbi
    nt = len(plan["terms"]) - tb
    if (bi, nid) in lw.gauss and node[0] == NODE_LEAF:
        # the latent sweep of the class owning this value: the referring rows' Gaussian observations, their own choices held at
        # their current values (evidence ctx)
        src = lw.gauss[(bi, nid)]
        lw.gauss[(plan["block_id"], new_id)] = dict(src, kinds=[(GSRC_EVCTX, k[1]) if k[0] == GSRC_LOCAL else k for k in src["kinds"]],
                                                    n_locals=0, transform=(GSRC_EVCTX, src["t_local"]))
        lw.latent_ev_locals[plan["cls"]] = bi
    if node[0] == NODE_FK:
        remap, kids = {}, []
        for k in range(node[4], node[4] + node[5]):
            c = blk.children[k]
            remap[c] = copy_subtree(lw, blk, c, plan, new_id, bi)
            kids.append(remap[c])
        cb = len(plan["children"])
        plan["children"].extend(kids)
        cmb = len(plan["colmap"]) // 2
        for j in range(len(lw.layout[blk.node_class[nid]])):
            cn, cc = blk.colmap[2 * (node[9] + j)], blk.colmap[2 * (node[9] + j) + 1]
            plan["colmap"].extend((remap[cn] if cn >= 0 else -1, cc))
        plan["nodes"][new_id] = (node[0], node[1], tb, nt, cb, len(kids), parent, node[7], 0, cmb, 0, 0)
    else:
        plan["nodes"][new_id] = (node[0], node[1], tb, nt, 0, 0, parent, -1, 0, 0, 0, 0)
    return new_id


def lower(model, query, data, extra_latent=None):
    """PCleanModel + Query + data -> Lowered (PCleanHIP.jl: lower; model.py: LoweredModel.__init__)"""
    lw = Lowered(model, query, extra_latent or {})
    build_domains(lw, data)
    build_layouts(lw)
    build_blocks(lw)
    build_latent_plans(lw)
    return lw


def score_block_args(lw, bi):
    """arguments of pclean_load_score_block for scoring block bi"""
    sb = lw.score_blocks[bi]
    t, pr = sb["terms"], sb["prob"]
    return ([x["obs"] for x in t], [x["pair"] for x in t], [c for x in t for c in x["val"]], [c for x in t for c in x["key"]],
            [x["nopt_fn"] for x in t], [x["other"] for x in t], pr["fn"], list(pr["a"]), list(pr["b"]))

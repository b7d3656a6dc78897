"""TEST INFRASTRUCTURE: the data structures of the reference's compiled model (src/model/model.jl:78-180) built from the
Python DSL mirror (pclean_amd/model.py) the way the reference's builder builds them from `@model` (src/dsl/builder.jl +
src/dsl/syntax.jl) — vertex numbering, argument nodes, reference-slot copies, blocks — so that the lowering of
julia/PCleanHIP.jl, which walks THOSE structures, can be transliterated into Python (tests/julia_lowering.py) and held
against the plan goldens without a Julia toolchain.

What it restates, statement by statement (syntax.jl:106-161 -> builder.jl):
  `x ~ Dist(args...)`    add_choice_node! (234-258): every argument that is not a plain name becomes a JuliaNode of its own
                         FIRST (resolve_argument!, 88-101: literals and compound expressions: `(names, f)`), then the choice;
  `x = expr`             add_julia_node! (205-231) over the PClean names the expression uses, in order of first appearance
                         (parse_compound_expression, syntax.jl:37-63 — a `@learned` name is such a name);
  `x ~ Class`            add_foreign_key! (123-175): the slot's vertex v, then a SubmodelNode copy of EVERY node of the target
                         class at v + i (arguments shifted by v: copy_node, 112-118), vmap = {i => v + i};
  `@learned`             add_basic_parameter! / add_indexed_parameter! (182-202): a ParameterNode, in no block;
  `begin ... end`        begin_block! / end_block! (13-20); a statement after a closed block opens a new one that stays open.
ExternalLikelihoodNodes (finish_class!) are appended BEHIND a class's own vertices and are skipped by everything the lowering
does (PCleanHIP.jl: value_vertices), so they are not generated here.

1-based vertex ids throughout (Julia's)."""


class JuliaNode:
    def __init__(self, f, arg_node_ids):
        self.f, self.arg_node_ids = f, list(arg_node_ids)


class RandomChoiceNode:
    def __init__(self, dist, arg_node_ids):
        self.dist, self.arg_node_ids = dist, list(arg_node_ids)


class ParameterNode:
    def __init__(self, prior):
        self.prior = prior


class ForeignKeyNode:
    def __init__(self, target_class, vmap):
        self.target_class, self.vmap = target_class, dict(vmap)


class SubmodelNode:
    def __init__(self, foreign_key_node_id, subnode_id, subnode):
        self.foreign_key_node_id, self.subnode_id, self.subnode = foreign_key_node_id, subnode_id, subnode


def strip_subnodes(n):  # model.jl: strip_subnodes
    while isinstance(n, SubmodelNode):
        n = n.subnode
    return n


def copy_node(n, v):  # builder.jl:112-118
    if isinstance(n, JuliaNode):
        return JuliaNode(n.f, [x + v for x in n.arg_node_ids])
    if isinstance(n, RandomChoiceNode):
        return RandomChoiceNode(n.dist, [x + v for x in n.arg_node_ids])
    if isinstance(n, ParameterNode):
        return n
    if isinstance(n, ForeignKeyNode):
        return ForeignKeyNode(n.target_class, {i: j + v for i, j in n.vmap.items()})
    return SubmodelNode(n.foreign_key_node_id + v, n.subnode_id, copy_node(n.subnode, v))


class PCleanClass:
    def __init__(self):
        self.nodes = []       # nodes[v - 1] = node of vertex v
        self.blocks = []      # lists of vertex ids
        self.names = {}       # declared name -> vertex id
        self.hash_keys = []

    def node(self, v):
        return self.nodes[v - 1]


class PCleanModel:
    def __init__(self):
        self.classes, self.class_order = {}, []


class Query:
    """dsl/query.jl:1-13: column -> vertex of the observed class (clean / dirty); `columns` keeps the @query's order, which
    the reference's Dict forgets (PCleanHIP.lower takes it as an argument)."""

    def __init__(self, model, cls, cleanmap, obsmap, columns):
        self.model, self.cls, self.cleanmap, self.obsmap, self.columns = model, cls, cleanmap, obsmap, columns


class KeyRef:
    """what an indexed `@learned` parameter hands back to the lowering's probe: the key that was asked for"""

    def __init__(self, key):
        self.key = key


class KeyProbe:
    def __getitem__(self, key):
        return KeyRef(key)


class Builder:
    def __init__(self):
        self.model = PCleanModel()
        self.open = False  # block_status

    def resolve_dot(self, cls, path):  # resolve_dot_expression (60-75)
        cm = self.model.classes[cls]
        if "." not in path:
            return cm.names[path]
        head, rest = path.split(".", 1)
        fk = cm.node(cm.names[head])
        return fk.vmap[self.resolve_dot(fk.target_class, rest)]

    def _place(self, cm, v):
        if self.open:
            cm.blocks[-1].append(v)
        else:
            cm.blocks.append([v])
            self.open = True

    def add_julia_node(self, cls, name, arg_ids, f):  # 205-231
        cm = self.model.classes[cls]
        cm.nodes.append(JuliaNode(f, arg_ids))
        v = len(cm.nodes)
        if name is not None:
            cm.names[name] = v
        self._place(cm, v)
        return v

    def const_arg(self, cls, value):  # a literal / outside constant: `(Symbol[], () -> value)` (resolve_argument!, 96-100)
        return self.add_julia_node(cls, None, [], lambda value=value: value)

    def add_choice_node(self, cls, name, dist, arg_ids):  # 234-258
        cm = self.model.classes[cls]
        cm.nodes.append(RandomChoiceNode(dist, arg_ids))
        v = len(cm.nodes)
        cm.names[name] = v
        self._place(cm, v)
        return v

    def add_parameter(self, cls, name, prior):  # 182-202
        cm = self.model.classes[cls]
        cm.nodes.append(ParameterNode(prior))
        cm.names[name] = len(cm.nodes)

    def add_foreign_key(self, cls, name, target):  # 123-175
        cm, tm = self.model.classes[cls], self.model.classes[target]
        tnodes = list(tm.nodes)  # (no ExternalLikelihoodNodes here)
        v = len(cm.nodes) + 1
        cm.names[name] = v
        cm.nodes.append(ForeignKeyNode(target, {i: i + v for i in range(1, len(tnodes) + 1)}))
        for i, node in enumerate(tnodes, start=1):
            cm.nodes.append(SubmodelNode(v, i, copy_node(node, v)))
        limit = len(cm.nodes)
        sampled = [v] + [x + v for blk in tm.blocks for x in blk if x + v <= limit]
        if self.open:
            cm.blocks[-1].extend(sampled)
        else:
            cm.blocks.append(sampled)
            self.open = True


def build_reference_model(pymodel, pyquery):
    """pclean_amd.model.Model / Query -> (PCleanModel, Query) as the reference's @model / @query would leave them."""
    from pclean_amd import model as M
    b = Builder()
    for cname in pymodel.class_order:
        pc = pymodel.classes[cname]
        b.model.classes[cname] = PCleanClass()
        b.model.class_order.append(cname)
        b.open = False
        first_of_block = {blk[0]: bi for bi, blk in enumerate(pc.blocks) if blk}
        for a in pc.attrs:
            if a.kind == "param":
                b.add_parameter(cname, a.name, a.prior)
                continue
            if a.name in first_of_block:  # `begin`, or the statement after an `end`: the next node opens a block
                b.open = False
            if a.kind == "fk":
                b.add_foreign_key(cname, a.name, a.target)
            elif a.kind == "julia":
                if isinstance(a.fn, M.ProbLookup):  # `cond ? const : param[key]` — names: the values, then the parameter
                    ids = [b.resolve_dot(cname, x) for x in a.args] + [b.model.classes[cname].names[a.fn.param]]

                    def f(*vals, fn=a.fn.fn):
                        r = fn(*vals[:-1])
                        return r if isinstance(r, float) else vals[-1][r]
                    b.add_julia_node(cname, a.name, ids, f)
                elif isinstance(a.fn, M.IndexedLookup):  # `param["$(x)_$(y)..."]` — names: the parameter, then the values
                    ids = [b.model.classes[cname].names[a.fn.param]] + [b.resolve_dot(cname, x) for x in a.args]
                    b.add_julia_node(cname, a.name, ids, lambda p, *vals: p[tuple(vals)])
                else:
                    b.add_julia_node(cname, a.name, [b.resolve_dot(cname, x) for x in a.args], a.fn)
            else:
                d = a.dist
                if isinstance(d, M.StringPrior):
                    ids = [b.const_arg(cname, d.min_len), b.const_arg(cname, d.max_len)]
                    if d.keyed_by:  # `possibilities[key]`: a JuliaNode of the key
                        ids.append(b.add_julia_node(cname, None, [b.resolve_dot(cname, d.keyed_by)], lambda k, atoms=d.atoms: atoms[k]))
                    else:
                        ids.append(b.const_arg(cname, list(d.atoms)))
                elif isinstance(d, M.TimePrior):
                    ids = [b.add_julia_node(cname, None, [b.resolve_dot(cname, d.keyed_by)], lambda k, atoms=d.atoms: atoms[k])]
                elif isinstance(d, M.ChooseUniformly):
                    ids = [b.const_arg(cname, list(d.options))]
                elif isinstance(d, M.ChooseProportionally):
                    ids = [b.const_arg(cname, list(d.options)), b.model.classes[cname].names[d.param]]
                elif isinstance(d, M.AddTypos):
                    ids = [b.resolve_dot(cname, d.ref)]
                    if d.max_typos is not None:
                        ids.append(b.const_arg(cname, d.max_typos))
                elif isinstance(d, M.MaybeSwap):
                    ids = [b.resolve_dot(cname, d.val),
                           b.add_julia_node(cname, None, [b.resolve_dot(cname, d.key)], lambda k, opts=d.options: opts[k]),
                           b.resolve_dot(cname, d.prob)]
                elif isinstance(d, M.TransformedGaussian):
                    ids = [b.resolve_dot(cname, d.mean), b.const_arg(cname, d.std), b.resolve_dot(cname, d.unit)]
                elif isinstance(d, M.Unmodeled):
                    ids = []
                else:
                    raise NotImplementedError(type(d).__name__)
                b.add_choice_node(cname, a.name, d, ids)
    q = Query(b.model, pyquery.cls, {c: b.resolve_dot(pyquery.cls, r) for c, r in pyquery.cleanmap.items()},
              {c: b.resolve_dot(pyquery.cls, r) for c, r in pyquery.obsmap.items()}, list(pyquery.obsmap))
    return b.model, q

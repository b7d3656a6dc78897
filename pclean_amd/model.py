"""Relational model definition mirroring PClean's `@model` / `@query` DSL, and its
lowering to the static enumeration plan consumed by libpclean_hip.so.

Reference mapping (files under /root/reference/src):
  Model / ClassDef        dsl/syntax.jl:106-161 (`@model`, `@class`), dsl/builder.jl:37-258
  ClassDef.param          `@learned x::ProportionsParameter`   (builder.jl:182-202)
  ClassDef.choice         `x ~ Dist(args...)`                  (builder.jl:234-258)
  ClassDef.fk             `x ~ OtherClass`  (builder.jl:123-175: the target's nodes are
                          inlined, i.e. a row stores a flattened copy of its referents)
  ClassDef.julia          `x = expr`                            (builder.jl:205-231)
  ClassDef.block()        `begin ... end`                       (builder.jl:13-20)
  Query                   dsl/query.jl:1-43
The reference JIT-compiles one Julia function per (class, block, observed set)
(inference/proposal_compiler.jl); here the same enumeration structure is emitted
once as flat arrays (nodes / terms / children / colmap, include/pclean_hip.h).
"""
from contextlib import contextmanager

import numpy as np

from . import _lib
from .encode import Domain, StringPool


# ---------------------------------------------------------------------------
# distributions (src/distributions/*.jl) — declarative descriptors
class StringPrior:
    """string_prior.jl: StringPrior(min_length, max_length, proposal_atoms).  With `keyed_by`
    the atoms are a dict keyed by the value of another attribute of the same class
    (`possibilities[countykey]`, experiments/rents/run.jl:13)."""

    def __init__(self, min_len, max_len, atoms, keyed_by=None):
        self.min_len, self.max_len, self.keyed_by = int(min_len), int(max_len), keyed_by
        self.atoms = {k: list(v) for k, v in atoms.items()} if keyed_by else list(atoms)

    def dummy_value(self):  # string_prior.jl:24-26
        return "*" * ((self.min_len + self.max_len) // 2)


class ChooseUniformly:
    """choose_uniformly.jl: ChooseUniformly(options)."""

    def __init__(self, options):
        self.options = list(options)


class ChooseProportionally:
    """choose_proportionally.jl: ChooseProportionally(options, probs::ProportionsParameter)."""

    def __init__(self, options, param):
        self.options, self.param = list(options), param


class AddTypos:
    """add_typos.jl: AddTypos(word[, max_typos]); `ref` names the clean value."""

    def __init__(self, ref, max_typos=None):
        self.ref, self.max_typos = ref, max_typos


class TimePrior:
    """time_prior.jl: TimePrior(proposal_atoms); atoms keyed by another attribute of the class
    (`times_for_flight["$flight_id-field"]`, experiments/flights/run.jl:17-20)."""
    keyed = True

    def __init__(self, atoms, keyed_by):
        self.atoms = {k: list(v) for k, v in atoms.items()}
        self.keyed_by = keyed_by

    def dummy_value(self):  # time_prior.jl:17-19
        return "**:** p.m."


class MaybeSwap:
    """maybe_swap.jl: MaybeSwap(val, options, prob). `val` references the clean value, `options`
    are keyed by the value of `key` (same dict as the TimePrior atoms), `prob` names a ProbLookup."""

    def __init__(self, val, options, key, prob):
        self.val, self.options, self.key, self.prob = val, {k: list(v) for k, v in options.items()}, key, prob


class IndexedProbParameter:
    """`@learned x::Dict{String, ProbParameter{a, b}}` (maybe_swap.jl:36-52)."""

    def __init__(self, a, b):
        self.a, self.b = float(a), float(b)


class ProbLookup:
    """Deterministic node choosing an error probability: fn(*args) returns a float constant or the
    key of the indexed ProbParameter (experiments/flights/run.jl:28)."""

    def __init__(self, param, fn):
        self.param, self.fn = param, fn


class Unmodeled:
    """unmodeled.jl: logdensity 0, no proposal; must be observed."""


class Transformation:
    """transformed_gaussian.jl:5-9 — forward, backward, |g'|."""

    def __init__(self, forward, backward, deriv):
        self.forward, self.backward, self.deriv = forward, backward, deriv


class TransformedGaussian:
    """transformed_gaussian.jl: TransformedGaussian(mean, std, t); `mean` names an IndexedLookup
    julia attribute, `unit` an own ChooseUniformly attribute over Transformations."""

    def __init__(self, mean, std, unit):
        self.mean, self.std, self.unit = mean, float(std), unit


class IndexedMeanParameter:
    """`@learned x::Dict{String, MeanParameter{mean, std}}` (add_noise.jl:15-45, distributions.jl:45-55)."""

    def __init__(self, mean, std):
        self.mean, self.std = float(mean), float(std)


class IndexedLookup:
    """Deterministic node `param[f(args...)]` (experiments/rents/run.jl:23): the parameter indexed by
    the tuple of its discrete arguments."""

    def __init__(self, param):
        self.param = param


class ProportionsParameter:
    """choose_proportionally.jl:31-74 (Dirichlet prior, default concentration 1.0)."""

    def __init__(self, concentration=1.0):
        self.concentration = float(concentration)


# ---------------------------------------------------------------------------
class Attr:
    def __init__(self, kind, name, **kw):
        self.kind, self.name = kind, name
        self.__dict__.update(kw)


class ClassDef:
    def __init__(self, model, name):
        self.model, self.name = model, name
        self.attrs = []        # in declaration order
        self.blocks = []       # list of lists of attr names; [] => whole class is one block
        self._open = None
        self.py_strength, self.py_discount = 1.0, 0.0  # builder.jl:39

    def _add(self, attr):
        if any(a.name == attr.name for a in self.attrs):
            raise ValueError(f"{self.name}.{attr.name} declared twice")
        self.attrs.append(attr)
        if attr.kind != "param":
            if self._open is not None:
                self._open.append(attr.name)
            elif self.blocks and self._implicit:
                self.blocks[-1].append(attr.name)
            else:
                self.blocks.append([attr.name])
                self._implicit = True
        return attr

    _implicit = False

    def param(self, name, prior):
        return self._add(Attr("param", name, prior=prior))

    def choice(self, name, dist):
        return self._add(Attr("choice", name, dist=dist))

    def fk(self, name, target):
        if target not in self.model.classes:
            raise ValueError(f"class {target} must be defined before it is referenced")
        return self._add(Attr("fk", name, target=target))

    def julia(self, name, fn, args):
        return self._add(Attr("julia", name, fn=fn, args=list(args)))

    @contextmanager
    def block(self):
        self._open = []
        self._implicit = False
        self.blocks.append(self._open)
        try:
            yield self
        finally:
            self._open = None

    def attr(self, name):
        for a in self.attrs:
            if a.name == name:
                return a
        raise KeyError(f"{self.name}.{name}")


class Model:
    def __init__(self):
        self.classes = {}
        self.class_order = []

    def add_class(self, name):
        c = ClassDef(self, name)
        self.classes[name] = c
        self.class_order.append(name)
        return c

    def resolve(self, cls, path):
        """Follow a dotted reference from class `cls`; returns (class name, attr)."""
        parts = path.split(".")
        c = self.classes[cls]
        for p in parts[:-1]:
            a = c.attr(p)
            if a.kind != "fk":
                raise ValueError(f"{path}: {p} is not a reference slot")
            c = self.classes[a.target]
        return c.name, c.attr(parts[-1])


class Query:
    """`@query Model.Class [ column clean_expr dirty_expr ]` (dsl/query.jl:15-38).
    bindings: column -> (clean reference, dirty attribute name)."""

    def __init__(self, model, cls, bindings):
        self.model, self.cls = model, cls
        self.cleanmap, self.obsmap = {}, {}
        for col, b in bindings.items():
            clean, dirty = (b, b) if isinstance(b, str) else b
            self.cleanmap[col] = clean
            self.obsmap[col] = dirty


# ---------------------------------------------------------------------------
# lowering
class Column:
    """One flattened column of a latent table."""

    def __init__(self, name, kind, cls, attr, target=None):
        self.name, self.kind, self.cls, self.attr, self.target = name, kind, cls, attr, target


class MappedDomain:
    """Values of a latent domain seen through a single-argument JuliaNode: entry v is the pool string f(value v)
    (repeats allowed) — what a pair table needs of its latent side (ids, lengths)."""

    def __init__(self, pool, strings):
        self.pool = pool
        self.ids = [pool.add(s) for s in strings]

    def __len__(self):
        return len(self.ids)

    def string(self, j):
        return self.pool.strings[self.ids[j]]

    def id_array(self):
        return np.array(self.ids, dtype=np.int32)


class LoweredModel:
    """Domains, flattened table layouts, option tables and per-block plans."""

    def __init__(self, model, query, dirty_columns, pool=None, extra_latent=None):
        """extra_latent {(class, attribute): [strings]}: values a latent attribute holds although they are neither
        proposal atoms nor the dummy — strings drawn by random(StringPrior / TimePrior) for a chosen
        ProposalDummyValue (block_proposal.jl:58-60).  They are appended to the attribute's domain (ids of the
        atoms and the dummy do not move) and are never options of a proposal."""
        self.model, self.query = model, query
        self._dirty_columns = dirty_columns
        self.extra_latent = {k: list(v) for k, v in (extra_latent or {}).items()}
        self.pool = pool or StringPool()
        self.latent_dom = {}   # (class, attr) -> Domain of latent values
        self.obs_dom = {}      # dirty attr name -> Domain of observed values
        self.layout = {}       # class -> [Column]
        self.colidx = {}       # class -> {dotted name: index}
        self.table_id = {}     # class -> candidate table id
        self.option_id = {}    # (class, attr) -> option table id
        self.pair_id = {}      # (dirty attr) -> pair table id
        self.fn_tables = {}    # fn id -> ndarray
        self.blocks = []       # dicts: nodes, terms, children, colmap, ctx_src_block, ctx_src_col, root_class
        self.obs_cols = []     # dirty attr name per observed column id
        self.obs_index = {}
        self._next_table = 0
        self._next_pair = 0
        self.cross_terms = []
        self.eq_pairs = {}
        self.same_pairs = {}   # pair id -> (observed domain, latent domain): 0 iff same string
        self.prob_spec = None
        self.gauss = {}        # (block id, node id) -> dict spec (resolved by the engine)
        self.locals = {}       # block index -> [own ChooseUniformly attrs enumerated with the Gaussian]
        self.latent_ev_locals = {}  # latent class -> block index whose locals feed the evidence ctx
        self.latent_ev_prob = {}    # latent class -> scoring block whose error-prob index feeds the evidence ctx
        self.latent_plans = {}
        self._build_domains(dirty_columns)
        self._build_layouts()
        self._build_blocks()
        self._build_latent_plans()

    # -- domains ------------------------------------------------------------
    def _build_domains(self, dirty_columns):
        m = self.model
        for cname in m.class_order:
            for a in m.classes[cname].attrs:
                if a.kind != "choice":
                    continue
                d = a.dist
                if (isinstance(d, StringPrior) and d.keyed_by) or isinstance(d, TimePrior):
                    dom = Domain(self.pool)
                    for k, atoms in d.atoms.items():
                        for s_ in atoms:
                            dom.add(s_)
                    dom.add(d.dummy_value())
                elif isinstance(d, StringPrior):
                    dom = Domain(self.pool, d.atoms)
                    dom.add(d.dummy_value())
                elif isinstance(d, ChooseProportionally):
                    dom = Domain(self.pool, d.options)
                elif isinstance(d, ChooseUniformly) and all(isinstance(o, str) for o in d.options):
                    dom = Domain(self.pool, d.options)
                else:
                    continue
                for s_ in self.extra_latent.get((cname, a.name), ()):
                    dom.add_extra(s_)  # ids after the dummy's: never an option (MaybeSwap: "val in options")
                self.latent_dom[(cname, a.name)] = dom
        ocls = m.classes[self.query.cls]
        self.direct_obs = {}   # obs name -> (path or own attr name) observed without noise (clean == dirty)
        self.numeric_obs = {}  # own TransformedGaussian attr -> numeric column index
        self.num_cols = []
        self.num_derived = []  # (source numeric column, Transformation, "backward" | "logabsderiv"): see _lower_gaussian
        for col, dirty in self.query.obsmap.items():
            own = None
            if "." not in dirty:
                own = ocls.attr(dirty)
            if own is not None and own.kind == "choice" and isinstance(own.dist, (AddTypos, MaybeSwap)):
                vals = [v for v in dirty_columns[col] if v is not None]
                self.obs_dom[dirty] = Domain(self.pool, list(dict.fromkeys(vals)))
            elif own is not None and own.kind == "choice" and isinstance(own.dist, TransformedGaussian):
                self.numeric_obs[dirty] = len(self.num_cols)
                self.num_cols.append(col)
                continue
            else:
                # direct observation of a latent value or of an own discrete choice: the observed
                # domain IS the latent domain (proposal_compiler.jl:277-293 compares values)
                if own is not None:
                    key = (self.query.cls, dirty)
                else:
                    head, rest = dirty.split(".", 1)
                    cn, la = m.resolve(ocls.attr(head).target, rest)
                    key = (cn, la.name)
                if key not in self.latent_dom:  # e.g. Unmodeled: its values are whatever is observed
                    vals = [v for v in dirty_columns[col] if v is not None]
                    self.latent_dom[key] = Domain(self.pool, list(dict.fromkeys(vals)))
                self.obs_dom[dirty] = self.latent_dom[key]
                self.direct_obs[dirty] = key
            self.obs_index[dirty] = len(self.obs_cols)
            self.obs_cols.append(dirty)
        self.query_columns = {dirty: col for col, dirty in self.query.obsmap.items()}
        self._never_missing = {dirty for col, dirty in self.query.obsmap.items()
                               if all(v is not None for v in dirty_columns[col])}

    def gauss_backward(self, rows, unit_idx):
        """unit.backward(x) of the Gaussian observation of `rows` under the Transformation options unit_idx (one per row):
        x * c for the linear ones, the derived column for the others (transformed_gaussian.jl:16, 27-34)."""
        spec = self.gauss_spec
        rows = np.asarray(rows)
        unit_idx = np.asarray(unit_idx)
        x = self.xnum[spec["x_col"], rows] * np.asarray(spec["t_scale"])[unit_idx]
        for ui, col in enumerate(spec.get("t_x_col", ())):
            if col >= 0:
                sel = unit_idx == ui
                x[sel] = self.xnum[col, rows[sel]]
        return x

    def relower(self, extra_latent):
        """Grow latent domains by the strings of extra_latent {(class, attribute): [strings]} and rebuild every
        derived table IN PLACE (pair / fn / equality tables, plans): value ids held by a trace stay valid, callers
        keep their reference.  The engine has to reload its static data afterwards (Engine.reload)."""
        merged = {k: list(v) for k, v in self.extra_latent.items()}
        for k, v in extra_latent.items():
            have = merged.setdefault(k, [])
            have.extend(x for x in dict.fromkeys(v) if x not in have)
        dirty = self._dirty_columns
        self.__init__(self.model, self.query, dirty, None, merged)
        self.encode_observations(dirty)

    def encode_observations(self, dirty_columns):
        """[n_cols][n_rows] int32 observed-domain indices, -1 = missing."""
        n_rows = len(next(iter(dirty_columns.values())))
        self.xnum = np.array([[np.nan if v is None else float(v) for v in dirty_columns[c]] for c in self.num_cols],
                             dtype=np.float64).reshape(len(self.num_cols), n_rows)
        derived = getattr(self, "num_derived", [])
        if derived:  # backward(x) / log|deriv(backward(x))| of the non-linear Transformations, row by row (_lower_gaussian)
            rows = []
            for src, unit, what in derived:
                col = np.full(n_rows, np.nan)
                for i, x in enumerate(self.xnum[src]):
                    if x == x:
                        bx = float(unit.backward(float(x)))
                        col[i] = bx if what == "backward" else float(np.log(abs(float(unit.deriv(bx)))))
                rows.append(col)
            self.xnum = np.vstack([self.xnum, np.array(rows, dtype=np.float64).reshape(len(rows), n_rows)])
        obs = np.full((len(self.obs_cols), n_rows), -1, dtype=np.int32)
        for j, dirty in enumerate(self.obs_cols):
            dom = self.obs_dom[dirty]
            col = dirty_columns[self.query_columns[dirty]]
            obs[j] = [(-1 if v is None else dom.get(v)) for v in col]
        self.obs_host = obs  # host copy for the sufficient statistics of observation-level parameters
        return obs

    # -- layouts ------------------------------------------------------------
    def _build_layouts(self):
        m = self.model
        for cname in m.class_order:
            if cname == self.query.cls:
                continue
            cols = []
            for a in m.classes[cname].attrs:
                if a.kind == "fk":
                    cols.append(Column(a.name, "fk", cname, a.name, a.target))
                    for c in self.layout[a.target]:
                        cols.append(Column(a.name + "." + c.name, c.kind, c.cls, c.attr, c.target))
                elif a.kind == "choice":
                    cols.append(Column(a.name, "val", cname, a.name))
            self.layout[cname] = cols
            self.colidx[cname] = {c.name: i for i, c in enumerate(cols)}
            self.table_id[cname] = self._next_table
            self._next_table += 1
        self.option_values = {}
        self.option_keycol = {}
        self.option_ncol = {}
        for (cname, aname), dom in self.latent_dom.items():
            self.option_id[(cname, aname)] = self._next_table
            self._next_table += 1
            # option k of discrete_proposal(dist, ...) is latent-domain value k (atoms are unique,
            # string_prior.jl:15; the dummy value, if any, is last)
            self.option_values[(cname, aname)] = np.arange(dom.n_base(), dtype=np.int32)  # (drawn strings are no options)
            if cname in self.model.classes:
                a = self.model.classes[cname].attr(aname)
                d = getattr(a, "dist", None)
                if (isinstance(d, StringPrior) and d.keyed_by) or isinstance(d, TimePrior):
                    # options = for every key: its atoms, then one dummy option; column 1 = key index
                    kdom = self.latent_dom[(cname, d.keyed_by)]
                    vals, keys = [], []
                    dummy = dom.get(d.dummy_value())
                    for k, atoms in d.atoms.items():
                        ki = kdom.get(k)
                        for s_ in atoms:
                            vals.append(dom.get(s_))
                            keys.append(ki)
                        vals.append(dummy)
                        keys.append(ki)
                    self.option_values[(cname, aname)] = np.array(vals, dtype=np.int32)
                    self.option_keycol[(cname, aname)] = np.array(keys, dtype=np.int32)
                    # column 2: number of atoms of the option's key group (MaybeSwap's length(options))
                    nk = {kdom.get(k): len(atoms) for k, atoms in d.atoms.items()}
                    self.option_ncol[(cname, aname)] = np.array([nk[k] for k in keys], dtype=np.int32)

    # -- plans --------------------------------------------------------------
    def _obs_terms_of_block(self, ocls, names):
        """Observation choices of a block as (dirty attr, kind, payload)."""
        out = []
        for n in names:
            a = ocls.attr(n)
            if a.kind == "choice" and isinstance(a.dist, AddTypos):
                out.append(a)
        return out

    def _eq_pair_for(self, dom_key):
        """0/1 identity table over a shared domain (observed value must equal the latent value)."""
        key = ("eq", dom_key)
        if key not in self.eq_pairs:
            self.eq_pairs[key] = (self._next_pair, len(self.latent_dom[dom_key]))
            self._next_pair += 1
        return self.eq_pairs[key][0]

    def _pair_for(self, dirty, lat_dom_key, lat_dom):
        key = (dirty, lat_dom_key)
        if key not in self.pair_id:
            self.pair_id[key] = (self._next_pair, self.obs_dom[dirty], lat_dom)
            self._next_pair += 1
        return self.pair_id[key][0]

    def _build_blocks(self):
        m = self.model
        ocls = m.classes[self.query.cls]
        root_of_block = []
        fk_block = {}
        self.score_blocks = {}
        # A block of the observed class with several reference slots (PCleanClass.blocks, model.jl:100-106) is lowered
        # into one ENGINE block per slot, proposed in declaration order with no resampling in between (block_group:
        # run_smc! resamples between the class's blocks only, row_inference.jl:152-155).  An observation belongs to the
        # last slot it mentions; a JuliaNode across two slots of one block reads the earlier slot's value as context.
        # Independent slots are enumerated independently by the reference as well (process_plan!,
        # proposal_compiler.jl:363-388); slots tied by a JuliaNode are proposed here one after the other, each given
        # the earlier ones, instead of jointly.
        eblocks, self.block_group = [], []
        for ub, names in enumerate(ocls.blocks):
            fks = [n for n in names if ocls.attr(n).kind == "fk"]
            if len(fks) <= 1:
                eblocks.append(list(names))
                self.block_group.append(ub)
                continue
            order = {f: i for i, f in enumerate(fks)}
            groups = [[f] for f in fks]

            def heads(a):
                if a.kind == "julia":
                    return [x.split(".", 1)[0] for x in a.args]
                ref = getattr(a.dist, "ref", None) if a.kind == "choice" else None
                if ref is None:
                    return []
                if "." in ref:
                    return [ref.split(".", 1)[0]]
                return heads(ocls.attr(ref))
            for n in names:
                a = ocls.attr(n)
                if a.kind == "fk":
                    continue
                hs = [order[h] for h in heads(a) if h in order]
                groups[max(hs) if hs else len(fks) - 1].append(n)
            for g in groups:
                eblocks.append(g)
                self.block_group.append(ub)
        self.engine_blocks = eblocks
        for bi, names in enumerate(eblocks):
            fks = [n for n in names if ocls.attr(n).kind == "fk"]
            root_of_block.append(fks[0] if fks else None)
            if fks:
                fk_block[fks[0]] = bi
        for bi, names in enumerate(eblocks):
            if root_of_block[bi] is None:
                self._lower_score_block(bi, ocls, names, fk_block)
                self.blocks.append(dict(score=True, nodes=[], terms=[], children=[], colmap=[], node_info=[],
                                        root_class=None, root_fk=None, ctx_src_block=[], ctx_src_col=[]))
                continue
            root_fk = ocls.attr(root_of_block[bi])
            blk = dict(nodes=[], terms=[], children=[], colmap=[], ctx_src_block=[], ctx_src_col=[],
                       root_class=root_fk.target, root_fk=root_fk.name, node_info=[])
            # collect observation terms of this block: (dirty attr, path below root, ctx spec)
            terms = []
            for a in self._obs_terms_of_block(ocls, names):
                ref = a.dist.ref
                if "." in ref:
                    head, rest = ref.split(".", 1)
                    if head != root_fk.name:
                        raise NotImplementedError(f"{a.name}: reference outside the block's root slot")
                    cname, la = m.resolve(root_fk.target, rest)
                    pid = self._pair_for(a.name, (cname, la.name), self.latent_dom[(cname, la.name)])
                    terms.append(dict(obs=a.name, path=rest, pair=pid, max_typos=a.dist.max_typos, ctx=None))
                else:
                    j = ocls.attr(ref)
                    if j.kind != "julia":
                        raise NotImplementedError(f"{a.name}: AddTypos of a non-reference")
                    local = [x for x in j.args if x.split(".", 1)[0] == root_fk.name]
                    other = [x for x in j.args if x.split(".", 1)[0] != root_fk.name]
                    if len(local) != 1 or len(other) > 1:
                        raise NotImplementedError(f"{j.name}: a JuliaNode under an AddTypos observation combines ONE value of "
                                                  "its block's slot with at most one value of an earlier slot; functions of "
                                                  "several values of one slot, or of three and more slots, are not lowered")
                    lc, la = m.resolve(root_fk.target, local[0].split(".", 1)[1])
                    ldom = self.latent_dom[(lc, la.name)]
                    if other:
                        ohead, orest = other[0].split(".", 1)
                        sb = fk_block[ohead]
                        if sb >= bi:
                            raise NotImplementedError("ctx must come from an earlier block")
                        ocn, oa = m.resolve(ocls.attr(ohead).target, orest)
                        odom = self.latent_dom[(ocn, oa.name)]
                        src = (sb, self.colidx[ocls.attr(ohead).target][orest])
                        have = list(zip(blk["ctx_src_block"], blk["ctx_src_col"]))
                        if src in have:  # two JuliaNodes reading the same earlier value share its ctx slot
                            slot = have.index(src)
                        else:
                            slot = len(have)
                            if slot >= _lib.MAX_CTX:
                                raise NotImplementedError(f"a block reads more than {_lib.MAX_CTX} earlier values through JuliaNodes "
                                                          "(PCLEAN_MAX_CTX)")
                            blk["ctx_src_block"].append(sb)
                            blk["ctx_src_col"].append(src[1])
                        order = [j.args.index(other[0]), j.args.index(local[0])]
                        jdom = Domain(self.pool)
                        fn = np.zeros((len(odom), len(ldom)), dtype=np.int32)
                        for x in range(len(odom)):
                            for y in range(len(ldom)):
                                argv = [None, None]
                                argv[order[0]] = odom.string(x)
                                argv[order[1]] = ldom.string(y)
                                fn[x, y] = jdom.add(j.fn(*argv))
                        fid = len(self.fn_tables)
                        self.fn_tables[fid] = fn
                        pid = self._pair_for(a.name, ("julia", j.name), jdom)
                        terms.append(dict(obs=a.name, path=local[0].split(".", 1)[1], pair=pid,
                                          max_typos=a.dist.max_typos, ctx=(slot, fid)))
                        # the same observation also constrains the OTHER argument's class (external
                        # likelihood of e.g. County.state through Record.stateavg_obs)
                        self.cross_terms.append(dict(obs=a.name, pair=pid, max_typos=a.dist.max_typos, fn=fid,
                                                     ctx_block=sb, ctx_path=orest, local_block=bi,
                                                     local_path=local[0].split(".", 1)[1]))
                    else:
                        # f(one value of this slot): a pair table whose latent string of value v is f(v) — the
                        # distances and word lengths AddTypos needs (add_typos.jl:56-63), nothing else changes
                        mdom = MappedDomain(self.pool, [j.fn(ldom.string(y)) for y in range(len(ldom))])
                        pid = self._pair_for(a.name, ("julia", j.name), mdom)
                        terms.append(dict(obs=a.name, path=local[0].split(".", 1)[1], pair=pid,
                                          max_typos=a.dist.max_typos, ctx=None))
            # direct (noise-free) observations of values below the root slot: equality constraints
            for obsname, key in self.direct_obs.items():
                if "." in obsname and obsname.split(".", 1)[0] == root_fk.name:
                    terms.append(dict(obs=obsname, path=obsname.split(".", 1)[1], pair=self._eq_pair_for(key),
                                      max_typos=None, ctx=None, dens=_lib.DENS_EQUAL))
            self._emit_fk_node(blk, root_fk.target, "", terms, parent=-1, parent_fk_col=-1)
            self.blocks.append(blk)
            self._lower_gaussian(bi, blk, ocls, names, root_fk)

    def _emit_term(self, blk, t, cand_col):
        blk["terms"].append((self.obs_index[t["obs"]], cand_col, t["pair"], t.get("dens", _lib.DENS_ADD_TYPOS),
                             -1 if t["max_typos"] is None else int(t["max_typos"]),
                             -1 if t["ctx"] is None else t["ctx"][0], -1 if t["ctx"] is None else t["ctx"][1], 0))

    def _emit_fk_node(self, blk, cname, prefix, terms, parent, parent_fk_col):
        """Node enumerating rows of latent class `cname`; `terms` are the observation
        terms whose clean value lives in this sub-tree (paths relative to it)."""
        m = self.model
        nid = len(blk["nodes"])
        blk["nodes"].append(None)
        blk["node_info"].append(dict(kind="fk", cls=cname, attr=None, path=prefix[:-1]))
        tb = len(blk["terms"])
        for t in terms:
            self._emit_term(blk, t, self.colidx[cname][t["path"]])
        nt = len(blk["terms"]) - tb
        # children: own attributes of the class, in declaration order
        kids = []
        colsrc = {}
        for a in m.classes[cname].attrs:
            if a.kind == "fk":
                sub = [dict(t, path=t["path"].split(".", 1)[1]) for t in terms if t["path"].split(".", 1)[0] == a.name
                       and "." in t["path"]]
                cid = self._emit_fk_node(blk, a.target, prefix + a.name + ".", sub, nid, self.colidx[cname][a.name])
                kids.append(cid)
                colsrc[a.name] = (-1, -1)
                for c in self.layout[a.target]:
                    colsrc[a.name + "." + c.name] = (cid, self.colidx[a.target][c.name])
            elif a.kind == "choice":
                sub = [t for t in terms if t["path"] == a.name]
                cid = len(blk["nodes"])
                ltb = len(blk["terms"])
                for t in sub:
                    self._emit_term(blk, t, 0)
                n_leaf_terms = len(sub)
                if (isinstance(a.dist, StringPrior) and a.dist.keyed_by) or isinstance(a.dist, TimePrior):
                    # atoms belong to the key they were listed under: the option's key must equal the
                    # (directly observed) key of this row
                    kt = [t for t in terms if t["path"] == a.dist.keyed_by and t.get("dens") == _lib.DENS_EQUAL]
                    if len(kt) != 1:
                        raise NotImplementedError("keyed atoms need their key attribute observed directly")
                    self._emit_term(blk, kt[0], 1)
                    n_leaf_terms += 1
                cacheable = int(n_leaf_terms == 1 and len(sub) == 1 and sub[0]["ctx"] is None)
                # the ProposalDummyValue of the proposal (string_prior.jl:16-26, time_prior.jl:15-19): which latent
                # value is the placeholder and what random(dist) samples when the dummy is chosen
                dval, dspec = 0, 0
                if isinstance(a.dist, (StringPrior, TimePrior)):
                    dval = self.latent_dom[(cname, a.name)].get(a.dist.dummy_value()) + 1
                    if isinstance(a.dist, TimePrior):
                        dspec = _lib.DUMMY_TIME_PRIOR
                    else:
                        if not (0 <= int(a.dist.min_len) <= 255 and 0 <= int(a.dist.max_len) <= 255):
                            raise NotImplementedError(f"{cname}.{a.name}: StringPrior lengths beyond 255 do not fit the "
                                                      "dummy specification (device draws use DUMMY_MAX_LEN = 255)")
                        dspec = _lib.DUMMY_STRING_PRIOR | (int(a.dist.min_len) << 8) | (int(a.dist.max_len) << 16)
                blk["nodes"].append((_lib.NODE_LEAF, self.option_id[(cname, a.name)], ltb, n_leaf_terms, 0, 0, nid, -1,
                                     cacheable, 0, dval, dspec))
                blk["node_info"].append(dict(kind="leaf", cls=cname, attr=a.name, path=prefix + a.name))
                kids.append(cid)
                colsrc[a.name] = (cid, 0)
        cb = len(blk["children"])
        blk["children"].extend(kids)
        cmb = len(blk["colmap"]) // 2
        for c in self.layout[cname]:
            blk["colmap"].extend(colsrc[c.name])
        blk["nodes"][nid] = (_lib.NODE_FK, self.table_id[cname], tb, nt, cb, len(kids), parent, parent_fk_col, 0, cmb,
                             0, 0)
        return nid

    def score_block_args(self, bi):
        """Arguments of pclean_load_score_block for scoring block bi."""
        sb = self.score_blocks[bi]
        t, pr = sb["terms"], sb["prob"]
        return (bi, [x["obs"] for x in t], [x["pair"] for x in t], [c for x in t for c in x["val"]],
                [c for x in t for c in x["key"]], [x["nopt_fn"] for x in t], [x["other"] for x in t], pr["fn"],
                list(pr["a"]), list(pr["b"]))

    def load_blocks_into(self, target):
        """Upload every block plan (observed-class blocks, scoring blocks, latent-class plans) into an
        object exposing load_block / load_score_block (the HIP context or the oracle world)."""
        for bi, blk in enumerate(self.blocks):
            if blk.get("score"):
                target.load_score_block(*self.score_block_args(bi))
            else:
                target.load_block(bi, *self.block_arrays(bi))
        for cname, pl in self.latent_plans.items():
            target.load_block(pl["block_id"], *self.latent_block_arrays(cname))
        if len(set(self.block_group)) < len(self.block_group):  # a model block with several reference slots
            for bi, g in enumerate(self.block_group):
                target.set_block_group(bi, g)

    def same_pair_table(self, pid):
        odom, vdom = self.same_pairs[pid]
        return (odom.id_array()[:, None] != vdom.id_array()[None, :]).astype(np.uint8)

    def _lower_score_block(self, bi, ocls, names, fk_block):
        """Block without a reference slot: MaybeSwap observations of values chosen in earlier blocks
        (experiments/flights/run.jl:29-34)."""
        m = self.model
        terms = []
        prob_spec = None
        for n in names:
            a = ocls.attr(n)
            if a.kind != "choice":
                continue
            if not isinstance(a.dist, MaybeSwap):
                raise NotImplementedError("a block without a reference slot may only hold MaybeSwap observations")
            vh, vrest = a.dist.val.split(".", 1)
            kh, krest = a.dist.key.split(".", 1)
            vcls, vattr = m.resolve(ocls.attr(vh).target, vrest)
            kcls, kattr = m.resolve(ocls.attr(kh).target, krest)
            vdom, kdom = self.latent_dom[(vcls, vattr.name)], self.latent_dom[(kcls, kattr.name)]
            # 0/1 "same string" table between the observed values and the latent domain
            odom = self.obs_dom[n]
            pid = self._next_pair
            self._next_pair += 1
            self.same_pairs[pid] = (odom, vdom)
            nopt = np.ones((len(kdom), 1), dtype=np.int32)
            for k, opts in a.dist.options.items():
                if kdom.get(k) >= 0:
                    nopt[kdom.get(k), 0] = len(opts)
            fid = len(self.fn_tables)
            self.fn_tables[fid] = nopt
            terms.append(dict(obs=self.obs_index[n], pair=pid, val=(fk_block[vh], self.colidx[ocls.attr(vh).target][vrest]),
                              key=(fk_block[kh], self.colidx[ocls.attr(kh).target][krest]), nopt_fn=fid,
                              other=vdom.get(vattr.dist.dummy_value()), attr=n, val_ref=a.dist.val))
            pl = ocls.attr(a.dist.prob)
            if pl.kind != "julia" or not isinstance(pl.fn, ProbLookup):
                raise NotImplementedError("MaybeSwap prob must be a ProbLookup")
            if prob_spec is None:
                (ah, arest), (bh, brest) = [x.split(".", 1) for x in pl.args]
                acls, aattr = m.resolve(ocls.attr(ah).target, arest)
                bcls, battr = m.resolve(ocls.attr(bh).target, brest)
                adom, bdom = self.latent_dom[(acls, aattr.name)], self.latent_dom[(bcls, battr.name)]
                keys, consts = [], []
                pf = np.zeros((len(adom), len(bdom)), dtype=np.int32)
                for x in range(len(adom)):
                    for y in range(len(bdom)):
                        r = pl.fn.fn(adom.string(x), bdom.string(y))
                        if isinstance(r, float):
                            if r not in consts:
                                consts.append(r)
                            pf[x, y] = consts.index(r)
                        else:
                            if r not in keys:
                                keys.append(r)
                            pf[x, y] = -1 - keys.index(r)
                # prob table layout: constants first, then one entry per parameter key
                pf = np.where(pf < 0, len(consts) + (-1 - pf), pf).astype(np.int32)
                pfid = len(self.fn_tables)
                self.fn_tables[pfid] = pf
                prob_spec = dict(fn=pfid, a=(fk_block[ah], self.colidx[ocls.attr(ah).target][arest]),
                                 b=(fk_block[bh], self.colidx[ocls.attr(bh).target][brest]), consts=consts, keys=keys,
                                 param=(self.query.cls, pl.fn.param))
        self.score_blocks[bi] = dict(terms=terms, prob=prob_spec)
        self.prob_spec = prob_spec

    def _lower_gaussian(self, bi, blk, ocls, names, root_fk):
        """`x ~ TransformedGaussian(param[f(root values, own choices)], std, unit)` with own
        ChooseUniformly choices (experiments/rents/run.jl:19-25) -> pclean_gauss specs."""
        m = self.model
        ga = [ocls.attr(n) for n in names if ocls.attr(n).kind == "choice" and isinstance(ocls.attr(n).dist, TransformedGaussian)]
        if not ga:
            return
        if len(ga) > 1:
            raise NotImplementedError("one Gaussian observation per block (so far)")
        g = ga[0]
        look = ocls.attr(g.dist.mean)
        if look.kind != "julia" or not isinstance(look.fn, IndexedLookup):
            raise NotImplementedError("TransformedGaussian mean must be an IndexedLookup")
        unit_attr = ocls.attr(g.dist.unit)
        units = unit_attr.dist.options
        if len(units) > 4:
            raise NotImplementedError("at most four Transformations to choose from (pclean_gauss::t_scale[4])")
        # A LINEAR unit (backward(x) = c x, |deriv| constant: the rents program's) is evaluated by the kernels as x * c and a
        # constant log|deriv|.  Any other Transformation (transformed_gaussian.jl:5-9 takes arbitrary functions, 15-16
        # evaluates them per observation): x is DATA, so backward(x) and log|deriv(backward(x))| of every row are two
        # more numeric columns, evaluated once on the host (encode_observations) and read by the kernels per row.
        t_linear = []
        for u in units:
            probes = (0.5, 2.0, -3.0, 1267.0)
            try:
                b1 = float(u.backward(1.0))
                d1 = float(u.deriv(b1))
                lin = abs(float(u.backward(0.0))) <= 1e-12 and all(
                    abs(float(u.backward(x)) - x * b1) <= 1e-9 * max(1.0, abs(x * b1)) and
                    abs(float(u.deriv(u.backward(x))) - d1) <= 1e-9 * max(1.0, abs(d1)) for x in probes)
            except (ValueError, ZeroDivisionError, OverflowError, FloatingPointError):
                lin = False  # (a probe outside the function's domain: log of a negative number ...)
            t_linear.append(bool(lin))
        locs = []  # own enumerated choices: index arguments that are own attrs, plus the unit
        dims = []  # (kind, payload, n_values)
        for arg in look.args:
            if "." in arg:
                head, rest = arg.split(".", 1)
                cn, la = m.resolve(root_fk.target, rest)
                dims.append(("cand", rest, len(self.latent_dom[(cn, la.name)]), (cn, la.name)))
            else:
                oa = ocls.attr(arg)
                if not isinstance(oa.dist, ChooseUniformly):
                    raise NotImplementedError("own index arguments must be ChooseUniformly choices")
                if arg not in locs:
                    locs.append(arg)
                dims.append(("local", locs.index(arg), len(oa.dist.options), None))
        if g.dist.unit not in locs:
            locs.append(g.dist.unit)
        if len(locs) > 2:
            raise NotImplementedError("at most two enumerated own choices")
        if int(np.prod([len(ocls.attr(l).dist.options) for l in locs])) > 16:
            raise NotImplementedError("at most 16 combinations of the enumerated own choices (gauss_combo_scores: sc[16])")
        strides, acc = [], 1
        for d in reversed(dims):
            strides.append(acc)
            acc *= d[2]
        strides = strides[::-1]
        self.locals[bi] = locs
        spec = dict(x_col=self.numeric_obs[g.name], param=(self.query.cls, look.fn.param), n_mean=acc, dims=dims,
                    strides=strides, locals=locs, local_n=[len(ocls.attr(l).dist.options) for l in locs],
                    local_obs=[self.obs_index.get(l, -1) if l in self.direct_obs else -1 for l in locs],
                    t_local=locs.index(g.dist.unit), sigma=g.dist.std, gauss_attr=g.name,
                    t_scale=[float(u.backward(1.0)) if lin else 1.0 for u, lin in zip(units, t_linear)],
                    t_lad=[float(np.log(abs(u.deriv(u.backward(1.0))))) if lin else 0.0 for u, lin in zip(units, t_linear)],
                    units=units, t_linear=t_linear, t_x_col=[-1] * len(units), t_lad_col=[-1] * len(units))
        # derived numeric columns of the non-linear units: appended behind the observed ones (encode_observations)
        self.num_derived = []
        for ui, lin in enumerate(t_linear):
            if not lin:
                spec["t_x_col"][ui] = len(self.num_cols) + len(self.num_derived)
                self.num_derived.append((spec["x_col"], units[ui], "backward"))
                spec["t_lad_col"][ui] = len(self.num_cols) + len(self.num_derived)
                self.num_derived.append((spec["x_col"], units[ui], "logabsderiv"))
        self.gauss_spec = spec
        # (a) block root: candidate-side index values come from the candidate's columns
        rc = root_fk.target
        self.gauss[(bi, 0)] = dict(spec, kinds=[("cand", self.colidx[rc][d[1]]) if d[0] == "cand" else ("local", d[1])
                                                 for d in dims], n_locals=len(locs), transform=("local", spec["t_local"]))
        # (b) new-row branch: the leaf of the one candidate-side value that is not always observed
        #     carries the term; the others are read from their direct observations
        open_dims = [d for d in dims if d[0] == "cand" and not self._always_observed(root_fk.name + "." + d[1])]
        if len(open_dims) != 1:
            raise NotImplementedError("exactly one candidate-side index value may be unobserved")
        for nid, info in enumerate(blk["node_info"]):
            if info["kind"] == "leaf" and info["path"] == open_dims[0][1]:
                kinds = []
                for d in dims:
                    if d[0] == "local":
                        kinds.append(("local", d[1]))
                    elif d is open_dims[0]:
                        kinds.append(("cand", 0))
                    else:
                        kinds.append(("obs", self.obs_index[root_fk.name + "." + d[1]]))
                self.gauss[(bi, nid)] = dict(spec, kinds=kinds, n_locals=len(locs), transform=("local", spec["t_local"]))
                node = list(blk["nodes"][nid])
                node[8] = 0  # not cacheable any more
                blk["nodes"][nid] = tuple(node)
                self.gauss_open = (bi, open_dims[0])

    def _always_observed(self, obsname):
        return obsname in self.direct_obs and obsname in self._never_missing

    # -- latent-class plans ------------------------------------------------------
    def _build_latent_plans(self):
        """For every latent class T: the sub-plans of its own attributes, scored against all
        observed rows that (transitively) refer to a row of T — the ExternalLikelihoodNodes that
        builder.jl:264-340 adds to T, restated as the children of T's node in the observed plan
        with evidence sets instead of a single row."""
        next_block = len(self.blocks)
        for bi, blk in enumerate(self.blocks):
            if blk.get("score"):
                continue
            for nid, info in enumerate(blk["node_info"]):
                if info["kind"] != "fk" or info["cls"] in self.latent_plans:
                    continue
                cname = info["cls"]
                # per-evidence-row context values of the plan's JuliaNode terms: slot s of the observed block keeps its
                # number (the evidence row's value of that earlier slot); every cross-block JuliaNode whose CONTEXT
                # argument lives in this class adds a slot holding the evidence row's LOCAL argument (_copy_subtree)
                plan = dict(block_id=next_block, src_block=bi, src_node=nid, cls=cname, path=info["path"], nodes=[],
                            terms=[], children=[], colmap=[], node_info=[], roots=[], root_attr=[],
                            ctx_sources=list(zip(blk["ctx_src_block"], blk["ctx_src_col"])))
                node = blk["nodes"][nid]
                for k in range(node[4], node[4] + node[5]):
                    child = blk["children"][k]
                    plan["roots"].append(self._copy_subtree(blk, child, plan, -1, bi))
                    ci = blk["node_info"][child]
                    # attribute of T this root re-proposes
                    plan["root_attr"].append(ci["attr"] if ci["kind"] == "leaf" else ci["path"].split(".")[-1])
                self.latent_plans[cname] = plan
                next_block += 1

    def _copy_subtree(self, blk, nid, plan, parent, bi):
        node, info = blk["nodes"][nid], blk["node_info"][nid]
        new_id = len(plan["nodes"])
        plan["nodes"].append(None)
        plan["node_info"].append(dict(info))
        tb = len(plan["terms"])
        for t in blk["terms"][node[2]:node[2] + node[3]]:
            t = list(t)
            if t[5] >= 0:
                t[7] = 1  # ctx now comes from the evidence row (fn[ctx][candidate])
            plan["terms"].append(tuple(t))
        # cross-block julia observations whose other argument lives in this sub-tree
        for ct in self.cross_terms:
            if ct["ctx_block"] != bi:
                continue
            q, p = ct["ctx_path"], info["path"]
            if info["kind"] == "leaf" and p == q:
                col = 0
            elif info["kind"] == "fk" and q.startswith(p + "."):
                col = self.colidx[info["cls"]][q[len(p) + 1:]]
            else:
                continue
            lb = ct["local_block"]
            src = (lb, self.colidx[self.blocks[lb]["root_class"]][ct["local_path"]])
            if src not in plan["ctx_sources"]:
                if len(plan["ctx_sources"]) >= _lib.MAX_CTX:
                    raise NotImplementedError(f"latent class {plan['cls']}: more than {_lib.MAX_CTX} per-evidence-row context "
                                              "values (PCLEAN_MAX_CTX)")
                plan["ctx_sources"].append(src)
            plan["terms"].append((self.obs_index[ct["obs"]], col, ct["pair"], _lib.DENS_ADD_TYPOS,
                                  -1 if ct["max_typos"] is None else int(ct["max_typos"]), plan["ctx_sources"].index(src),
                                  ct["fn"], 2))
        # MaybeSwap observations (scoring blocks) of this value: external likelihood of the referring rows,
        # each with its own error probability (evidence ctx slot 0 = index into the prob table)
        for sbi, sb in getattr(self, "score_blocks", {}).items():
            for t in sb["terms"]:
                if info["kind"] == "leaf" and t["val"][0] == bi and t["val_ref"].split(".", 1)[1] == info["path"]:
                    plan["terms"].append((t["obs"], 0, t["pair"], _lib.DENS_MAYBE_SWAP, 2, 0, t["other"], 1))
                    self.latent_ev_prob[plan["cls"]] = sbi
        nt = len(plan["terms"]) - tb
        if (bi, nid) in self.gauss and info["kind"] == "leaf":
            # latent sweep of the class owning this value: external likelihood of the referring rows'
            # Gaussian observations, their own choices held at their current values (evidence ctx)
            src = self.gauss[(bi, nid)]
            kinds = [("evctx", k[1]) if k[0] == "local" else k for k in src["kinds"]]
            self.gauss[(plan["block_id"], new_id)] = dict(src, kinds=kinds, n_locals=0,
                                                          transform=("evctx", src["t_local"]))
            self.latent_ev_locals[plan["cls"]] = bi
        kids = []
        if node[0] == _lib.NODE_FK:
            remap = {}
            for k in range(node[4], node[4] + node[5]):
                c = blk["children"][k]
                remap[c] = self._copy_subtree(blk, c, plan, new_id, bi)
                kids.append(remap[c])
            cb = len(plan["children"])
            plan["children"].extend(kids)
            cmb = len(plan["colmap"]) // 2
            ncols = len(self.layout[info["cls"]])
            for j in range(ncols):
                cn, cc = blk["colmap"][2 * (node[9] + j)], blk["colmap"][2 * (node[9] + j) + 1]
                plan["colmap"].extend((remap[cn] if cn >= 0 else -1, cc))
            plan["nodes"][new_id] = (node[0], node[1], tb, nt, cb, len(kids), parent, node[7], 0, cmb, 0, 0)
        else:
            plan["nodes"][new_id] = (node[0], node[1], tb, nt, 0, 0, parent, -1, 0, 0, 0, 0)
        return new_id

    def latent_block_arrays(self, cname):
        pl = self.latent_plans[cname]
        nodes = np.array(pl["nodes"], dtype=_lib.NODE_DTYPE)
        terms = np.array(pl["terms"], dtype=_lib.TERM_DTYPE) if pl["terms"] else np.zeros(0, dtype=_lib.TERM_DTYPE)
        # latent-mode ctx terms read the evidence row's ctx slots (ctx_sources; MaybeSwap / Gaussian evidence: slots 0, 1)
        n_ctx = max(2 if (pl["cls"] in self.latent_ev_locals) else 1, len(pl.get("ctx_sources", [])))
        return (nodes, terms, np.array(pl["children"], dtype=np.int32), np.array(pl["colmap"], dtype=np.int32),
                np.zeros(n_ctx, dtype=np.int32), np.zeros(n_ctx, dtype=np.int32))

    # -- arrays for the C ABI -------------------------------------------------
    def block_arrays(self, bi):
        blk = self.blocks[bi]
        nodes = np.array(blk["nodes"], dtype=_lib.NODE_DTYPE)
        terms = np.array(blk["terms"], dtype=_lib.TERM_DTYPE) if blk["terms"] else np.zeros(0, dtype=_lib.TERM_DTYPE)
        return (nodes, terms, np.array(blk["children"], dtype=np.int32), np.array(blk["colmap"], dtype=np.int32),
                np.array(blk["ctx_src_block"], dtype=np.int32), np.array(blk["ctx_src_col"], dtype=np.int32))

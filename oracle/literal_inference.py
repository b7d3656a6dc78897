"""oracle/literal_inference.py — TEST INFRASTRUCTURE ONLY.

A LITERAL, SEQUENTIAL restatement of PClean's inference loop on a dict-of-rows trace of plain strings — the reference's
own schedule, one row at a time, creation and garbage collection on the spot:

    initialize_trace           src/inference/inference.jl:3-58
    run_inference! / pgibbs_sweep!                inference.jl:60-88
    run_smc! (particles, blocks, resampling, final choice, commit)      row_inference.jl:108-187
    incorporate / unincorporate, reference counting, cascading deletion  model/dependency_tracking.jl:26-236
    Pitman-Yor hyper-parameter moves                                     model/trace.jl:65-108
    ProportionsParameter Gibbs move                                      distributions/choose_proportionally.jl:70-74
    evaluate_accuracy                                                    src/analysis.jl:36-88

It scores with oracle/literal.py (BlockProposal / LatentProposal: the model description walked on strings) and shares
NOTHING with the product's host code: no LoweredModel, no pclean_amd.trace / inference / parallel / analysis — the run
that tests/golden/literal_sequential.json records is an independent end-to-end reference for the F1 of the product's
sequential (batch_rows=1) and batched runs (its random numbers are its own: agreement is statistical, +-0.5 pt of F1,
never bit for bit).

Scope: hospital-shaped programs — reference slots, AddTypos observations (plain or through a JuliaNode), StringPrior /
ChooseUniformly / ChooseProportionally choices, one reference slot per block of the observed class, latent classes of
one block each.  Every latent choice of such a program is enumerated, so a particle's weight increment is its block's
log marginal and particles differ in their draws only.
"""
import math

import numpy as np

import literal as L

NEW = "NEW"


class LiteralSampler:
    def __init__(self, model, query, dirty, config, seed, restricted=False):
        """dirty: {column: [string | None per row]}; config: num_iters, num_particles, use_mh_instead_of_pg,
        rejuv_frequency (infer_config.jl:1-16)."""
        from pclean_amd.model import ChooseProportionally
        self.model, self.query, self.cfg, self.restricted = model, query, config, restricted
        self.rng = np.random.default_rng(seed)
        self.ocls = model.classes[query.cls]
        self.n = len(next(iter(dirty.values())))
        self.observed = [{query.obsmap[c]: dirty[c][i] for c in query.obsmap} for i in range(self.n)]
        self.tr = L.LitTrace(model)
        for cname in model.class_order:  # parameters from their priors (choose_proportionally.jl:48-55)
            for a in model.classes[cname].attrs:
                if a.kind == "choice" and isinstance(a.dist, ChooseProportionally):
                    prior = model.classes[cname].attr(a.dist.param).prior
                    self.tr.params[(cname, a.dist.param)] = self.rng.dirichlet(np.full(len(a.dist.options), prior.concentration))
        self.blocks = [b for b in self.ocls.blocks]
        self.slot_of_block = []
        for b in self.blocks:
            fks = [a for a in b if self.ocls.attr(a).kind == "fk"]
            assert len(fks) == 1, "one reference slot per block of the observed class"
            self.slot_of_block.append(fks[0])
        self.cur = [None] * self.n      # per observed row: {slot attribute: key}
        self.gensym = 0
        self.P = 2 if config.use_mh_instead_of_pg else config.num_particles

    # ---- the latent database -------------------------------------------------------------------------------------
    def _fresh_key(self):
        self.gensym += 1
        return f"row_{self.gensym}"

    def _snapshot(self, cls, key):
        """(key, row with nested snapshots) of row key and everything below it: what re-creates it if it is collected"""
        row = dict(self.tr.tables[cls][key])
        for a in self.model.classes[cls].attrs:
            if a.kind == "fk":
                row[a.name] = self._snapshot(a.target, row[a.name])
        return (key, row)

    def _refer(self, cls, spec):
        """refer_to_row! (dependency_tracking.jl:205-236).  spec = existing key | (key or None, {attr: string | spec}):
        a row to create (under the given key when it is a collected row coming back).  Returns the key."""
        if not isinstance(spec, tuple):
            self.tr.counts[cls][spec] += 1
            return spec
        key, row = spec
        if key is not None and key in self.tr.tables[cls]:  # (still there: somebody else kept it alive)
            self.tr.counts[cls][key] += 1
            return key
        key = key if key is not None else self._fresh_key()
        new = {}
        for a in self.model.classes[cls].attrs:
            if a.kind == "fk":
                new[a.name] = self._refer(a.target, row[a.name])
            elif a.kind == "choice":
                new[a.name] = row[a.name]
        self.tr.tables[cls][key] = new
        self.tr.counts[cls][key] = 1
        return key

    def _flat_spec(self, cls, spec, prefix):
        """{path: string} of the flattened values of an existing row or of a row to create"""
        if not isinstance(spec, tuple):
            out = {}
            row = self.tr.tables[cls][spec]
            for a in self.model.classes[cls].attrs:
                if a.kind == "choice":
                    out[prefix + a.name] = row[a.name]
                elif a.kind == "fk":
                    out.update(self._flat_spec(a.target, row[a.name], prefix + a.name + "."))
            return out
        out = {}
        for a in self.model.classes[cls].attrs:
            if a.kind == "choice":
                out[prefix + a.name] = spec[1][a.name]
            elif a.kind == "fk":
                out.update(self._flat_spec(a.target, spec[1][a.name], prefix + a.name + "."))
        return out

    # ---- sampling from an enumerated proposal -----------------------------------------------------------------------
    def _pick(self, keys, scores):
        m = max(scores)
        w = np.exp(np.asarray(scores) - m)
        return keys[int(self.rng.choice(len(keys), p=w / w.sum()))]

    def _sample_new(self, prop, cls, prefix):
        """the contents of a NEW row of cls at `prefix`, every own choice and nested slot from its conditional given the
        likelihood terms below it (the per-branch draws of proposal_compiler.jl:115-127, 233-245)"""
        row = {}
        for a in self.model.classes[cls].attrs:
            if a.kind == "choice":
                path = prefix + a.name
                terms = [t for t in prop.terms if path in t["paths"]]
                options, lps, dummy = L.discrete_proposal(self.tr, cls, a)
                vals = [dummy if o is None else o for o in options]
                sc = [lp + sum(prop._lik(t, {path: v}) for t in terms) for v, lp in zip(vals, lps)]
                row[a.name] = self._pick(vals, sc)
            elif a.kind == "fk":
                row[a.name] = self._sample_slot(prop, a.target, prefix + a.name + ".")
        return (None, row)

    def _sample_slot(self, prop, cls, prefix, scores=None):
        scores = prop._slot_scores(cls, prefix, {}) if scores is None else scores
        keys = list(scores)
        k = self._pick(keys, [scores[x] for x in keys])
        return self._sample_new(prop, cls, prefix) if k == NEW else k

    # ---- run_smc! for a row of the observed class --------------------------------------------------------------------
    def smc_observed(self, i):
        csmc = self.cur[i] is not None
        retained = None
        if csmc:  # unincorporate_row!: the row's references go, referents nobody else holds are collected
            retained = {s: self._snapshot(self.ocls.attr(s).target, k) for s, k in self.cur[i].items()}
            for s, k in self.cur[i].items():
                self.tr.unrefer(self.ocls.attr(s).target, k)
        P = self.P
        parts = [dict() for _ in range(P)]  # slot -> spec
        logw = np.zeros(P)
        memo = {}
        for b, attrs in enumerate(self.blocks):
            slot = self.slot_of_block[b]
            tgt = self.ocls.attr(slot).target
            for p in range(P):
                ctx = {}
                for s2, spec in parts[p].items():
                    ctx.update(self._flat_spec(self.ocls.attr(s2).target, spec, s2 + "."))
                ck = (b, tuple(sorted(ctx.items())))
                if ck not in memo:
                    prop = L.BlockProposal(self.tr, self.query, attrs, self.observed[i], ctx, self.restricted)
                    memo[ck] = (prop, prop.scores())
                prop, scores = memo[ck]
                logw[p] += L.logsumexp(list(scores.values()))
                if p == 0 and csmc:  # the retained particle keeps its referent (re-created if it was collected)
                    key, _ = retained[slot]
                    parts[p][slot] = key if key in self.tr.tables[tgt] else retained[slot]
                else:
                    parts[p][slot] = self._sample_slot(prop, tgt, "", scores)
            if not self.cfg.use_mh_instead_of_pg and b < len(self.blocks) - 1:  # maybe_resample, row_inference.jl:87-105
                w = np.exp(logw - logw.max())
                w /= w.sum()
                if 1.0 / np.sum(w * w) < P / 2:
                    idx = self.rng.choice(P, size=P, p=w)
                    if csmc:
                        idx[0] = 0
                    parts = [dict(parts[j]) for j in idx]
                    logw[:] = 0.0
        w = np.exp(logw - logw.max())
        w /= w.sum()
        if self.cfg.use_mh_instead_of_pg and csmc:
            chosen = 1 if self.rng.random() < min(1.0, w[1] / (1e-10 + w[0])) else 0
        else:
            chosen = int(self.rng.choice(P, p=w))
        self.cur[i] = {s: self._refer(self.ocls.attr(s).target, spec) for s, spec in parts[chosen].items()}

    # ---- run_smc! for a row of a latent class ---------------------------------------------------------------------------
    def _paths_to(self, cls):
        """(block index, slot, sub path) of every place class cls hangs below a slot of the observed class"""
        out = []

        def walk(c, b, slot, sub):
            if c == cls:
                out.append((b, slot, sub))
            for a in self.model.classes[c].attrs:
                if a.kind == "fk":
                    walk(a.target, b, slot, sub + a.name + ".")

        for b, slot in enumerate(self.slot_of_block):
            walk(self.ocls.attr(slot).target, b, slot, "")
        return out

    def _key_at(self, i, slot, sub):
        c, k = self.ocls.attr(slot).target, self.cur[i][slot]
        for part in [p for p in sub.split(".") if p]:
            k = self.tr.tables[c][k][part]
            c = self.model.classes[c].attr(part).target
        return k

    def smc_latent(self, cls, key):
        places = self._paths_to(cls)
        assert len(places) == 1, "a latent class below one slot"
        b, slot, sub = places[0]
        rows = [i for i in range(self.n) if self.cur[i] is not None and self._key_at(i, slot, sub) == key]
        evidence = []
        for i in rows:
            ctx = {}
            for s2, k2 in self.cur[i].items():
                if s2 != slot:
                    ctx.update(self._flat_spec(self.ocls.attr(s2).target, k2, s2 + "."))
            evidence.append((self.observed[i], ctx))
        cdef = self.model.classes[cls]
        row = self.tr.tables[cls][key]
        # unincorporate: the row's own references go (a referent nobody else holds is collected, and comes back if the
        # retained particle wins)
        retained = {a.name: self._snapshot(a.target, row[a.name]) for a in cdef.attrs if a.kind == "fk"}
        for a in cdef.attrs:
            if a.kind == "fk":
                self.tr.unrefer(a.target, row[a.name])
        prop = L.LatentProposal(self.tr, self.query, self.blocks[b], sub, evidence, self.restricted)
        fresh = {}
        for a in cdef.attrs:  # independent sub-plans (proposal_compiler.jl:363-388), each from its conditional
            if a.kind == "choice":
                sc = prop.leaf_scores(cls, a.name)
                vals = list(sc)
                fresh[a.name] = self._pick(vals, [sc[v] for v in vals])
            elif a.kind == "fk":
                fresh[a.name] = self._sample_slot(prop, a.target, sub + a.name + ".")
        # every particle carries the same weight (all sub-plans enumerated): the retained one is kept with probability
        # 1 / P under particle Gibbs; Metropolis-Hastings accepts the fresh one with min(1, w2 / (1e-10 + w1)), w1 = w2
        if self.cfg.use_mh_instead_of_pg:
            keep = not (self.rng.random() < min(1.0, 0.5 / (1e-10 + 0.5)))
        else:
            keep = int(self.rng.integers(self.P)) == 0
        for a in cdef.attrs:
            if a.kind == "choice" and not keep:
                row[a.name] = fresh[a.name]
            elif a.kind == "fk":
                spec = retained[a.name] if keep else fresh[a.name]
                if keep and spec[0] in self.tr.tables[a.target]:
                    spec = spec[0]
                row[a.name] = self._refer(a.target, spec)

    # ---- parameter moves ----------------------------------------------------------------------------------------------
    def resample_class(self, cls):
        """pgibbs_sweep!'s move (inference.jl:72-77): the class's learned parameters and its table's Pitman-Yor
        hyper-parameters"""
        from pclean_amd.model import ChooseProportionally
        cdef = self.model.classes[cls]
        for a in cdef.attrs:
            if a.kind == "choice" and isinstance(a.dist, ChooseProportionally):
                prior = cdef.attr(a.dist.param).prior
                counts = np.zeros(len(a.dist.options))
                index = {o: j for j, o in enumerate(a.dist.options)}
                for row in self.tr.tables[cls].values():
                    counts[index[row[a.name]]] += 1
                self.tr.params[(cls, a.dist.param)] = self.rng.dirichlet(prior.concentration + counts)
        if cls != self.query.cls:
            self._resample_py(cls)

    @staticmethod
    def _py_score(strength, discount, counts):  # trace.jl:65-78
        lp, n_ref = 0.0, 0
        for n_obj, size in enumerate(counts, start=1):
            lp += math.log(n_obj * discount + strength) - math.log(n_ref + strength)
            for j in range(1, size):
                lp += math.log(j - discount) - math.log(n_ref + j + strength)
            n_ref += size
        return lp

    def _resample_py(self, cls):  # trace.jl:80-108
        counts = list(self.tr.counts[cls].values())
        if not counts:
            return
        s, d = self.tr.py[cls]
        old = self._py_score(s, d, counts)
        s_new = self.rng.gamma(1.0, 1.0)
        new = self._py_score(s_new, d, counts)
        if math.log(self.rng.random()) < new + (-s) - old - (-s_new):  # logpdf(Gamma(1,1), x) = -x
            s, old = s_new, new
        d_new = self.rng.random()
        new = self._py_score(s, d_new, counts)
        if math.log(self.rng.random()) < new - old:
            d = d_new
        self.tr.py[cls] = (s, d)

    # ---- drivers -------------------------------------------------------------------------------------------------------
    def initialize(self):
        rf = max(int(self.cfg.rejuv_frequency), 1)
        for i in range(self.n):
            self.smc_observed(i)
            if (i + 1) % rf == 0:  # inference.jl:40-47: every class's parameters
                for c in self.model.class_order:
                    self.resample_class(c)

    def sweep(self):
        rf = max(int(self.cfg.rejuv_frequency), 1)
        for cls in self.model.class_order:
            if cls == self.query.cls:
                for i in range(self.n):
                    if i and i % rf == 0:
                        self.resample_class(cls)
                    self.smc_observed(i)
            else:
                for j, key in enumerate(list(self.tr.tables[cls])):
                    if j and j % rf == 0:
                        self.resample_class(cls)
                    if key in self.tr.tables[cls]:  # (collected meanwhile: nothing to rejuvenate)
                        self.smc_latent(cls, key)

    def run(self):
        self.initialize()
        for _ in range(self.cfg.num_iters):
            self.sweep()
        return self

    # ---- evaluate_accuracy (analysis.jl:36-88) ----------------------------------------------------------------------------
    def _row_value(self, i, ref):
        """value of `ref` in observed row i: a path below one of its reference slots, or a JuliaNode of the row"""
        if "." in ref:
            slot, rest = ref.split(".", 1)
            return self.tr.value(self.ocls.attr(slot).target, self.cur[i][slot], rest)
        a = self.ocls.attr(ref)
        assert a.kind == "julia", ref
        return a.fn(*[self._row_value(i, arg) for arg in a.args])

    def cleaned_value(self, i, col):
        return self._row_value(i, self.query.cleanmap[col])

    def accuracy(self, dirty, clean):
        errors = changed = cleaned = imputed = imputed_ok = 0
        for i in range(self.n):
            for col in clean:
                if col not in dirty:
                    continue
                d, c = dirty[col][i], clean[col][i]
                in_query = col in self.query.cleanmap
                if d is None:
                    if in_query and c is not None:
                        imputed += 1
                        imputed_ok += int(self.cleaned_value(i, col) == c)
                    continue
                errors += int(d != c)
                if in_query:
                    ours = self.cleaned_value(i, col)
                    if ours != d:
                        changed += 1
                        cleaned += int(ours == c)
        precision = (cleaned + imputed_ok) / max(changed + imputed, 1)
        recall = (cleaned + imputed_ok) / max(errors + imputed, 1)
        f1 = 0.0 if precision == 0 or recall == 0 else 2.0 / (1 / precision + 1 / recall)
        return dict(f1=f1, errors=errors, changed=changed, cleaned=cleaned, precision=precision, recall=recall,
                    imputed=imputed, correctly_imputed=imputed_ok)

    def latent_rows(self):
        return {c: len(t) for c, t in self.tr.tables.items() if c != self.query.cls}

    def check(self):
        """reference counts == number of referring slots (observed rows + latent rows)"""
        want = {c: {k: 0 for k in t} for c, t in self.tr.tables.items()}
        for cur in self.cur:
            if cur is not None:
                for s, k in cur.items():
                    want[self.ocls.attr(s).target][k] += 1
        for c, t in self.tr.tables.items():
            for row in t.values():
                for a in self.model.classes[c].attrs:
                    if a.kind == "fk":
                        want[a.target][row[a.name]] += 1
        for c in want:
            assert want[c] == {k: v for k, v in self.tr.counts[c].items()}, c

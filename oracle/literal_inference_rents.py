"""oracle/literal_inference_rents.py — TEST INFRASTRUCTURE ONLY.

The LITERAL, SEQUENTIAL sampler of oracle/literal_inference.py for rents-shaped programs (experiments/rents/run.jl): one
block holding a reference slot with noise-free observations of some of the referent's attributes and a typo-corrupted one,
own uniform choices that are enumerated INSIDE every candidate branch (room type when it is missing, the unit), and a
TransformedGaussian observation whose mean is a learned parameter indexed by (referent's attributes, own choice).  One row
at a time, creation and garbage collection on the spot; strings, floats and the model description only — none of the
product's lowering, plan arrays, trace, inference or analysis code.  Scores through oracle/literal.py (GaussBlockProposal).
Its runs (tests/golden/literal_sequential.json: "rents", "rents_pg20") are an independent end-to-end reference for
BASELINE.json configs[2]; its random numbers are its own.

    run_smc! for an observed row            src/inference/row_inference.jl:108-187 (every choice of the block enumerated:
                                            all particles carry the block's marginal; a chosen ProposalDummyValue adds
                                            -log(dummy mass) + the drawn string's likelihood - the placeholder's,
                                            block_proposal.jl:58-60)
    the enumeration                         proposal_compiler.jl:96-113 (own choices), 131-252 (slot), 277-293 (noise-free)
    latent County rows                      proposal_compiler.jl:306-350 (external likelihoods incl. the Gaussian one)
    MeanParameter Gibbs move                add_noise.jl:74-82;  ProportionsParameter: choose_proportionally.jl:70-74
    random(StringPrior)                     string_prior.jl:28-39
"""
import math

import numpy as np

import literal as L
from literal_inference import LiteralSampler, NEW


class RentsLiteralSampler(LiteralSampler):
    def __init__(self, model, query, dirty, config, seed):
        from pclean_amd.model import ChooseProportionally, TransformedGaussian
        self.model, self.query, self.cfg, self.restricted = model, query, config, False
        self.rng = np.random.default_rng(seed)
        self.ocls = model.classes[query.cls]
        self.n = len(next(iter(dirty.values())))
        self.dirty = dirty
        self.tr = L.LitTrace(model)
        for cname in model.class_order:  # parameters from their priors (choose_proportionally.jl:48-55)
            for a in model.classes[cname].attrs:
                if a.kind == "choice" and isinstance(a.dist, ChooseProportionally):
                    prior = model.classes[cname].attr(a.dist.param).prior
                    self.tr.params[(cname, a.dist.param)] = self.rng.dirichlet(np.full(len(a.dist.options), prior.concentration))
        assert len(self.ocls.blocks) == 1
        self.block = self.ocls.blocks[0]
        fks = [a for a in self.block if self.ocls.attr(a).kind == "fk"]
        assert len(fks) == 1
        self.slot = fks[0]
        self.tgt = self.ocls.attr(self.slot).target
        self.g = next(self.ocls.attr(a) for a in self.block if self.ocls.attr(a).kind == "choice"
                      and isinstance(self.ocls.attr(a).dist, TransformedGaussian))
        self.look = self.ocls.attr(self.g.dist.mean)
        self.mean_prior = self.ocls.attr(self.look.fn.param).prior
        self.own_names = [a for a in self.look.args if "." not in a]
        if self.g.dist.unit not in self.own_names:
            self.own_names.append(self.g.dist.unit)
        self.x_col = next(c for c, da in query.obsmap.items() if da == self.g.name)
        self.own_col = {da: c for c, da in query.obsmap.items() if da in self.own_names}
        self.means = {}                 # (lookup arguments) -> current value, created from the prior when first looked up
        self.cur = [None] * self.n      # {slot: key}
        self.own = [None] * self.n      # {own choice: option}
        self.gensym = 0
        self.P = 2 if config.use_mh_instead_of_pg else config.num_particles
        self.rows_of = {}               # key of a County row -> set of observed rows referring to it

    # ---- the learned means -----------------------------------------------------------------------------------------
    def mean_of(self, args):
        key = tuple(args[a] for a in self.look.args)
        if key not in self.means:
            self.means[key] = float(self.rng.normal(self.mean_prior.mean, self.mean_prior.std))
        return self.means[key]

    def _x(self, i):
        v = self.dirty[self.x_col][i]
        return None if v is None else float(v)

    def _gauss_at(self, i, below, own):
        """log density of row i's number given the referent's values `below` and the own choices `own`"""
        x = self._x(i)
        if x is None:
            return 0.0
        unit = own[self.g.dist.unit]
        args = {a: (below[a.split(".", 1)[1]] if "." in a else own[a]) for a in self.look.args}
        xb = unit.backward(x)
        return L.normal_logpdf(xb, self.mean_of(args), self.g.dist.std) - math.log(abs(unit.deriv(xb)))

    def _sample_own(self, i, below):
        """the row's own choices from their conditional given the chosen referent (proposal_compiler.jl:96-127): observed
        ones are fixed, the others proportional to uniform prior x Gaussian density"""
        names = self.own_names
        opts = []
        for nme in names:
            o = self.ocls.attr(nme).dist.options
            seen = self.dirty[self.own_col[nme]][i] if nme in self.own_col else None
            opts.append([seen] if seen is not None else list(o))
        combos = [dict()]
        for nme, o in zip(names, opts):
            combos = [dict(c, **{nme: v}) for c in combos for v in o]
        sc = [self._gauss_at(i, below, c) for c in combos]
        return combos[self._pick(list(range(len(combos))), sc)]

    def _random_string(self, d):  # string_prior.jl:28-39
        n = int(self.rng.integers(d.min_len, d.max_len + 1))
        out, prev = [], None
        for _ in range(n):
            p = L._INIT if prev is None else L._TRANS[:, prev]
            prev = int(self.rng.choice(len(p), p=p / p.sum()))
            out.append(L.ALPHABET[prev])
        return "".join(out)

    def _typo_lik(self, prop, path, value):
        return sum(L.add_typos_logpdf(v, value, mt) for p, v, mt in prop.typos if p == path and v is not None)

    def _new_spec(self, i, prop):
        """contents of a NEW referent for row i, every choice from its conditional (GaussBlockProposal._new restated as a
        sampler); returns (spec, weight correction of a chosen dummy)"""
        from pclean_amd.model import ChooseProportionally, StringPrior, Unmodeled
        cdef = self.model.classes[self.tgt]
        fixed = {p: v for p, v in prop.direct.items() if v is not None}
        gauss_paths = [a.split(".", 1)[1] for a in self.look.args if "." in a]
        open_paths = [p for p in gauss_paths if p not in fixed]
        row, corr = dict(fixed), 0.0
        for a in cdef.attrs:
            if a.kind != "choice" or isinstance(a.dist, Unmodeled) or a.name in fixed:
                continue
            d = a.dist
            if isinstance(d, StringPrior):
                atoms = d.atoms[row[d.keyed_by]] if d.keyed_by else d.atoms
                lps = [L.string_prior_logpdf(s_, d.min_len, d.max_len) for s_ in atoms]
                options = list(atoms) + [d.dummy_value()]
                lps = lps + [math.log1p(-math.exp(L.logsumexp(lps)))]
            elif isinstance(d, ChooseProportionally):
                probs = self.tr.params[(self.tgt, d.param)]
                options = list(d.options)
                lps = [math.log(p) if p > 0 else -math.inf for p in probs]
            else:
                raise NotImplementedError(type(d))
            sc = []
            for o, lp in zip(options, lps):
                s_ = lp + self._typo_lik(prop, a.name, o)
                if a.name in open_paths:
                    s_ += prop._gauss(dict(row, **{a.name: o}))
                sc.append(s_)
            j = self._pick(list(range(len(options))), sc)
            v = options[j]
            if isinstance(d, StringPrior) and v == d.dummy_value():  # block_proposal.jl:58-60
                v = self._random_string(d)
                corr += -lps[j] + self._typo_lik(prop, a.name, v) - self._typo_lik(prop, a.name, d.dummy_value())
            row[a.name] = v
        return (None, row), corr

    # ---- run_smc! for a row of the observed class ---------------------------------------------------------------------
    def smc_observed(self, i):
        csmc = self.cur[i] is not None
        retained = retained_own = None
        if csmc:
            key = self.cur[i][self.slot]
            retained = self._snapshot(self.tgt, key)
            retained_own = self.own[i]
            self.rows_of[key].discard(i)
            self._touch(self.tr.unrefer, self.tgt, key)
        row = {c: self.dirty[c][i] for c in self.query.obsmap}
        prop = L.GaussBlockProposal(self.tr, self.query, self.block, row, self.mean_of)
        scores = self._scores(prop)
        keys = list(scores)
        P = self.P
        specs, logw = [None] * P, np.zeros(P)
        for p in range(P):
            if p == 0 and csmc:
                k0, _ = retained
                specs[p] = k0 if k0 in self.tr.tables[self.tgt] else retained
                continue
            k = self._pick(keys, [scores[x] for x in keys])
            if k == NEW:
                specs[p], corr = self._new_spec(i, prop)
                logw[p] += corr
            else:
                specs[p] = k
        w = np.exp(logw - logw.max())
        w /= w.sum()
        if self.cfg.use_mh_instead_of_pg and csmc:
            chosen = 1 if self.rng.random() < min(1.0, w[1] / (1e-10 + w[0])) else 0
        else:
            chosen = int(self.rng.choice(P, p=w))
        key = self._touch(self._refer, self.tgt, specs[chosen])
        self.cur[i] = {self.slot: key}
        self.rows_of.setdefault(key, set()).add(i)
        if chosen == 0 and csmc:
            self.own[i] = retained_own
        else:
            self.own[i] = self._sample_own(i, self.tr.tables[self.tgt][key])

    def _scores(self, prop):
        """GaussBlockProposal.scores() over the rows that can match the noise-free observations only (a County of another key
        scores -inf: it takes no part in the log-sum-exp or the draw) — an index on the Unmodeled key, not a change of the
        scores"""
        cls = self.tgt
        s, d = self.tr.py[cls]
        counts = self.tr.counts[cls]
        tot = sum(counts.values())
        out = {}
        keyattr = next((p for p, v in prop.direct.items() if v is not None and p in self._index_attr()), None)
        cands = self.by_key.get(prop.direct[keyattr], ()) if keyattr else counts
        for k in cands:
            c = counts[k]
            out[k] = (math.log(c - d) - math.log(tot + s)) + prop._existing(cls, k)
        out[NEW] = (math.log(s + d * len(counts)) - math.log(tot + s)) + prop._new(cls)
        return out

    def _index_attr(self):
        from pclean_amd.model import Unmodeled
        return {a.name for a in self.model.classes[self.tgt].attrs if a.kind == "choice" and isinstance(a.dist, Unmodeled)}

    # (the index: rows of the slot's class by their Unmodeled key — kept by _refer / unrefer through the two overrides below)
    @property
    def by_key(self):
        idx = getattr(self, "_by_key", None)
        if idx is None or self._by_key_version != self._table_version:
            ka = next(iter(self._index_attr()))
            idx = {}
            for k, r in self.tr.tables[self.tgt].items():
                idx.setdefault(r[ka], []).append(k)
            self._by_key, self._by_key_version = idx, self._table_version
        return idx

    _table_version = 0

    def _touch(self, fn, *args):
        """run a call that may create or collect rows of the slot's class; the index above is rebuilt when it did"""
        before = set(self.tr.tables[self.tgt]) if len(self.tr.tables[self.tgt]) < 64 else None
        n0 = len(self.tr.tables[self.tgt])
        out = fn(*args)
        if len(self.tr.tables[self.tgt]) != n0 or (before is not None and before != set(self.tr.tables[self.tgt])):
            self._table_version += 1
        return out

    # ---- run_smc! for a row of the latent class -------------------------------------------------------------------------
    def smc_latent(self, cls, key):
        from pclean_amd.model import ChooseProportionally, StringPrior, Unmodeled
        assert cls == self.tgt
        rows = sorted(self.rows_of.get(key, ()))
        row = self.tr.tables[cls][key]
        cdef = self.model.classes[cls]
        pre = self.slot + "."
        typo_of = {}   # attribute -> (dirty column, max typos) of its AddTypos observation
        direct_col = {}  # attribute -> dirty column of its noise-free observation
        for col, da in self.query.obsmap.items():
            if da.startswith(pre):
                direct_col[da[len(pre):]] = col
            elif da in self.block:
                a = self.ocls.attr(da)
                if hasattr(a.dist, "ref") and a.dist.ref.startswith(pre):
                    typo_of[a.dist.ref[len(pre):]] = (col, a.dist.max_typos)
        gauss_paths = [a.split(".", 1)[1] for a in self.look.args if "." in a]
        P = self.P
        fresh, logw = [dict(row) for _ in range(P)], np.zeros(P)
        for a in cdef.attrs:
            if a.kind != "choice" or isinstance(a.dist, Unmodeled):
                continue
            d = a.dist
            if isinstance(d, StringPrior):
                atoms = d.atoms[row[d.keyed_by]] if d.keyed_by else d.atoms
                lps = [L.string_prior_logpdf(s_, d.min_len, d.max_len) for s_ in atoms]
                options = list(atoms) + [d.dummy_value()]
                lps = lps + [math.log1p(-math.exp(L.logsumexp(lps)))]
            elif isinstance(d, ChooseProportionally):
                options = list(d.options)
                lps = [math.log(p) if p > 0 else -math.inf for p in self.tr.params[(cls, d.param)]]
            else:
                raise NotImplementedError(type(d))

            def lik(v):
                s_ = 0.0
                if a.name in direct_col:
                    for i in rows:
                        o = self.dirty[direct_col[a.name]][i]
                        if o is not None and o != v:
                            return -math.inf
                if a.name in typo_of:
                    col, mt = typo_of[a.name]
                    for i in rows:
                        o = self.dirty[col][i]
                        if o is not None:
                            s_ += L.add_typos_logpdf(o, v, mt)
                if a.name in gauss_paths:
                    for i in rows:
                        s_ += self._gauss_at(i, dict(row, **{a.name: v}), self.own[i])
                return s_

            sc = [lp + lik(o) for o, lp in zip(options, lps)]
            logw += L.logsumexp(sc)
            is_sp = isinstance(d, StringPrior)
            if is_sp and row[a.name] not in options[:-1]:  # a string drawn for a dummy earlier stands for the dummy (49-52)
                logw[0] += -lps[-1] + lik(row[a.name]) - lik(d.dummy_value())
            for p in range(1, P):
                v = options[self._pick(list(range(len(options))), sc)]
                if is_sp and v == d.dummy_value():
                    v = self._random_string(d)
                    logw[p] += -lps[-1] + lik(v) - lik(d.dummy_value())
                fresh[p][a.name] = v
        w = np.exp(logw - logw.max())
        w /= w.sum()
        if self.cfg.use_mh_instead_of_pg:
            chosen = 1 if self.rng.random() < min(1.0, w[1] / (1e-10 + w[0])) else 0
        else:
            chosen = int(self.rng.choice(P, p=w))
        if chosen:
            row.update(fresh[chosen])

    # ---- parameter moves ----------------------------------------------------------------------------------------------
    def resample_class(self, cls):
        if cls != self.query.cls:
            return super().resample_class(cls)
        # add_noise.jl:74-82: every mean given the (back-transformed) numbers of the rows that look it up
        n, sm = {}, {}
        for i in range(self.n):
            if self.cur[i] is None or self._x(i) is None:
                continue
            below = self.tr.tables[self.tgt][self.cur[i][self.slot]]
            args = {a: (below[a.split(".", 1)[1]] if "." in a else self.own[i][a]) for a in self.look.args}
            key = tuple(args[a] for a in self.look.args)
            xb = self.own[i][self.g.dist.unit].backward(self._x(i))
            n[key] = n.get(key, 0) + 1
            sm[key] = sm.get(key, 0.0) + xb
        var0, sig2 = self.mean_prior.std ** 2, self.g.dist.std ** 2
        for key in sorted(set(self.means) | set(n), key=repr):
            var = 1.0 / (1.0 / var0 + n.get(key, 0) / sig2)
            mean = var * (self.mean_prior.mean / var0 + sm.get(key, 0.0) / sig2)
            self.means[key] = float(self.rng.normal(mean, math.sqrt(var)))

    # ---- evaluate_accuracy (analysis.jl:36-88; numbers compared as numbers) ---------------------------------------------
    def cleaned_value(self, i, col):
        ref = self.query.cleanmap[col]
        if "." in ref:
            slot, rest = ref.split(".", 1)
            return self.tr.value(self.ocls.attr(slot).target, self.cur[i][slot], rest)
        a = self.ocls.attr(ref)
        if a.kind == "choice":
            return self.own[i][ref]
        assert a.kind == "julia"
        vals = [self.own[i][arg] if arg in self.own_names else self._x(i) for arg in a.args]
        return a.fn(*vals)

    def accuracy(self, dirty, clean):
        errors = changed = cleaned = imputed = imputed_ok = 0
        for i in range(self.n):
            for col in clean:
                if col not in dirty:
                    continue
                d, c = dirty[col][i], clean[col][i]
                in_query = col in self.query.cleanmap
                numeric = in_query and self.query.cleanmap[col] not in self.own_names and "." not in self.query.cleanmap[col]
                if numeric:
                    d = None if d is None else float(d)
                    c = None if c is None else float(c)
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

    def check(self):
        want = {k: 0 for k in self.tr.tables[self.tgt]}
        for c in self.cur:
            if c is not None:
                want[c[self.slot]] += 1
        assert want == dict(self.tr.counts[self.tgt])
        for k, rows in self.rows_of.items():
            assert not rows or k in want
            assert all(self.cur[i][self.slot] == k for i in rows)

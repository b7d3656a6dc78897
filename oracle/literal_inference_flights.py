"""oracle/literal_inference_flights.py — TEST INFRASTRUCTURE ONLY.

The LITERAL, SEQUENTIAL sampler of oracle/literal_inference.py for flights-shaped programs (experiments/flights/run.jl):
reference slots whose only observations are noise-free ones of the referent's attributes, new rows whose other choices
are proposed from (keyed) prior proposals with a ProposalDummyValue, ONE block of MaybeSwap observations scored through a
learned per-website error probability, latent rows rejuvenated against the MaybeSwap evidence of every referring row, and
the Beta-Bernoulli Gibbs move of the error probabilities.  One row at a time, creation and garbage collection on the
spot (the reference's schedule); strings and the model description only — none of the product's lowering, plan arrays,
trace, inference or analysis code.  Its runs (tests/golden/literal_sequential.json: "flights") are an independent
end-to-end reference for the F1 of the product's sequential and batched runs; its random numbers are its own.

    run_smc! for an observed row            src/inference/row_inference.jl:108-187
      slot blocks (PriorSlotProposal)       proposal_compiler.jl:131-252, 277-293
      the MaybeSwap block: p += logdensity  block_proposal.jl:62-64; maybe_swap.jl:13-28
      a chosen ProposalDummyValue           block_proposal.jl:58-60 (weight - log(dummy mass), value = random(dist))
      maybe_resample / final choice         row_inference.jl:87-105, 152-165
    latent rows                             proposal_compiler.jl:306-350 (external likelihoods), inference.jl:60-81
    error probabilities                     maybe_swap.jl:36-52 (heads = the observation differs from the clean value)
    Pitman-Yor moves, evaluate_accuracy     literal_inference.py (shared: they do not depend on the program's shape)
"""
import math

import numpy as np

import literal as L
from literal_inference import LiteralSampler, NEW


class FlightsLiteralSampler(LiteralSampler):
    def __init__(self, model, query, dirty, config, seed, latent_dummy_correction=True):
        from pclean_amd.model import IndexedProbParameter, MaybeSwap
        # False: a latent row's particles all carry the enumeration's marginal — a chosen ProposalDummyValue keeps the
        # placeholder's likelihood in its weight (what the product's latent sweeps do)
        self.latent_dummy_correction = latent_dummy_correction
        self.model, self.query, self.cfg, self.restricted = model, query, config, False
        self.rng = np.random.default_rng(seed)
        self.ocls = model.classes[query.cls]
        self.n = len(next(iter(dirty.values())))
        self.dirty = dirty
        self.tr = L.LitTrace(model)
        self.cur = [None] * self.n
        self.gensym = 0
        self.P = 2 if config.use_mh_instead_of_pg else config.num_particles
        self.slot_blocks, self.score_attrs = [], []
        for b in self.ocls.blocks:
            fks = [a for a in b if self.ocls.attr(a).kind == "fk"]
            if fks:  # (a deterministic node may share the block: nothing is proposed or scored for it)
                assert len(fks) == 1 and all(self.ocls.attr(a).kind in ("fk", "julia") for a in b), "a slot block holds one reference slot"
                self.slot_blocks.append(fks[0])
            else:
                self.score_attrs += [a for a in b if self.ocls.attr(a).kind == "choice"]
        for a in self.score_attrs:
            assert isinstance(self.ocls.attr(a).dist, MaybeSwap)
        # dirty column of every observed attribute / noise-free observation
        self.direct = {s: {} for s in self.slot_blocks}   # slot -> {attribute of the referent: dirty column}
        self.col_of = {}                                   # MaybeSwap attribute -> dirty column
        for col, da in query.obsmap.items():
            if "." in da:
                slot, rest = da.split(".", 1)
                self.direct[slot][rest] = col
            else:
                self.col_of[da] = col
        par = [a for a in self.ocls.attrs if a.kind == "param" and isinstance(a.prior, IndexedProbParameter)]
        assert len(par) == 1
        self.prob_attr = par[0]
        self.tr.params[(query.cls, self.prob_attr.name)] = {}

    # ---- the learned error probabilities: created from the prior the first time a key is looked up ------------------
    def _prob(self, src, fid):
        ms = self.ocls.attr(self.score_attrs[0]).dist
        j = self.ocls.attr(ms.prob)
        r = j.fn.fn(src, fid)
        if isinstance(r, float):
            return r
        table = self.tr.params[(self.query.cls, self.prob_attr.name)]
        if r not in table:
            table[r] = float(self.rng.beta(self.prob_attr.prior.a, self.prob_attr.prior.b))
        return table[r]

    def _random_time(self):  # time_prior.jl:21-23
        return f"{self.rng.integers(1, 13)}:{self.rng.integers(1, 61)} {'a.m.' if self.rng.random() < 0.5 else 'p.m.'}"

    def _score_block(self, i, flat):
        """the MaybeSwap block of observed row i given the flattened values of a particle's referents"""
        total = 0.0
        fid = flat["flight.flight_id"]
        prob = self._prob(flat["src.name"], fid)
        for a in self.score_attrs:
            d = self.ocls.attr(a).dist
            total += L.maybe_swap_logpdf(self.dirty[self.col_of[a]][i], flat[d.val], d.options[flat[d.key]], prob)
        return total

    # ---- new rows ---------------------------------------------------------------------------------------------------
    def _new_spec(self, cls, fixed):
        """contents of a NEW row of cls: noise-free observed attributes take the observed value, every other choice is
        drawn from its (keyed) prior proposal; returns (spec, weight correction of the chosen dummies)"""
        row, corr = dict(fixed), 0.0
        for a in self.model.classes[cls].attrs:
            if a.kind != "choice" or a.name in row:
                continue
            options, lps = L.own_choice_proposal(self.tr, cls, a, row)
            w = np.exp(np.asarray(lps) - max(lps))
            j = int(self.rng.choice(len(options), p=w / w.sum()))
            v = options[j]
            if hasattr(a.dist, "dummy_value") and v == a.dist.dummy_value():
                corr += -lps[j]
                v = self._random_time()
            row[a.name] = v
        return (None, row), corr

    # ---- run_smc! for a row of the observed class ---------------------------------------------------------------------
    def smc_observed(self, i):
        csmc = self.cur[i] is not None
        retained = None
        if csmc:
            retained = {s: self._snapshot(self.ocls.attr(s).target, k) for s, k in self.cur[i].items()}
            for s, k in self.cur[i].items():
                self.tr.unrefer(self.ocls.attr(s).target, k)
        P = self.P
        parts = [dict() for _ in range(P)]
        logw = np.zeros(P)

        def resample():
            nonlocal parts
            if self.cfg.use_mh_instead_of_pg:
                return
            w = np.exp(logw - logw.max())
            w /= w.sum()
            if 1.0 / np.sum(w * w) < P / 2:
                idx = self.rng.choice(P, size=P, p=w)
                if csmc:
                    idx[0] = 0
                parts = [dict(parts[j]) for j in idx]
                logw[:] = 0.0

        for slot in self.slot_blocks:
            tgt = self.ocls.attr(slot).target
            direct = {attr: self.dirty[col][i] for attr, col in self.direct[slot].items()}
            scores = L.PriorSlotProposal(self.tr, tgt, direct).scores()
            lse = L.logsumexp(list(scores.values()))
            keys = list(scores)
            for p in range(P):
                logw[p] += lse
                if p == 0 and csmc:
                    key, _ = retained[slot]
                    parts[p][slot] = key if key in self.tr.tables[tgt] else retained[slot]
                    continue
                k = self._pick(keys, [scores[x] for x in keys])
                if k == NEW:
                    spec, corr = self._new_spec(tgt, {a: v for a, v in direct.items() if v is not None})
                    parts[p][slot] = spec
                    logw[p] += corr
                else:
                    parts[p][slot] = k
            resample()
        for p in range(P):
            flat = {}
            for slot, spec in parts[p].items():
                flat.update(self._flat_spec(self.ocls.attr(slot).target, spec, slot + "."))
            logw[p] += self._score_block(i, flat)
        if not np.isfinite(logw).any():
            logw[:] = 0.0
        w = np.exp(logw - logw.max())
        w /= w.sum()
        if self.cfg.use_mh_instead_of_pg and csmc:
            chosen = 1 if self.rng.random() < min(1.0, w[1] / (1e-10 + w[0])) else 0
        else:
            chosen = int(self.rng.choice(P, p=w))
        self.cur[i] = {s: self._refer(self.ocls.attr(s).target, spec) for s, spec in parts[chosen].items()}

    # ---- run_smc! for a row of a latent class ---------------------------------------------------------------------------
    def smc_latent(self, cls, key):
        slot = next(s for s in self.slot_blocks if self.ocls.attr(s).target == cls)
        rows = [i for i in range(self.n) if self.cur[i] is not None and self.cur[i][slot] == key]
        row = self.tr.tables[cls][key]
        cdef = self.model.classes[cls]
        # attributes the referring rows observe without noise: every referring row observed exactly this value (an existing
        # row must carry it, proposal_compiler.jl:277-293), so their enumeration has one live option — the value stays
        noise_free = set(self.direct[slot])
        swap_of = {self.ocls.attr(a).dist.val.split(".", 1)[1]: a for a in self.score_attrs
                   if self.ocls.attr(a).dist.val.startswith(slot + ".")}
        P = self.P
        fresh, logw = [dict(row) for _ in range(P)], np.zeros(P)   # (particle 0: the retained one, the row as it is)
        ev = []
        for i in rows:
            flat = {}
            for s2, k2 in self.cur[i].items():
                flat.update(self._flat_spec(self.ocls.attr(s2).target, k2, s2 + "."))
            ev.append((i, flat))
        for a in cdef.attrs:
            if a.kind != "choice" or a.name in noise_free or a.name not in swap_of:
                continue
            d = self.ocls.attr(swap_of[a.name]).dist
            col = self.col_of[swap_of[a.name]]
            options, lps = L.own_choice_proposal(self.tr, cls, a, row)

            def lik(v):
                return sum(L.maybe_swap_logpdf(self.dirty[col][i], v, d.options[flat[d.key]], self._prob(flat["src.name"], flat[d.key]))
                           for i, flat in ev)

            dummy = a.dist.dummy_value()
            sc = [lp + lik(o) for o, lp in zip(options, lps)]
            logw += L.logsumexp(sc)
            # retained particle: a value that is no option (a time drawn for a dummy earlier) stands for the dummy
            # (block_proposal.jl:49-52): its proposal probability is the dummy's, its likelihood its own
            if self.latent_dummy_correction and row[a.name] not in options[:-1]:
                logw[0] += -lps[-1] + lik(row[a.name]) - lik(dummy)
            pr = np.exp(np.asarray(sc) - max(sc))
            pr /= pr.sum()
            for p in range(1, P):
                v = options[int(self.rng.choice(len(options), p=pr))]
                if v == dummy:  # block_proposal.jl:58-60
                    v = self._random_time()
                    if self.latent_dummy_correction:
                        logw[p] += -lps[-1] + lik(v) - lik(dummy)
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
        if cls == self.query.cls:  # maybe_swap.jl:44-52: heads = the observation differs from the clean value
            table = self.tr.params[(cls, self.prob_attr.name)]
            heads, tails = {k: 0 for k in table}, {k: 0 for k in table}
            ms0 = self.ocls.attr(self.score_attrs[0]).dist
            j = self.ocls.attr(ms0.prob)
            for i in range(self.n):
                if self.cur[i] is None:
                    continue
                flat = {}
                for s2, k2 in self.cur[i].items():
                    flat.update(self._flat_spec(self.ocls.attr(s2).target, k2, s2 + "."))
                r = j.fn.fn(flat["src.name"], flat["flight.flight_id"])
                if isinstance(r, float):
                    continue
                for a in self.score_attrs:
                    d = self.ocls.attr(a).dist
                    o = self.dirty[self.col_of[a]][i]
                    if o is None:
                        continue
                    if o == flat[d.val]:
                        tails[r] = tails.get(r, 0) + 1
                    else:
                        heads[r] = heads.get(r, 0) + 1
            pr = self.prob_attr.prior
            for k in sorted(set(heads) | set(tails) | set(table)):  # (sorted: the draws must not depend on the hash seed)
                table[k] = float(self.rng.beta(pr.a + heads.get(k, 0), pr.b + tails.get(k, 0)))
        else:
            self._resample_py(cls)

"""oracle/literal.py — TEST INFRASTRUCTURE ONLY (never imported by pclean_amd, bench.py's timed region or the C ABI).

A second, LITERAL restatement of PClean's enumerated block proposal: it walks the MODEL DESCRIPTION (classes,
attributes, distributions, reference slots — pclean_amd.model's mirror of the `@model` DSL) and a dict-of-rows trace
holding plain Python strings, the way the reference's generated proposals do:

  * candidates of a reference slot = the rows of the target class + one "new row"     proposal_compiler.jl:131-252
  * CRP prior  log(count - d) - log(total + s),  new:  log(s + d K) - log(total + s)   proposal_compiler.jl:165-171,
                                                                                        model/trace.jl:53-61
  * `prob += logdensity(dist, observed, args...)` for every observed choice below it   proposal_compiler.jl:75
  * the new row's own choices enumerated over their discrete proposals, sibling
    sub-plans independently, log-marginals added                                        proposal_compiler.jl:55-129,363-388
  * StringPrior proposal = atoms + ProposalDummyValue with mass log1p(-exp(lse)); the
    dummy is scored against the placeholder string "*"^mid                              string_prior.jl:16-26
  * densities from the strings themselves: Damerau-Levenshtein + NegativeBinomial       add_typos.jl:50-66
    (scipy's nbinom = Distributions.jl's parametrisation), bigram StringPrior from the
    lmparams CSVs                                                                        string_prior.jl:43-61

It shares NOTHING with the product's lowering (LoweredModel's nodes / terms / colmap / pair tables / fn tables /
option tables): plans are derived here from the attribute references, JuliaNodes are evaluated by calling the
model's Python function on strings.  tests/golden/literal_scores.json holds its per-candidate scores for rows of
hospital_dirty.csv; the C++ oracle (CPU suite) and the HIP path (-m gpu) must reproduce them to 1e-12 relative.
Scope: programs built from reference slots, AddTypos, StringPrior (plain and keyed atoms), ChooseUniformly,
ChooseProportionally, JuliaNodes (hospital) and — GaussBlockProposal below — directly observed latent attributes, own
ChooseUniformly choices and a TransformedGaussian observation with an IndexedLookup mean (rents), — PriorSlotProposal / score_block — slots with nothing but noise-free observations,
TimePrior and a block of MaybeSwap observations (flights), and — LatentProposal — the rejuvenation of a LATENT row's
own choices / reference slots against its evidence set (ExternalLikelihoodNodes, hospital).
"""
import math
import os

import numpy as np
from scipy.stats import nbinom

HERE = os.path.dirname(os.path.abspath(__file__))
LM_DIR = os.path.join(os.path.dirname(HERE), "pclean_amd", "lmparams")
ALPHABET = [chr(c) for c in range(ord("a"), ord("z") + 1)] + [" ", "."]
IMPOSSIBLE = -1e5  # utils.jl


def _lm():
    init = np.loadtxt(os.path.join(LM_DIR, "letter_probabilities.csv"), delimiter=",").reshape(-1)
    trans = np.loadtxt(os.path.join(LM_DIR, "letter_transition_matrix.csv"), delimiter=",")
    return init, trans


_INIT, _TRANS = _lm()
_IDX = {c: i for i, c in enumerate(ALPHABET)}


def damerau_levenshtein(a, b, restricted=False):
    """Unrestricted (Lowrance-Wagner) or restricted (optimal string alignment) Damerau-Levenshtein distance."""
    la, lb = len(a), len(b)
    if restricted:
        d = [[0] * (lb + 1) for _ in range(la + 1)]
        for i in range(la + 1):
            d[i][0] = i
        for j in range(lb + 1):
            d[0][j] = j
        for i in range(1, la + 1):
            for j in range(1, lb + 1):
                c = 0 if a[i - 1] == b[j - 1] else 1
                v = min(d[i - 1][j] + 1, d[i][j - 1] + 1, d[i - 1][j - 1] + c)
                if i > 1 and j > 1 and a[i - 1] == b[j - 2] and a[i - 2] == b[j - 1]:
                    v = min(v, d[i - 2][j - 2] + 1)
                d[i][j] = v
        return d[la][lb]
    da = {}
    maxdist = la + lb
    H = [[0] * (lb + 2) for _ in range(la + 2)]
    H[0][0] = maxdist
    for i in range(la + 1):
        H[i + 1][0] = maxdist
        H[i + 1][1] = i
    for j in range(lb + 1):
        H[0][j + 1] = maxdist
        H[1][j + 1] = j
    for i in range(1, la + 1):
        db = 0
        for j in range(1, lb + 1):
            k = da.get(b[j - 1], 0)
            l = db
            cost = 1
            if a[i - 1] == b[j - 1]:
                cost = 0
                db = j
            H[i + 1][j + 1] = min(H[i][j] + cost, H[i + 1][j] + 1, H[i][j + 1] + 1,
                                  H[k][l] + (i - k - 1) + 1 + (j - l - 1))
        da[a[i - 1]] = i
    return H[la + 1][lb + 1]


_typo_memo = {}


def add_typos_logpdf(observed, word, max_typos=None, restricted=False):
    """add_typos.jl:50-66."""
    if observed is None:
        return 0.0
    key = (observed, word, max_typos, restricted)
    v = _typo_memo.get(key)
    if v is None:
        d = damerau_levenshtein(observed, word, restricted)
        if max_typos is not None and d > max_typos:
            v = IMPOSSIBLE
        else:
            v = float(nbinom.logpmf(d, math.ceil(len(word) / 5.0), 0.9))
            v -= math.log(len(word)) * d
            v -= math.log(26) * d / 2
        _typo_memo[key] = v
    return v


def string_prior_logpdf(s, min_len, max_len):
    """string_prior.jl:43-61."""
    if len(s) < min_len or len(s) > max_len:
        return -math.inf
    score = -math.log(max_len - min_len + 1)
    prev = None
    for ch in s:
        dist = _INIT if prev is None else _TRANS[:, prev]
        prev = _IDX.get(ch.lower())
        if prev is None:
            score += -math.log(28)
        else:
            p = dist[prev]
            score += max(math.log(p), -1000.0) if p > 0 else -1000.0
    return score


def logsumexp(xs):
    if not xs:
        return -math.inf
    m = max(xs)
    if m == -math.inf:
        return m
    return m + math.log(sum(math.exp(x - m) for x in xs))


class LitTrace:
    """Dict-of-rows database: tables[cls][key] = {attr: string | referenced key}; counts[cls][key]; (strength,
    discount) per class; params[(cls, param)] = probability vector aligned with the distribution's options."""

    def __init__(self, model):
        self.model = model
        self.tables = {c: {} for c in model.classes}
        self.counts = {c: {} for c in model.classes}
        self.py = {c: (model.classes[c].py_strength, model.classes[c].py_discount) for c in model.classes}
        self.params = {}

    def value(self, cls, key, path):
        """Follow a dotted path from row `key` of class cls; the last step is an own attribute."""
        row = self.tables[cls][key]
        c = self.model.classes[cls]
        parts = path.split(".")
        for p in parts[:-1]:
            a = c.attr(p)
            key = row[p]
            c = self.model.classes[a.target]
            row = self.tables[c.name][key]
        return row[parts[-1]]

    def unrefer(self, cls, key):
        """unrefer_to_row! (dependency_tracking.jl:162-201): drop one reference; a row nobody refers to any more is
        deleted and releases its own referents."""
        self.counts[cls][key] -= 1
        if self.counts[cls][key] == 0:
            row = self.tables[cls].pop(key)
            del self.counts[cls][key]
            for a in self.model.classes[cls].attrs:
                if a.kind == "fk":
                    self.unrefer(a.target, row[a.name])


def discrete_proposal(trace, cls, attr):
    """(options, log-probabilities, dummy string or None) of an own choice (distributions.jl:16)."""
    from pclean_amd.model import ChooseProportionally, ChooseUniformly, StringPrior
    d = attr.dist
    if isinstance(d, StringPrior):
        lps = [string_prior_logpdf(s, d.min_len, d.max_len) for s in d.atoms]
        total = logsumexp(lps)
        return list(d.atoms) + [None], lps + [math.log1p(-math.exp(total))], d.dummy_value()
    if isinstance(d, ChooseUniformly):
        return list(d.options), [-math.log(len(d.options))] * len(d.options), None
    if isinstance(d, ChooseProportionally):
        probs = trace.params[(cls, d.param)]
        return list(d.options), [math.log(p) if p > 0 else -math.inf for p in probs], None
    raise NotImplementedError(type(d))


class BlockProposal:
    """The enumerated proposal of one block of the observed class for one observed row."""

    def __init__(self, trace, query, block_attrs, observed, other_block_values, restricted=False):
        """observed: {dirty attribute name: string | None}; other_block_values: {path: string} for paths below the
        reference slots of OTHER blocks (values chosen earlier in the row: the JuliaNode arguments that are not
        below this block's slot)."""
        self.trace, self.model, self.query = trace, trace.model, query
        self.ocls = self.model.classes[query.cls]
        self.observed, self.ctx, self.restricted = observed, other_block_values, restricted
        fks = [a for a in block_attrs if self.ocls.attr(a).kind == "fk"]
        assert len(fks) == 1, "one reference slot per block"
        self.fk = self.ocls.attr(fks[0])
        # likelihood terms: (observed string, function of {path below the slot: string} -> latent word, max_typos, paths used)
        self.terms = []
        from pclean_amd.model import AddTypos
        for an in block_attrs:
            a = self.ocls.attr(an)
            if a.kind != "choice":
                continue
            assert isinstance(a.dist, AddTypos)
            self.terms.append(self._term(a))

    def _term(self, a):
        ref = a.dist.ref
        pre = self.fk.name + "."
        if ref.startswith(pre):
            path = ref[len(pre):]
            return dict(obs=self.observed[a.name], paths=[path], word=lambda vals, p=path: vals[p], max_typos=a.dist.max_typos)
        j = self.ocls.attr(ref)  # a JuliaNode of the row: its arguments are below this slot or another block's
        assert j.kind == "julia"
        mine = [arg[len(pre):] for arg in j.args if arg.startswith(pre)]

        def word(vals, j=j):
            args = [vals[arg[len(pre):]] if arg.startswith(pre) else self.ctx[arg] for arg in j.args]
            return j.fn(*args)

        return dict(obs=self.observed[a.name], paths=mine, word=word, max_typos=a.dist.max_typos)

    # -- helpers -------------------------------------------------------------------------------------------------
    def _crp(self, cls):
        s, d = self.trace.py[cls]
        counts = self.trace.counts[cls]
        total = sum(counts.values())
        return s, d, counts, total

    def _lik(self, term, vals):
        return add_typos_logpdf(term["obs"], term["word"](vals), term["max_typos"], self.restricted)

    def _flat(self, cls, key, prefix=""):
        """{path: string} of every own attribute reachable from row key (the flattened row)."""
        out = {}
        for a in self.model.classes[cls].attrs:
            if a.kind == "choice":
                out[prefix + a.name] = self.trace.tables[cls][key][a.name]
            elif a.kind == "fk":
                out.update(self._flat(a.target, self.trace.tables[cls][key][a.name], prefix + a.name + "."))
        return out

    def _terms_below(self, prefix):
        """terms all of whose own paths start with `prefix` (prefix '' = every term)."""
        return [t for t in self.terms if t["paths"] and all(p.startswith(prefix) for p in t["paths"])]

    def _new_marginal(self, cls, prefix, fixed):
        """log-marginal of a fresh row of cls reached through `prefix` (path below the slot, '' or 'loc.' ...):
        its own choices and nested reference slots are independent sub-plans, enumerated one by one.  `fixed` =
        values of paths outside this row that the terms may need (none in the programs covered)."""
        total = 0.0
        for a in self.model.classes[cls].attrs:
            if a.kind == "choice":
                path = prefix + a.name
                terms = [t for t in self.terms if path in t["paths"]]
                assert all(t["paths"] == [path] for t in terms), "a term below two sub-plans is not enumerable independently"
                options, lps, dummy = discrete_proposal(self.trace, cls, a)
                sc = []
                for o, lp in zip(options, lps):
                    vals = dict(fixed)
                    vals[path] = dummy if o is None else o
                    sc.append(lp + sum(self._lik(t, vals) for t in terms))
                total += logsumexp(sc)
            elif a.kind == "fk":
                sub = prefix + a.name + "."
                total += logsumexp(list(self._slot_scores(a.target, sub, fixed).values()))
        return total

    def _slot_scores(self, cls, prefix, fixed):
        """{key | 'NEW': score} of the candidates of a reference slot to class cls whose row sits at `prefix`."""
        s, d, counts, total = self._crp(cls)
        terms = self._terms_below(prefix)
        out = {}
        for key, c in counts.items():
            vals = dict(fixed)
            vals.update(self._flat(cls, key, prefix))
            out[key] = (math.log(c - d) - math.log(total + s)) + sum(self._lik(t, vals) for t in terms)
        out["NEW"] = (math.log(s + d * len(counts)) - math.log(total + s)) + self._new_marginal(cls, prefix, fixed)
        return out

    def scores(self):
        """{candidate key of the block's slot | 'NEW': log score}; the block's log-marginal is their logsumexp."""
        return self._slot_scores(self.fk.target, "", {})


# ---- rents-shaped blocks: reference slot + directly observed latent attributes + own uniform choices + Gaussian ----
def normal_logpdf(x, mean, std):
    z = (x - mean) / std
    return -0.5 * z * z - math.log(std) - 0.5 * math.log(2.0 * math.pi)


class GaussBlockProposal:
    """The enumerated proposal of a block `slot ~ Class; noisy ~ AddTypos(slot.attr); own ~ ChooseUniformly(...);
    x ~ TransformedGaussian(param[f(slot values, own choices)], std, unit)` (experiments/rents/run.jl:13-25) for one
    observed row.  Dirty columns bound to a latent path (`CountyKey => county.countykey`) are noise-free observations:
    an existing row must carry exactly that value, a new row takes it (proposal_compiler.jl:277-293).  The own
    choices that are not observed (the unit, a missing room type) are enumerated INSIDE every candidate branch
    (proposal_compiler.jl:96-113); each contributes its ChooseUniformly density (choose_uniformly.jl:7-10).

    mean_of(values: {lookup argument: string}) -> the MeanParameter's current value (add_noise.jl:15-21)."""

    def __init__(self, trace, query, block_attrs, row, mean_of):
        from pclean_amd.model import AddTypos, ChooseUniformly, TransformedGaussian
        self.trace, self.model, self.query, self.mean_of = trace, trace.model, query, mean_of
        self.ocls = self.model.classes[query.cls]
        fks = [a for a in block_attrs if self.ocls.attr(a).kind == "fk"]
        assert len(fks) == 1
        self.fk = self.ocls.attr(fks[0])
        pre = self.fk.name + "."
        self.typos, self.direct, self.own_obs = [], {}, {}
        for col, dirty_attr in query.obsmap.items():
            v = row[col]
            if dirty_attr.startswith(pre):          # noise-free observation of a value below the slot
                self.direct[dirty_attr[len(pre):]] = v
            elif dirty_attr in block_attrs:
                a = self.ocls.attr(dirty_attr)
                if isinstance(a.dist, AddTypos):
                    assert a.dist.ref.startswith(pre)
                    self.typos.append((a.dist.ref[len(pre):], v, a.dist.max_typos))
                elif isinstance(a.dist, ChooseUniformly):
                    self.own_obs[dirty_attr] = v     # observed own choice (None = missing)
                elif isinstance(a.dist, TransformedGaussian):
                    self.x = None if v is None else float(v)
                    self.g = a
        look = self.ocls.attr(self.g.dist.mean)
        self.look_args = list(look.args)
        self.own = [a for a in self.look_args if "." not in a]
        if self.g.dist.unit not in self.own:
            self.own.append(self.g.dist.unit)

    def _gauss(self, below):
        """log-sum over the unobserved own choices of: their uniform densities + the Gaussian density of x
        (transformed_gaussian.jl:15-16: logpdf(Normal(mean, std), t.backward(x)) - log|t.deriv(t.backward(x))|).
        `below` = {path below the slot: string} of the candidate."""
        def rec(i, vals, lp):
            if i == len(self.own):
                unit = vals[self.g.dist.unit]
                args = {a: (below[a.split(".", 1)[1]] if "." in a else vals[a]) for a in self.look_args}
                xb = unit.backward(self.x)
                return [lp + normal_logpdf(xb, self.mean_of(args), self.g.dist.std) - math.log(abs(unit.deriv(xb)))]
            name = self.own[i]
            opts = self.ocls.attr(name).dist.options
            seen = self.own_obs.get(name)
            out = []
            for o in opts:
                if seen is not None and o != seen:
                    continue
                out += rec(i + 1, dict(vals, **{name: o}), lp - math.log(len(opts)))
            return out
        return logsumexp(rec(0, {}, 0.0))

    def _existing(self, cls, key):
        row = self.trace.tables[cls][key]
        sc = 0.0
        for path, v in self.direct.items():
            if v is not None and row[path] != v:
                return -math.inf
        for path, v, mt in self.typos:
            if v is not None:
                sc += add_typos_logpdf(v, row[path], mt)
        return sc + self._gauss(row)

    def _new(self, cls):
        """fresh row: directly observed attributes take the observed value; every other own choice is enumerated over
        its discrete proposal — independent sub-plans multiply, the one the Gaussian depends on carries it."""
        from pclean_amd.model import ChooseProportionally, StringPrior, Unmodeled
        c = self.model.classes[cls]
        fixed = {p: v for p, v in self.direct.items() if v is not None}
        gauss_paths = [a.split(".", 1)[1] for a in self.look_args if "." in a]
        open_paths = [p for p in gauss_paths if p not in fixed]
        assert len(open_paths) <= 1
        total, gauss_done = 0.0, False
        for a in c.attrs:
            if a.kind != "choice":
                continue
            d = a.dist
            if isinstance(d, Unmodeled):
                assert a.name in fixed, "an Unmodeled attribute must be observed"
                continue
            if isinstance(d, StringPrior):
                atoms = d.atoms[fixed[d.keyed_by]] if d.keyed_by else d.atoms
                lps = [string_prior_logpdf(s_, d.min_len, d.max_len) for s_ in atoms]
                options = list(atoms) + [d.dummy_value()]
                lps = lps + [math.log1p(-math.exp(logsumexp(lps)))]
            elif isinstance(d, ChooseProportionally):
                probs = self.trace.params[(cls, d.param)]
                options = list(d.options)
                lps = [math.log(p) if p > 0 else -math.inf for p in probs]
            else:
                raise NotImplementedError(type(d))
            sc = []
            for o, lp in zip(options, lps):
                if a.name in self.direct and self.direct[a.name] is not None and o != self.direct[a.name]:
                    continue  # observed without noise: the other options are impossible
                s_ = lp
                for path, v, mt in self.typos:
                    if path == a.name and v is not None:
                        s_ += add_typos_logpdf(v, o, mt)
                if a.name in open_paths or (not open_paths and a.name == gauss_paths[0]):
                    s_ += self._gauss(dict(fixed, **{a.name: o}))
                    gauss_done = True
                sc.append(s_)
            total += logsumexp(sc)
        assert gauss_done
        return total

    def scores(self):
        cls = self.fk.target
        s, d = self.trace.py[cls]
        counts = self.trace.counts[cls]
        tot = sum(counts.values())
        out = {k: (math.log(c - d) - math.log(tot + s)) + self._existing(cls, k) for k, c in counts.items()}
        out["NEW"] = (math.log(s + d * len(counts)) - math.log(tot + s)) + self._new(cls)
        return out


# ---- flights-shaped blocks: slots with noise-free observations only, TimePrior, a block of MaybeSwap observations ----
import re

_TIME_RE = re.compile(r"^[0-9]?[0-9]:[0-9][0-9] [ap]\.m\.$")  # time_prior.jl:10


def maybe_swap_logpdf(observed, val, options, prob):
    """maybe_swap.jl:13-28 (a missing observation is an observed value)."""
    if observed is None:
        return 0.0 if val in options else -1000.0
    if val == observed:
        return math.log1p(-prob)
    return math.log(prob) - math.log(len(options))


def own_choice_proposal(trace, cls, attr, row_values):
    """(options, log-probabilities) of the discrete proposal of an own choice of a NEW row whose already fixed
    values are row_values (the key of keyed atoms)."""
    from pclean_amd.model import StringPrior, TimePrior
    d = attr.dist
    if isinstance(d, TimePrior):  # time_prior.jl:8-14
        atoms = d.atoms[row_values[d.keyed_by]]
        lps = [(-math.log(1440.0) if _TIME_RE.match(a) else -math.inf) for a in atoms]
        return list(atoms) + [d.dummy_value()], lps + [math.log1p(-math.exp(logsumexp(lps)))]
    if isinstance(d, StringPrior) and d.keyed_by:
        atoms = d.atoms[row_values[d.keyed_by]]
        lps = [string_prior_logpdf(a, d.min_len, d.max_len) for a in atoms]
        return list(atoms) + [d.dummy_value()], lps + [math.log1p(-math.exp(logsumexp(lps)))]
    options, lps, dummy = discrete_proposal(trace, cls, attr)
    return [dummy if o is None else o for o in options], lps


class PriorSlotProposal:
    """Block `slot ~ Class` whose only observations are noise-free ones of the class's attributes
    (`flight ~ Flight` with `flight => flight.flight_id`, experiments/flights/run.jl:24-26,43): existing rows must
    carry the observed values (proposal_compiler.jl:277-293), scored by the CRP prior alone; the new row takes the
    observed values — each at its prior density under its distribution's discrete proposal — and its other choices
    are enumerated with nothing to score them: their proposals' masses (which sum to 1)."""

    def __init__(self, trace, cls, direct):
        self.trace, self.model, self.cls, self.direct = trace, trace.model, cls, direct

    def scores(self):
        s, d = self.trace.py[self.cls]
        counts = self.trace.counts[self.cls]
        tot = sum(counts.values())
        out = {}
        for k, c in counts.items():
            row = self.trace.tables[self.cls][k]
            ok = all(v is None or row[a] == v for a, v in self.direct.items())
            out[k] = (math.log(c - d) - math.log(tot + s)) if ok else -math.inf
        new = math.log(s + d * len(counts)) - math.log(tot + s)
        fixed = {a: v for a, v in self.direct.items() if v is not None}
        for a in self.model.classes[self.cls].attrs:
            if a.kind != "choice":
                continue
            options, lps = own_choice_proposal(self.trace, self.cls, a, fixed)
            if a.name in fixed:
                new += logsumexp([lp for o, lp in zip(options, lps) if o == fixed[a.name]] or [-math.inf])
            else:
                new += logsumexp(lps)
        out["NEW"] = new
        return out


def score_block(trace, query, block_attrs, row, referents):
    """Block of observed choices without a latent choice (the four MaybeSwap observations of flights,
    run.jl:29-34): sum of logdensity(MaybeSwap, observed, val, options, prob) with val / options / prob read through
    the row's CURRENT referents; referents = {slot attribute: key}."""
    from pclean_amd.model import MaybeSwap
    model = trace.model
    ocls = model.classes[query.cls]

    def value(path):
        head, rest = path.split(".", 1)
        return trace.value(ocls.attr(head).target, referents[head], rest)

    total = 0.0
    for col, dirty_attr in query.obsmap.items():
        if dirty_attr not in block_attrs:
            continue
        a = ocls.attr(dirty_attr)
        assert isinstance(a.dist, MaybeSwap)
        j = ocls.attr(a.dist.prob)  # ProbLookup JuliaNode
        r = j.fn.fn(*[value(arg) for arg in j.args])
        prob = r if isinstance(r, float) else trace.params[(query.cls, j.fn.param)][r]
        total += maybe_swap_logpdf(row[col], value(a.dist.val), a.dist.options[value(a.dist.key)], prob)
    return total


# ---- latent-class rows: own choices and reference slots scored against the EVIDENCE SET ---------------------------------
class LatentProposal(BlockProposal):
    """Rejuvenation of a row of a latent class (pgibbs_sweep! over the class's own blocks, inference.jl:60-81) whose
    choices are observed only THROUGH the rows that refer to it: every likelihood term of every referring observed row
    (ExternalLikelihoodNodes, proposal_compiler.jl:306-350; block_proposal.jl:119-155) is summed.

    top_attrs: the observed class's block the latent class hangs below (`['hosp', 'service', ...]`);
    sub: the latent class's path below that block's reference slot ('' for the slot's own class, 'loc.', 'loc.county.');
    evidence: [(observed: {dirty attribute: string | None}, ctx: {path of another block: string})] per referring row."""

    def __init__(self, trace, query, top_attrs, sub, evidence, restricted=False):
        self.trace, self.model, self.query = trace, trace.model, query
        self.ocls = self.model.classes[query.cls]
        self.restricted, self.sub = restricted, sub
        fks = [a for a in top_attrs if self.ocls.attr(a).kind == "fk"]
        assert len(fks) == 1
        self.fk = self.ocls.attr(fks[0])
        from pclean_amd.model import AddTypos
        self.terms = []
        pre = self.fk.name + "."
        import copy
        for observed, ctx in evidence:
            row = copy.copy(self)  # (_term's closures read .observed / .ctx of the object they were made on: one per row)
            row.observed, row.ctx = observed, ctx
            for a in self.ocls.attrs:  # EVERY observed choice of the referring row that depends on a value below the slot,
                if a.kind != "choice" or not isinstance(a.dist, AddTypos):  # whatever block of the observed class it sits in
                    continue
                ref = a.dist.ref
                if "." in ref and not ref.startswith(pre):
                    continue  # the clean value lives below another slot
                t = row._term(a)
                if t["paths"]:
                    self.terms.append(t)

    def leaf_scores(self, cls, attr_name):
        """{option string: score} of the discrete proposal of the latent row's own choice (dummy under its string)."""
        a = self.model.classes[cls].attr(attr_name)
        path = self.sub + attr_name
        terms = [t for t in self.terms if path in t["paths"]]
        assert all(t["paths"] == [path] for t in terms)
        options, lps, dummy = discrete_proposal(self.trace, cls, a)
        return {(dummy if o is None else o): lp + sum(self._lik(t, {path: (dummy if o is None else o)}) for t in terms)
                for o, lp in zip(options, lps)}

    def slot_scores(self, attr_name):
        """{key | 'NEW': score} of the candidates of the latent row's reference slot (its own reference already
        removed from the trace by the caller)."""
        cls = self.sub_class()
        a = self.model.classes[cls].attr(attr_name)
        return self._slot_scores(a.target, self.sub + attr_name + ".", {})

    def sub_class(self):
        c = self.model.classes[self.fk.target]
        for part in [p for p in self.sub.split(".") if p]:
            c = self.model.classes[c.attr(part).target]
        return c.name


def lit_trace_from(lowered, trace):
    """LitTrace holding the rows of a pclean_amd Trace as strings (decoding only: latent_dom gives the string of a
    value index, layout names the columns — no plan arrays involved).  Keys are the product's row ids."""
    lw = lowered
    lt = LitTrace(lw.model)
    for cname, t in trace.tables.items():
        for k in range(t.n):
            if not t.live[k]:
                continue
            row = {}
            for j, col in enumerate(lw.layout[cname]):
                if "." in col.name:
                    continue
                if col.kind == "fk":
                    row[col.name] = int(t.cols[j, k])
                else:
                    row[col.name] = lw.latent_dom[(cname, col.name)].string(int(t.cols[j, k]))
            lt.tables[cname][k] = row
            lt.counts[cname][k] = int(t.counts[k])
        lt.py[cname] = (float(t.strength), float(t.discount))
    for (cname, pname), p in trace.params.items():
        lt.params[(cname, pname)] = np.asarray(p.value, dtype=np.float64)
    if getattr(lw, "prob_spec", None) is not None:  # Dict{String, ProbParameter}: key string -> current value
        pr = lw.prob_spec
        lt.params[pr["param"]] = {k: float(v) for k, v in zip(pr["keys"], trace.prob_param.value)}
    return lt

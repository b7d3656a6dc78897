"""Latent database state (host side) and its batched commit.

Mirrors src/model/trace.jl:24-50 (TableTrace / PCleanTrace) with flat arrays:
per latent class a column-major int32 table of flattened rows
(dependency_tracking.jl:88-96), int64 reference counts, Pitman–Yor parameters;
per observed row the current referent of each block's root reference slot.
`commit` restates incorporate_row!/unincorporate_row!/refer_to_row!/unrefer_to_row!
(dependency_tracking.jl:26-126,162-236) for a whole sweep at once: reference
counts move, new latent rows are created recursively, rows whose count reaches
zero are garbage-collected recursively, conjugate counts follow.
"""
import numpy as np

from .model import ChooseProportionally, ChooseUniformly, StringPrior

CHOICE_NEW = -1


class ProportionsState:
    """choose_proportionally.jl:36-74: Dirichlet-categorical parameter."""

    def __init__(self, n_options, concentration, rng):
        self.alpha = np.full(n_options, concentration, dtype=np.float64)
        self.counts = np.zeros(n_options, dtype=np.int64)
        self.value = rng.dirichlet(self.alpha)  # lazy init in the reference (48-55), eager here

    def resample(self, rng):  # resample_value!, 70-74
        self.value = rng.dirichlet(self.alpha + self.counts)


class MeanTableState:
    """IndexedParameter of MeanParameters (add_noise.jl:15-82, distributions.jl:45-55), dense over
    the index tuple; values are initialised eagerly from the prior (the reference does it lazily)."""

    def __init__(self, n, mean, std, sigma, rng):
        self.prior_mean, self.prior_std, self.sigma = mean, std, sigma
        self.value = rng.normal(mean, std, size=n)

    def resample(self, rng, idx, xs):
        """Gibbs update given observations xs (already transformed back) of entries idx (74-82)."""
        n = np.bincount(idx, minlength=len(self.value)).astype(np.float64)
        sm = np.bincount(idx, weights=xs, minlength=len(self.value))
        var0 = self.prior_std ** 2
        var = 1.0 / (1.0 / var0 + n / self.sigma ** 2)
        mean = var * (self.prior_mean / var0 + sm / self.sigma ** 2)
        self.value = rng.normal(mean, np.sqrt(var))


class ProbTableState:
    """Dict{String, ProbParameter{a, b}} (maybe_swap.jl:36-52): one Beta-Bernoulli error probability per key;
    heads = the observation differs from the clean value.  Eager initialisation from the prior."""

    def __init__(self, n_keys, a, b, rng):
        self.a, self.b = a, b
        self.value = rng.beta(a, b, size=n_keys)

    def resample(self, rng, heads, tails):  # resample_value!, maybe_swap.jl:50-52
        self.value = rng.beta(self.a + heads, self.b + tails)


def unique_rows(vals):
    """(index of the first occurrence of every distinct row of vals [k][c], group id of every row), groups numbered in
    order of first occurrence — what np.unique(vals, axis=0, return_index=True, return_inverse=True) gives after
    re-ordering its groups, without sorting the rows: one 64-bit hash per row, a 2^16-bucket histogram of its low bits
    (a row alone in its bucket is distinct from every other row), exact grouping of the few rows that share a bucket.
    Most proposals of a sweep are distinct, so almost nothing is ever sorted."""
    vals = np.ascontiguousarray(vals)
    k = len(vals)
    if k == 0:
        return np.zeros(0, np.int64), np.zeros(0, np.int64)
    mult = ((np.arange(1, vals.shape[1] + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) | np.uint64(1)).view(np.int64)
    h = (vals.astype(np.int64) @ mult).view(np.uint64)  # (integer matmul wraps modulo 2^64)
    h ^= h >> np.uint64(29)
    bucket = (h & np.uint64(0xFFFF)).astype(np.int64)
    crowded = np.bincount(bucket, minlength=1 << 16)[bucket] > 1
    is_first = ~crowded
    rep = np.arange(k, dtype=np.int64)  # index of the first occurrence of every row's group
    idx = np.flatnonzero(crowded)
    if len(idx):  # exact grouping of the rows whose bucket holds more than one row
        sub = vals[idx]
        _, f, inv = np.unique(h[idx], return_index=True, return_inverse=True)
        inv = np.asarray(inv).reshape(-1)
        if not np.array_equal(sub[f][inv], sub):  # two different rows share the 64-bit hash: exact grouping
            _, f, inv = np.unique(sub, axis=0, return_index=True, return_inverse=True)
            inv = np.asarray(inv).reshape(-1)
        rep[idx] = idx[f][inv]
        is_first[idx[f]] = True
    first = np.flatnonzero(is_first)
    rank = np.cumsum(is_first) - 1
    return first, rank[rep]


class LatentTable:
    def __init__(self, n_cols, strength=1.0, discount=0.0, cap=16):
        self.n_cols = n_cols
        self.cols = np.zeros((n_cols, cap), dtype=np.int32)
        self.counts = np.zeros(cap, dtype=np.int64)
        self.live = np.zeros(cap, dtype=bool)
        self.n = 0          # high-water mark
        self.free = []
        self.strength, self.discount = strength, discount
        self.cols_dirty = True  # value columns changed since the last upload to the device

    def alloc(self):
        if self.free:
            return self.free.pop()
        if self.n == self.cols.shape[1]:
            cap = max(16, 2 * self.n)
            self.cols = np.concatenate([self.cols, np.zeros((self.n_cols, cap - self.n), np.int32)], axis=1)
            self.counts = np.concatenate([self.counts, np.zeros(cap - self.n, np.int64)])
            self.live = np.concatenate([self.live, np.zeros(cap - self.n, bool)])
        self.n += 1
        return self.n - 1

    def alloc_many(self, k):
        """The ids k successive alloc() calls would return, without the Python call per row."""
        take = min(k, len(self.free))
        ids = np.empty(k, dtype=np.int64)
        if take:
            ids[:take] = self.free[:-take - 1:-1] if take < len(self.free) else self.free[::-1]
            del self.free[len(self.free) - take:]
        rest = k - take
        if rest:
            need = self.n + rest
            if need > self.cols.shape[1]:
                cap = max(16, 2 * self.n, need)
                grow = cap - self.cols.shape[1]
                self.cols = np.concatenate([self.cols, np.zeros((self.n_cols, grow), np.int32)], axis=1)
                self.counts = np.concatenate([self.counts, np.zeros(grow, np.int64)])
                self.live = np.concatenate([self.live, np.zeros(grow, bool)])
            ids[take:] = np.arange(self.n, need)
            self.n = need
        return ids

    def view(self):
        return self.cols[:, :self.n], self.counts[:self.n]

    @property
    def n_live(self):
        return int(np.sum(self.counts[:self.n] > 0))


def _device_synced(name):
    """Attribute `name` of a Trace whose arrays may be BEHIND the device (Engine.sweep_commit_device commits sweeps in
    HBM and leaves the host arrays alone): any access first pulls the device state (Engine.pull)."""
    private = "_" + name

    def get(self):
        if self._dev is not None:
            self._sync()
        if name == "cur":
            self._cur_version += 1  # (conservative: a reader may write through the array it gets)
        return getattr(self, private)

    def put(self, value):
        if self._dev is not None:
            self._sync()
        if name == "cur":
            self._cur_version += 1
        setattr(self, private, value)

    return property(get, put)


class Trace:
    tables = _device_synced("tables")
    params = _device_synced("params")
    row_origin = _device_synced("row_origin")
    locals = _device_synced("locals")
    cur = _device_synced("cur")

    def _sync(self):
        """Bring the host arrays up to date with the device-resident state (no-op when nothing is ahead)."""
        dev = self._dev
        if dev is not None:
            self._dev = None  # (first: the pull itself goes through these attributes)
            dev.pull(self)

    def __deepcopy__(self, memo):
        import copy
        self._sync()
        new = object.__new__(type(self))
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            # (the lowered model is the program, not trace state: shared, as every engine of the trace shares it)
            new.__dict__[k] = None if k == "_dev" else (v if k == "lw" else copy.deepcopy(v, memo))
        return new

    def __init__(self, lowered, n_rows, seed=0):
        self._dev = None       # the Engine whose device state is ahead of the host arrays (None: the host is current)
        self._cur_version = 0  # bumped by every host access to `cur` (the engine re-uploads the referents when it moved)
        self.n_rows = n_rows
        self.lw = lowered
        self.rng = np.random.default_rng(seed)
        m = lowered.model
        self.tables = {c: LatentTable(len(lowered.layout[c]), m.classes[c].py_strength, m.classes[c].py_discount)
                       for c in lowered.layout}
        self.params = {}
        for cname in m.class_order:
            for a in m.classes[cname].attrs:
                if a.kind == "choice" and isinstance(a.dist, ChooseProportionally):
                    prior = m.classes[cname].attr(a.dist.param).prior
                    self.params[(cname, a.dist.param)] = ProportionsState(len(a.dist.options), prior.concentration,
                                                                         self.rng)
        self._class_plans, self._node_plans = {}, {}
        # latent rows created by an observed-class sweep: (class, row) -> (creating observed row, chosen particle,
        # sweep index, block) — names the draw stream of the values sampled for their chosen ProposalDummyValues
        # (inference.resample_dummies; include/pclean_philox.h: pclean_dummy_seed)
        self.row_origin = {}
        self._origin = None
        self.cur = np.full((len(lowered.blocks), n_rows), -1, dtype=np.int32)
        # own enumerated choices of the observed class (e.g. br, unit) and the Gaussian mean parameter
        self.locals = {bi: np.full((n_rows, 2), -1, dtype=np.int32) for bi in getattr(lowered, "locals", {})}
        self.pending_locals = {}
        self.mean_param = None
        self.prob_param = None
        if lowered.prob_spec is not None:
            pp = m.classes[lowered.prob_spec["param"][0]].attr(lowered.prob_spec["param"][1]).prior
            self.prob_param = ProbTableState(len(lowered.prob_spec["keys"]), pp.a, pp.b, self.rng)
        spec = getattr(lowered, "gauss_spec", None)
        if spec is not None:
            prior = m.classes[spec["param"][0]].attr(spec["param"][1]).prior
            self.mean_param = MeanTableState(spec["n_mean"], prior.mean, prior.std, spec["sigma"], self.rng)

    def on_relower(self):
        """The lowered model was rebuilt in place with larger latent domains (LoweredModel.relower): drop the plans
        cached from its arrays; a keyed parameter table gains prior draws for the keys that appeared."""
        self._class_plans, self._node_plans = {}, {}
        for t in self.tables.values():
            t.cols_dirty = True
        pr = self.lw.prob_spec
        if pr is not None and self.prob_param is not None and len(pr["keys"]) > len(self.prob_param.value):
            extra = self.rng.beta(self.prob_param.a, self.prob_param.b, size=len(pr["keys"]) - len(self.prob_param.value))
            self.prob_param.value = np.concatenate([self.prob_param.value, extra])

    # -- MaybeSwap error probabilities -------------------------------------------------------------
    def _root_values(self, src):
        bi, col = src
        t = self.tables[self.lw.blocks[bi]["root_class"]]
        return t.cols[col, np.maximum(self.cur[bi], 0)]

    def prob_index(self):
        """Index into prob_table() of every observed row's error probability (the JuliaNode `error_prob`)."""
        pr = self.lw.prob_spec
        return self.lw.fn_tables[pr["fn"]][self._root_values(pr["a"]), self._root_values(pr["b"])]

    def prob_table(self):
        return np.concatenate([np.asarray(self.lw.prob_spec["consts"], dtype=np.float64), self.prob_param.value])

    def resample_prob_param(self):
        """heads/tails of every parameter key from the current trace (update_sufficient_statistics!
        of MaybeSwap, maybe_swap.jl:44-48), then the conjugate Beta draw."""
        lw = self.lw
        pr = lw.prob_spec
        nc = len(pr["consts"])
        pidx = self.prob_index()
        assigned = np.all(self.cur[[b for b, blk in enumerate(lw.blocks) if not blk.get("score")]] >= 0, axis=0)
        heads = np.zeros(len(pr["keys"]), dtype=np.int64)
        tails = np.zeros(len(pr["keys"]), dtype=np.int64)
        for sb in lw.score_blocks.values():
            for t in sb["terms"]:
                o = lw.obs_host[t["obs"]]
                v = self._root_values(t["val"])
                ok = assigned & (o >= 0) & (pidx >= nc)
                same = lw.same_pair_table(t["pair"])[np.maximum(o, 0), v] == 0
                heads += np.bincount(pidx[ok & ~same] - nc, minlength=len(heads))
                tails += np.bincount(pidx[ok & same] - nc, minlength=len(tails))
        self.prob_param.resample(self.rng, heads, tails)

    def commit_locals(self, begin=0, count=None):
        for bi, loc in self.pending_locals.items():
            n = len(loc) if count is None else count
            self.locals[bi][begin:begin + n] = loc[:n]
        self.pending_locals = {}

    def gaussian_index(self):
        """(rows, mean-table index, backward-transformed x) of every assigned observed row."""
        lw = self.lw
        spec = lw.gauss_spec
        bi = next(iter(lw.locals))
        rows = np.nonzero((self.cur[bi] >= 0) & (self.locals[bi][:, 0] >= 0))[0]
        root = lw.blocks[bi]["root_class"]
        t = self.tables[root]
        idx = np.zeros(len(rows), dtype=np.int64)
        for d, st in zip(spec["dims"], spec["strides"]):
            if d[0] == "cand":
                idx += st * t.cols[lw.colidx[root][d[1]], self.cur[bi][rows]]
            else:
                idx += st * self.locals[bi][rows, d[1]]
        x = lw.gauss_backward(rows, self.locals[bi][rows, spec["t_local"]])
        ok = ~np.isnan(x)
        return rows[ok], idx[ok], x[ok]

    # -- own-choice sufficient statistics (update_sufficient_statistics!, dependency_tracking.jl:6-21)
    def _class_plan(self, cname):
        """(direct reference-slot columns [(col, target class)], conjugate choices [(col, parameter state)])."""
        p = self._class_plans.get(cname)
        if p is None:
            lw = self.lw
            fks = [(j, c.target) for j, c in enumerate(lw.layout[cname]) if c.kind == "fk" and "." not in c.name]
            props = [(lw.colidx[cname][a.name], self.params[(cname, a.dist.param)])
                     for a in lw.model.classes[cname].attrs
                     if a.kind == "choice" and isinstance(a.dist, ChooseProportionally)]
            p = self._class_plans[cname] = (fks, props)
        return p

    def _own_choice_stats(self, cname, row, sign):
        t = self.tables[cname]
        for j, state in self._class_plan(cname)[1]:
            state.counts[t.cols[j, row]] += sign

    def _dedup_key(self, cname, values):
        return (cname,) + tuple(int(v) for v in values)

    def insert_row(self, cname, values):
        """Create one latent row with flattened `values` (FK columns hold target row
        ids whose counts are incremented) — refer_to_row!, dependency_tracking.jl:205-236."""
        t = self.tables[cname]
        r = t.alloc()
        t.cols[:, r] = values
        t.cols_dirty = True
        t.counts[r] = 0
        t.live[r] = True
        for j, target in self._class_plan(cname)[0]:
            self.tables[target].counts[values[j]] += 1
        self._own_choice_stats(cname, r, +1)
        if self._origin is not None:
            self.row_origin[(cname, int(r))] = self._origin
        else:
            self.row_origin.pop((cname, int(r)), None)
        return r

    def delete_row(self, cname, r):
        """unrefer_to_row! tail (dependency_tracking.jl:189-201): drop the row, release its referents."""
        t = self.tables[cname]
        assert t.counts[r] == 0 and t.live[r]
        self._own_choice_stats(cname, r, -1)
        t.live[r] = False
        t.free.append(int(r))
        for j, c in enumerate(self.lw.layout[cname]):
            if c.kind == "fk" and "." not in c.name:
                tgt = self.tables[c.target]
                k = int(t.cols[j, r])
                tgt.counts[k] -= 1
                if tgt.counts[k] == 0:
                    self.delete_row(c.target, k)

    # -- bulk variants (many rows of one class at once, numpy instead of a Python call per row) ------
    def insert_rows_bulk(self, cname, values):
        """insert_row for the rows of `values` [k][n_cols], in order; returns their row ids."""
        t = self.tables[cname]
        k = len(values)
        if k == 0:
            return np.zeros(0, dtype=np.int64)
        ids = t.alloc_many(k)
        t.cols[:, ids] = values.T
        t.cols_dirty = True
        t.counts[ids] = 0
        t.live[ids] = True
        fks, props = self._class_plan(cname)
        for j, target in fks:
            np.add.at(self.tables[target].counts, values[:, j], 1)
        for j, state in props:
            np.add.at(state.counts, values[:, j], 1)
        self._bulk_ids = ids
        return ids

    def delete_rows_bulk(self, cname, ids):
        """delete_row for every row of `ids` (all unreferenced and live), cascading to referents that
        lose their last reference."""
        ids = np.asarray(ids, dtype=np.int64)
        if len(ids) == 0:
            return
        t = self.tables[cname]
        assert np.all(t.counts[ids] == 0) and np.all(t.live[ids])
        fks, props = self._class_plan(cname)
        for j, state in props:
            np.subtract.at(state.counts, t.cols[j, ids], 1)
        t.live[ids] = False
        t.free.extend(ids.tolist())
        for j, target in fks:
            tgt = self.tables[target]
            ref = t.cols[j, ids]
            np.subtract.at(tgt.counts, ref, 1)
            gone = ref[(tgt.counts[ref] == 0) & tgt.live[ref]]  # (usually none: sort only those)
            if len(gone):
                self.delete_rows_bulk(target, np.unique(gone))

    def materialise_bulk(self, bi, vals, reuse=None, origin=None):
        """Rows of block bi's root class for the node choices vals [k][n_nodes] (row_inference.jl:169-185 for
        many rows).  Proposals without a nested NEW referent are built with array operations; the rest
        go through _materialise.  Returns the row ids, in the order of `vals`.

        reuse [k] (optional): per proposal a row of the class that is about to lose its last reference (the
        proposing row's old referent), -1 for none.  Where the new row's flattened values EQUAL that row's, the row
        is kept instead of being collected and re-created under another id — in the reference the observed row is
        unincorporated, its singleton referent garbage-collected and an identical row created under a fresh gensym
        key (row_inference.jl:115-126, 169-185): the same table up to the row's name.  It keeps the table's columns
        (and everything the device derives from them) unchanged when rows re-propose their own private referent.

        origin (optional): (creating observed rows [k], chosen particles [k], sweep index) — recorded for the rows
        created (row_origin) when the block has choices that may hold a ProposalDummyValue."""
        k = len(vals)
        out = np.empty(k, dtype=np.int64)
        if k == 0:
            return out
        track = origin is not None and self._block_has_dummy(bi)
        cname, n_cols, leaves, copies, slots = self._node_plan(bi, 0)
        nested = [cn for _, cn, _, _ in copies] + [cid for _, cid in slots]
        simple = np.ones(k, dtype=bool)
        for cn in set(nested):
            simple &= vals[:, cn] >= 0
        idx = np.flatnonzero(simple)
        if len(idx):
            v = vals[idx]
            values = np.zeros((len(idx), n_cols), dtype=np.int32)
            for j, cn, opts in leaves:
                values[:, j] = opts[v[:, cn]]
            for j, cn, ccls, cc in copies:
                values[:, j] = self.tables[ccls].cols[cc, v[:, cn]]
            for j, cid in slots:
                values[:, j] = v[:, cid]
            if reuse is not None:
                t = self.tables[cname]
                old = np.asarray(reuse, dtype=np.int64)[idx]
                same = old >= 0
                same[same] = np.all(t.cols[:, old[same]] == values[same].T, axis=0)
                out[idx[same]] = old[same]
                fresh = ~same
                out[idx[fresh]] = self.insert_rows_bulk(cname, values[fresh])
                made = idx[fresh]
            else:
                out[idx] = self.insert_rows_bulk(cname, values)
                made = idx
            if track:
                for i in made:
                    self.row_origin[(cname, int(out[i]))] = (int(origin[0][i]), int(origin[1][i]), int(origin[2]), bi)
        for i in np.flatnonzero(~simple):
            self._origin = (int(origin[0][i]), int(origin[1][i]), int(origin[2]), bi) if track else None
            out[i] = self._materialise(bi, 0, vals[i])
            self._origin = None
        return out

    def _block_has_dummy(self, bi):
        return any(n[0] == 1 and n[10] != 0 for n in self.lw.blocks[bi]["nodes"])

    # -- building a new row from the sampled node choices of a block ---------
    def _node_plan(self, bi, node):
        """Static recipe for building a row of `node`'s class from the sampled node choices:
        (class, n_cols, leaf columns [(col, node, option values)], copied columns [(col, child node,
        child class, child column)], reference-slot columns [(col, child node)])."""
        key = (bi, node)
        p = self._node_plans.get(key)
        if p is None:
            blk = self.lw.blocks[bi]
            nodes, info = blk["nodes"], blk["node_info"]
            cname = info[node]["cls"]
            layout = self.lw.layout[cname]
            cmb = nodes[node][9]
            leaves, copies, slots = [], [], []
            for j, c in enumerate(layout):
                cn, cc = blk["colmap"][2 * (cmb + j)], blk["colmap"][2 * (cmb + j) + 1]
                if cn >= 0:
                    if nodes[cn][0] == 1:  # leaf: option index -> latent value
                        leaves.append((j, cn, self.lw.option_values[(info[cn]["cls"], info[cn]["attr"])]))
                    else:
                        copies.append((j, cn, info[cn]["cls"], cc))
                if c.kind == "fk" and "." not in c.name:
                    for k in range(nodes[node][4], nodes[node][4] + nodes[node][5]):
                        cid = blk["children"][k]
                        if nodes[cid][0] == 0 and nodes[cid][7] == j:  # the direct child node of this slot
                            slots.append((j, cid))
            p = self._node_plans[key] = (cname, len(layout), leaves, copies, slots)
        return p

    def _materialise(self, bi, node, vals):
        cname, n_cols, leaves, copies, slots = self._node_plan(bi, node)
        values = np.zeros(n_cols, dtype=np.int32)
        child_rows = {}

        def child_row(cn):
            r = child_rows.get(cn)
            if r is None:
                r = int(vals[cn])
                if r == CHOICE_NEW:
                    r = self._materialise(bi, cn, vals)
                child_rows[cn] = r
            return r

        for j, cn, opts in leaves:
            values[j] = opts[vals[cn]]
        for j, cn, ccls, cc in copies:
            r = child_row(cn)  # may create the child (and grow its table) first
            values[j] = self.tables[ccls].cols[cc, r]
        for j, cid in slots:
            values[j] = child_row(cid)
        return self.insert_row(cname, values)

    def commit(self, choice, new_rows):
        """Apply one batched sweep: choice [n_blocks][n_rows] (row id or CHOICE_NEW),
        new_rows[b] = (rows, vals[n][n_nodes]).  Returns the number of changed referents."""
        changed = 0
        for bi, blk in enumerate(self.lw.blocks):
            if blk.get("score"):
                continue
            cname = blk["root_class"]
            t = self.tables[cname]
            ch = np.array(choice[bi], dtype=np.int64)
            rows_new, vals_new = new_rows.get(bi, (np.zeros(0, np.int32), None))
            for j, i in enumerate(rows_new):
                ch[i] = self._materialise(bi, 0, vals_new[j])
            cur = self.cur[bi]
            moved = np.nonzero(ch != cur)[0]
            changed += len(moved)
            if len(moved):
                olds = cur[moved]
                np.add.at(t.counts, ch[moved], 1)
                np.subtract.at(t.counts, olds[olds >= 0], 1)
                self.cur[bi, moved] = ch[moved]
                for k in np.unique(olds[olds >= 0]):
                    if t.counts[k] == 0 and t.live[k]:
                        self.delete_row(cname, int(k))
            # rows created but immediately unreferenced cannot happen: each new row has its creator
        return changed

    def check_consistency(self):
        """Reference counts == number of referring slots; flattened copies == referent's values;
        conjugate counts == live rows' choices.  Raises AssertionError with a description."""
        lw = self.lw
        want = {c: np.zeros(t.n, dtype=np.int64) for c, t in self.tables.items()}
        for bi, blk in enumerate(lw.blocks):
            if blk.get("score"):
                continue
            cur = self.cur[bi]
            cur = cur[cur >= 0]
            want[blk["root_class"]] += np.bincount(cur, minlength=self.tables[blk["root_class"]].n)
        for cname, t in self.tables.items():
            live = np.nonzero(t.live[:t.n])[0]
            for j, c in enumerate(lw.layout[cname]):
                if c.kind == "fk" and "." not in c.name:
                    tgt = self.tables[c.target]
                    ref = t.cols[j, live]
                    assert np.all(tgt.live[ref]), f"{cname}.{c.name} refers to a deleted {c.target} row"
                    want[c.target] += np.bincount(ref, minlength=tgt.n)
                    for jj, cc in enumerate(lw.layout[cname]):
                        if cc.name.startswith(c.name + "."):
                            sub = cc.name[len(c.name) + 1:]
                            assert np.array_equal(t.cols[jj, live], tgt.cols[lw.colidx[c.target][sub], ref]), \
                                f"{cname}.{cc.name} is not the flattened copy of its referent"
        for cname, t in self.tables.items():
            assert np.array_equal(want[cname], np.where(t.live[:t.n], t.counts[:t.n], want[cname])), \
                f"{cname}: reference counts differ from the referring slots"
            assert np.all((t.counts[:t.n] > 0) == t.live[:t.n]), f"{cname}: live flags differ from counts"
        for (cname, pname), p in self.params.items():
            t = self.tables[cname]
            live = np.nonzero(t.live[:t.n])[0]
            for a in lw.model.classes[cname].attrs:
                if a.kind == "choice" and isinstance(a.dist, ChooseProportionally) and a.dist.param == pname:
                    got = np.bincount(t.cols[lw.colidx[cname][a.name], live], minlength=len(p.counts))
                    assert np.array_equal(got, p.counts), f"{cname}.{pname}: Dirichlet counts out of sync"

    def commit_batch(self, begin, count, choice, new_rows, dedup=False, sweep_idx=0):
        """Commit a batch of observed rows [begin, begin+count) (initialize_trace: rows had no
        referent before).  With dedup, identical new-row proposals of the batch become one row —
        the sequential reference would have let the second row join the first row's new referent.
        Returns the number of latent rows of the blocks' root classes the batch created."""
        created = 0
        for bi, blk in enumerate(self.lw.blocks):
            if blk.get("score"):
                self.cur[bi, begin:begin + count] = 0
                continue
            cname = blk["root_class"]
            ch = np.array(choice[bi], dtype=np.int64)
            rows_new, vals_new = new_rows.get(bi, (np.zeros(0, np.int32), None))
            if len(rows_new):
                vals_new = np.array(vals_new)
                part = -1 - vals_new[:, 0]  # entry 0 of a record: -1 - chosen particle (pclean_get_new_rows)
                vals_new[:, 0] = CHOICE_NEW
                rows_g = np.asarray(rows_new, dtype=np.int64) + begin
                if dedup:  # one row per distinct proposal, created in order of first occurrence
                    first, grp = unique_rows(vals_new)
                    ch[rows_new] = self.materialise_bulk(bi, vals_new[first], origin=(rows_g[first], part[first], sweep_idx))[grp]
                    created += len(first)
                else:
                    ch[rows_new] = self.materialise_bulk(bi, vals_new, origin=(rows_g, part, sweep_idx))
                    created += len(vals_new)
            t = self.tables[cname]
            old = self.cur[bi, begin:begin + count]
            np.add.at(t.counts, ch, 1)
            np.subtract.at(t.counts, old[old >= 0], 1)
            self.cur[bi, begin:begin + count] = ch
            cand = np.unique(old[old >= 0])
            self.delete_rows_bulk(cname, cand[(t.counts[cand] == 0) & t.live[cand]])
        return created

    # -- parameter moves (inference.jl:72-77 -> resample_value!) ----------------
    def has_parameters(self, cname):
        """True when pgibbs_sweep!'s move every rejuv_frequency rows (inference.jl:72-77) has anything to resample for
        class cname: a learned parameter declared in it, or Pitman-Yor hyper-parameters of its own table (every
        latent class; the observed class has no table)."""
        return (cname in self._tables or any(c == cname for c, _ in self._params)  # (static keys: no device pull)
                or (self.prob_param is not None and self.lw.prob_spec["param"][0] == cname)
                or (self.mean_param is not None and self.lw.gauss_spec["param"][0] == cname))

    def has_learned_parameters(self, cname):
        """True when class cname declares a learned parameter (its value depends on the rows of the class, so the move
        of inference.jl:72-77 has to interleave with the row updates).  A class whose only "parameters" are its table's
        Pitman-Yor hyper-parameters is different: their conditional depends on the table's reference counts alone, and
        a sweep of the class's OWN rows changes neither those counts nor the set of rows."""
        return (any(c == cname for c, _ in self._params)
                or (self.prob_param is not None and self.lw.prob_spec["param"][0] == cname)
                or (self.mean_param is not None and self.lw.gauss_spec["param"][0] == cname))

    def resample_parameters(self, cname=None):
        """resample_value! of every learned parameter (cname=None: initialize_trace, inference.jl:40-47) or
        of the parameters declared in class cname (pgibbs_sweep!, inference.jl:72-77)."""
        for (c, _), p in self.params.items():
            if cname is None or c == cname:
                p.resample(self.rng)
        if self.prob_param is not None and (cname is None or self.lw.prob_spec["param"][0] == cname):
            self.resample_prob_param()
        if self.mean_param is not None and (cname is None or self.lw.gauss_spec["param"][0] == cname):
            _, idx, x = self.gaussian_index()
            self.mean_param.resample(self.rng, idx, x)

    @staticmethod
    def _py_prepare(counts):
        """The count-only pieces of pitman_yor_score (shared by the three evaluations of one hyper-parameter move):
        number of clusters K, number of customers N, and the counts-of-counts of the clusters with more than one customer
        (a table of 10^4 latent rows holds a few hundred distinct counts)."""
        counts = np.asarray(counts, dtype=np.int64)
        big = counts[counts > 1]
        c, h = np.unique(big, return_counts=True) if big.size else (np.zeros(0, np.int64), np.zeros(0, np.int64))
        return int(counts.size), float(counts.sum()), c.astype(np.float64), h.astype(np.float64)

    @staticmethod
    def _py_score_prepared(strength, discount, prep):
        """trace.jl:65-78 in closed form, O(number of DISTINCT counts).  The reference walks the clusters in table order:
        cluster j (1-based) starts with log(j d + s) - log(n + s) and its i-th join adds log(i - d) - log(n + i + s), n = the
        customers before it.  The denominators are log(n' + s) for n' = 0 .. N-1, each exactly once whatever the order:
        lgamma(N + s) - lgamma(s).  The joins of a cluster of size c: lgamma(c - d) - lgamma(1 - d), by counts-of-counts.
        The cluster starts: sum_j log(j d + s) = K log d + lgamma(K + 1 + s/d) - lgamma(1 + s/d) (d > 0), K log s (d = 0) —
        summed term by term when s/d is so large that the two log-gammas cancel badly."""
        from scipy.special import gammaln
        K, N, c, h = prep
        if K == 0:
            return 0.0
        if discount > 0.0 and strength / discount < 1e6:
            r = strength / discount
            lp = K * np.log(discount) + float(gammaln(K + 1.0 + r) - gammaln(1.0 + r))
        else:
            lp = float(np.sum(np.log(np.arange(1, K + 1, dtype=np.float64) * discount + strength)))
        if c.size:
            lp += float(np.sum(h * (gammaln(c - discount) - gammaln(1.0 - discount))))
        lp -= float(gammaln(N + strength) - gammaln(strength))
        return float(lp)

    @staticmethod
    def pitman_yor_score(strength, discount, counts):
        """trace.jl:65-78 (the score does not depend on the order of the clusters: see _py_score_prepared)."""
        return Trace._py_score_prepared(strength, discount, Trace._py_prepare(counts))

    def resample_py_params(self, t):
        """resample_py_params! (trace.jl:80-108): independence MH on strength ~ Gamma(1,1) and
        discount ~ U(0,1)."""
        counts = t.counts[:t.n][t.counts[:t.n] > 0]
        if counts.size == 0:
            return
        prep = self._py_prepare(counts)
        old = self._py_score_prepared(t.strength, t.discount, prep)
        s_new = self.rng.gamma(1.0, 1.0)
        new = self._py_score_prepared(s_new, t.discount, prep)
        # logpdf(Gamma(1,1), x) = -x
        alpha = new + (-t.strength) - old - (-s_new)
        if np.log(self.rng.random()) < alpha:
            t.strength, old = s_new, new
        d_new = self.rng.random()
        new = self._py_score_prepared(t.strength, d_new, prep)
        if np.log(self.rng.random()) < new - old:
            t.discount = d_new

    # -- initial state from known clean values (tests / synthetic bench) --------
    @classmethod
    def from_clean_values(cls, lowered, clean_by_path, n_rows, seed=0):
        """clean_by_path: {block index: {path below root: list of strings per row}}.
        Latent tables are the de-duplicated tuples; counts follow."""
        tr = cls(lowered, n_rows, seed)
        memo = {}

        def build(cname, getter):
            """row id of the (possibly new) latent row of class cname whose values come from getter(path)."""
            layout = lowered.layout[cname]
            values = np.zeros(len(layout), dtype=np.int32)
            for j, c in enumerate(layout):
                if c.kind == "fk" and "." not in c.name:
                    values[j] = build(c.target, lambda p, pre=c.name: getter(pre + "." + p))
            for j, c in enumerate(layout):
                if c.kind == "val":
                    if "." in c.name:
                        head, rest = c.name.split(".", 1)
                        values[j] = tr.tables[lowered.layout[cname][lowered.colidx[cname][head]].target].cols[
                            lowered.colidx[lowered.layout[cname][lowered.colidx[cname][head]].target][rest],
                            values[lowered.colidx[cname][head]]]
                    else:
                        values[j] = lowered.latent_dom[(cname, c.name)].index_of(getter(c.name))
                elif "." in c.name:  # nested fk id
                    head, rest = c.name.split(".", 1)
                    tgt = lowered.layout[cname][lowered.colidx[cname][head]].target
                    values[j] = tr.tables[tgt].cols[lowered.colidx[tgt][rest], values[lowered.colidx[cname][head]]]
            key = tr._dedup_key(cname, values)
            r = memo.get(key)
            if r is None:
                r = tr.insert_row(cname, values)
                memo[key] = r
            return r

        for bi, blk in enumerate(lowered.blocks):
            if blk.get("score"):
                tr.cur[bi] = 0
                continue
            cname = blk["root_class"]
            t = tr.tables[cname]
            paths = clean_by_path[bi]
            names = sorted(paths)
            cols = [paths[p] for p in names]
            root_memo = {}
            cur = np.empty(n_rows, dtype=np.int32)
            for i, key in enumerate(zip(*cols)):
                r = root_memo.get(key)
                if r is None:
                    vals = dict(zip(names, key))
                    r = build(cname, lambda p, vals=vals: vals[p])
                    root_memo[key] = r
                cur[i] = r
            tr.cur[bi] = cur
            t = tr.tables[cname]
            t.counts[:t.n] += np.bincount(cur, minlength=t.n)
        return tr

"""initialize_trace / run_inference! on the HIP path (src/inference/inference.jl:3-88).

Schedule differences from the reference: rows are processed in batches
against frozen tables and committed per (sub-)batch; a class sweep is cut into sub-batches of
max(rejuv_frequency, n / max_sub_batches) rows and the class's parameters are re-sampled between them
(the reference: every `rejuv_frequency` rows, inference.jl:72-77).

  initialize_trace : SMC over the observed rows (inference.jl:20-37) in geometrically
                     growing batches; within a batch identical new-row proposals are merged.
  run_inference    : for every class in class_order (latent classes first, observed class
                     last, inference.jl:62) one rejuvenation sweep: latent classes through
                     pclean_sweep_latent (external likelihood over referring rows),
                     the observed class through pclean_sweep.
"""
import os

import numpy as np

from .model import ChooseProportionally
from .parallel import Comm, exchange_and_commit, shard_bounds
from .trace import CHOICE_NEW


# ---------------------------------------------------------------------------
# wall-clock accounting of the host phases (bench.py / scripts print it; negligible cost)
import time as _time
from collections import defaultdict as _dd
from contextlib import contextmanager as _cm

TIMERS = _dd(float)
# PCLEAN_HOST_COMMIT=1: commit every observed-class sweep on the host (the path of several ranks, of refused device commits)
DEVICE_COMMIT = not os.environ.get("PCLEAN_HOST_COMMIT")
# observed rows from which a latent class's evidence sets are built on the device (Engine.build_evidence_device; below:
# NumPy on the host is as fast as the round trip; PCLEAN_DEVICE_EVIDENCE_MIN_ROWS overrides); PCLEAN_HOST_EVIDENCE=1 keeps the host
# path at every size
DEVICE_EVIDENCE_MIN_ROWS = int(os.environ.get("PCLEAN_DEVICE_EVIDENCE_MIN_ROWS", 1 << 17))


@_cm
def _timed(name):
    t0 = _time.perf_counter()
    try:
        yield
    finally:
        TIMERS[name] += _time.perf_counter() - t0


# ---------------------------------------------------------------------------
# evidence sets
def _follow(lw, trace, start_cls, keys, path):
    """Row ids reached from rows `keys` of start_cls along the reference-slot path."""
    cname = start_cls
    for step in [p for p in path.split(".") if p]:
        t = trace.tables[cname]
        j = lw.colidx[cname][step]
        keys = t.cols[j, keys]
        cname = lw.layout[cname][j].target
    return keys


def stable_argsort_ids(keys):
    """np.argsort(keys, kind="stable") for small integer ids (latent row ids, -1 = none): NumPy sorts 16-bit integers
    with a radix sort (4x faster than its merge sort on 1M int32 keys), so the ids are sorted as one or two 16-bit
    digits, least significant first."""
    keys = np.asarray(keys)
    if keys.size == 0:
        return np.zeros(0, dtype=np.int64)
    lo, hi = int(keys.min()), int(keys.max())
    if hi - lo < (1 << 16):
        return np.argsort((keys - lo).astype(np.uint16), kind="stable")
    if hi - lo < (1 << 32):
        k = (keys.astype(np.int64) - lo).astype(np.uint32)
        o1 = np.argsort((k & 0xffff).astype(np.uint16), kind="stable")
        return o1[np.argsort((k >> 16).astype(np.uint16)[o1], kind="stable")]
    return np.argsort(keys, kind="stable")


def build_evidence(lw, trace, cname, argsort=None):
    """CSR of observed rows referring (transitively) to each live row of latent class cname,
    plus the per-evidence-row ctx value of the cross-block JuliaNode terms.  argsort(ids, id_max): a stable argsort of small
    ids (the engine's device radix sort, pclean_argsort_ids, from 2^17 rows on; default: NumPy's, stable_argsort_ids)."""
    pl = lw.latent_plans[cname]
    bi = pl["src_block"]
    root_cls = lw.blocks[bi]["root_class"]
    t = trace.tables[cname]
    keys = _follow(lw, trace, root_cls, trace.cur[bi], pl["path"])
    if argsort is not None and len(keys) >= (1 << 17):
        order = argsort(keys, t.n)
    else:
        order = stable_argsort_ids(keys).astype(np.int32)
    counts = np.bincount(keys, minlength=t.n)
    live = np.nonzero(t.live[:t.n])[0].astype(np.int32)
    off_all = np.zeros(t.n + 1, dtype=np.int64)
    np.cumsum(counts, out=off_all[1:])
    # evidence rows grouped by latent row, restricted to live rows (dead rows have none anyway)
    ev_off = np.zeros(len(live) + 1, dtype=np.int32)
    np.cumsum(counts[live], out=ev_off[1:])
    # `order` is grouped by latent row in ascending id; keep the groups of live rows (vectorised)
    if not counts[:t.n][~t.live[:t.n]].any():  # the consistent case: only live rows are referred to -> every row stays
        ev_rows = order
    else:
        sk = keys[order]
        ev_rows = order[(sk >= 0) & t.live[np.maximum(sk, 0)]].astype(np.int32)
    ev_ctx = None
    sources = pl.get("ctx_sources", [])
    if sources and (cname in lw.latent_ev_prob or cname in getattr(lw, "latent_ev_locals", {})):
        raise NotImplementedError(f"latent class {cname}: JuliaNode contexts together with MaybeSwap / Gaussian evidence contexts")
    if sources:  # slot s = the evidence row's value of source s (model.py: _build_latent_plans / _copy_subtree)
        from . import _lib
        assert len(sources) <= _lib.MAX_CTX
        ev_ctx = np.zeros((len(ev_rows), _lib.MAX_CTX), dtype=np.int32)  # (the width the library indexes)
        for s_, (ob, col) in enumerate(sources):
            rc = lw.blocks[ob]["root_class"]
            ev_ctx[:, s_] = trace.tables[rc].cols[col, trace.cur[ob]][ev_rows]
    if cname in lw.latent_ev_prob:  # MaybeSwap external likelihood: the rows' error-probability index
        ev_ctx = np.zeros((len(ev_rows), 2), dtype=np.int32)
        ev_ctx[:, 0] = trace.prob_index()[ev_rows]
    if cname in getattr(lw, "latent_ev_locals", {}):  # Gaussian external likelihood: the rows' own choices
        ev_ctx = np.ascontiguousarray(trace.locals[lw.latent_ev_locals[cname]][ev_rows]).astype(np.int32)
    return live, ev_off, ev_rows, ev_ctx


# ---------------------------------------------------------------------------
def refresh_flattened(lw, trace):
    """Re-copy the inlined (flattened) values of every reference slot from its referent —
    update_referring_rows_with_new_values_for_updated_row! (dependency_tracking.jl:239-257),
    done for whole tables in class order (targets before sources)."""
    for cname in lw.model.class_order:
        if cname not in trace.tables:
            continue
        t = trace.tables[cname]
        if t.n == 0:
            continue
        for j, c in enumerate(lw.layout[cname]):
            if c.kind == "fk" and "." not in c.name:
                tgt = trace.tables[c.target]
                ref = t.cols[j, :t.n]
                for jj, cc in enumerate(lw.layout[cname]):
                    if cc.name.startswith(c.name + "."):
                        sub = cc.name[len(c.name) + 1:]
                        new = tgt.cols[lw.colidx[c.target][sub], ref]
                        if not np.array_equal(new, t.cols[jj, :t.n]):
                            t.cols[jj, :t.n] = new
                            t.cols_dirty = True


def _after_commit(engine, trace, seed):
    """what follows every commit: chosen dummy values get their prior draw (resample_dummies)"""
    trace.dummy_stamp = getattr(trace, "dummy_stamp", 0) + 1
    return resample_dummies(engine, trace, seed, trace.dummy_stamp)


def _count_dummy_cases(lw, trace, todo, obs):
    """Diagnostics of resample_dummies (trace.dummy_cases): how many of the chosen dummies had NO observation of their
    attribute among the rows referring to the latent row (the sweep's particle weight is the reference's, bit for bit)
    and how many had one in the block that enumerated them (the sweep's weight used the placeholder's likelihood where
    the reference re-scores the drawn string — DESIGN.md §11).  Observations scored in a LATER block (flights'
    MaybeSwap block) count as unobserved: that block re-scores whatever string the row holds."""
    m, q = lw.model, lw.query
    ocls = m.classes[q.cls]
    cases = getattr(trace, "dummy_cases", None)
    if cases is None:
        cases = trace.dummy_cases = {"unobserved": 0, "observed": 0}
    block_of = {a: bi for bi, b in enumerate(ocls.blocks) for a in b}
    for cname, j, an, d, rows, stream in todo:
        cols = []
        for col, ref in q.cleanmap.items():  # dirty columns whose clean value is this attribute, scored in its slot's block
            if "." not in ref:
                continue
            head, rest = ref.split(".", 1)
            try:
                cn, la = m.resolve(ocls.attr(head).target, rest)
            except (KeyError, ValueError):
                continue
            if (cn, la.name) == (cname, an) and block_of.get(q.obsmap[col]) == block_of.get(head):
                cols.append(q.obsmap[col])
        if not cols or cname not in lw.latent_plans:
            cases["unobserved"] += len(rows)
            continue
        pl = lw.latent_plans[cname]
        bi = pl["src_block"]
        assigned = np.flatnonzero(trace.cur[bi] >= 0)  # (during the initialisation later rows have no referent yet)
        keys = _follow(lw, trace, lw.blocks[bi]["root_class"], trace.cur[bi][assigned], pl["path"])
        for r in rows:
            er = assigned[keys == r]
            seen = any((obs[lw.obs_index[c], er] >= 0).any() for c in cols if c in lw.obs_index)
            cases["observed" if seen else "unobserved"] += 1


def _leaf_node_of(lw, bi, cname, aname):
    """node id of the option list of attribute cname.aname in observed-class block bi (None if it has none)"""
    for nid, info in enumerate(lw.blocks[bi]["node_info"]):
        if info["kind"] == "leaf" and info["cls"] == cname and info["attr"] == aname:
            return nid
    return None


def resample_dummies(engine, trace, seed, stamp):
    """block_proposal.jl:58-60: a RandomChoiceNode whose enumerated proposal chose the ProposalDummyValue gets
    `random(node.dist, args...)` — a string from the bigram StringPrior / a random TimePrior time — as its value.
    The sweeps keep the placeholder while they run (its mass and its likelihood are what the enumeration scored);
    this step, run after every commit, replaces the placeholders of the rows that were just created: the strings
    are drawn by the device samplers (counter = (seed, class / attribute / stamp, i): identical on every rank),
    appended to the attribute's latent domain (LoweredModel.relower: pair / fn / equality tables grow a value) and
    the engine reloads its static data.  Rare: the dummy wins only for unobserved or very short strings.
    Particle weights: with no observation below the node the reference's weight p - q equals the block marginal up
    to -log(dummy mass) ~ 1e-40, i.e. bit-identical in fp64; with an observation below it the reference re-scores
    the sampled string while the sweep's weight used the placeholder's likelihood.
    Returns the number of values replaced."""
    from .model import StringPrior, TimePrior
    from .sampling import dummy_seed
    lw = engine.lw
    m = lw.model
    run_seed = int(seed)
    todo = []
    for ci, cname in enumerate(m.class_order):
        t = trace.tables.get(cname)
        if t is None or t.n == 0:
            continue
        for j, col in enumerate(lw.layout[cname]):
            if col.kind != "val" or "." in col.name:
                continue
            d = m.classes[cname].attr(col.name).dist
            if not isinstance(d, (StringPrior, TimePrior)):
                continue
            dummy = lw.latent_dom[(cname, col.name)].get(d.dummy_value())
            rows = np.flatnonzero((t.cols[j, :t.n] == dummy) & t.live[:t.n])
            if len(rows):
                todo.append((cname, j, col.name, d, rows, (ci * 64 + j) * 65536 + (stamp & 0xffff)))
    # the 32-bit Philox stream holds (class, attribute, low 16 bits of the commit number); the commit number's high
    # bits go into the key, so that runs of more than 65 536 commits (the sequential schedule) never reuse a counter
    seed = (int(seed) + (int(stamp) >> 16) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    if not todo:
        return 0
    _count_dummy_cases(lw, trace, todo, engine.obs)
    drawn = []
    for cname, j, an, d, rows, stream in todo:
        strings = engine.sample_prior_strings(d, len(rows), seed, stream)
        if isinstance(d, StringPrior) and hasattr(engine, "sample_prior_strings_at"):
            # rows created by an observed-class sweep: the string the sweep already drew for the creating particle's
            # weight (private stream pclean_dummy_seed(seed, site of the option list, particle, sweep) at the creating row)
            seeds, elems, where = [], [], []
            for i, r in enumerate(rows):
                org = trace.row_origin.get((cname, int(r)))
                if org is None:
                    continue
                row_o, particle, sweep_idx, bi = org
                node = _leaf_node_of(lw, bi, cname, an)
                if node is None:
                    continue
                seeds.append(dummy_seed(run_seed, (bi << 16) | node, particle, sweep_idx))
                elems.append(row_o)
                where.append(i)
            if where:
                for i, s_ in zip(where, engine.sample_prior_strings_at(d, seeds, elems)):
                    strings[i] = s_
        drawn.append(strings)
    # a draw that happens to be one of the row's OWN proposal atoms is that option; anything else is a value outside
    # the options (MaybeSwap asks `val in options`, maybe_swap.jl:18) and gets an id after the dummy's — even when the
    # string equals an atom listed under another key
    own, extras = [], {}
    for (cname, j, an, d, rows, stream), strings in zip(todo, drawn):
        t = trace.tables[cname]
        if getattr(d, "keyed_by", None):
            kdom = lw.latent_dom[(cname, d.keyed_by)]
            keys = [kdom.string(int(v)) for v in t.cols[lw.colidx[cname][d.keyed_by], rows]]
            is_own = [s_ in d.atoms.get(k, ()) for s_, k in zip(strings, keys)]
        else:
            atoms = set(d.atoms)
            is_own = [s_ in atoms for s_ in strings]
        own.append(is_own)
        extras[(cname, an)] = [s_ for s_, o in zip(strings, is_own) if not o]
    lw.relower(extras)
    engine.reload()
    trace.on_relower()
    n = 0
    for (cname, j, an, d, rows, stream), strings, is_own in zip(todo, drawn, own):
        dom = lw.latent_dom[(cname, an)]
        t = trace.tables[cname]
        t.cols[j, rows] = [dom.index_of(s_) if o else dom.extra[s_] for s_, o in zip(strings, is_own)]
        t.cols_dirty = True
        n += len(rows)
    refresh_flattened(lw, trace)
    if os.environ.get("PCLEAN_DEBUG_DUMMY"):
        print(f"[pclean] {n} chosen dummy values resampled (so far {trace.dummy_cases}): " +
              ", ".join(f"{cname}.{an} x{len(rows)} (e.g. {strings[0]!r})" for (cname, j, an, d, rows, st), strings in zip(todo, drawn)),
              flush=True)
    return n


def _materialise_latent(lw, trace, pl, node, vals):
    """Create the latent row proposed as NEW at `node` of a latent plan (recursively)."""
    nodes, info = pl["nodes"], pl["node_info"]
    cname = info[node]["cls"]
    layout = lw.layout[cname]
    cmb = nodes[node][9]
    values = np.zeros(len(layout), dtype=np.int32)
    child_rows = {}

    def child_row(cn):
        if cn not in child_rows:
            ch = int(vals[cn])
            if ch == CHOICE_NEW:
                ch = _materialise_latent(lw, trace, pl, cn, vals)
            child_rows[cn] = ch
        return child_rows[cn]

    for j, c in enumerate(layout):
        cn, cc = pl["colmap"][2 * (cmb + j)], pl["colmap"][2 * (cmb + j) + 1]
        if cn < 0:
            continue
        if nodes[cn][0] == 1:
            values[j] = lw.option_values[(info[cn]["cls"], info[cn]["attr"])][vals[cn]]
        else:
            r = child_row(cn)  # may create the child row (and grow its table) first
            values[j] = trace.tables[info[cn]["cls"]].cols[cc, r]
    for j, c in enumerate(layout):
        if c.kind == "fk" and "." not in c.name:
            for k in range(nodes[node][4], nodes[node][4] + nodes[node][5]):
                cid = pl["children"][k]
                if nodes[cid][0] == 0 and nodes[cid][7] == j:
                    values[j] = child_row(cid)
    return trace.insert_row(cname, values)


def commit_latent(lw, trace, cname, live, chosen, vals):
    """Apply a latent-class sweep: rows whose chosen particle is fresh take the sampled values
    (run_smc! tail, row_inference.jl:169-185, for a latent row).  Array operations over all changed rows;
    only proposals of a brand-new referent are built one by one."""
    pl = lw.latent_plans[cname]
    t = trace.tables[cname]
    idx = np.flatnonzero(np.asarray(chosen) > 0)
    if len(idx) == 0:
        return 0
    h = np.asarray(live, dtype=np.int64)[idx]
    vals = np.asarray(vals)
    fks, props = trace._class_plan(cname)
    for j, state in props:  # own-choice sufficient statistics: take the old values out ...
        np.subtract.at(state.counts, t.cols[j, h], 1)
    changed = 0
    released = []  # (class, rows) referents to release AFTER every new reference has been counted:
    # another row of this batch may have joined a referent that this row leaves (batched schedule)
    for r, root in enumerate(pl["roots"]):
        attr = pl["root_attr"][r]
        j = lw.colidx[cname][attr]
        v = vals[idx, root]
        old = t.cols[j, h].copy()
        if pl["nodes"][root][0] == 1:  # leaf: option index -> latent value
            new = lw.option_values[(cname, attr)][v]
        else:
            new = v.astype(np.int64)
            for k in np.flatnonzero(v == CHOICE_NEW):  # ascending row order -> deterministic row ids
                new[k] = _materialise_latent(lw, trace, pl, root, vals[idx[k]])
            tgt_cls = lw.layout[cname][j].target
            moved = new != old
            np.add.at(trace.tables[tgt_cls].counts, new[moved], 1)
            released.append((tgt_cls, old[moved]))
        moved = new != old
        if moved.any():
            t.cols[j, h[moved]] = new[moved]
            t.cols_dirty = True
            changed += int(moved.sum())
    for j, state in props:  # ... and put the new ones in
        np.add.at(state.counts, t.cols[j, h], 1)
    for tgt_cls, old in released:
        np.subtract.at(trace.tables[tgt_cls].counts, old, 1)
    for tgt_cls, old in released:
        tgt = trace.tables[tgt_cls]
        cand = np.unique(old)
        trace.delete_rows_bulk(tgt_cls, cand[(tgt.counts[cand] == 0) & tgt.live[cand]])
    refresh_flattened(lw, trace)
    return changed


def sub_batches(n, config, max_sub_batches, batch_rows=None, has_parameters=True):
    """Row ranges of one class sweep, each swept against frozen tables and committed before the next.  The
    reference resamples the class's parameters and Pitman-Yor hyper-parameters every `rejuv_frequency` rows
    (inference.jl:72-77); the batched schedule does it between sub-batches of max(rejuv_frequency,
    ceil(n / max_sub_batches)) rows — exactly the reference's cadence whenever n / rejuv_frequency <=
    max_sub_batches.  A class with NOTHING to resample (has_parameters=False: an observed class without learned
    parameters, e.g. hospital's Record — rejuv_frequency has no effect on it in the reference either) is swept in
    one batch.  batch_rows overrides the size: batch_rows=1 is the reference's SEQUENTIAL schedule (every
    row sees the commits of all rows before it; parameters still move every rejuv_frequency rows)."""
    if n <= 0:
        return []
    size = max(int(config.rejuv_frequency), 1, -(-n // max(int(max_sub_batches), 1)))
    if not has_parameters:
        size = n
    if batch_rows:
        size = max(int(batch_rows), 1)
    return [(b, min(b + size, n)) for b in range(0, n, size)]


def _crosses_rejuv(b0, b1, config):
    """True when rows (b0, b1] contain a multiple of rejuv_frequency: time for a parameter move."""
    rf = max(int(config.rejuv_frequency), 1)
    return b1 // rf != b0 // rf


def latent_current_choices(lw, trace, cname, rows, config):
    """excl argument of pclean_sweep_latent for latent rows `rows` of class cname: per sub-plan root the row's current
    referent (reference slots; -1 for choices).  With use_dd_proposals = false the retained particle must also name the
    current OPTION of every choice: the index of the row's value in the proposal's options — a value that is no option
    (a string drawn for a chosen dummy) counts as the ProposalDummyValue, as in block_proposal.jl:49-52."""
    pl = lw.latent_plans[cname]
    t = trace.tables[cname]
    excl = np.full((len(pl["roots"]), len(rows)), -1, dtype=np.int32)
    for r, root in enumerate(pl["roots"]):
        col = lw.colidx[cname][pl["root_attr"][r]]
        if pl["nodes"][root][0] == 0:
            excl[r] = t.cols[col, rows]
        elif not getattr(config, "use_dd_proposals", True):
            opts = lw.option_values[(cname, pl["root_attr"][r])]
            index = np.full(len(lw.latent_dom[(cname, pl["root_attr"][r])]) + 1, -1, dtype=np.int32)
            index[opts[::-1]] = np.arange(len(opts) - 1, -1, -1)  # first option holding each value
            cur = index[t.cols[col, rows]]
            d = lw.model.classes[cname].attr(pl["root_attr"][r]).dist
            if (cur < 0).any():
                dummy = lw.latent_dom[(cname, pl["root_attr"][r])].get(d.dummy_value()) if hasattr(d, "dummy_value") else -1
                cur = np.where(cur < 0, index[dummy] if dummy >= 0 else 0, cur)
            excl[r] = cur
    return excl


def latent_sweep(engine, trace, cname, config, seed, sweep_idx, comm=None, max_sub_batches=32, verbose=False,
                 batch_rows=None):
    """One rejuvenation sweep of latent class cname.  With several ranks the live latent rows are
    block-partitioned: a rank scores its rows against their complete evidence sets (observations and
    trace are replicated), then (chosen particle, sampled values) are all-gathered and every rank
    applies the same commit — no floating-point reduction, identical result for any rank count."""
    comm = comm or Comm()
    lw = engine.lw
    pl = lw.latent_plans[cname]
    from ._lib import _ctx_cols
    dev_sort = getattr(getattr(engine, "hip", None), "argsort_ids", None) if not os.environ.get("PCLEAN_HOST_ARGSORT") else None

    def evidence():
        # on the device when the observed rows' referents and the tables live there (Engine.build_evidence_device: the
        # ordered rows and their ctx values never cross PCIe; ev_rows is None then), else with NumPy on the host
        dev = getattr(engine, "build_evidence_device", None)
        if dev is not None and trace.cur.shape[1] >= DEVICE_EVIDENCE_MIN_ROWS:
            got = dev(trace, cname)
            if got is not None:
                return got
        live_, off_, rows_, ctx_ = build_evidence(lw, trace, cname, dev_sort)
        return live_, off_, rows_, _ctx_cols(ctx_)  # padded to the library's width once, not in every sub-batch's call

    with _timed(f"latent/{cname}/build_evidence"):
        live, ev_off, ev_rows, ev_ctx = evidence()
    if len(live) == 0:
        return 0
    t = trace.tables[cname]
    changed = 0
    prev = 0
    # A latent class WITHOUT learned parameters is swept in one batch: the only thing the reference resamples every
    # rejuv_frequency rows of it are its table's Pitman-Yor hyper-parameters, whose conditional (reference counts of
    # the table, number of rows) no update of the class's own rows touches — those moves commute with the row
    # updates and are made, as many as the cut schedule would make, after the batch.
    learned = trace.has_learned_parameters(cname)
    deferred_moves = 0
    if not learned and not batch_rows:
        cut = sub_batches(len(live), config, max_sub_batches)
        deferred_moves = sum(1 for i in range(1, len(cut)) if _crosses_rejuv(cut[i - 1][0], cut[i][0], config))
    for bn, (b0, b1) in enumerate(sub_batches(len(live), config, max_sub_batches, batch_rows, has_parameters=learned)):
        if bn and _crosses_rejuv(prev, b0, config):  # inference.jl:72-77: this class's parameters and PY hyper-parameters
            prev = b0
            resample_class_parameters(trace, cname)
            if verbose and (b0 // max(config.reporting_frequency, 1)) != ((b0 - 1) // max(config.reporting_frequency, 1)):
                print(f"{cname}: Cleaning row {b0} of {len(live)}", flush=True)
        excl = latent_current_choices(lw, trace, cname, live[b0:b1], config)
        with _timed(f"latent/{cname}/upload"):
            engine.upload_trace(trace)
        lo, hi = shard_bounds(b1 - b0, comm.rank, comm.world)
        lo, hi = lo + b0, hi + b0
        e0, e1 = int(ev_off[lo]), int(ev_off[hi])
        chosen = np.zeros(0, np.int32)
        vals = np.zeros((0, len(pl["nodes"])), np.int32)
        if hi > lo:
            with _timed(f"latent/{cname}/gpu_sweep"):
                if ev_rows is None:  # (resident evidence)
                    chosen, vals = engine.sweep_latent(trace, cname, config, seed, sweep_idx, live[lo:hi],
                                                       ev_off[lo:hi + 1] - e0, None, None,
                                                       np.ascontiguousarray(excl[:, lo - b0:hi - b0]), ev_begin=e0)
                else:
                    chosen, vals = engine.sweep_latent(trace, cname, config, seed, sweep_idx, live[lo:hi],
                                                       ev_off[lo:hi + 1] - e0, ev_rows[e0:e1],
                                                       None if ev_ctx is None else ev_ctx[e0:e1],
                                                       np.ascontiguousarray(excl[:, lo - b0:hi - b0]))
        if comm.world > 1:
            chosen = comm.allgather_varlen_i32(chosen)
            vals = comm.allgather_varlen_i32(vals).reshape(-1, len(pl["nodes"]))
        with _timed(f"latent/{cname}/commit"):
            changed += commit_latent(lw, trace, cname, live[b0:b1], chosen, vals)
            if _after_commit(engine, trace, seed):
                pl = lw.latent_plans[cname]  # (the lowered model was rebuilt in place)
                # placeholders became drawn strings: per-evidence-row ctx values may have held a dummy's id
                live2, ev_off, ev_rows, ev_ctx = evidence()
                assert np.array_equal(live2, live)
    for _ in range(deferred_moves):
        resample_class_parameters(trace, cname)
    return changed


# ---------------------------------------------------------------------------
def _gather_locals(trace, comm, begin, n_local, lo):
    """Own enumerated choices (e.g. br, unit) of the rows just swept: every rank learns all of them."""
    for bi in sorted(trace.pending_locals):  # same keys, same order on every rank (collectives inside)
        loc = trace.pending_locals[bi]
        if comm.world > 1:
            loc = comm.allgather_varlen_i32(loc[:n_local]).reshape(-1, 2)
            trace.locals[bi][begin:begin + len(loc)] = loc
        else:
            trace.locals[bi][lo:lo + n_local] = loc[:n_local]
    trace.pending_locals = {}


def _sweep_window(engine, trace, config, seed, sweep_idx, b0, b1, comm):
    """Rejuvenation of the observed rows [b0, b1) against the current (frozen) tables + commit on every rank:
    the rows are block-partitioned over the ranks.  Returns the global number of rows whose referent changed."""
    lo, hi = shard_bounds(b1 - b0, comm.rank, comm.world)
    lo, hi = lo + b0, hi + b0
    light = hasattr(engine, "sweep_moved")  # the HIP engine reports the moved rows: no per-row outputs needed
    fetched = False
    if (hi > lo or comm.world > 1) and DEVICE_COMMIT and hasattr(engine, "enable_device_commit") and \
            engine.enable_device_commit(trace, comm):
        # the sweep AND its commit on the device (csrc/commit.hip): tables, counts and referents stay in HBM, the host
        # arrays of the trace fall behind until something reads them (Trace._sync).  Several ranks: every rank's moved
        # rows and new-row records are all-gathered in HBM and the same commit kernel runs everywhere (collective: a
        # rank that owns no row of the window takes part with empty lists)
        with _timed("observed/device_sweep_commit"):
            changed = engine.sweep_commit_device(trace, config, seed, sweep_idx, lo, hi, comm=comm, window=(b0, b1))
        if changed is not None:
            # (no created row holds a ProposalDummyValue — the device refuses such commits — so resample_dummies has
            # nothing to draw; the commit still counts for the draw streams of later ones)
            trace.dummy_stamp = getattr(trace, "dummy_stamp", 0) + 1
            return changed
        fetched = True  # refused, nothing modified: the sweep's outputs are on the host, the commit runs there
    if not fetched:
        with _timed("observed/upload"):
            engine.upload_trace(trace)
    with _timed("observed/gpu_sweep"):
        if fetched:
            choice, chosen, logml, new_rows = None, None, None, engine.fetched_new_rows(trace, lo, hi)
        else:
            choice, chosen, logml, new_rows = engine.sweep(trace, config, seed, sweep_idx, lo, hi, reuse_buffers=True,
                                                           **({"light": True} if light else {}))
    with _timed("observed/stats_moved"):
        stats = engine.sweep_stats_reduced(trace) if hasattr(engine, "sweep_stats_reduced") else None
        reduced = stats is not None  # summed over the ranks on the device (one RCCL all-reduce over xGMI)
        if not reduced:
            stats = engine.sweep_stats(trace)
        moved = engine.sweep_moved() if light else None
        _gather_locals(trace, comm, b0, hi - lo, lo)
    with _timed("observed/exchange_commit"):
        changed = exchange_and_commit(trace, engine.lw, comm, lo, choice, stats, new_rows, global_cur=True,
                                      moved_local=moved, n_local=hi - lo, stats_reduced=reduced, sweep_idx=sweep_idx)
        _after_commit(engine, trace, seed)
        return changed


def observed_sweep(engine, trace, config, seed, sweep_idx, comm=None, max_sub_batches=32, verbose=False,
                   batch_rows=None):
    """One rejuvenation sweep of the observed class; the rows of every sub-batch are block-partitioned over
    the ranks, the class's parameters are resampled between sub-batches (inference.jl:72-77).  An observed class
    without learned parameters (hospital's Record) has nothing to resample and is swept in ONE batch
    (sub_batches); max_sub_batches only bounds the number of parameter moves of a class that has some."""
    comm = comm or Comm()
    n = trace._cur.shape[1]  # (the shape only: no pull of a trace the device is ahead of)
    changed = 0
    prev = 0
    has_par = trace.has_parameters(engine.lw.query.cls)
    for bn, (b0, b1) in enumerate(sub_batches(n, config, max_sub_batches, batch_rows, has_par)):
        if bn and has_par and _crosses_rejuv(prev, b0, config):
            prev = b0
            resample_class_parameters(trace, engine.lw.query.cls)
            if verbose and (b0 // max(config.reporting_frequency, 1)) != ((b0 - 1) // max(config.reporting_frequency, 1)):
                print(f"{engine.lw.query.cls}: Cleaning row {b0} of {n}", flush=True)
        changed += _sweep_window(engine, trace, config, seed, sweep_idx, b0, b1, comm)
    return changed


def resample_parameters(trace):
    """resample_value! for every learned parameter + Pitman–Yor hyper-parameters of every class
    (initialize_trace's move, inference.jl:40-47; distributions.jl:57-61; trace.jl:80-108)."""
    trace.resample_parameters()
    for t in trace.tables.values():
        trace.resample_py_params(t)


def resample_class_parameters(trace, cname):
    """pgibbs_sweep!'s move (inference.jl:72-77): only the swept class's parameters and its table's
    Pitman–Yor hyper-parameters."""
    with _timed("resample_parameters"):
        trace.resample_parameters(cname)
        if cname in trace.tables:
            trace.resample_py_params(trace.tables[cname])


def initialize_trace(engine, trace, config, seed, max_batch=256, comm=None, merge_rounds=0):
    """SMC initialisation of the observed rows (inference.jl:3-58), batched: batch b sees the
    latent rows created by batches < b; identical new-row proposals inside a batch are merged.
    With several ranks each batch is block-partitioned; choices and new-row records of the batch are
    all-gathered and every rank applies the same commit.

    In-batch sequential emulation (merge_rounds > 0, off by default): in the reference row i of a batch sees the rows its
    predecessors created (inference.jl:20-37).  After the commit of a batch that created new latent rows, the
    batch is swept again in two halves — first half, commit, second half, commit — as a rejuvenation against the
    tables that now hold the batch's own new rows: rows whose new row duplicates an entity another row of the
    batch created move there (their emptied row is collected), so the result no longer depends on whether one
    entity's rows arrive in one batch (tables sorted by entity) or spread over many (random order).  Round r
    splits the batch at a different point (1/2, 1/3, 2/3 ...) so that rows which shared a half get separated.
    Measured (CPU oracle engine = the GPU path bit for bit): hospital in file order 306 -> 58 latent hospitals
    with two rounds (random order: 46-49), F1 after initialisation 0.37 -> 0.71; on tables in random order it buys
    nothing and the extra synchronous moves cost rents up to 2 pt of F1 after one iteration — hence opt-in
    (scripts/run_hospital.py --file-order uses it)."""
    comm = comm or Comm()
    lw = engine.lw
    n = trace.cur.shape[1]
    nb = len(lw.blocks)
    trace.cur[:] = -1
    begin, size = 0, 1
    cuts = (0.5, 1.0 / 3.0, 2.0 / 3.0, 0.25, 0.75)
    while begin < n:
        count = min(size, n - begin)
        lo, hi = shard_bounds(count, comm.rank, comm.world)
        engine.upload_trace(trace)
        choice, chosen, logml, new_rows = engine.sweep(trace, config, seed, 0x7fffffff, begin + lo, begin + hi)
        _gather_locals(trace, comm, begin, hi - lo, begin + lo)
        if comm.world > 1:
            choice = comm.allgather_varlen_i32(np.ascontiguousarray(choice.T)).reshape(-1, nb).T
            merged = {}
            for bi, blk in enumerate(lw.blocks):
                if blk.get("score"):
                    continue
                nn = len(blk["nodes"])
                rows, vals = new_rows.get(bi, (np.zeros(0, np.int32), np.zeros((0, nn), np.int32)))
                g_rows = comm.allgather_varlen_i32(np.asarray(rows, np.int32) + lo)
                g_vals = comm.allgather_varlen_i32(np.asarray(vals, np.int32)).reshape(-1, nn)
                if len(g_rows):
                    merged[bi] = (g_rows, g_vals)  # rank order == row order (contiguous shards)
            new_rows = merged
        created = trace.commit_batch(begin, count, choice, new_rows, dedup=True, sweep_idx=0x7fffffff)
        _after_commit(engine, trace, seed)
        if created >= max(2, count // 64) and count >= 4:  # (identical on every rank: the commit is replicated)
            for r in range(min(merge_rounds, len(cuts))):
                mid = begin + min(max(int(count * cuts[r]), 1), count - 1)
                for b0, b1 in ((begin, mid), (mid, begin + count)):
                    _sweep_window(engine, trace, config, seed, 0x7ffffffe - r, b0, b1, comm)
        begin += count
        size = min(max_batch, size * 2)
        if begin % max(config.rejuv_frequency, 1) < count:
            resample_parameters(trace)
    return trace


def run_inference(engine, trace, config, seed, verbose=False, comm=None, max_sub_batches=32, batch_rows=None):
    """run_inference! (inference.jl:83-88): config.num_iters sweeps over all classes.  `comm` shards
    every class sweep over the ranks of a torch.distributed job (one process per GPU).  A class sweep is
    cut into sub-batches between which the class's parameters are resampled (`sub_batches`): at most
    max_sub_batches per class, so tables with n / rejuv_frequency <= max_sub_batches follow the reference's
    cadence exactly; batch_rows=1 is the reference's sequential schedule (sub_batches).  use_lo_sweeps is, as in the reference, only read by instrumented_inference.jl (out of
    scope): pgibbs_sweep! sweeps the latent classes regardless of it."""
    lw = engine.lw
    if hasattr(engine, "prepare") and not getattr(engine, "_prepared", False):
        with _timed("prepare"):
            engine.prepare(trace, comm)  # (one-time: compact tables and caches of every class, device-resident commit)
    for it in range(config.num_iters):
        if verbose:
            print(f"Iteration {it + 1}/{config.num_iters}", flush=True)
        for cname in lw.model.class_order:
            if cname in lw.latent_plans:
                ch = latent_sweep(engine, trace, cname, config, seed, it, comm, max_sub_batches, verbose, batch_rows)
            elif cname == lw.query.cls:
                ch = observed_sweep(engine, trace, config, seed, it, comm, max_sub_batches, verbose, batch_rows)
            else:
                continue
            if verbose:
                print(f"iteration {it + 1}/{config.num_iters} {cname}: {ch} changes", flush=True)
                trace.check_consistency()
    return trace

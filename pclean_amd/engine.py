"""Host driver of the HIP path: uploads a lowered model + trace and runs batched
rejuvenation sweeps of the observed class (pgibbs_sweep!'s loop over the rows of
the observed class, src/inference/inference.jl:60-81 + row_inference.jl:108-187).
"""
import os
import re

import numpy as np

from . import _lib
from ._lib import HipContext, InferConfig
from .encode import lm_log_tables
from .model import ChooseProportionally, ChooseUniformly, StringPrior, TimePrior, Unmodeled

# time_prior.jl:10 — atoms matching this pattern score -log(1440), everything else -Inf
_TIME_RE = re.compile(r"^[0-9]?[0-9]:[0-9][0-9] [ap]\.m\.$")


def make_gauss(spec, mean_table_id=0):
    """pclean_gauss from a lowered Gaussian spec (model.LoweredModel.gauss)."""
    g = _lib.Gauss()
    g.x_col, g.mean_table, g.n_dims = spec["x_col"], mean_table_id, len(spec["kinds"])
    for i, (kind, payload) in enumerate(spec["kinds"]):
        g.src_kind[i], g.src[i], g.stride[i] = _lib.GSRC[kind], int(payload), int(spec["strides"][i])
    g.n_locals = spec["n_locals"]
    for l in range(2):
        g.local_n[l] = spec["local_n"][l] if l < len(spec["local_n"]) else 1
        g.local_obs_col[l] = spec["local_obs"][l] if l < len(spec["local_obs"]) else -1
    g.transform_src_kind, g.transform_src = _lib.GSRC[spec["transform"][0]], int(spec["transform"][1])
    for u in range(4):
        g.t_scale[u] = spec["t_scale"][u] if u < len(spec["t_scale"]) else 1.0
        g.t_logabsderiv[u] = spec["t_lad"][u] if u < len(spec["t_lad"]) else 0.0
        g.t_x_col[u] = spec["t_x_col"][u] if u < len(spec.get("t_x_col", ())) else -1  # (non-linear units: derived columns)
        g.t_lad_col[u] = spec["t_lad_col"][u] if u < len(spec.get("t_lad_col", ())) else -1
    g.sigma = spec["sigma"]
    return g


def lw_locals(lw):
    return getattr(lw, "locals", {})


def _logsumexp(x):
    m = np.max(x)
    if not np.isfinite(m):
        return m
    return m + np.log(np.sum(np.exp(x - m)))


class InferenceConfig:
    """src/inference/infer_config.jl:1-16 — same 7 fields, same defaults."""

    def __init__(self, num_iters, num_particles, use_dd_proposals=True, use_lo_sweeps=True,
                 use_mh_instead_of_pg=False, rejuv_frequency=50, reporting_frequency=100):
        if use_mh_instead_of_pg:
            num_particles = 2
        self.num_iters, self.num_particles = int(num_iters), int(num_particles)
        self.use_dd_proposals, self.use_lo_sweeps = bool(use_dd_proposals), bool(use_lo_sweeps)
        self.use_mh_instead_of_pg = bool(use_mh_instead_of_pg)
        self.rejuv_frequency, self.reporting_frequency = int(rejuv_frequency), int(reporting_frequency)

    def as_c(self):
        return InferConfig(self.num_iters, self.num_particles, int(self.use_dd_proposals), int(self.use_lo_sweeps),
                           int(self.use_mh_instead_of_pg), self.rejuv_frequency, self.reporting_frequency)


class Engine:
    def __init__(self, lowered, obs, device=0, dist_mode=_lib.DIST_DL, row_offset=0):
        self.lw = lowered
        self.obs = np.ascontiguousarray(obs, dtype=np.int32)
        self.device, self.row_offset = device, row_offset
        self.hip = HipContext(device)
        self.dist_mode = dist_mode
        self.option_logp = {}
        self._uploaded_shape = {}
        self._last_upload = {}
        self._dc = None  # state of the device-resident commit (enable_device_commit)
        self._upload_static()
        if row_offset:
            _lib.check(self.hip.h, self.hip.lib.pclean_set_row_offset(self.hip.h, _lib.C.c_int64(row_offset)),
                       "pclean_set_row_offset")

    def _sync_ahead_trace(self):
        """A trace the device-resident commit left behind (Trace._dev is this engine) holds its only current copy in this
        engine's HBM: pull it before the context goes away (close / reload)."""
        ref = getattr(self, "_ahead", None)
        tr = ref() if ref is not None else None
        if tr is not None and getattr(tr, "_dev", None) is self:
            tr._sync()
        self._ahead = None

    def close(self):
        self._sync_ahead_trace()
        self.hip.close()

    def reload(self):
        """The lowered model grew (LoweredModel.relower: strings drawn for chosen dummy values joined latent
        domains): a fresh context with the new string pool, pair / fn / option tables and plans.  Rare (a
        ProposalDummyValue is only ever chosen for unobserved or very short strings), so nothing is patched
        incrementally.  A device-side RCCL communicator bound with init_device_comm is re-bound to the new context —
        collectively: every rank reloads at the same point (the dummy draws are replicated)."""
        self._sync_ahead_trace()
        self.hip.close()
        self.hip = HipContext(self.device)
        self.option_logp = {}
        self._uploaded_shape = {}
        self._last_upload = {}
        self._upload_static()
        if self.row_offset:
            _lib.check(self.hip.h, self.hip.lib.pclean_set_row_offset(self.hip.h, _lib.C.c_int64(self.row_offset)),
                       "pclean_set_row_offset")
        self._dev_comm = False
        if getattr(self, "_comm", None) is not None:
            self.init_device_comm(self._comm)
        self._dc = None  # (a fresh context: the device-resident commit is set up again on demand)

    def sample_prior_strings(self, dist, n, seed, stream):
        """n draws of random(StringPrior) (string_prior.jl:28-40: length uniform on [min, max], bigram letters) or
        random(TimePrior) (time_prior.jl:21-23) by the device samplers, as strings; counter = (seed, stream, i)."""
        from . import sampling
        if isinstance(dist, TimePrior):
            return sampling.random_time_prior(self.hip, n, seed=seed, stream=stream)
        return sampling.random_string_prior(self.hip, n, dist.min_len, dist.max_len, seed=seed, stream=stream)

    def sample_prior_strings_at(self, dist, seeds, elems):
        """random(StringPrior) with the private streams the sweep used for the weights of the particles that chose the
        dummy (pclean_dummy_seed keys, global observed rows): the strings those particles' new rows hold."""
        from . import sampling
        return sampling.random_string_prior_at(self.hip, seeds, elems, dist.min_len, dist.max_len)

    # -- static data ----------------------------------------------------------
    def _upload_gauss(self):
        lw, hip = self.lw, self.hip
        for (bid, nid), spec in getattr(lw, "gauss", {}).items():
            hip.set_node_gauss(bid, nid, make_gauss(spec))

    def _upload_static(self):
        lw, hip = self.lw, self.hip
        sym, off, lm, _ = lw.pool.arrays()
        hip.load_strings(sym, off)
        from .encode import load_lm_params
        hip.set_lm_tables(*load_lm_params(), lw.pool.letter_symbols())
        hip.load_columns(self.obs)
        import time
        t0 = time.perf_counter()
        self.pair_cells = 0  # DP cells of the AddTypos pair tables: sum over tables of (sum of obs lengths) x (sum of latent lengths)
        self.pair_count = 0
        for key, (pid, odom, ldom) in lw.pair_id.items():
            oi, li = odom.id_array(), ldom.id_array()
            hip.build_pair_table(pid, oi, li, self.dist_mode)
            self.pair_cells += int(lw.pool.lens[oi].astype(np.int64).sum()) * int(lw.pool.lens[li].astype(np.int64).sum())
            self.pair_count += len(oi) * len(li)
        self.pair_build_s = time.perf_counter() - t0
        for fid, fn in lw.fn_tables.items():
            hip.set_fn_table(fid, fn)
        for key, (pid, n) in lw.eq_pairs.items():  # equality constraints: 0 on the diagonal, 1 elsewhere
            hip.set_pair_table(pid, (1 - np.eye(n, dtype=np.uint8)))
        for pid in lw.same_pairs:  # MaybeSwap: 0 iff the observed string is the latent value
            hip.set_pair_table(pid, lw.same_pair_table(pid))
        if getattr(lw, "xnum", None) is not None and lw.xnum.shape[0]:
            hip.load_numeric_columns(lw.xnum)
        init_l, trans_l = lm_log_tables()
        m = lw.model
        for (cname, aname), dom in lw.latent_dom.items():
            d = m.classes[cname].attr(aname).dist
            tid = lw.option_id[(cname, aname)]
            if isinstance(d, TimePrior):
                # per key: its atoms' scores + that key's dummy mass (time_prior.jl:8-22)
                vals = lw.option_values[(cname, aname)]
                keys = lw.option_keycol[(cname, aname)]
                dummy = dom.get(d.dummy_value())
                scores = np.array([-np.log(1440.0) if _TIME_RE.match(dom.string(v)) else -np.inf for v in vals])
                logp = scores.copy()
                for k in np.unique(keys):
                    sel = (keys == k) & (vals != dummy)
                    with np.errstate(divide="ignore"):
                        logp[(keys == k) & (vals == dummy)] = np.log1p(-np.exp(_logsumexp(scores[sel])))
                self.option_logp[(cname, aname)] = logp
                hip.set_options_cols(tid, np.stack([vals, keys, lw.option_ncol[(cname, aname)]]), logp)
                continue
            if d is None or isinstance(d, Unmodeled) or not isinstance(d, (StringPrior, ChooseUniformly, ChooseProportionally)):
                if cname == lw.query.cls:
                    continue  # own choices of the observed class are enumerated as locals, not as leaves
                logp = np.zeros(len(dom))  # Unmodeled: logdensity 0 (unmodeled.jl:7-10)
            elif isinstance(d, StringPrior) and d.keyed_by:
                # per key: scores of its atoms + that key's dummy mass (string_prior.jl:16-22)
                vals = lw.option_values[(cname, aname)]
                keys = lw.option_keycol[(cname, aname)]
                ids = dom.id_array()[vals]
                offs = np.zeros(len(ids) + 1, dtype=np.int64)
                np.cumsum(lw.pool.lens[ids], out=offs[1:])
                lmcat = np.concatenate([lm[off[i]:off[i + 1]] for i in ids])
                scores = hip.string_prior_scores(lmcat, offs, d.min_len, d.max_len, init_l, trans_l)
                logp = scores.copy()
                dummy = dom.get(d.dummy_value())
                for k in np.unique(keys):
                    sel = (keys == k) & (vals != dummy)
                    with np.errstate(divide="ignore"):
                        logp[(keys == k) & (vals == dummy)] = np.log1p(-np.exp(_logsumexp(scores[sel])))
                self.option_logp[(cname, aname)] = logp
                hip.set_options_cols(tid, np.stack([vals, keys]), logp)
                continue
            elif isinstance(d, StringPrior):
                # discrete_proposal(::StringPrior): atom scores + dummy mass (string_prior.jl:16-22)
                ids = dom.id_array()[:dom.n_base() - 1]  # the atoms (the dummy is the last value before any drawn string)
                offs = np.zeros(len(ids) + 1, dtype=np.int64)
                lens = lw.pool.lens[ids]
                np.cumsum(lens, out=offs[1:])
                lmcat = np.concatenate([lm[off[i]:off[i + 1]] for i in ids]) if len(ids) else np.zeros(0, np.uint8)
                scores = hip.string_prior_scores(lmcat, offs, d.min_len, d.max_len, init_l, trans_l)
                with np.errstate(divide="ignore"):
                    dummy = np.log1p(-np.exp(_logsumexp(scores)))
                logp = np.concatenate([scores, [dummy]])
            elif isinstance(d, ChooseUniformly):
                logp = np.full(len(dom), -np.log(len(d.options)))
            elif isinstance(d, ChooseProportionally):
                continue  # depends on the parameter value: uploaded with the trace
            else:
                raise NotImplementedError(type(d))
            self.option_logp[(cname, aname)] = logp
            hip.set_options(tid, lw.option_values[(cname, aname)], logp)
        lw.load_blocks_into(hip)
        self._gauss_pending = bool(getattr(lw, "gauss", {}))

    # -- dynamic data ---------------------------------------------------------
    def upload_trace(self, trace):
        """Latent tables, parameter-dependent option priors and parameter tables of `trace` -> device.  Whatever is
        byte-identical to the last upload is skipped: besides the copies that keeps the device-side versions — and
        with them the per-value caches of option-list marginals and the compact byte tables — valid (between two
        observed-class sweeps of one rejuvenation period only the root tables' counts move)."""
        lw, hip = self.lw, self.hip
        m = lw.model
        last = self._last_upload
        dc = getattr(self, "_dc", None)
        dbg = os.environ.get("PCLEAN_DEBUG_UPLOAD")  # diagnostic: uploads of a table that take more than a millisecond
        t_dbg = None
        for cname, t in trace.tables.items():
            if dbg:
                t_dbg = self._debug_upload(t_dbg, cname)
            cols, counts = t.view()
            cap = t.n
            in_dc = dc is not None and cname in dc["tables"]
            if in_dc:
                # device-resident commit: the table is uploaded with spare rows (count 0: dead candidates, weight exactly 0)
                # so that the device can create rows without changing any array's shape
                cap = dc["cap"].get(cname, 0)
                grown = max(0, t.n - dc["alloc"].get(cname, (t.n, 0))[0])  # what the host added to the high-water mark
                dc["created"][cname] = max(grown, dc["created"].get(cname, 0))
                if t.n + self._slack_min(cname, t) > cap:
                    cap = dc["cap"][cname] = self._capacity(cname, t)
            key = (cname, cap)
            prev = last.get(("table", cname))
            alloc = (t.n, len(t.free))
            need_cols = t.cols_dirty or self._uploaded_shape.get(cname) != key
            # (the comparison comes before any padding: most tables of most sub-batches have not moved)
            if not need_cols and prev is not None and prev[1] == (t.strength, t.discount) and len(prev[0]) >= t.n \
                    and np.array_equal(prev[0][:t.n], counts) and not prev[0][t.n:].any() \
                    and (not in_dc or dc["alloc"].get(cname) == alloc):
                continue  # nothing moved in this table
            if in_dc and cap > t.n:
                pk = np.zeros(cap, dtype=np.int64)
                pk[:t.n] = counts
                counts = pk
                if need_cols:
                    pc = np.zeros((t.n_cols, cap), dtype=np.int32)
                    pc[:, :t.n] = cols
                    cols = pc
            if need_cols:
                hip.set_table(lw.table_id[cname], np.ascontiguousarray(cols), counts, t.strength, t.discount)
                t.cols_dirty = False
                self._uploaded_shape[cname] = key
            else:  # only reference counts moved: keep the device columns and their compact byte tables
                hip.set_table(lw.table_id[cname], None, counts, t.strength, t.discount, n_cols=t.n_cols)
            last[("table", cname)] = (counts.copy(), (t.strength, t.discount))
            if in_dc:
                hip.commit_set_table_state(lw.table_id[cname], t.n, t.free)
                dc["alloc"][cname] = alloc
        if dbg:
            self._debug_upload(t_dbg, None)
        for (cname, aname), dom in lw.latent_dom.items():
            d = m.classes[cname].attr(aname).dist
            if isinstance(d, ChooseProportionally):
                with np.errstate(divide="ignore"):
                    logp = np.log(trace.params[(cname, d.param)].value)  # logprobs(), utils.jl:33-36
                self.option_logp[(cname, aname)] = logp
                prev = last.get(("options", cname, aname))
                if prev is not None and np.array_equal(prev, logp):
                    continue
                hip.set_options(lw.option_id[(cname, aname)], lw.option_values[(cname, aname)], logp)
                last[("options", cname, aname)] = logp.copy()
        if lw.prob_spec is not None:
            pt = trace.prob_table()
            if last.get("prob") is None or not np.array_equal(last["prob"], pt):
                hip.set_prob_table(pt)
                last["prob"] = pt.copy()
        if getattr(lw, "gauss", None):
            mv = trace.mean_param.value
            if last.get("mean") is None or not np.array_equal(last["mean"], mv):
                hip.set_mean_table(0, mv)
                last["mean"] = mv.copy()
            if self._gauss_pending:  # needs the mean table to exist
                self._upload_gauss()
                self._gauss_pending = False

    @staticmethod
    def _debug_upload(prev, cname):
        """PCLEAN_DEBUG_UPLOAD: prints the table whose upload just ended when it took more than a millisecond"""
        import time
        now = time.perf_counter()
        if prev is not None and now - prev[1] > 1e-3:
            print(f"[upload] {prev[0]}: {1e3 * (now - prev[1]):.1f} ms", flush=True)
        return (cname, now)

    def _upload_cur_locals(self, trace, cfg):
        """prior proposals (use_dd_proposals = false) of a class with own enumerated choices: the retained particle keeps the
        row's current ones (block_proposal.jl:42-56) — the data-driven proposal enumerates them and needs none"""
        if cfg.use_dd_proposals or not self.lw.locals:
            return
        loc = trace._locals  # (host-owned even while the device is ahead of the trace: sweep_commit_device keeps it current)
        for bi in self.lw.locals:
            self.hip.set_cur_locals(bi, loc[bi])

    def sweep(self, trace, config, seed, sweep_idx, lo=0, hi=None, reuse_buffers=False, light=False):
        """One batched sweep over the observed rows [lo, hi) of the trace (default: all of them).
        Returns (choice, chosen_particle, logml, new_rows), all indexed relative to lo.  reuse_buffers:
        trace.cur is passed in place (page-locked once) and the results are views of page-locked buffers
        that the next sweep overwrites (callers that commit the result right away: inference.py, bench.py).
        light: no per-row outputs are copied back (choice / chosen / logml are None); the commit reads the
        moved rows (sweep_moved) and the new-row records only."""
        cfg = config.as_c() if isinstance(config, InferenceConfig) else config
        hi = trace.cur.shape[1] if hi is None else hi
        if hi <= lo:  # a rank may own no row of a small batch
            for bi in self.lw.locals:
                trace.pending_locals[bi] = np.zeros((0, 2), dtype=np.int32)
            self._empty_sweep = True
            nb = trace.cur.shape[0]
            return (None if light else np.zeros((nb, 0), np.int32)), np.zeros(0, np.int32), np.zeros(0), {}
        self._empty_sweep = False
        self.hip.set_active_rows(lo, hi - lo)
        self._upload_cur_locals(trace, cfg)
        if reuse_buffers and trace.cur.dtype == np.int32 and trace.cur.flags.c_contiguous:
            choice, chosen, logml = self.hip.sweep(cfg, seed, sweep_idx, trace.cur, True, window=(lo, hi), light=light)
        else:
            choice, chosen, logml = self.hip.sweep(cfg, seed, sweep_idx, trace.cur[:, lo:hi], reuse_buffers, light=light)
        new_rows = {}
        for bi, blk in enumerate(self.lw.blocks):
            if blk.get("score"):
                continue
            rows, vals = self.hip.get_new_rows(bi, len(blk["nodes"]))
            if len(rows):
                new_rows[bi] = (rows, vals)
        if self.lw.locals:
            for bi in self.lw.locals:
                trace.pending_locals[bi] = self.hip.get_locals(bi, hi - lo)
        return choice, chosen, logml, new_rows

    # -- device-resident commit (csrc/commit.hip) ------------------------------------------------------------------
    def _slack_min(self, cname, t):
        """spare rows a table needs before the next device commit: a few times what the last commits added to its
        high-water mark (rows re-created from the free list need no room).  Dead candidates cost every enumeration —
        a one-row table padded to hundreds of rows would make its reference slots hundreds of times dearer — so the
        margin stays proportional to the table"""
        dc = self._dc
        if os.environ.get("PCLEAN_DC_NOPAD"):
            return 0
        return max(8, 4 * dc["created"].get(cname, 0) + 8)

    def _capacity(self, cname, t):
        if os.environ.get("PCLEAN_DC_NOPAD"):  # diagnostic: no spare rows (every growth is a refused commit)
            return t.n
        slack = max(16, t.n // 16, 2 * self._slack_min(cname, t))
        return -(-(t.n + slack) // 16) * 16

    def enable_device_commit(self, trace, comm=None):
        """Switch the engine to the device-resident commit of observed-class sweeps (pclean_commit_*): latent tables
        are uploaded with spare capacity, their allocation state and the observed rows' referents live in HBM and
        sweep_commit_device() commits a sweep without any per-row read-back.  Returns False (and changes nothing) when
        the plan / the run is one the device commit does not take: several ranks (the new-row records of the other
        ranks are exchanged through the host), or a plan shape pclean_commit_enable refuses."""
        if getattr(self, "_dc", None) is not None:
            return True
        if getattr(self, "_dc_refused", False):
            return False
        self._dc_dist = False
        if comm is not None and (comm.world > 1 or (getattr(self, "_dev_comm", False) and os.environ.get("PCLEAN_FORCE_DIST"))):
            # several ranks: every rank's moved rows and new-row records are all-gathered on the device and the same commit
            # kernel runs everywhere (pclean_commit_device_dist) — needs the library's own RCCL communicator
            if not getattr(self, "_dev_comm", False) or os.environ.get("PCLEAN_DIST_HOST_COMMIT"):
                self._dc_refused = True
                return False
            self._dc_dist = True
        if self.row_offset:
            self._dc_refused = True
            return False
        lw = self.lw
        self.upload_trace(trace)  # (the library checks the blocks' tables exist)
        ok, why = self.hip.commit_enable(len(lw.blocks))
        if not ok:
            self._dc_refused, self._dc_why = True, why
            return False
        by_id = {tid: c for c, tid in lw.table_id.items()}
        self._dc = dict(tables=[by_id[t] for t in self.hip.commit_tables()], cap={}, alloc={}, created={}, cur_version=None,
                        commits=0, fallbacks=0)
        for cname in self._dc["tables"]:  # re-upload with capacity
            self._uploaded_shape.pop(cname, None)
        self.upload_trace(trace)
        return True

    def prepare(self, trace, comm=None):
        """One-time set-up before the first run_inference iteration: the device-resident commit (tables uploaded with their
        spare capacity from the start, so that no table changes shape later) and every compact table / per-value cache the
        sweeps of all classes will ask for (pclean_prepare).  Optional: results do not depend on it."""
        from . import inference as inf
        if inf.DEVICE_COMMIT:
            self.enable_device_commit(trace, comm)
        self.upload_trace(trace)
        ev = 0
        for pl in self.lw.latent_plans.values():
            ev |= 1 << int(pl["block_id"])
        self.hip.prepare(ev)
        self._prepared = True

    def _sync_cur(self, trace):
        dc = self._dc
        if dc["cur_version"] != (id(trace), trace._cur_version):
            self.hip.set_cur(trace._cur)
            dc["cur_version"] = (id(trace), trace._cur_version)

    def sweep_commit_device(self, trace, config, seed, sweep_idx, lo=0, hi=None, comm=None, window=None):
        """One batched sweep of the observed rows [lo, hi) against the device-resident tables AND its commit on the
        device: ONE stream synchronisation, a summary comes back.  Returns the number of rows whose referent changed,
        or None when the device refused the commit (nothing was modified; the sweep's outputs are on the host as after
        a plain sweep(..., light=True): the caller commits on the host).  The host arrays of `trace` are left alone:
        the trace is marked as behind the device and pulls the state when something reads it (Trace._sync)."""
        dc = self._dc
        cfg = config.as_c() if isinstance(config, InferenceConfig) else config
        hi = trace._cur.shape[1] if hi is None else hi
        if trace._dev is None:  # the host arrays are current: the device gets whatever moved since the last upload
            self.upload_trace(trace)
            self._sync_cur(trace)
        elif trace._dev is not self:
            raise _lib.PCleanHipError("the trace is ahead on another engine")
        dist = getattr(self, "_dc_dist", False)
        local_empty = hi <= lo  # (several ranks: a rank may own no row of a small window; it still takes part in the exchange)
        self._empty_sweep = local_empty
        self.hip.set_sweep_mode(True)
        try:
            if not local_empty:
                self.hip.set_active_rows(lo, hi - lo)
                self._upload_cur_locals(trace, cfg)
                self.hip.sweep_device_cur(cfg, seed, sweep_idx, len(self.lw.blocks))
            if dist:
                world = comm.world if comm is not None else 1
                b0, b1 = window if window is not None else (lo, hi)
                summ = self.hip.commit_device_dist(len(self.lw.blocks), sweep_idx, local_empty, -(-(b1 - b0) // world))
            else:
                summ = self.hip.commit_device(len(self.lw.blocks), sweep_idx)
        finally:
            self.hip.set_sweep_mode(False)
        dc["commits"] += 1
        by_id = {tid: c for c, tid in self.lw.table_id.items()}
        if summ.fallback:
            dc["fallbacks"] += 1
            dc["last_fallback"] = int(summ.fallback)
            if not local_empty:
                self.hip.sweep_fetch()
            # (several ranks: the delta reference counts in the device buffers are already summed over the ranks)
            self._stats_reduced_on_device = bool(summ.stats_reduced)
            return None  # (a table about to outgrow its capacity gets more room at the next upload: upload_trace)
        if os.environ.get("PCLEAN_DEBUG_COMMIT"):
            print("[pclean] device commit: changed", summ.n_changed, "records", list(summ.n_records[:len(self.lw.blocks)]),
                  "distinct", list(summ.n_distinct[:len(self.lw.blocks)]),
                  {by_id[summ.slot[i].table_id]: (summ.slot[i].n_hw, summ.slot[i].n_free, summ.slot[i].created, summ.slot[i].deleted,
                                                  summ.slot[i].cols_changed) for i in range(summ.n_slots)}, flush=True)
        for si in range(summ.n_slots):
            sl = summ.slot[si]
            c = by_id[sl.table_id]
            grown = max(0, int(sl.n_hw) - dc["alloc"].get(c, (int(sl.n_hw), 0))[0])  # rows the high-water mark moved
            dc["created"][c] = max(grown, dc["created"].get(c, 0) // 2)
            dc["alloc"][c] = (int(sl.n_hw), int(sl.n_free))
        for bi in lw_locals(self.lw):  # own enumerated choices of the chosen particles: host-owned (parameter moves read them)
            loc = self.hip.get_locals(bi, hi - lo) if not local_empty else np.zeros((0, 2), dtype=np.int32)
            if dist and comm is not None and comm.world > 1:  # every rank learns all of them (rank order = row order)
                b0 = window[0] if window is not None else lo
                loc = comm.allgather_varlen_i32(np.ascontiguousarray(loc, dtype=np.int32)).reshape(-1, 2)
                trace._locals[bi][b0:b0 + len(loc)] = loc
            else:
                trace._locals[bi][lo:hi] = loc
        trace._dev = self
        import weakref
        self._ahead = weakref.ref(trace)  # (close / reload pull it first: the committed sweeps exist nowhere else)
        return int(summ.n_changed)

    def fetched_new_rows(self, trace, lo, hi):
        """new-row records of the last sweep once its outputs are on the host (a refused device commit: pclean_sweep_fetch
        ran) — what sweep(..., light=True) returns as new_rows; own enumerated choices go to trace.pending_locals"""
        new_rows = {}
        if getattr(self, "_empty_sweep", False):  # (several ranks: this rank swept no row of the window)
            for bi in lw_locals(self.lw):
                trace.pending_locals[bi] = np.zeros((0, 2), dtype=np.int32)
            return new_rows
        for bi, blk in enumerate(self.lw.blocks):
            if blk.get("score"):
                continue
            rows, vals = self.hip.get_new_rows(bi, len(blk["nodes"]))
            if len(rows):
                new_rows[bi] = (rows, vals)
        for bi in lw_locals(self.lw):
            trace.pending_locals[bi] = self.hip.get_locals(bi, hi - lo)
        return new_rows

    def pull(self, trace):
        """Device-resident state -> host arrays of `trace` (called through Trace._sync when something reads a trace that
        device commits left behind): latent tables with their allocation state, the observed rows' referents, the
        Dirichlet counts of own choices (recomputed from the tables), the origins of created rows."""
        lw, dc = self.lw, self._dc
        if dc is None or not self.hip.h:
            raise _lib.PCleanHipError("the trace is behind a device-resident commit whose context is gone (Engine.close / "
                                      "reload without a pull): its committed sweeps are lost")
        for cname in dc["tables"]:
            t = trace._tables[cname]
            state, cols, counts, live, free, origin = self.hip.commit_pull_table(lw.table_id[cname])
            n = int(state[0])
            if t.cols.shape[1] < n:
                grow = max(n, 2 * t.cols.shape[1]) - t.cols.shape[1]
                t.cols = np.concatenate([t.cols, np.zeros((t.n_cols, grow), np.int32)], axis=1)
                t.counts = np.concatenate([t.counts, np.zeros(grow, np.int64)])
                t.live = np.concatenate([t.live, np.zeros(grow, bool)])
            t.cols[:, :n] = cols[:, :n]
            t.counts[:n] = counts[:n]
            t.counts[n:] = 0
            t.live[:n] = live[:n]
            t.live[n:] = False
            t.n = n
            t.free = [int(r) for r in free]
            t.cols_dirty = False
            for r in np.flatnonzero(origin[:, 0]):
                mark = int(origin[r, 0])
                if mark > 0:
                    trace._row_origin[(cname, int(r))] = (int(origin[r, 1]), int(origin[r, 2]), int(origin[r, 3]), mark - 1)
                else:
                    trace._row_origin.pop((cname, int(r)), None)
            cap = dc["cap"].get(cname, n)
            pk = np.zeros(max(cap, n), dtype=np.int64)
            pk[:n] = counts[:n]
            self._last_upload[("table", cname)] = (pk if cap > n else pk[:n].copy(), (t.strength, t.discount))
            dc["alloc"][cname] = (n, len(t.free))
        if trace._cur.flags.c_contiguous and trace._cur.dtype == np.int32:
            self.hip.get_cur(trace._cur.shape[0], trace._cur.shape[1], out=trace._cur)  # in place (the array may be page-locked)
        else:
            trace._cur[...] = self.hip.get_cur(*trace._cur.shape)
        dc["cur_version"] = (id(trace), trace._cur_version)
        for (cname, pname), state in trace._params.items():  # own-choice sufficient statistics (check_consistency's identity)
            t = trace._tables[cname]
            rows = np.flatnonzero(t.live[:t.n])
            for a in lw.model.classes[cname].attrs:
                if a.kind == "choice" and isinstance(a.dist, ChooseProportionally) and a.dist.param == pname:
                    state.counts = np.bincount(t.cols[lw.colidx[cname][a.name], rows], minlength=len(state.counts)).astype(np.int64)

    def sweep_moved(self):
        """{block: (rows relative to the swept window, new referent)} of the last sweep, rows ascending."""
        if getattr(self, "_empty_sweep", False):
            return {bi: (np.zeros(0, np.int32), np.zeros(0, np.int32))
                    for bi, blk in enumerate(self.lw.blocks) if not blk.get("score")}
        return {bi: self.hip.get_moved(bi) for bi, blk in enumerate(self.lw.blocks) if not blk.get("score")}

    def sweep_latent(self, trace, cname, config, seed, sweep_idx, live, ev_off, ev_rows, ev_ctx, excl, ev_begin=None):
        """Rejuvenation of the latent rows `live` of class cname against their evidence sets
        (pclean_sweep_latent).  Returns (chosen particle, sampled node values) per latent row.  ev_begin (with ev_rows
        None): the rows' evidence starts at that position of the evidence build_evidence_device left on the device."""
        pl = self.lw.latent_plans[cname]
        cfg = config.as_c() if isinstance(config, InferenceConfig) else config
        self.hip.set_active_rows(0, -1)
        if ev_begin is not None:
            return self.hip.sweep_latent_resident(cfg, seed, sweep_idx, pl["block_id"], pl["roots"], live, ev_off, ev_begin,
                                                  excl, len(pl["nodes"]))
        return self.hip.sweep_latent(cfg, seed, sweep_idx, pl["block_id"], pl["roots"], live, ev_off, ev_rows, ev_ctx,
                                     excl, len(pl["nodes"]))

    def build_evidence_device(self, trace, cname):
        """inference.build_evidence on the device (pclean_build_evidence): returns (live, ev_off, None, None) with the ordered
        evidence rows and their per-row ctx values left in HBM for sweep_latent(..., ev_rows=None, ev_begin=...), or None
        when this class / state takes the host path (no device-resident referents, MaybeSwap / Gaussian evidence contexts,
        a walk longer than the library's, references to dead rows).  Same arrays as the host path, element for element
        (tests/test_gpu_edges.py::test_device_evidence_equals_host)."""
        lw = self.lw
        dc = getattr(self, "_dc", None)
        if dc is None or os.environ.get("PCLEAN_HOST_EVIDENCE"):
            return None
        pl = lw.latent_plans[cname]
        if cname in lw.latent_ev_prob or cname in getattr(lw, "latent_ev_locals", {}):
            return None
        bi = pl["src_block"]
        steps, cn = [], lw.blocks[bi]["root_class"]
        for step in [p for p in pl["path"].split(".") if p]:
            j = lw.colidx[cn][step]
            steps.append((lw.table_id[cn], j))
            cn = lw.layout[cn][j].target
        sources = []
        for ob, col in pl.get("ctx_sources", []):
            sources.append((ob, lw.table_id[lw.blocks[ob]["root_class"]], col))
        if len(steps) > _lib.EV_MAX_STEPS or len(sources) > _lib.MAX_CTX or cn != cname:
            return None
        t = trace.tables[cname]  # (a trace that device commits left behind pulls here)
        self.upload_trace(trace)
        self._sync_cur(trace)
        off = self.hip.build_evidence(bi, steps, t.n, sources)
        counts = np.diff(off)
        if off[0] != 0 or int(off[-1]) != trace.cur.shape[1] or counts[~t.live[:t.n]].any():
            return None  # rows without a referent / referring to dead rows: the host path filters them
        live = np.nonzero(t.live[:t.n])[0].astype(np.int32)
        ev_off = np.zeros(len(live) + 1, dtype=np.int32)
        np.cumsum(counts[live], out=ev_off[1:])
        # (dead rows hold no evidence: the live rows' groups are contiguous in the resident order, live row j's at ev_off[j])
        return live, ev_off, None, None

    def init_device_comm(self, comm):
        """Bind the library's own RCCL communicator to the ranks of `comm` (parallel.Comm over torch.distributed):
        rank 0's rendezvous id travels through torch's object broadcast.  Afterwards sweep_stats_reduced sums the
        delta reference counts on the device with ONE all-reduce over xGMI.  Returns False (and the torch path stays
        in use) when RCCL cannot be bound."""
        self._dev_comm = False
        self._comm = None
        if comm.dist is None:
            return False
        self._comm = comm  # reload() binds the fresh context to the same ranks
        try:
            box = [self.hip.comm_unique_id() if comm.rank == 0 else None]
            comm.dist.broadcast_object_list(box, src=0)
            self.hip.comm_init(comm.world, comm.rank, box[0])
            self._dev_comm = True
        except Exception as e:  # noqa: BLE001 — any failure: keep the torch.distributed exchange
            import sys
            print(f"[pclean] device-side RCCL exchange unavailable ({e}); using torch.distributed", file=sys.stderr)
        return self._dev_comm

    def sweep_stats_reduced(self, trace):
        """Delta reference counts of the last sweep summed over ALL ranks, per block root table — one fused RCCL
        all-reduce of the device-resident buffers (pclean_allreduce_stats_fused); None without init_device_comm."""
        if not getattr(self, "_dev_comm", False):
            return None
        blocks = [bi for bi, blk in enumerate(self.lw.blocks) if not blk.get("score")]
        if getattr(self, "_stats_reduced_on_device", False):  # a refused pclean_commit_device_dist already summed them in place
            self._stats_reduced_on_device = False
            return {bi: self.hip.get_stats(self.lw.table_id[self.lw.blocks[bi]["root_class"]],
                                           trace.tables[self.lw.blocks[bi]["root_class"]].n) for bi in blocks}
        tids = [self.lw.table_id[self.lw.blocks[bi]["root_class"]] for bi in blocks]
        ns = [trace.tables[self.lw.blocks[bi]["root_class"]].n for bi in blocks]
        # a rank that swept nothing contributes zeros (its device buffers hold stale counts)
        return dict(zip(blocks, self.hip.allreduce_stats_fused(tids, ns, getattr(self, "_empty_sweep", False))))

    def sweep_stats(self, trace):
        """Delta reference counts of the last sweep per block root table (the all-reduce payload)."""
        if getattr(self, "_empty_sweep", False):
            return {bi: np.zeros(trace.tables[blk["root_class"]].n, dtype=np.int64)
                    for bi, blk in enumerate(self.lw.blocks) if not blk.get("score")}
        return {bi: self.hip.get_stats(self.lw.table_id[blk["root_class"]], trace.tables[blk["root_class"]].n)
                for bi, blk in enumerate(self.lw.blocks) if not blk.get("score")}

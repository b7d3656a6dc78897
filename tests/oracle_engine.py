"""Stand-in for pclean_amd.engine.Engine whose compute is the CPU oracle (TEST INFRASTRUCTURE).

It lets the CPU suite drive the product's host code — inference.initialize_trace / run_inference,
parallel.exchange_and_commit, trace commits, parameter moves — end to end, single process or
gloo world_size 2, without a GPU.  Only the four methods inference.py uses are provided."""
import ctypes as C

import numpy as np

import helpers
from pclean_amd._lib import InferConfig


class OracleEngine:
    def __init__(self, oracle, lowered, obs, cached=False):
        """cached: keep ONE oracle World alive and refresh only what a commit can change (row window, latent tables,
        parameter-dependent option priors) — what makes row-at-a-time (sequential-schedule) reference runs affordable."""
        self.oracle, self.lw, self.obs = oracle, lowered, np.ascontiguousarray(obs, dtype=np.int32)
        self._choice = None
        self.cached, self._w, self._logp = cached, None, None

    def upload_trace(self, trace):
        pass  # worlds are rebuilt from the trace at every sweep

    def reload(self):
        """the lowered model grew (LoweredModel.relower): drop the cached world, re-read the observations"""
        self._w, self._logp = None, None

    def sample_prior_strings(self, dist, n, seed, stream):
        """same draws as Engine.sample_prior_strings (the oracle's samplers are bit-identical to the device's)"""
        from pclean_amd import sampling
        from pclean_amd.model import TimePrior
        ro = self.oracle.RandomOracle()
        if isinstance(dist, TimePrior):
            return sampling.random_time_prior(ro, n, seed=seed, stream=stream)
        return sampling.random_string_prior(ro, n, dist.min_len, dist.max_len, seed=seed, stream=stream)

    def sample_prior_strings_at(self, dist, seeds, elems):
        from pclean_amd import sampling
        return sampling.random_string_prior_at(self.oracle.RandomOracle(), seeds, elems, dist.min_len, dist.max_len)

    def _cfg(self, config):
        return InferConfig(config.num_iters, config.num_particles, int(getattr(config, "use_dd_proposals", True)), 1,
                           int(config.use_mh_instead_of_pg),
                           config.rejuv_frequency, config.reporting_frequency)

    def _world(self, trace, lo, hi):
        orc, lw = self.oracle, self.lw
        win = np.ascontiguousarray(self.obs[:, lo:hi])
        if not self.cached or self._w is None:
            self._logp = helpers.option_logp_cpu(orc, lw, trace)
            self._w = helpers.mirror_world(orc, lw, win, trace, None, 1, self._logp, row_lo=lo)
            return self._w
        from pclean_amd.model import ChooseProportionally
        w = self._w
        w.set_obs(win)
        if getattr(lw, "xnum", None) is not None and lw.xnum.shape[0]:
            w.set_numeric(lw.xnum[:, lo:hi])
        for cname, t in trace.tables.items():
            cols, counts = t.view()
            full, m1, scal = orc.table_priors(counts, t.strength, t.discount)
            w.set_table(lw.table_id[cname], np.ascontiguousarray(cols), counts, full, m1, scal)
        for (cname, aname), dom in lw.latent_dom.items():
            d = lw.model.classes[cname].attr(aname).dist
            if isinstance(d, ChooseProportionally):
                with np.errstate(divide="ignore"):
                    lp = np.log(trace.params[(cname, d.param)].value)
                self._logp[(cname, aname)] = lp
                w.set_options(lw.option_id[(cname, aname)], lw.option_values[(cname, aname)], lp)
        if lw.prob_spec is not None:
            w.set_prob(trace.prob_table())
        if getattr(lw, "gauss", None):
            w.set_mean(0, trace.mean_param.value)
        return w

    def sweep(self, trace, config, seed, sweep_idx, lo=0, hi=None, reuse_buffers=False, light=False):
        orc, lw = self.oracle, self.lw
        hi = trace.cur.shape[1] if hi is None else hi
        n, nb = hi - lo, trace.cur.shape[0]
        choice = np.empty((nb, n), dtype=np.int32)
        chosen = np.empty(n, dtype=np.int32)
        logml = np.empty(n)
        new_rows = {}
        if n:
            w = self._world(trace, lo, hi)
            cfg = self._cfg(config)
            cur = np.ascontiguousarray(trace.cur[:, lo:hi])
            for bi in lw.locals:  # (only a prior-proposal sweep looks at them: its retained particle keeps the row's own choices)
                w.set_cur_locals(bi, trace.locals[bi][lo:hi])
            rc = orc.lib().pco_sweep_batched(w.h, C.byref(cfg), C.c_uint64(seed), C.c_uint32(sweep_idx), nb, C.c_int64(lo),
                                             orc._p(cur, C.c_int32), orc._p(choice, C.c_int32), orc._p(chosen, C.c_int32),
                                             orc._p(logml, C.c_double))
            if rc:
                raise ValueError("use_dd_proposals = false: plan shape not supported (Gaussian term off the slot or with context sources)")
            for b, blk in enumerate(lw.blocks):
                if blk.get("score"):
                    continue
                k = orc.lib().pco_new_rows_count(b)
                if k:
                    rows = np.empty(k, dtype=np.int32)
                    vals = np.empty((k, len(blk["nodes"])), dtype=np.int32)
                    orc.lib().pco_new_rows_get(b, len(blk["nodes"]), orc._p(rows, C.c_int32), orc._p(vals, C.c_int32))
                    new_rows[b] = (rows, vals)
            for bi in lw.locals:
                trace.pending_locals[bi] = w.get_locals(bi, n)
        else:
            for bi in lw.locals:
                trace.pending_locals[bi] = np.zeros((0, 2), dtype=np.int32)
        self._choice, self._cur = choice, np.ascontiguousarray(trace.cur[:, lo:hi])
        return choice, chosen, logml, new_rows

    def sweep_stats(self, trace):
        """Delta reference counts of the last sweep per block root table (what stats_kernel produces)."""
        out = {}
        for b, blk in enumerate(self.lw.blocks):
            if blk.get("score"):
                continue
            t = trace.tables[blk["root_class"]]
            ch, cur = self._choice[b], self._cur[b]
            moved = ch != cur
            d = -np.bincount(cur[moved & (cur >= 0)], minlength=t.n).astype(np.int64)
            out[b] = d + np.bincount(ch[moved & (ch >= 0)], minlength=t.n)
        return out

    def sweep_moved(self):
        """What pclean_get_moved returns: rows (relative to the swept window, ascending) whose referent changed."""
        out = {}
        for b, blk in enumerate(self.lw.blocks):
            if blk.get("score"):
                continue
            rows = np.flatnonzero(self._choice[b] != self._cur[b]).astype(np.int32)
            out[b] = (rows, self._choice[b][rows].astype(np.int32))
        return out

    def sweep_latent(self, trace, cname, config, seed, sweep_idx, live, ev_off, ev_rows, ev_ctx, excl):
        pl = self.lw.latent_plans[cname]
        w = self._world(trace, 0, self.obs.shape[1])
        return w.sweep_latent(self._cfg(config), seed, sweep_idx, pl["block_id"], pl["roots"], live, ev_off, ev_rows,
                              ev_ctx, excl, len(pl["nodes"]))

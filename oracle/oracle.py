"""ctypes wrapper of oracle/libpclean_oracle.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package `pclean_amd` never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpclean_oracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".cpp", ".h"))]
    srcs += [os.path.join(HERE, "..", "include", f) for f in os.listdir(os.path.join(HERE, "..", "include"))]
    stale = force or not os.path.exists(LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(LIB_PATH) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-C", HERE, "-B", "libpclean_oracle.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        dbl = C.c_double
        for name in ["pco_negbin_logpdf", "pco_normal_logpdf", "pco_add_typos", "pco_string_prior",
                     "pco_dummy_logmass", "pco_choose_uniformly", "pco_choose_proportionally",
                     "pco_transformed_gaussian", "pco_maybe_swap", "pco_time_prior", "pco_logsumexp",
                     "pco_py_existing", "pco_py_new", "pco_pitman_yor_score", "pco_ess", "pco_det_exp",
                     "pco_det_log"]:
            getattr(L, name).restype = dbl
        L.pco_negbin_logpdf.argtypes = [dbl, dbl, C.c_int]
        L.pco_normal_logpdf.argtypes = [dbl, dbl, dbl]
        L.pco_add_typos.argtypes = [C.c_int, C.c_int, C.c_int]
        L.pco_transformed_gaussian.argtypes = [dbl, dbl, dbl, dbl]
        L.pco_maybe_swap.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, dbl]
        L.pco_py_existing.argtypes = [C.c_int64, C.c_int64, dbl, dbl]
        L.pco_py_new.argtypes = [C.c_int64, C.c_int64, dbl, dbl]
        L.pco_det_exp.argtypes = [dbl]
        L.pco_det_log.argtypes = [dbl]
        L.pco_fixw.argtypes = [dbl]
        L.pco_fixw.restype = C.c_uint64
        L.pco_rand64.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.pco_rand64.restype = C.c_uint64
        L.pco_dummy_seed.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
        L.pco_dummy_seed.restype = C.c_uint64
        L.pco_world_create.restype = C.c_void_p
        _lib = L
    return _lib


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _ctx_cols(a):
    """per-item / per-evidence-row context values padded to the PCLEAN_MAX_CTX columns the library indexes"""
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.int32)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    if a.shape[1] < 4:
        a = np.concatenate([a, np.zeros((a.shape[0], 4 - a.shape[1]), dtype=np.int32)], axis=1)
    return np.ascontiguousarray(a)



def _sym(s, symmap=None):
    return np.array([ord(c) for c in s], dtype=np.uint16)


def osa(a, b):
    a, b = _sym(a), _sym(b)
    return lib().pco_osa(_p(a, C.c_uint16), len(a), _p(b, C.c_uint16), len(b))


def dl(a, b):
    a, b = _sym(a), _sym(b)
    return lib().pco_dl(_p(a, C.c_uint16), len(a), _p(b, C.c_uint16), len(b))


def pair_table(sym, off, obs_ids, lat_ids, mode):
    sym = np.ascontiguousarray(sym, np.uint16)
    off = np.ascontiguousarray(off, np.int64)
    obs_ids = np.ascontiguousarray(obs_ids, np.int32)
    lat_ids = np.ascontiguousarray(lat_ids, np.int32)
    out = np.empty((len(obs_ids), len(lat_ids)), dtype=np.uint16)
    lib().pco_pair_table(_p(sym, C.c_uint16), _p(off, C.c_int64), len(obs_ids), _p(obs_ids, C.c_int32), len(lat_ids),
                         _p(lat_ids, C.c_int32), int(mode), _p(out, C.c_uint16))
    return out


def add_typos(d, word_len, max_typos=-1):
    return lib().pco_add_typos(int(d), int(word_len), int(max_typos))


def add_typos_strings(observed, word, max_typos=None, mode="dl"):
    """logdensity(AddTypos(), observed, word, max_typos) — add_typos.jl:50-66."""
    if observed is None:
        return 0.0
    d = dl(observed, word) if mode == "dl" else osa(observed, word)
    return add_typos(d, len(word), -1 if max_typos is None else max_typos)


def string_prior(lm, min_len, max_len, init_p, trans_p):
    lm = np.ascontiguousarray(lm, np.uint8)
    init_p = np.ascontiguousarray(init_p, np.float64)
    trans_p = np.ascontiguousarray(trans_p, np.float64)
    return lib().pco_string_prior(_p(lm, C.c_uint8), len(lm), int(min_len), int(max_len), _p(init_p, C.c_double),
                                  _p(trans_p, C.c_double))


def dummy_logmass(atom_logps):
    a = np.ascontiguousarray(atom_logps, np.float64)
    return lib().pco_dummy_logmass(_p(a, C.c_double), len(a))


def choose_proportionally(observed, options, probs):
    o = np.ascontiguousarray(options, np.int32)
    p = np.ascontiguousarray(probs, np.float64)
    return lib().pco_choose_proportionally(int(observed), _p(o, C.c_int32), _p(p, C.c_double), len(o))


def logsumexp(x):
    x = np.ascontiguousarray(x, np.float64)
    return lib().pco_logsumexp(_p(x, C.c_double), len(x))


def ess(logw):
    x = np.ascontiguousarray(logw, np.float64)
    return lib().pco_ess(_p(x, C.c_double), len(x))


def pitman_yor_score(strength, discount, counts):
    c = np.ascontiguousarray(counts, np.int64)
    return lib().pco_pitman_yor_score(C.c_double(strength), C.c_double(discount), _p(c, C.c_int64), len(c))


def time_regex(s):
    a = np.array([ord(c) for c in s], dtype=np.uint32)
    return bool(lib().pco_time_regex(_p(a, C.c_uint32), len(a)))


def time_prior_atom(s):
    """discrete_proposal(::TimePrior) atom score (time_prior.jl:10)."""
    return lib().pco_time_prior() if time_regex(s) else -np.inf


class RandomOracle:
    """Drop-in for the six HipContext.random_* methods, computed by the CPU restatement (random.h)."""

    def random_add_typos(self, cp, off, max_typos, seed, stream, stride):
        cp = np.ascontiguousarray(cp, np.uint32)
        off = np.ascontiguousarray(off, np.int64)
        n = len(off) - 1
        out = np.zeros((n, stride), np.uint32)
        lens = np.zeros(n, np.int32)
        lib().pco_random_add_typos(n, _p(cp, C.c_uint32), _p(off, C.c_int64), int(max_typos), C.c_uint64(seed),
                                   C.c_uint32(stream), int(stride), _p(out, C.c_uint32), _p(lens, C.c_int32))
        return out, lens

    def random_string_prior(self, n, min_len, max_len, init_p, trans_p, seed, stream):
        init_p = np.ascontiguousarray(init_p, np.float64)
        trans_p = np.ascontiguousarray(trans_p, np.float64)
        stride = max(int(max_len), 1)
        out = np.zeros((n, stride), np.uint8)
        lens = np.zeros(n, np.int32)
        lib().pco_random_string_prior(n, int(min_len), int(max_len), _p(init_p, C.c_double), _p(trans_p, C.c_double),
                                      C.c_uint64(seed), C.c_uint32(stream), stride, _p(out, C.c_uint8),
                                      _p(lens, C.c_int32))
        return out, lens

    def random_string_prior_at(self, seeds, elems, min_len, max_len, init_p, trans_p, stream=0):
        seeds = np.ascontiguousarray(seeds, np.uint64)
        elems = np.ascontiguousarray(elems, np.uint32)
        init_p = np.ascontiguousarray(init_p, np.float64)
        trans_p = np.ascontiguousarray(trans_p, np.float64)
        n, stride = len(seeds), max(int(max_len), 1)
        out = np.zeros((n, stride), np.uint8)
        lens = np.zeros(n, np.int32)
        lib().pco_random_string_prior_at(n, _p(seeds, C.c_uint64), _p(elems, C.c_uint32), int(min_len), int(max_len),
                                         _p(init_p, C.c_double), _p(trans_p, C.c_double), C.c_uint32(stream), stride,
                                         _p(out, C.c_uint8), _p(lens, C.c_int32))
        return out, lens

    def random_categorical(self, n, logp, seed, stream):
        logp = np.ascontiguousarray(logp, np.float64)
        out = np.zeros(n, np.int32)
        lib().pco_random_categorical(n, len(logp), _p(logp, C.c_double), C.c_uint64(seed), C.c_uint32(stream),
                                     _p(out, C.c_int32))
        return out

    def random_normal(self, mean, std, fwd_scale, seed, stream):
        mean = np.ascontiguousarray(mean, np.float64)
        out = np.zeros(len(mean), np.float64)
        lib().pco_random_normal(len(mean), _p(mean, C.c_double), C.c_double(std), C.c_double(fwd_scale), C.c_uint64(seed),
                                C.c_uint32(stream), _p(out, C.c_double))
        return out

    def random_maybe_swap(self, prob, n_options, seed, stream):
        prob = np.ascontiguousarray(prob, np.float64)
        n_options = np.ascontiguousarray(n_options, np.int32)
        out = np.zeros(len(prob), np.int32)
        lib().pco_random_maybe_swap(len(prob), _p(prob, C.c_double), _p(n_options, C.c_int32), C.c_uint64(seed),
                                    C.c_uint32(stream), _p(out, C.c_int32))
        return out

    def random_time_prior(self, n, seed, stream):
        out = np.zeros((n, 3), np.int32)
        lib().pco_random_time_prior(n, C.c_uint64(seed), C.c_uint32(stream), _p(out, C.c_int32))
        return out


def philox(c, k):
    out = np.empty(4, dtype=np.uint32)
    lib().pco_philox(C.c_uint32(c[0]), C.c_uint32(c[1]), C.c_uint32(c[2]), C.c_uint32(c[3]), C.c_uint32(k[0]),
                     C.c_uint32(k[1]), _p(out, C.c_uint32))
    return out


def table_priors(counts, strength, discount):
    c = np.ascontiguousarray(counts, np.int64)
    full = np.empty(len(c))
    m1 = np.empty(len(c))
    scal = np.empty(4)
    lib().pco_table_priors(len(c), _p(c, C.c_int64), C.c_double(strength), C.c_double(discount), _p(full, C.c_double),
                           _p(m1, C.c_double), _p(scal, C.c_double))
    return full, m1, scal


class World:
    """Host-side copy of everything the product uploads, for the batched spec."""

    def __init__(self):
        self.L = lib()
        self.h = C.c_void_p(self.L.pco_world_create())

    def __del__(self):
        try:
            self.L.pco_world_destroy(self.h)
        except Exception:
            pass

    def set_obs(self, obs):
        obs = np.ascontiguousarray(obs, np.int32)
        self.L.pco_world_set_obs(self.h, obs.shape[1], obs.shape[0], _p(obs, C.c_int32))

    def set_density(self, max_r, max_d, max_len, nb, logl):
        nb = np.ascontiguousarray(nb, np.float64)
        logl = np.ascontiguousarray(logl, np.float64)
        self.L.pco_world_set_density(self.h, max_r, max_d, max_len, _p(nb, C.c_double), _p(logl, C.c_double))

    def set_pair(self, pid, d, lat_len):
        d = np.ascontiguousarray(d, np.uint16)
        lat_len = np.ascontiguousarray(lat_len, np.uint16)
        self.L.pco_world_set_pair(self.h, pid, d.shape[0], d.shape[1], _p(d, C.c_uint16), _p(lat_len, C.c_uint16))

    def set_ev_row_by_row(self, on):
        """evidence sets summed row by row (the reference's order) instead of aggregated per term"""
        self.L.pco_world_set_ev_row_by_row(self.h, int(bool(on)))

    def set_block_group(self, block_id, group):
        self.L.pco_world_set_block_group(self.h, int(block_id), int(group))

    def set_strings(self, sym, off):
        sym = np.ascontiguousarray(sym, np.uint16)
        off = np.ascontiguousarray(off, np.int64)
        self.L.pco_world_set_strings(self.h, len(off) - 1, _p(sym, C.c_uint16), _p(off, C.c_int64))

    def set_pair_strings(self, pid, obs_ids, dist_mode):
        obs_ids = np.ascontiguousarray(obs_ids, np.int32)
        self.L.pco_world_set_pair_strings(self.h, pid, len(obs_ids), _p(obs_ids, C.c_int32), int(dist_mode))

    def set_lm(self, init_p, trans_p, letter_sym):
        init_p = np.ascontiguousarray(init_p, np.float64)
        trans_p = np.ascontiguousarray(trans_p, np.float64)
        letter_sym = np.ascontiguousarray(letter_sym, np.uint16)
        self.L.pco_world_set_lm(self.h, _p(init_p, C.c_double), _p(trans_p, C.c_double), _p(letter_sym, C.c_uint16))

    def set_table(self, tid, cols, counts, logc_full, logc_m1, scal):
        cols = np.ascontiguousarray(cols, np.int32)
        counts = np.ascontiguousarray(counts, np.int64)
        logc_full = np.ascontiguousarray(logc_full, np.float64)
        logc_m1 = np.ascontiguousarray(logc_m1, np.float64)
        scal = np.ascontiguousarray(scal, np.float64)
        n_cols = cols.shape[0] if cols.ndim == 2 else 0
        self.L.pco_world_set_table(self.h, tid, len(counts), n_cols, _p(cols, C.c_int32), _p(counts, C.c_int64),
                                   _p(logc_full, C.c_double), _p(logc_m1, C.c_double), _p(scal, C.c_double))

    def set_options(self, tid, values, logp):
        values = np.ascontiguousarray(values, np.int32)
        logp = np.ascontiguousarray(logp, np.float64)
        self.L.pco_world_set_options(self.h, tid, len(values), _p(values, C.c_int32), _p(logp, C.c_double))

    def set_options_cols(self, tid, cols, logp):
        cols = np.ascontiguousarray(cols, np.int32)
        logp = np.ascontiguousarray(logp, np.float64)
        self.L.pco_world_set_options_cols(self.h, tid, cols.shape[1], cols.shape[0], _p(cols, C.c_int32),
                                          _p(logp, C.c_double))

    def set_numeric(self, x):
        x = np.ascontiguousarray(x, np.float64)
        self.L.pco_world_set_numeric(self.h, x.shape[1], x.shape[0], _p(x, C.c_double))

    def set_mean(self, tid, mean):
        mean = np.ascontiguousarray(mean, np.float64).reshape(-1)
        self.L.pco_world_set_mean(self.h, tid, len(mean), _p(mean, C.c_double))

    def set_gauss(self, block, node, g):
        self.L.pco_world_set_gauss(self.h, block, node, C.byref(g))

    def set_cur_locals(self, block, locals_):
        """current own choices ([n][2]) of the sweep window's rows: kept by the retained particle of a prior-proposal sweep"""
        a = np.ascontiguousarray(locals_, dtype=np.int32).reshape(-1, 2)
        self.L.pco_world_set_cur_locals(self.h, block, len(a), _p(a, C.c_int32))

    PRUNED_STATS = ("root_evaluations", "served_by_the_memo", "candidates_scored_exactly", "candidates_pruned",
                    "new_row_branches_skipped", "new_row_branches_evaluated", "child_memo_hits", "child_memo_misses",
                    "full_enumerations")

    def sweep_batched(self, cfg, seed, sweep, cur, n_rows=None, pruned=False, row_offset=0):
        """the batched sweep over the first n_rows rows of the World (default: all); pruned: with grouping + exact pruning
        (oracle/pruned.h).  Returns (choice [nb][n_rows], chosen, logml, stats or None)."""
        cur = np.ascontiguousarray(cur, dtype=np.int32)
        nb, n = cur.shape
        n_rows = n if n_rows is None else n_rows
        choice = np.full((nb, n), -3, dtype=np.int32)
        chosen = np.zeros(n, dtype=np.int32)
        logml = np.zeros(n)
        stats = None
        if pruned:
            st = np.zeros(9, dtype=np.uint64)
            rc = self.L.pco_sweep_batched_pruned(self.h, C.byref(cfg), C.c_uint64(seed), C.c_uint32(sweep), nb, C.c_int64(row_offset),
                                                 C.c_int(n_rows), _p(cur, C.c_int32), _p(choice, C.c_int32), _p(chosen, C.c_int32),
                                                 _p(logml, C.c_double), _p(st, C.c_uint64))
            stats = dict(zip(self.PRUNED_STATS, (int(x) for x in st)))
        else:
            assert n_rows == n
            rc = self.L.pco_sweep_batched(self.h, C.byref(cfg), C.c_uint64(seed), C.c_uint32(sweep), nb, C.c_int64(row_offset),
                                          _p(cur, C.c_int32), _p(choice, C.c_int32), _p(chosen, C.c_int32), _p(logml, C.c_double))
        if rc:
            raise ValueError("the oracle refused the sweep")
        return choice[:, :n_rows], chosen[:n_rows], logml[:n_rows], stats

    def get_locals(self, block, n_rows):
        out = np.empty((n_rows, 2), dtype=np.int32)
        self.L.pco_get_locals(block, n_rows, _p(out, C.c_int32))
        return out

    def set_prob(self, p):
        p = np.ascontiguousarray(p, np.float64)
        self.L.pco_world_set_prob(self.h, len(p), _p(p, C.c_double))

    def load_score_block(self, bid, obs_col, pair_table, val_src, key_src, nopt_fn, other_val, prob_fn, prob_a_src,
                         prob_b_src):
        a = [np.ascontiguousarray(x, np.int32).reshape(-1) for x in
             (obs_col, pair_table, val_src, key_src, nopt_fn, other_val, prob_a_src, prob_b_src)]
        self.L.pco_world_load_score_block(self.h, bid, len(a[0]), _p(a[0], C.c_int32), _p(a[1], C.c_int32),
                                          _p(a[2], C.c_int32), _p(a[3], C.c_int32), _p(a[4], C.c_int32),
                                          _p(a[5], C.c_int32), prob_fn, _p(a[6], C.c_int32), _p(a[7], C.c_int32))

    def set_fn(self, fid, fn):
        fn = np.ascontiguousarray(fn, np.int32)
        self.L.pco_world_set_fn(self.h, fid, fn.shape[0], fn.shape[1], _p(fn, C.c_int32))

    def load_block(self, bid, nodes, terms, children, colmap, ctx_src_block=(), ctx_src_col=()):
        children = np.ascontiguousarray(children, np.int32)
        colmap = np.ascontiguousarray(colmap, np.int32)
        csb = np.ascontiguousarray(ctx_src_block, np.int32)
        csc = np.ascontiguousarray(ctx_src_col, np.int32)
        self.L.pco_world_load_block(self.h, bid, len(nodes), nodes.ctypes.data_as(C.c_void_p), len(terms),
                                    terms.ctypes.data_as(C.c_void_p), len(children), _p(children, C.c_int32),
                                    len(colmap), _p(colmap, C.c_int32), len(csb), _p(csb, C.c_int32),
                                    _p(csc, C.c_int32))

    def sweep_latent(self, cfg, seed, sweep, block_id, roots, keys, ev_off, ev_rows, ev_ctx, excl, n_nodes):
        roots = np.ascontiguousarray(roots, np.int32)
        keys = np.ascontiguousarray(keys, np.int32)
        ev_off = np.ascontiguousarray(ev_off, np.int32)
        ev_rows = np.ascontiguousarray(ev_rows, np.int32)
        ev_ctx = _ctx_cols(ev_ctx)
        excl = np.ascontiguousarray(excl, np.int32)
        chosen = np.zeros(len(keys), dtype=np.int32)
        vals = np.full((len(keys), n_nodes), -2, dtype=np.int32)
        self.L.pco_sweep_latent(self.h, C.byref(cfg), C.c_uint64(seed), C.c_uint32(sweep), block_id, len(roots),
                                _p(roots, C.c_int32), len(keys), _p(keys, C.c_int32), _p(ev_off, C.c_int32),
                                _p(ev_rows, C.c_int32), _p(ev_ctx, C.c_int32), _p(excl, C.c_int32),
                                _p(chosen, C.c_int32), _p(vals, C.c_int32))
        return chosen, vals

    def eval_tree(self, block_id, node_id, row, ctxv, excl, n_scores):
        """(log-marginal, scores of every candidate + the new-row candidate last) of one row; the new-row branch is
        evaluated recursively."""
        ctxv = _ctx_cols(ctxv if ctxv is not None else np.zeros(2, np.int32)).reshape(-1)
        sc = np.empty(n_scores)
        self.L.pco_eval_tree.restype = C.c_double
        lse = self.L.pco_eval_tree(self.h, block_id, node_id, int(row), _p(ctxv, C.c_int32), int(excl), _p(sc, C.c_double),
                                   n_scores)
        return lse, sc

    def eval_tree_ev(self, block_id, node_id, ev_rows, ev_ctx, excl, n_scores):
        """eval_tree for a latent-class work item: node of a latent plan scored against an evidence set."""
        ev_rows = np.ascontiguousarray(ev_rows, np.int32)
        ev_ctx = _ctx_cols(ev_ctx)
        sc = np.empty(n_scores)
        self.L.pco_eval_tree_ev.restype = C.c_double
        lse = self.L.pco_eval_tree_ev(self.h, block_id, node_id, len(ev_rows), _p(ev_rows, C.c_int32), _p(ev_ctx, C.c_int32),
                                      int(excl), _p(sc, C.c_double), n_scores)
        return lse, sc

    def score_node(self, block_id, node_id, rows, ctxv=None, excl=None, snew=None, seed=0, sweep=0, n_draws=0,
                   n_cand=None, want_scores=False):
        rows = np.ascontiguousarray(rows, np.int32)
        n = len(rows)
        ctxv = _ctx_cols(ctxv)
        excl = None if excl is None else np.ascontiguousarray(excl, np.int32)
        snew = None if snew is None else np.ascontiguousarray(snew, np.float64)
        lse = np.empty(n)
        scores = np.empty((n, n_cand)) if want_scores else None
        draws = np.empty((n, n_draws), dtype=np.int32) if n_draws else None
        self.L.pco_score_node(self.h, block_id, node_id, n, _p(rows, C.c_int32), _p(ctxv, C.c_int32),
                              _p(excl, C.c_int32), _p(snew, C.c_double), C.c_uint64(seed), C.c_uint32(sweep), n_draws,
                              _p(lse, C.c_double), _p(scores, C.c_double), _p(draws, C.c_int32))
        return lse, scores, draws

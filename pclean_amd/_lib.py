"""ctypes binding of libpclean_hip.so (the C ABI declared in include/pclean_hip.h).

There is deliberately no CPU fallback: if the shared object is missing or no
gfx950 device is visible, every compute call raises `PCleanHipError`.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# PCLEAN_HIP_LIB: another build of the same ABI (e.g. the -DWAVE_PHASE_CLOCK measurement build, scripts/build_phase_clock.sh)
LIB_PATH = os.environ.get("PCLEAN_HIP_LIB") or os.path.join(HERE, "libpclean_hip.so")

MAX_CTX = 4
EV_MAX_STEPS = 4  # PCLEAN_EV_MAX_STEPS
CHOICE_NEW = -1
DIST_OSA, DIST_DL = 0, 1
DENS_ADD_TYPOS, DENS_EQUAL, DENS_MAYBE_SWAP = 0, 1, 2
NODE_FK, NODE_LEAF = 0, 1
DUMMY_STRING_PRIOR, DUMMY_TIME_PRIOR = 1, 2


class PCleanHipError(RuntimeError):
    pass


class Term(C.Structure):
    _fields_ = [("obs_col", C.c_int32), ("cand_col", C.c_int32), ("pair_table", C.c_int32),
                ("dens_kind", C.c_int32), ("max_typos", C.c_int32), ("ctx_slot", C.c_int32),
                ("fn_table", C.c_int32), ("ctx_mode", C.c_int32)]


class Node(C.Structure):
    _fields_ = [("kind", C.c_int32), ("table", C.c_int32), ("term_begin", C.c_int32), ("n_terms", C.c_int32),
                ("child_begin", C.c_int32), ("n_children", C.c_int32), ("parent", C.c_int32),
                ("parent_fk_col", C.c_int32), ("cacheable", C.c_int32), ("colmap_begin", C.c_int32),
                ("dummy_value", C.c_int32), ("dummy_spec", C.c_int32)]


class Gauss(C.Structure):
    _fields_ = [("x_col", C.c_int32), ("mean_table", C.c_int32), ("n_dims", C.c_int32), ("src_kind", C.c_int32 * 4),
                ("src", C.c_int32 * 4), ("stride", C.c_int32 * 4), ("n_locals", C.c_int32), ("local_n", C.c_int32 * 2),
                ("local_obs_col", C.c_int32 * 2), ("transform_src_kind", C.c_int32), ("transform_src", C.c_int32),
                ("fixed_locals", C.c_int32), ("pad", C.c_int32), ("t_scale", C.c_double * 4),
                ("t_logabsderiv", C.c_double * 4), ("sigma", C.c_double), ("t_x_col", C.c_int32 * 4),
                ("t_lad_col", C.c_int32 * 4)]


GSRC = {"cand": 0, "obs": 1, "local": 2, "itemctx": 3, "evctx": 4}


class InferConfig(C.Structure):
    _fields_ = [("num_iters", C.c_int32), ("num_particles", C.c_int32), ("use_dd_proposals", C.c_int32),
                ("use_lo_sweeps", C.c_int32), ("use_mh_instead_of_pg", C.c_int32), ("rejuv_frequency", C.c_int32),
                ("reporting_frequency", C.c_int32)]


class Timing(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("hot_kernel_ms", C.c_float), ("hot_kernel_launches", C.c_int32),
                ("reserved", C.c_int32), ("hot_kernel_alg_bytes", C.c_double)]


class RootStats(C.Structure):
    _fields_ = [("fast", C.c_int32), ("n_items", C.c_int32), ("n_groups", C.c_int32), ("n_cand", C.c_int32),
                ("kpad", C.c_int32), ("n_terms", C.c_int32), ("n_pre", C.c_int32), ("n_draws", C.c_int32),
                ("pre_obs_col", C.c_int32 * 3), ("overflow_items", C.c_int32), ("cstride", C.c_int32),
                ("full_scans", C.c_int32), ("fine_blocks", C.c_int32), ("scored_terms", C.c_int32),
                ("resolved_groups", C.c_int32), ("pre_scored", C.c_int32), ("lazy_entries", C.c_int32)]


class CommitSlot(C.Structure):
    _fields_ = [("table_id", C.c_int32), ("n_hw", C.c_int32), ("n_free", C.c_int32), ("cols_changed", C.c_int32),
                ("created", C.c_int32), ("deleted", C.c_int32), ("total", C.c_int64), ("live", C.c_int64),
                ("max_count", C.c_int64)]


class CommitSummary(C.Structure):
    _fields_ = [("fallback", C.c_int32), ("n_changed", C.c_int32), ("n_slots", C.c_int32), ("stats_reduced", C.c_int32),
                ("n_records", C.c_int32 * 16), ("n_distinct", C.c_int32 * 16), ("slot", CommitSlot * 16)]


COMMIT_FB_RECORDS, COMMIT_FB_DUMMY, COMMIT_FB_CAPACITY = 1, 2, 4

TERM_DTYPE = np.dtype([("obs_col", "<i4"), ("cand_col", "<i4"), ("pair_table", "<i4"), ("dens_kind", "<i4"),
                       ("max_typos", "<i4"), ("ctx_slot", "<i4"), ("fn_table", "<i4"), ("ctx_mode", "<i4")])
NODE_DTYPE = np.dtype([("kind", "<i4"), ("table", "<i4"), ("term_begin", "<i4"), ("n_terms", "<i4"),
                       ("child_begin", "<i4"), ("n_children", "<i4"), ("parent", "<i4"), ("parent_fk_col", "<i4"),
                       ("cacheable", "<i4"), ("colmap_begin", "<i4"), ("dummy_value", "<i4"), ("dummy_spec", "<i4")])

_lib = None


def _p(arr, ctype):
    if arr is None:
        return None
    return arr.ctypes.data_as(C.POINTER(ctype))


def _ctx_cols(a):
    """per-item / per-evidence-row context values padded to the PCLEAN_MAX_CTX columns the library indexes"""
    if a is None:
        return None
    a = np.ascontiguousarray(a, dtype=np.int32)
    if a.ndim == 1:
        a = a.reshape(1, -1)
    if a.shape[1] < MAX_CTX:
        a = np.concatenate([a, np.zeros((a.shape[0], MAX_CTX - a.shape[1]), dtype=np.int32)], axis=1)
    return np.ascontiguousarray(a)



def load_library(path=None):
    """dlopen the HIP library; raises PCleanHipError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise PCleanHipError(
            f"{path} not found: build it with `python -m pclean_amd.build` (hipcc, gfx950). "
            "pclean_amd has no CPU fallback.")
    lib = C.CDLL(path)
    lib.pclean_last_error.restype = C.c_char_p
    lib.pclean_version.restype = C.c_char_p
    _lib = lib
    return lib


def check(ctx_handle, rc, what):
    if rc != 0:
        lib = load_library()
        msg = lib.pclean_last_error(ctx_handle).decode() if ctx_handle else ""
        raise PCleanHipError(f"{what} failed with status {rc}: {msg}")


class HipContext:
    """Owns one `pclean_ctx` (one GPU)."""

    def __init__(self, device=0):
        self.lib = load_library()
        self._pinned, self._io = [], {}
        self.h = C.c_void_p()
        rc = self.lib.pclean_ctx_create(C.c_int(device), C.byref(self.h))
        if rc != 0:
            raise PCleanHipError(
                f"pclean_ctx_create(device={device}) failed with status {rc}"
                + (" (no gfx950 device visible; pclean_amd has no CPU fallback)" if rc == -3 else ""))

    def close(self):
        if self.h:
            for a in self._pinned:
                self.lib.pclean_unpin_host(self.h, C.c_void_p(a.ctypes.data))
            if self._io.get("cur_pinned"):  # a trace's `cur` registered in place by sweep(window=...)
                self.lib.pclean_unpin_host(self.h, C.c_void_p(self._io["cur_pinned"][0]))
            self._pinned, self._io = [], {}
            self.lib.pclean_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- data ---------------------------------------------------------------
    def load_strings(self, sym, off):
        sym = np.ascontiguousarray(sym, dtype=np.uint16)
        off = np.ascontiguousarray(off, dtype=np.int64)
        check(self.h, self.lib.pclean_load_strings(self.h, C.c_int32(len(off) - 1), _p(sym, C.c_uint16),
                                                   _p(off, C.c_int64)), "pclean_load_strings")

    def load_columns(self, obs):
        obs = np.ascontiguousarray(obs, dtype=np.int32)  # [n_cols][n_rows]
        n_cols, n_rows = obs.shape
        check(self.h, self.lib.pclean_load_columns(self.h, C.c_int32(n_rows), C.c_int32(n_cols), _p(obs, C.c_int32)),
              "pclean_load_columns")

    def build_pair_table(self, table_id, obs_ids, lat_ids, mode=DIST_OSA):
        obs_ids = np.ascontiguousarray(obs_ids, dtype=np.int32)
        lat_ids = np.ascontiguousarray(lat_ids, dtype=np.int32)
        check(self.h, self.lib.pclean_build_pair_table(self.h, C.c_int32(table_id), C.c_int32(len(obs_ids)),
                                                       _p(obs_ids, C.c_int32), C.c_int32(len(lat_ids)),
                                                       _p(lat_ids, C.c_int32), C.c_int32(mode)),
              "pclean_build_pair_table")

    def set_pair_table(self, table_id, table):
        table = np.ascontiguousarray(table, dtype=np.uint8)
        check(self.h, self.lib.pclean_set_pair_table(self.h, C.c_int32(table_id), C.c_int32(table.shape[0]),
                                                     C.c_int32(table.shape[1]), _p(table, C.c_uint8)),
              "pclean_set_pair_table")

    def get_pair_table(self, table_id, n_obs, n_lat):
        out = np.empty((n_obs, n_lat), dtype=np.uint16)
        check(self.h, self.lib.pclean_get_pair_table(self.h, C.c_int32(table_id), _p(out, C.c_uint16)),
              "pclean_get_pair_table")
        return out

    def get_pair_rows(self, table_id, obs_rows, n_lat):
        obs_rows = np.ascontiguousarray(obs_rows, dtype=np.int32)
        out = np.empty((len(obs_rows), n_lat), dtype=np.uint16)
        check(self.h, self.lib.pclean_get_pair_rows(self.h, C.c_int32(table_id), C.c_int32(len(obs_rows)),
                                                    _p(obs_rows, C.c_int32), _p(out, C.c_uint16)),
              "pclean_get_pair_rows")
        return out

    def set_row_offset(self, row_offset):
        check(self.h, self.lib.pclean_set_row_offset(self.h, C.c_int64(row_offset)), "pclean_set_row_offset")

    def get_density_tables(self):
        mr, md, ml = C.c_int32(), C.c_int32(), C.c_int32()
        check(self.h, self.lib.pclean_get_density_tables(self.h, C.byref(mr), C.byref(md), C.byref(ml), None, None),
              "pclean_get_density_tables")
        nb = np.empty((mr.value + 1, md.value + 1), dtype=np.float64)
        logl = np.empty(ml.value + 1, dtype=np.float64)
        check(self.h, self.lib.pclean_get_density_tables(self.h, C.byref(mr), C.byref(md), C.byref(ml),
                                                         _p(nb, C.c_double), _p(logl, C.c_double)),
              "pclean_get_density_tables")
        return mr.value, md.value, ml.value, nb, logl

    def string_prior_scores(self, lm, off, min_len, max_len, init_logp, trans_logp):
        lm = np.ascontiguousarray(lm, dtype=np.uint8)
        off = np.ascontiguousarray(off, dtype=np.int64)
        init_logp = np.ascontiguousarray(init_logp, dtype=np.float64)
        trans_logp = np.ascontiguousarray(trans_logp, dtype=np.float64)
        n = len(off) - 1
        out = np.empty(n, dtype=np.float64)
        check(self.h, self.lib.pclean_string_prior_scores(self.h, C.c_int32(n), _p(lm, C.c_uint8), _p(off, C.c_int64),
                                                          C.c_int32(min_len), C.c_int32(max_len),
                                                          _p(init_logp, C.c_double), _p(trans_logp, C.c_double),
                                                          _p(out, C.c_double)), "pclean_string_prior_scores")
        return out

    # -- candidate tables -----------------------------------------------------
    def root_flags(self, n_rows):
        """per row of the last sweep's block-0 root scan: bit 0 = re-run by the generic kernel, bit 1 = guess-and-refine"""
        out = np.empty(n_rows, dtype=np.int32)
        check(self.h, self.lib.pclean_debug_root_flags(self.h, C.c_int32(n_rows), _p(out, C.c_int32)), "pclean_debug_root_flags")
        return out

    def global_evidence_sort(self, on):
        check(self.h, self.lib.pclean_debug_global_evidence_sort(self.h, C.c_int32(int(on))), "pclean_debug_global_evidence_sort")

    def force_generic(self, on):
        check(self.h, self.lib.pclean_debug_force_generic(self.h, C.c_int32(int(on))), "pclean_debug_force_generic")

    def set_table(self, table_id, cols, counts, strength, discount, n_cols=None):
        """cols None (with n_cols given) keeps the columns uploaded before and refreshes the counts only."""
        counts = np.ascontiguousarray(counts, dtype=np.int64)
        if cols is None:
            n_rows = len(counts)
        else:
            cols = np.ascontiguousarray(cols, dtype=np.int32)  # [n_cols][n_rows]
            n_cols, n_rows = cols.shape if cols.ndim == 2 else (0, len(counts))
        check(self.h, self.lib.pclean_set_table(self.h, C.c_int32(table_id), C.c_int32(n_rows), C.c_int32(n_cols),
                                                _p(cols, C.c_int32), _p(counts, C.c_int64), C.c_double(strength),
                                                C.c_double(discount)), "pclean_set_table")

    def set_block_group(self, block_id, group):
        check(self.h, self.lib.pclean_set_block_group(self.h, C.c_int32(block_id), C.c_int32(group)), "pclean_set_block_group")

    def set_lm_tables(self, init_p, trans_p, letter_sym):
        init_p = np.ascontiguousarray(init_p, dtype=np.float64)
        trans_p = np.ascontiguousarray(trans_p, dtype=np.float64)
        letter_sym = np.ascontiguousarray(letter_sym, dtype=np.uint16)
        assert init_p.size == 28 and trans_p.size == 28 * 28 and letter_sym.size == 28
        check(self.h, self.lib.pclean_set_lm_tables(self.h, _p(init_p, C.c_double), _p(trans_p, C.c_double),
                                                    _p(letter_sym, C.c_uint16)), "pclean_set_lm_tables")

    def set_options(self, table_id, values, logp):
        values = np.ascontiguousarray(values, dtype=np.int32)
        logp = np.ascontiguousarray(logp, dtype=np.float64)
        check(self.h, self.lib.pclean_set_options(self.h, C.c_int32(table_id), C.c_int32(len(values)),
                                                  _p(values, C.c_int32), _p(logp, C.c_double)), "pclean_set_options")

    def set_options_cols(self, table_id, cols, logp):
        cols = np.ascontiguousarray(cols, dtype=np.int32)  # [n_cols][n_options]
        logp = np.ascontiguousarray(logp, dtype=np.float64)
        check(self.h, self.lib.pclean_set_options_cols(self.h, C.c_int32(table_id), C.c_int32(cols.shape[1]),
                                                       C.c_int32(cols.shape[0]), _p(cols, C.c_int32),
                                                       _p(logp, C.c_double)), "pclean_set_options_cols")

    def load_numeric_columns(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)  # [n_cols][n_rows]
        check(self.h, self.lib.pclean_load_numeric_columns(self.h, C.c_int32(x.shape[1]), C.c_int32(x.shape[0]),
                                                           _p(x, C.c_double)), "pclean_load_numeric_columns")

    def set_mean_table(self, table_id, mean):
        mean = np.ascontiguousarray(mean, dtype=np.float64).reshape(-1)
        check(self.h, self.lib.pclean_set_mean_table(self.h, C.c_int32(table_id), C.c_int32(len(mean)),
                                                     _p(mean, C.c_double)), "pclean_set_mean_table")

    def set_node_gauss(self, block_id, node_id, g):
        check(self.h, self.lib.pclean_set_node_gauss(self.h, C.c_int32(block_id), C.c_int32(node_id), C.byref(g)),
              "pclean_set_node_gauss")

    def set_cur_locals(self, block_id, locals_):
        """current own choices of EVERY observed row ([n_rows][2] int32; None clears): the retained particle of a sweep with
        use_dd_proposals = false keeps them"""
        if locals_ is None:
            check(self.h, self.lib.pclean_set_cur_locals(self.h, C.c_int32(block_id), None, C.c_int32(0)), "pclean_set_cur_locals")
            return
        a = np.ascontiguousarray(locals_, dtype=np.int32).reshape(-1, 2)
        check(self.h, self.lib.pclean_set_cur_locals(self.h, C.c_int32(block_id), _p(a, C.c_int32), C.c_int32(len(a))),
              "pclean_set_cur_locals")

    def get_locals(self, block_id, n_rows):
        out = np.empty((n_rows, 2), dtype=np.int32)
        check(self.h, self.lib.pclean_get_locals(self.h, C.c_int32(block_id), _p(out, C.c_int32)), "pclean_get_locals")
        return out

    def set_fn_table(self, fn_id, fn):
        fn = np.ascontiguousarray(fn, dtype=np.int32)
        check(self.h, self.lib.pclean_set_fn_table(self.h, C.c_int32(fn_id), C.c_int32(fn.shape[0]),
                                                   C.c_int32(fn.shape[1]), _p(fn, C.c_int32)), "pclean_set_fn_table")

    def set_prob_table(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        check(self.h, self.lib.pclean_set_prob_table(self.h, C.c_int32(len(p)), _p(p, C.c_double)),
              "pclean_set_prob_table")

    def load_score_block(self, block_id, obs_col, pair_table, val_src, key_src, nopt_fn, other_val, prob_fn,
                         prob_a_src, prob_b_src):
        a = [np.ascontiguousarray(x, dtype=np.int32).reshape(-1) for x in
             (obs_col, pair_table, val_src, key_src, nopt_fn, other_val, prob_a_src, prob_b_src)]
        check(self.h, self.lib.pclean_load_score_block(
            self.h, C.c_int32(block_id), C.c_int32(len(a[0])), _p(a[0], C.c_int32), _p(a[1], C.c_int32),
            _p(a[2], C.c_int32), _p(a[3], C.c_int32), _p(a[4], C.c_int32), _p(a[5], C.c_int32), C.c_int32(prob_fn),
            _p(a[6], C.c_int32), _p(a[7], C.c_int32)), "pclean_load_score_block")

    # -- random(dist, args...) -------------------------------------------------------
    def random_add_typos(self, cp, off, max_typos, seed, stream, stride):
        cp = np.ascontiguousarray(cp, dtype=np.uint32)
        off = np.ascontiguousarray(off, dtype=np.int64)
        n = len(off) - 1
        out = np.zeros((n, stride), dtype=np.uint32)
        lens = np.zeros(n, dtype=np.int32)
        check(self.h, self.lib.pclean_random_add_typos(
            self.h, C.c_int32(n), _p(cp, C.c_uint32), _p(off, C.c_int64), C.c_int32(max_typos), C.c_uint64(seed),
            C.c_uint32(stream), C.c_int32(stride), _p(out, C.c_uint32), _p(lens, C.c_int32)), "pclean_random_add_typos")
        return out, lens

    def random_string_prior(self, n, min_len, max_len, init_p, trans_p, seed, stream):
        init_p = np.ascontiguousarray(init_p, dtype=np.float64)
        trans_p = np.ascontiguousarray(trans_p, dtype=np.float64)
        stride = max(int(max_len), 1)
        out = np.zeros((n, stride), dtype=np.uint8)
        lens = np.zeros(n, dtype=np.int32)
        check(self.h, self.lib.pclean_random_string_prior(
            self.h, C.c_int32(n), C.c_int32(min_len), C.c_int32(max_len), _p(init_p, C.c_double),
            _p(trans_p, C.c_double), C.c_uint64(seed), C.c_uint32(stream), C.c_int32(stride), _p(out, C.c_uint8),
            _p(lens, C.c_int32)), "pclean_random_string_prior")
        return out, lens

    def random_string_prior_at(self, seeds, elems, min_len, max_len, init_p, trans_p, stream=0):
        """draw i with the private stream (seeds[i], elems[i]) — the value of a chosen ProposalDummyValue"""
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        elems = np.ascontiguousarray(elems, dtype=np.uint32)
        init_p = np.ascontiguousarray(init_p, dtype=np.float64)
        trans_p = np.ascontiguousarray(trans_p, dtype=np.float64)
        n, stride = len(seeds), max(int(max_len), 1)
        out = np.zeros((n, stride), dtype=np.uint8)
        lens = np.zeros(n, dtype=np.int32)
        check(self.h, self.lib.pclean_random_string_prior_at(
            self.h, C.c_int32(n), _p(seeds, C.c_uint64), _p(elems, C.c_uint32), C.c_int32(min_len), C.c_int32(max_len),
            _p(init_p, C.c_double), _p(trans_p, C.c_double), C.c_uint32(stream), C.c_int32(stride), _p(out, C.c_uint8),
            _p(lens, C.c_int32)), "pclean_random_string_prior_at")
        return out, lens

    def random_categorical(self, n, logp, seed, stream):
        logp = np.ascontiguousarray(logp, dtype=np.float64)
        out = np.zeros(n, dtype=np.int32)
        check(self.h, self.lib.pclean_random_categorical(self.h, C.c_int32(n), C.c_int32(len(logp)), _p(logp, C.c_double),
                                                         C.c_uint64(seed), C.c_uint32(stream), _p(out, C.c_int32)),
              "pclean_random_categorical")
        return out

    def random_normal(self, mean, std, fwd_scale, seed, stream):
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        out = np.zeros(len(mean), dtype=np.float64)
        check(self.h, self.lib.pclean_random_normal(self.h, C.c_int32(len(mean)), _p(mean, C.c_double), C.c_double(std),
                                                    C.c_double(fwd_scale), C.c_uint64(seed), C.c_uint32(stream),
                                                    _p(out, C.c_double)), "pclean_random_normal")
        return out

    def random_maybe_swap(self, prob, n_options, seed, stream):
        prob = np.ascontiguousarray(prob, dtype=np.float64)
        n_options = np.ascontiguousarray(n_options, dtype=np.int32)
        out = np.zeros(len(prob), dtype=np.int32)
        check(self.h, self.lib.pclean_random_maybe_swap(self.h, C.c_int32(len(prob)), _p(prob, C.c_double),
                                                        _p(n_options, C.c_int32), C.c_uint64(seed), C.c_uint32(stream),
                                                        _p(out, C.c_int32)), "pclean_random_maybe_swap")
        return out

    def random_time_prior(self, n, seed, stream):
        out = np.zeros((n, 3), dtype=np.int32)
        check(self.h, self.lib.pclean_random_time_prior(self.h, C.c_int32(n), C.c_uint64(seed), C.c_uint32(stream),
                                                        _p(out, C.c_int32)), "pclean_random_time_prior")
        return out

    def table_shape(self, table_id):
        """(rows, columns) of a candidate table as the library holds it (a latent table uploaded with spare capacity
        for the device-resident commit has more rows than the trace's table)"""
        nr, nc = C.c_int32(), C.c_int32()
        check(self.h, self.lib.pclean_table_shape(self.h, C.c_int32(table_id), C.byref(nr), C.byref(nc)), "pclean_table_shape")
        return nr.value, nc.value

    def get_table_priors(self, table_id, n_rows, is_options=False):
        cap = max(self.table_shape(table_id)[0], n_rows)
        full = np.empty(cap, dtype=np.float64)
        m1 = np.empty(cap, dtype=np.float64)
        scal = np.empty(4, dtype=np.float64)
        check(self.h, self.lib.pclean_get_table_priors(self.h, C.c_int32(table_id), _p(full, C.c_double),
                                                       None if is_options else _p(m1, C.c_double),
                                                       _p(scal, C.c_double)), "pclean_get_table_priors")
        return full[:n_rows], m1[:n_rows], scal

    def load_block(self, block_id, nodes, terms, children, colmap, ctx_src_block=(), ctx_src_col=()):
        nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
        terms = np.ascontiguousarray(terms, dtype=TERM_DTYPE)
        children = np.ascontiguousarray(children, dtype=np.int32)
        colmap = np.ascontiguousarray(colmap, dtype=np.int32)
        csb = np.ascontiguousarray(ctx_src_block, dtype=np.int32)
        csc = np.ascontiguousarray(ctx_src_col, dtype=np.int32)
        check(self.h, self.lib.pclean_load_block(
            self.h, C.c_int32(block_id), C.c_int32(len(nodes)), nodes.ctypes.data_as(C.c_void_p),
            C.c_int32(len(terms)), terms.ctypes.data_as(C.c_void_p), C.c_int32(len(children)),
            _p(children, C.c_int32), C.c_int32(len(colmap)), _p(colmap, C.c_int32), C.c_int32(len(csb)),
            _p(csb, C.c_int32), _p(csc, C.c_int32)), "pclean_load_block")

    # -- enumeration / sweep ---------------------------------------------------
    def score_node(self, block_id, node_id, rows, ctxv=None, excl=None, snew=None, seed=0, sweep=0, n_draws=0,
                   n_cand=None, want_scores=False):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        n = len(rows)
        ctxv = _ctx_cols(ctxv)
        excl = None if excl is None else np.ascontiguousarray(excl, dtype=np.int32)
        snew = None if snew is None else np.ascontiguousarray(snew, dtype=np.float64)
        lse = np.empty(n, dtype=np.float64)
        scores = np.empty((n, n_cand), dtype=np.float64) if want_scores else None
        draws = np.empty((n, n_draws), dtype=np.int32) if n_draws else None
        check(self.h, self.lib.pclean_score_node(
            self.h, C.c_int32(block_id), C.c_int32(node_id), C.c_int32(n), _p(rows, C.c_int32), _p(ctxv, C.c_int32),
            _p(excl, C.c_int32), _p(snew, C.c_double), C.c_uint64(seed), C.c_uint32(sweep), C.c_int32(n_draws),
            _p(lse, C.c_double), _p(scores, C.c_double), _p(draws, C.c_int32)), "pclean_score_node")
        return lse, scores, draws

    def score_node_ev(self, block_id, node_id, keys, ev_off, ev_rows, ev_ctx=None, excl=None, seed=0, sweep=0, n_draws=0,
                      n_cand=None, want_scores=False):
        """pclean_score_node_ev: one node of a latent plan for latent rows `keys` against their evidence sets."""
        keys = np.ascontiguousarray(keys, dtype=np.int32)
        n = len(keys)
        ev_off = np.ascontiguousarray(ev_off, dtype=np.int32)
        ev_rows = np.ascontiguousarray(ev_rows, dtype=np.int32)
        ev_ctx = _ctx_cols(ev_ctx)
        excl = None if excl is None else np.ascontiguousarray(excl, dtype=np.int32)
        lse = np.empty(n, dtype=np.float64)
        scores = np.empty((n, n_cand), dtype=np.float64) if want_scores else None
        draws = np.empty(n, dtype=np.int32) if n_draws else None
        check(self.h, self.lib.pclean_score_node_ev(
            self.h, C.c_int32(block_id), C.c_int32(node_id), C.c_int32(n), _p(keys, C.c_int32), _p(ev_off, C.c_int32),
            _p(ev_rows, C.c_int32), _p(ev_ctx, C.c_int32), _p(excl, C.c_int32), C.c_uint64(seed), C.c_uint32(sweep),
            C.c_int32(n_draws), _p(lse, C.c_double), _p(scores, C.c_double), _p(draws, C.c_int32)), "pclean_score_node_ev")
        return lse, scores, draws

    def pinned_empty(self, shape, dtype):
        """numpy array whose buffer is page-locked for this context (released with the context)."""
        a = np.empty(shape, dtype=dtype)
        if a.nbytes:
            check(self.h, self.lib.pclean_pin_host(self.h, C.c_void_p(a.ctypes.data), C.c_size_t(a.nbytes)),
                  "pclean_pin_host")
            self._pinned.append(a)
        return a

    def _io_buffers(self, n_blocks, n_rows):
        key = (n_blocks, n_rows)
        if self._io.get("key") != key:
            for a in self._io.get("arrays", ()):
                self.lib.pclean_unpin_host(self.h, C.c_void_p(a.ctypes.data))
                self._pinned = [p for p in self._pinned if p is not a]
            arrays = (self.pinned_empty((n_blocks, n_rows), np.int32), self.pinned_empty((n_blocks, n_rows), np.int32),
                      self.pinned_empty(n_rows, np.int32), self.pinned_empty(n_rows, np.float64))
            self._io.update(key=key, arrays=arrays)
        return self._io["arrays"]

    def sweep(self, cfg, seed, sweep_idx, cur, reuse_buffers=False, window=None, light=False):
        """cur [n_blocks][n_rows] -> (choice, chosen particle, log marginal likelihood).
        window=(lo, hi): cur is the array over ALL rows; rows [lo, hi) are passed in place (no copy) through
        pclean_set_cur_stride — with reuse_buffers the whole array is page-locked once.
        reuse_buffers: outputs go through page-locked buffers owned by the context (views that the next sweep
        of the same shape overwrites).  light: no output copies at all (the caller reads pclean_get_moved /
        pclean_get_new_rows); returns (None, None, None)."""
        n_blocks = np.shape(cur)[0]
        if window is not None:
            lo, hi = window
            assert cur.dtype == np.int32 and cur.flags.c_contiguous
            n_rows = hi - lo
            if reuse_buffers and self._io.get("cur_pinned") != (cur.ctypes.data, cur.nbytes):
                old = self._io.get("cur_pinned")
                if old:
                    self.lib.pclean_unpin_host(self.h, C.c_void_p(old[0]))
                check(self.h, self.lib.pclean_pin_host(self.h, C.c_void_p(cur.ctypes.data), C.c_size_t(cur.nbytes)),
                      "pclean_pin_host")
                self._io["cur_pinned"] = (cur.ctypes.data, cur.nbytes)
                self._io["cur_ref"] = cur
            check(self.h, self.lib.pclean_set_cur_stride(self.h, C.c_int64(cur.shape[1])), "pclean_set_cur_stride")
            cur_ptr = C.cast(C.c_void_p(cur.ctypes.data + 4 * lo), C.POINTER(C.c_int32))
        else:
            n_rows = np.shape(cur)[1]
            check(self.h, self.lib.pclean_set_cur_stride(self.h, C.c_int64(0)), "pclean_set_cur_stride")
            cur = np.ascontiguousarray(cur, dtype=np.int32)
            cur_ptr = _p(cur, C.c_int32)
        if light:
            choice = chosen = logml = None
        elif reuse_buffers:
            _, choice, chosen, logml = self._io_buffers(n_blocks, n_rows)
        else:
            choice = np.empty((n_blocks, n_rows), dtype=np.int32)
            chosen = np.empty(n_rows, dtype=np.int32)
            logml = np.empty(n_rows, dtype=np.float64)
        check(self.h, self.lib.pclean_sweep(self.h, C.byref(cfg), C.c_uint64(seed), C.c_uint32(sweep_idx),
                                            C.c_int32(n_blocks), cur_ptr, _p(choice, C.c_int32),
                                            _p(chosen, C.c_int32), _p(logml, C.c_double)), "pclean_sweep")
        return choice, chosen, logml

    def set_active_rows(self, begin, count):
        check(self.h, self.lib.pclean_set_active_rows(self.h, C.c_int32(begin), C.c_int32(count)),
              "pclean_set_active_rows")

    def sweep_latent(self, cfg, seed, sweep_idx, block_id, roots, keys, ev_off, ev_rows, ev_ctx, excl, n_nodes):
        roots = np.ascontiguousarray(roots, dtype=np.int32)
        keys = np.ascontiguousarray(keys, dtype=np.int32)
        ev_off = np.ascontiguousarray(ev_off, dtype=np.int32)
        ev_rows = np.ascontiguousarray(ev_rows, dtype=np.int32)
        ev_ctx = _ctx_cols(ev_ctx)
        excl = np.ascontiguousarray(excl, dtype=np.int32)  # [n_roots][n_items]
        n = len(keys)
        chosen = np.zeros(n, dtype=np.int32)
        vals = np.full((n, n_nodes), -2, dtype=np.int32)
        check(self.h, self.lib.pclean_sweep_latent(
            self.h, C.byref(cfg), C.c_uint64(seed), C.c_uint32(sweep_idx), C.c_int32(block_id), C.c_int32(len(roots)),
            _p(roots, C.c_int32), C.c_int32(n), _p(keys, C.c_int32), _p(ev_off, C.c_int32), _p(ev_rows, C.c_int32),
            _p(ev_ctx, C.c_int32), _p(excl, C.c_int32), _p(chosen, C.c_int32), _p(vals, C.c_int32)),
            "pclean_sweep_latent")
        return chosen, vals

    # -- C-level RCCL exchange (for hosts without torch.distributed; the Python host uses parallel.Comm) --
    def comm_unique_id(self):
        buf = (C.c_ubyte * 128)()
        check(self.h, self.lib.pclean_comm_unique_id(self.h, buf), "pclean_comm_unique_id")
        return bytes(buf)

    def comm_init(self, n_ranks, rank, unique_id):
        buf = (C.c_ubyte * 128).from_buffer_copy(unique_id)
        check(self.h, self.lib.pclean_comm_init(self.h, C.c_int32(n_ranks), C.c_int32(rank), buf), "pclean_comm_init")

    def allreduce_stats(self, table_id, n_rows):
        out = np.zeros(n_rows, dtype=np.int64)
        check(self.h, self.lib.pclean_allreduce_stats(self.h, C.c_int32(table_id), _p(out, C.c_int64)),
              "pclean_allreduce_stats")
        return out

    def allreduce_stats_fused(self, table_ids, n_rows, local_is_zero=False):
        """One RCCL all-reduce of the concatenated delta counts of `table_ids`; returns the list of summed vectors."""
        ids = np.ascontiguousarray(table_ids, dtype=np.int32)
        caps = [self.table_shape(int(t))[0] for t in ids]  # (the library's tables may hold spare capacity)
        out = np.zeros(int(sum(caps)), dtype=np.int64)
        check(self.h, self.lib.pclean_allreduce_stats_fused(self.h, C.c_int32(len(ids)), _p(ids, C.c_int32),
                                                            C.c_int32(int(local_is_zero)), _p(out, C.c_int64)),
              "pclean_allreduce_stats_fused")
        return [part[:n] for part, n in zip(np.split(out, np.cumsum(caps)[:-1]), n_rows)]

    def comm_destroy(self):
        check(self.h, self.lib.pclean_comm_destroy(self.h), "pclean_comm_destroy")

    def comm_stats(self):
        """what this context's collectives moved: calls, bytes of the last one, device milliseconds of all of them"""
        out = np.zeros(8, dtype=np.int64)
        check(self.h, self.lib.pclean_comm_get_stats(self.h, _p(out, C.c_int64)), "pclean_comm_get_stats")
        return {"allgather_calls": int(out[0]), "allgather_bytes_per_rank_last": int(out[1]), "allgather_device_ms": out[2] / 1e6,
                "allreduce_calls": int(out[3]), "allreduce_bytes_last": int(out[4]), "allreduce_device_ms": out[5] / 1e6,
                "n_ranks": int(out[6]), "rank": int(out[7])}

    def get_moved(self, block_id):
        n = C.c_int32()
        check(self.h, self.lib.pclean_get_moved(self.h, C.c_int32(block_id), C.byref(n), None, None), "pclean_get_moved")
        rows = np.empty(n.value, dtype=np.int32)
        ch = np.empty(n.value, dtype=np.int32)
        if n.value:
            check(self.h, self.lib.pclean_get_moved(self.h, C.c_int32(block_id), C.byref(n), _p(rows, C.c_int32),
                                                    _p(ch, C.c_int32)), "pclean_get_moved")
        return rows, ch

    def get_new_rows(self, block_id, n_nodes):
        n = C.c_int32()
        check(self.h, self.lib.pclean_get_new_rows(self.h, C.c_int32(block_id), C.byref(n), None, None),
              "pclean_get_new_rows")
        rows = np.empty(n.value, dtype=np.int32)
        vals = np.empty((n.value, n_nodes), dtype=np.int32)
        if n.value:
            check(self.h, self.lib.pclean_get_new_rows(self.h, C.c_int32(block_id), C.byref(n), _p(rows, C.c_int32),
                                                       _p(vals, C.c_int32)), "pclean_get_new_rows")
        return rows, vals

    def get_stats(self, table_id, n_rows):
        out = np.zeros(max(self.table_shape(table_id)[0], n_rows), dtype=np.int64)
        check(self.h, self.lib.pclean_get_stats(self.h, C.c_int32(table_id), _p(out, C.c_int64)), "pclean_get_stats")
        return out[:n_rows]

    # -- device-resident commit (include/pclean_hip.h) -------------------------------------------------------------
    def prepare(self, ev_blocks=0):
        check(self.h, self.lib.pclean_prepare(self.h, C.c_uint32(int(ev_blocks))), "pclean_prepare")

    def commit_enable(self, n_blocks):
        """(supported, reason)"""
        ok = C.c_int32()
        check(self.h, self.lib.pclean_commit_enable(self.h, C.c_int32(n_blocks), C.byref(ok)), "pclean_commit_enable")
        return bool(ok.value), ("" if ok.value else self.lib.pclean_last_error(self.h).decode())

    def commit_tables(self):
        n = C.c_int32()
        ids = np.zeros(16, dtype=np.int32)
        check(self.h, self.lib.pclean_commit_n_slots(self.h, C.byref(n), _p(ids, C.c_int32)), "pclean_commit_n_slots")
        return [int(t) for t in ids[:n.value]]

    def commit_set_table_state(self, table_id, n_hw, free):
        free = np.ascontiguousarray(free, dtype=np.int32)
        check(self.h, self.lib.pclean_commit_set_table_state(self.h, C.c_int32(table_id), C.c_int32(n_hw), C.c_int32(len(free)),
                                                             _p(free, C.c_int32) if len(free) else None),
              "pclean_commit_set_table_state")

    def commit_device(self, n_blocks, sweep_idx):
        out = CommitSummary()
        check(self.h, self.lib.pclean_commit_device(self.h, C.c_int32(n_blocks), C.c_uint32(sweep_idx), C.byref(out)),
              "pclean_commit_device")
        return out

    def commit_device_dist(self, n_blocks, sweep_idx, local_empty, max_local_rows):
        """collective over the ranks of comm_init: the commit of a sweep whose rows are sharded over them"""
        out = CommitSummary()
        check(self.h, self.lib.pclean_commit_device_dist(self.h, C.c_int32(n_blocks), C.c_uint32(sweep_idx),
                                                         C.c_int32(int(bool(local_empty))), C.c_int32(int(max_local_rows)),
                                                         C.byref(out)), "pclean_commit_device_dist")
        return out

    def commit_pull_table(self, table_id):
        """(state words, cols [n_cols][cap], counts, live, free stack, origin [cap][4]) of a latent table's device state"""
        cap, nc = self.table_shape(table_id)
        state = np.zeros(8, dtype=np.int32)
        cols = np.zeros((nc, cap), dtype=np.int32)
        counts = np.zeros(cap, dtype=np.int64)
        live = np.zeros(cap, dtype=np.uint8)
        free = np.zeros(cap, dtype=np.int32)
        origin = np.zeros((cap, 4), dtype=np.int32)
        check(self.h, self.lib.pclean_commit_pull_table(self.h, C.c_int32(table_id), _p(state, C.c_int32), _p(cols, C.c_int32),
                                                        _p(counts, C.c_int64), _p(live, C.c_uint8), _p(free, C.c_int32),
                                                        _p(origin, C.c_int32)), "pclean_commit_pull_table")
        return state, cols, counts, live.astype(bool), free[:state[1]], origin

    def set_cur(self, cur):
        cur = np.ascontiguousarray(cur, dtype=np.int32)
        check(self.h, self.lib.pclean_set_cur(self.h, C.c_int32(cur.shape[0]), _p(cur, C.c_int32)), "pclean_set_cur")

    def get_cur(self, n_blocks, n_rows, out=None):
        out = np.empty((n_blocks, n_rows), dtype=np.int32) if out is None else out
        assert out.dtype == np.int32 and out.flags.c_contiguous and out.shape == (n_blocks, n_rows)
        check(self.h, self.lib.pclean_get_cur(self.h, C.c_int32(n_blocks), _p(out, C.c_int32)), "pclean_get_cur")
        return out

    def drop_cur(self):
        check(self.h, self.lib.pclean_drop_cur(self.h), "pclean_drop_cur")

    def set_sweep_mode(self, deferred):
        check(self.h, self.lib.pclean_set_sweep_mode(self.h, C.c_int32(1 if deferred else 0)), "pclean_set_sweep_mode")

    def sweep_fetch(self):
        check(self.h, self.lib.pclean_sweep_fetch(self.h), "pclean_sweep_fetch")

    def sweep_device_cur(self, cfg, seed, sweep_idx, n_blocks):
        """pclean_sweep on the device-resident referents (cur == NULL), no output buffers"""
        check(self.h, self.lib.pclean_set_cur_stride(self.h, C.c_int64(0)), "pclean_set_cur_stride")
        check(self.h, self.lib.pclean_sweep(self.h, C.byref(cfg), C.c_uint64(seed), C.c_uint32(sweep_idx), C.c_int32(n_blocks),
                                            None, None, None, None), "pclean_sweep")

    def stats_device_ptr(self, table_id):
        ptr = C.c_void_p()
        n = C.c_int64()
        check(self.h, self.lib.pclean_stats_device_ptr(self.h, C.c_int32(table_id), C.byref(ptr), C.byref(n)),
              "pclean_stats_device_ptr")
        return ptr.value, n.value

    def get_root_stats(self):
        r = RootStats()
        check(self.h, self.lib.pclean_get_root_stats(self.h, C.byref(r)), "pclean_get_root_stats")
        return r

    def argsort_ids(self, ids, id_max):
        """np.argsort(ids, kind="stable") for ids in [-1, id_max] on the device (pclean_argsort_ids)"""
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        out = np.empty(len(ids), dtype=np.int32)
        check(self.h, self.lib.pclean_argsort_ids(self.h, C.c_int32(len(ids)), _p(ids, C.c_int32), C.c_int32(int(id_max)),
                                                  _p(out, C.c_int32)), "pclean_argsort_ids")
        return out

    def build_evidence(self, cur_block, steps, n_target_rows, sources):
        """pclean_build_evidence: steps = [(table id, reference-slot column)], sources = [(block, table id, value column)];
        returns off (int32 [n_target_rows + 1]); the ordered rows and their ctx values stay on the device"""
        st = np.ascontiguousarray([s[0] for s in steps], dtype=np.int32)
        sc = np.ascontiguousarray([s[1] for s in steps], dtype=np.int32)
        sb = np.ascontiguousarray([s[0] for s in sources], dtype=np.int32)
        stb = np.ascontiguousarray([s[1] for s in sources], dtype=np.int32)
        scl = np.ascontiguousarray([s[2] for s in sources], dtype=np.int32)
        off = np.empty(int(n_target_rows) + 1, dtype=np.int32)
        check(self.h, self.lib.pclean_build_evidence(
            self.h, C.c_int32(int(cur_block)), C.c_int32(len(st)), _p(st, C.c_int32) if len(st) else None,
            _p(sc, C.c_int32) if len(sc) else None, C.c_int32(int(n_target_rows)), C.c_int32(len(sb)),
            _p(sb, C.c_int32) if len(sb) else None, _p(stb, C.c_int32) if len(sb) else None,
            _p(scl, C.c_int32) if len(sb) else None, _p(off, C.c_int32)), "pclean_build_evidence")
        return off

    def get_evidence(self, begin, n, with_ctx=True):
        """rows [begin, begin + n) of the resident evidence: (observed row ids, ctx values [n][MAX_CTX] or None)"""
        rows = np.empty(int(n), dtype=np.int32)
        cx = np.empty((int(n), MAX_CTX), dtype=np.int32) if with_ctx else None
        check(self.h, self.lib.pclean_get_evidence(self.h, C.c_int32(int(begin)), C.c_int32(int(n)), _p(rows, C.c_int32),
                                                   _p(cx, C.c_int32) if with_ctx else None), "pclean_get_evidence")
        return rows, cx

    def sweep_latent_resident(self, cfg, seed, sweep_idx, block_id, roots, keys, ev_off, ev_begin, excl, n_nodes):
        """pclean_sweep_latent over rows [ev_begin + ev_off[t], ev_begin + ev_off[t + 1]) of the resident evidence"""
        roots = np.ascontiguousarray(roots, dtype=np.int32)
        keys = np.ascontiguousarray(keys, dtype=np.int32)
        ev_off = np.ascontiguousarray(ev_off, dtype=np.int32)
        excl = np.ascontiguousarray(excl, dtype=np.int32)  # [n_roots][n_items]
        n = len(keys)
        chosen = np.zeros(n, dtype=np.int32)
        vals = np.full((n, n_nodes), -2, dtype=np.int32)
        check(self.h, self.lib.pclean_sweep_latent_resident(
            self.h, C.byref(cfg), C.c_uint64(seed), C.c_uint32(sweep_idx), C.c_int32(block_id), C.c_int32(len(roots)),
            _p(roots, C.c_int32), C.c_int32(n), _p(keys, C.c_int32), _p(ev_off, C.c_int32), C.c_int32(int(ev_begin)),
            _p(excl, C.c_int32), _p(chosen, C.c_int32), _p(vals, C.c_int32)), "pclean_sweep_latent_resident")
        return chosen, vals

    def set_timed_block(self, block_id):
        """which block's root launch group get_timing().hot_kernel_* / get_root_stats() describe (default 0)"""
        check(self.h, self.lib.pclean_set_timed_block(self.h, C.c_int32(int(block_id))), "pclean_set_timed_block")

    def set_profiling(self, on):
        check(self.h, self.lib.pclean_set_profiling(self.h, C.c_int32(int(on))), "pclean_set_profiling")

    def get_profile(self):
        """{phase name: (milliseconds, recorded intervals)} accumulated since set_profiling(True)."""
        n = C.c_int32()
        check(self.h, self.lib.pclean_get_profile(self.h, C.c_int32(0), None, None, None, C.byref(n)), "pclean_get_profile")
        names = C.create_string_buffer(32 * max(n.value, 1))
        ms = np.zeros(max(n.value, 1), dtype=np.float32)
        cnt = np.zeros(max(n.value, 1), dtype=np.int32)
        check(self.h, self.lib.pclean_get_profile(self.h, n, names, _p(ms, C.c_float), _p(cnt, C.c_int32), C.byref(n)),
              "pclean_get_profile")
        raw = names.raw
        return {raw[32 * i:32 * i + 32].split(b"\0")[0].decode(): (float(ms[i]), int(cnt[i])) for i in range(n.value)}

    def get_timing(self):
        t = Timing()
        check(self.h, self.lib.pclean_get_timing(self.h, C.byref(t)), "pclean_get_timing")
        return t

    def maybe_resample(self, logw, retain_first, seed, sweep, block):
        logw = np.ascontiguousarray(logw, dtype=np.float64)
        n, p = logw.shape
        anc = np.empty((n, p), dtype=np.int32)
        inc = np.empty(n, dtype=np.float64)
        ess = np.empty(n, dtype=np.float64)
        check(self.h, self.lib.pclean_maybe_resample(self.h, C.c_int32(n), C.c_int32(p), _p(logw, C.c_double),
                                                     C.c_int32(int(retain_first)), C.c_uint64(seed), C.c_uint32(sweep),
                                                     C.c_uint32(block), _p(anc, C.c_int32), _p(inc, C.c_double),
                                                     _p(ess, C.c_double)), "pclean_maybe_resample")
        return anc, inc, ess

    def final_choice(self, logw, use_mh, is_csmc, seed, sweep):
        logw = np.ascontiguousarray(logw, dtype=np.float64)
        n, p = logw.shape
        chosen = np.empty(n, dtype=np.int32)
        tot = np.empty(n, dtype=np.float64)
        check(self.h, self.lib.pclean_final_choice(self.h, C.c_int32(n), C.c_int32(p), _p(logw, C.c_double),
                                                   C.c_int32(int(use_mh)), C.c_int32(int(is_csmc)), C.c_uint64(seed),
                                                   C.c_uint32(sweep), _p(chosen, C.c_int32), _p(tot, C.c_double)),
              "pclean_final_choice")
        return chosen, tot

    def debug_detmath(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        e = np.empty_like(x)
        l = np.empty_like(x)
        f = np.empty(len(x), dtype=np.uint64)
        check(self.h, self.lib.pclean_debug_detmath(self.h, C.c_int32(len(x)), _p(x, C.c_double), _p(e, C.c_double),
                                                    _p(l, C.c_double), _p(f, C.c_uint64)), "pclean_debug_detmath")
        return e, l, f

    def debug_rand64(self, seed, rows, site, particle, sweep):
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        out = np.empty(len(rows), dtype=np.uint64)
        check(self.h, self.lib.pclean_debug_rand64(self.h, C.c_int32(len(rows)), C.c_uint64(seed), _p(rows, C.c_uint32),
                                                   C.c_uint32(site), C.c_uint32(particle), C.c_uint32(sweep),
                                                   _p(out, C.c_uint64)), "pclean_debug_rand64")
        return out

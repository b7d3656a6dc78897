"""TEST INFRASTRUCTURE: drives tests/commit_host (the host build of pclean_amd/csrc/commit_core.h — the algorithm of the
device-resident commit, one thread, barriers as no-ops) on a host Trace, the way pclean_amd/csrc/commit.hip drives the
HIP build of the same header on the device state.  Lets the CPU suite hold the device commit's algorithm against the
product's host commit (parallel.exchange_and_commit) on real sweeps."""
import ctypes as C
import os
import subprocess

import numpy as np

from pclean_amd._lib import NODE_DTYPE

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "commit_host", "commit_host.cpp")
LIB = os.path.join(HERE, "commit_host", "libcommit_host.so")
CORE = os.path.join(os.path.dirname(HERE), "pclean_amd", "csrc", "commit_core.h")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(SRC), os.path.getmtime(CORE)):
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", LIB, SRC])
        _lib = C.CDLL(LIB)
        _lib.pcch_create.restype = C.c_void_p
    return _lib


def _ptr(a):
    return 0 if a is None else a.ctypes.data


class EmulatedDevice:
    """Device-side state of the commit for one trace: latent tables padded to a capacity, free stacks, referents."""

    def __init__(self, lw, trace, slack=64):
        L = lib()
        self.lw, self.L = lw, L
        self.h = C.c_void_p(L.pcch_create())
        self._keep = []
        for bi, blk in enumerate(lw.blocks):
            if blk.get("score"):
                nodes = np.zeros(0, dtype=NODE_DTYPE)
                ch = cm = np.zeros(0, dtype=np.int32)
                ov = np.zeros(0, dtype=np.int64)
            else:
                nodes, terms, ch, cm = lw.block_arrays(bi)[:4]
                nodes = np.ascontiguousarray(nodes, dtype=NODE_DTYPE)
                ov = np.zeros(len(nodes), dtype=np.int64)
                for i, info in enumerate(blk["node_info"]):
                    if info["kind"] == "leaf":
                        vals = np.ascontiguousarray(lw.option_values[(info["cls"], info["attr"])], dtype=np.int32)
                        self._keep.append(vals)
                        ov[i] = vals.ctypes.data
            ch = np.ascontiguousarray(ch, dtype=np.int32)
            cm = np.ascontiguousarray(cm, dtype=np.int32)
            self._keep += [nodes, ch, cm, ov]
            L.pcch_add_block(self.h, len(nodes), C.c_void_p(_ptr(nodes)), len(ch), C.c_void_p(_ptr(ch)), len(cm),
                             C.c_void_p(_ptr(cm)), C.c_void_p(_ptr(ov)), int(bool(blk.get("score"))))
        why = C.c_char_p()
        self.supported = L.pcch_build(self.h, C.byref(why)) == 0
        self.why = why.value.decode() if why.value else ""
        if not self.supported:
            return
        by_id = {tid: c for c, tid in lw.table_id.items()}
        self.slots = [by_id[L.pcch_slot_table(self.h, s)] for s in range(L.pcch_n_slots(self.h))]
        self.plans = [L.pcch_plan_block(self.h, p) for p in range(L.pcch_n_plans(self.h))]
        self.tab = {}
        for s, cname in enumerate(self.slots):
            t = trace.tables[cname]
            cap = t.n + slack
            cols = np.zeros((t.n_cols, cap), dtype=np.int32)
            cols[:, :t.n] = t.cols[:, :t.n]
            counts = np.zeros(cap, dtype=np.int64)
            counts[:t.n] = t.counts[:t.n]
            live = np.zeros(cap, dtype=np.uint8)
            live[:t.n] = t.live[:t.n]
            free = np.zeros(cap, dtype=np.int32)
            free[:len(t.free)] = t.free
            state = np.zeros(8, dtype=np.int32)
            state[0], state[1] = t.n, len(t.free)
            origin = np.zeros((cap, 4), dtype=np.int32)
            self.tab[cname] = dict(cols=cols, counts=counts, live=live, free=free, state=state, origin=origin, cap=cap)
            L.pcch_set_table(self.h, s, cap, t.n_cols, C.c_void_p(_ptr(cols)), C.c_void_p(_ptr(counts)), C.c_void_p(_ptr(live)),
                             C.c_void_p(_ptr(free)), C.c_void_p(_ptr(state)), C.c_void_p(_ptr(origin)))
        self.cur = np.ascontiguousarray(trace.cur.copy(), dtype=np.int32)
        self.row_origin = dict(trace.row_origin)

    def close(self):
        self.L.pcch_destroy(self.h)

    def commit(self, choice, chosen, new_rows, stats, sweep_idx, row_lo=0, kcap=4096):
        """choice [n_blocks][N], chosen [N], new_rows {block: (rows, vals)}, stats {block: delta counts}: one sweep's outputs.
        Returns (fallback bits, n_changed, records per block, distinct proposals per block)."""
        lw = self.lw
        N = choice.shape[1]
        P = len(self.plans)
        keep = []
        ptrs = {k: np.zeros(P, dtype=np.int64) for k in
                ("choice", "chosen", "newpos", "vals", "moved", "newl", "counts2", "cur", "delta")}
        nn = np.zeros(P, dtype=np.int32)
        chosen = np.ascontiguousarray(chosen, dtype=np.int32)
        for p, bi in enumerate(self.plans):
            blk = lw.blocks[bi]
            nn[p] = len(blk["nodes"])
            ch = np.ascontiguousarray(choice[bi], dtype=np.int32)
            rows, vals = new_rows.get(bi, (np.zeros(0, np.int32), np.zeros((0, nn[p]), np.int32)))
            rows = np.ascontiguousarray(rows, dtype=np.int32)
            vals = np.ascontiguousarray(np.array(vals, dtype=np.int32).reshape(-1, nn[p]))
            # the device keeps the records of EVERY proposed new row in an arbitrary order: shuffle them behind an indirection
            perm = np.random.default_rng(bi + 17).permutation(len(rows))
            store = np.ascontiguousarray(vals[perm]) if len(rows) else np.zeros((1, nn[p]), np.int32)
            newpos = np.full(N, -1, dtype=np.int32)
            inv = np.empty(len(rows), dtype=np.int32)
            inv[perm] = np.arange(len(rows), dtype=np.int32)
            newpos[rows] = inv
            moved = np.flatnonzero(ch != self.cur[bi]).astype(np.int32)
            counts2 = np.array([len(moved), len(rows)], dtype=np.int32)
            cap = self.tab[blk["root_class"]]["cap"]
            delta = np.zeros(cap, dtype=np.int64)
            d = np.asarray(stats[bi], dtype=np.int64)
            delta[:len(d)] = d
            curb = self.cur[bi]
            assert curb.flags.c_contiguous
            for k, a in (("choice", ch), ("chosen", chosen), ("newpos", newpos), ("vals", store), ("moved", moved), ("newl", rows),
                         ("counts2", counts2), ("cur", curb), ("delta", delta)):
                keep.append(a)
                ptrs[k][p] = a.ctypes.data
        res = np.zeros(34, dtype=np.int32)
        rc = self.L.pcch_commit(self.h, N, int(sweep_idx), int(row_lo), int(kcap), C.c_void_p(_ptr(nn)),
                                *[C.c_void_p(_ptr(ptrs[k])) for k in ("choice", "chosen", "newpos", "vals", "moved", "newl",
                                                                      "counts2", "cur", "delta")], C.c_void_p(_ptr(res)))
        assert rc == 0
        if not res[0]:  # what Engine.pull does with the origin marks
            for cname, tb in self.tab.items():
                for r in np.flatnonzero(tb["origin"][:, 0]):
                    mark = int(tb["origin"][r, 0])
                    if mark > 0:
                        self.row_origin[(cname, int(r))] = (int(tb["origin"][r, 1]), int(tb["origin"][r, 2]), int(tb["origin"][r, 3]),
                                                            mark - 1)
                    else:
                        self.row_origin.pop((cname, int(r)), None)
                tb["origin"][:] = 0
        return int(res[0]), int(res[1]), res[2:18].copy(), res[18:34].copy()

    def commit_gathered(self, shards, stats, sweep_idx, cap_m=None, cap_k=None):
        """The commit of a sweep whose rows were sharded over len(shards) ranks (pclean_commit_device_dist: pcc_pack per rank,
        the segments side by side, pcc_merge, the commit in its gathered form).  shards: [(lo, hi, choice [n_blocks][hi - lo],
        chosen [hi - lo], new_rows {block: (rows relative to lo, vals)})], contiguous and ascending; stats {block: delta counts
        summed over the shards}.  Returns what commit() returns."""
        lw = self.lw
        R, P = len(shards), len(self.plans)
        keep = []
        rp = {k: np.zeros(R * P, dtype=np.int64) for k in ("choice", "newpos", "vals", "moved", "newl", "counts2")}
        chosen_p = np.zeros(R, dtype=np.int64)
        N_r = np.array([hi - lo for lo, hi, *_ in shards], dtype=np.int32)
        lo_r = np.array([lo for lo, *_ in shards], dtype=np.int32)
        empty_r = (N_r == 0).astype(np.int32)
        nn = np.array([len(lw.blocks[bi]["nodes"]) for bi in self.plans], dtype=np.int32)
        max_m = np.zeros(P, dtype=np.int32)
        max_k = np.zeros(P, dtype=np.int32)
        for r, (lo, hi, choice, chosen, new_rows) in enumerate(shards):
            n = hi - lo
            ch_all = np.ascontiguousarray(chosen, dtype=np.int32) if n else np.zeros(1, np.int32)
            keep.append(ch_all)
            chosen_p[r] = ch_all.ctypes.data
            for p, bi in enumerate(self.plans):
                ch = np.ascontiguousarray(choice[bi], dtype=np.int32) if n else np.zeros(1, np.int32)
                rows, vals = new_rows.get(bi, (np.zeros(0, np.int32), np.zeros((0, nn[p]), np.int32)))
                rows = np.ascontiguousarray(rows, dtype=np.int32)
                vals = np.ascontiguousarray(np.array(vals, dtype=np.int32).reshape(-1, nn[p]))
                perm = np.random.default_rng(bi + 17 + r).permutation(len(rows))
                store = np.ascontiguousarray(vals[perm]) if len(rows) else np.zeros((1, nn[p]), np.int32)
                newpos = np.full(max(n, 1), -1, dtype=np.int32)
                inv = np.empty(len(rows), dtype=np.int32)
                inv[perm] = np.arange(len(rows), dtype=np.int32)
                newpos[rows] = inv
                moved = np.flatnonzero(ch[:n] != self.cur[bi][lo:hi]).astype(np.int32) if n else np.zeros(0, np.int32)
                counts2 = np.array([len(moved), len(rows)], dtype=np.int32)
                max_m[p] = max(max_m[p], len(moved))
                max_k[p] = max(max_k[p], len(rows))
                for k, a in (("choice", ch), ("newpos", newpos), ("vals", store), ("moved", moved if len(moved) else np.zeros(1, np.int32)),
                             ("newl", rows if len(rows) else np.zeros(1, np.int32)), ("counts2", counts2)):
                    keep.append(a)
                    rp[k][r * P + p] = a.ctypes.data
        cap_m = np.maximum(max_m, 4).astype(np.int32) if cap_m is None else np.full(P, cap_m, dtype=np.int32)
        cap_k = np.maximum(max_k, 4).astype(np.int32) if cap_k is None else np.full(P, cap_k, dtype=np.int32)
        cur_p = np.zeros(P, dtype=np.int64)
        delta_p = np.zeros(P, dtype=np.int64)
        for p, bi in enumerate(self.plans):
            cap = self.tab[lw.blocks[bi]["root_class"]]["cap"]
            delta = np.zeros(cap, dtype=np.int64)
            d = np.asarray(stats[bi], dtype=np.int64)
            delta[:len(d)] = d
            keep.append(delta)
            delta_p[p] = delta.ctypes.data
            assert self.cur[bi].flags.c_contiguous
            cur_p[p] = self.cur[bi].ctypes.data
        res = np.zeros(34, dtype=np.int32)
        rc = self.L.pcch_commit_gathered(self.h, R, int(self.cur.shape[1]), int(sweep_idx), C.c_void_p(_ptr(N_r)), C.c_void_p(_ptr(lo_r)),
                                         C.c_void_p(_ptr(empty_r)), C.c_void_p(_ptr(nn)), C.c_void_p(_ptr(cap_m)), C.c_void_p(_ptr(cap_k)),
                                         C.c_void_p(_ptr(rp["choice"])), C.c_void_p(_ptr(chosen_p)), C.c_void_p(_ptr(rp["newpos"])),
                                         C.c_void_p(_ptr(rp["vals"])), C.c_void_p(_ptr(rp["moved"])), C.c_void_p(_ptr(rp["newl"])),
                                         C.c_void_p(_ptr(rp["counts2"])), C.c_void_p(_ptr(cur_p)), C.c_void_p(_ptr(delta_p)),
                                         C.c_void_p(_ptr(res)))
        assert rc == 0
        if not res[0]:
            for cname, tb in self.tab.items():
                for r in np.flatnonzero(tb["origin"][:, 0]):
                    mark = int(tb["origin"][r, 0])
                    if mark > 0:
                        self.row_origin[(cname, int(r))] = (int(tb["origin"][r, 1]), int(tb["origin"][r, 2]), int(tb["origin"][r, 3]),
                                                            mark - 1)
                    else:
                        self.row_origin.pop((cname, int(r)), None)
                tb["origin"][:] = 0
        return int(res[0]), int(res[1]), res[2:18].copy(), res[18:34].copy()

    def assert_equals_trace(self, trace, what=""):
        """the emulated device state == the host trace after the host commit of the same sweeps"""
        for cname, tb in self.tab.items():
            t = trace.tables[cname]
            n = int(tb["state"][0])
            assert n == t.n, (what, cname, "high-water mark", n, t.n)
            assert list(tb["free"][:tb["state"][1]]) == list(t.free), (what, cname, "free list")
            assert np.array_equal(tb["live"][:n].astype(bool), t.live[:n]), (what, cname, "live flags")
            assert np.array_equal(tb["counts"][:n], t.counts[:n]), (what, cname, "reference counts")
            assert np.array_equal(tb["cols"][:, :n], t.cols[:, :n]), (what, cname, "columns")
            # rows beyond the high-water mark were never written
            assert not tb["live"][n:].any() and not tb["counts"][n:].any(), (what, cname, "rows beyond the high-water mark")
        assert np.array_equal(self.cur, trace.cur), (what, "current referents")
        assert self.row_origin == trace.row_origin, (what, "row origins")

"""diagnostic: where does the device-resident state diverge from a fresh upload?"""
import copy, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from pclean_amd import _lib, inference as inf
from pclean_amd.engine import Engine, InferenceConfig
from pclean_amd.trace import Trace

S = helpers.hospital_setup(n_rows=500)
lw, obs = S["lw"], S["obs"]
cfg = InferenceConfig(1, 8)
eng = Engine(lw, obs, dist_mode=_lib.DIST_DL)
ref = Engine(lw, obs, dist_mode=_lib.DIST_DL)
tr = Trace(lw, obs.shape[1], 1)
inf.initialize_trace(eng, tr, cfg, 11, max_batch=64)
assert eng.enable_device_commit(tr)
for sweep in range(4):
    host = copy.deepcopy(tr)
    ref.upload_trace(host)
    for cname, t in host.tables.items():
        tid = lw.table_id[cname]
        a = eng.hip.get_table_priors(tid, t.n)
        b = ref.hip.get_table_priors(tid, t.n)
        for k, nm in enumerate(("logc_full", "logc_m1", "scal")):
            if not np.array_equal(a[k], b[k]):
                bad = np.flatnonzero(a[k] != b[k])
                print(f"sweep {sweep} {cname} {nm} differs at {bad[:10]}: dev {a[k][bad[:5]]} ref {b[k][bad[:5]]} counts {t.counts[bad[:5]] if k<2 else ''}")
        cap = eng.hip.table_shape(tid)[0]
        st, cols, counts, live, free, origin = eng.hip.commit_pull_table(tid)
        if not np.array_equal(counts[:t.n], t.counts[:t.n]) or counts[t.n:].any():
            print(f"sweep {sweep} {cname}: device counts differ")
        if not np.array_equal(cols[:, :t.n], t.cols[:, :t.n]):
            print(f"sweep {sweep} {cname}: device cols differ")
    if sweep == 2:
        rows = np.array([312, 473, 493, 0, 15], dtype=np.int32)
        blk = lw.blocks[1]
        col = blk["ctx_src_col"][0]
        hosp = host.tables[lw.blocks[0]["root_class"]]
        ctxv = np.stack([hosp.cols[col, host.cur[0][rows]], np.zeros(len(rows))], axis=1).astype(np.int32)
        excl = host.cur[1][rows].astype(np.int32)
        n = host.tables[blk["root_class"]].n
        cap = eng.hip.table_shape(lw.table_id[blk["root_class"]])[0]
        eng.hip.set_active_rows(0, -1); ref.hip.set_active_rows(0, -1)
        a = eng.hip.score_node(1, 0, rows, ctxv, excl, None, seed=1, sweep=1, n_draws=0, n_cand=cap + 1, want_scores=True)
        b = ref.hip.score_node(1, 0, rows, ctxv, excl, None, seed=1, sweep=1, n_draws=0, n_cand=n + 1, want_scores=True)
        print("lse dev", a[0], "ref", b[0])
        for i, r in enumerate(rows):
            sa, sb = a[1][i], b[1][i]
            print(f" row {r}: existing-candidate scores equal {np.array_equal(sa[:n], sb[:n])}; padding all -inf {np.all(np.isneginf(sa[n:cap]))}; new dev {sa[cap]} ref {sb[n]}")
            if not np.array_equal(sa[:n], sb[:n]):
                d = np.flatnonzero(sa[:n] != sb[:n]); print("   cand", d, sa[d], sb[d])
        for node in range(1, len(blk["nodes"])):
            try:
                a = eng.hip.score_node(1, node, rows, ctxv, None, None, seed=1, sweep=1, n_draws=0)
                b = ref.hip.score_node(1, node, rows, ctxv, None, None, seed=1, sweep=1, n_draws=0)
                print("  node", node, blk["node_info"][node], "lse equal", np.array_equal(a[0], b[0]), a[0][:3], b[0][:3])
            except Exception as e:
                print("  node", node, "err", e)
    _, _, _, ref_new = ref.sweep(host, cfg, 42, sweep, light=True)
    ref_moved = ref.sweep_moved()
    changed = eng.sweep_commit_device(tr, cfg, 42, sweep)
    eng.hip.sweep_fetch()
    moved = eng.sweep_moved()
    for bi in ref_moved:
        same = np.array_equal(moved[bi][0], ref_moved[bi][0]) and np.array_equal(moved[bi][1], ref_moved[bi][1])
        print(f"sweep {sweep} block {bi}: moved lists equal: {same}; changed {changed}; dc {eng._dc['alloc']}")
        if not same and np.array_equal(moved[bi][0], ref_moved[bi][0]):
            d = np.flatnonzero(moved[bi][1] != ref_moved[bi][1])
            print("   rows", moved[bi][0][d], "dev", moved[bi][1][d], "ref", ref_moved[bi][1][d], "cur", host.cur[bi][moved[bi][0][d]])
            root = lw.blocks[bi]["root_class"]
            t = host.tables[root]
            print("   ref-chosen rows: counts", t.counts[ref_moved[bi][1][d]], "live", t.live[ref_moved[bi][1][d]], "n", t.n)
eng.close(); ref.close()

"""Full-size (BASELINE.json configs[4]: 1M rows, 10k latent hospitals, 20 particles)
checks through size-independent properties: determinism, bit-exact parity with the
oracle on row chunks spread over the table (first / middle / last rows), conservation
of the reference-count deltas, and sanity of the moved fraction."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


DIST_MODES = ["dl", "osa"]  # unrestricted Damerau-Levenshtein (the Engine's default, what bench.py times) / restricted (OSA)


def _dist(name):
    from pclean_amd import _lib
    return {"dl": _lib.DIST_DL, "osa": _lib.DIST_OSA}[name]


def test_million_row_sweep_properties(oracle):
    sys.path.insert(0, ROOT)
    import bench
    from pclean_amd import _lib
    from pclean_amd._lib import InferConfig
    from pclean_amd.engine import Engine, InferenceConfig
    n_rows, n_hosp, P, seed = 1_000_000, 10_000, 20, 20250926
    import helpers
    dirty, clean, lw, obs, tr = helpers.truth_workload(n_rows, n_hosp, seed)
    eng = Engine(lw, obs)  # (the default distance: unrestricted DL, as bench.py)
    assert eng.dist_mode == _lib.DIST_DL
    try:
        eng.upload_trace(tr)
        cfg = InferenceConfig(1, P)
        choice, chosen, logml, new_rows = eng.sweep(tr, cfg, seed, 0)
        stats = eng.sweep_stats(tr)
        # pclean_get_moved = exactly the rows whose referent differs from cur, ascending, with their choice
        for bi, (rows_m, ch_m) in eng.sweep_moved().items():
            want = np.flatnonzero(choice[bi] != tr.cur[bi])
            assert np.array_equal(rows_m, want) and np.array_equal(ch_m, choice[bi][want])
        # (a) determinism
        choice2, chosen2, logml2, _ = eng.sweep(tr, cfg, seed, 0)
        assert np.array_equal(choice, choice2) and np.array_equal(chosen, chosen2) and np.array_equal(logml, logml2)
        # (b) parity on chunks spread over the table
        c = InferConfig(1, P, 1, 1, 0, 50, 100)
        # ... plus rows the sweep moved and rows that picked a brand-new referent (grouped launches, the
        # integer pre-filter and the generic fallback all meet here); chunks must be contiguous for the
        # oracle's row-keyed RNG, so take the 120-row window around such rows
        moved = np.nonzero((choice != tr.cur).any(axis=0))[0]
        fresh = np.nonzero((choice < 0).any(axis=0))[0]
        starts = [0, n_rows // 2 - 37, n_rows - 120]
        for special in (moved[:1], moved[len(moved) // 2:len(moved) // 2 + 1], fresh[:1], fresh[-1:]):
            if len(special):
                starts.append(int(min(max(special[0] - 60, 0), n_rows - 120)))
        assert len(moved) > 100 and len(fresh) > 10
        for start in starts:
            rows = np.arange(start, start + 120)
            w, _ = bench.oracle_world_for_rows(oracle, lw, obs, tr, eng, rows)
            cur = np.ascontiguousarray(tr.cur[:, rows])
            och = np.empty((2, len(rows)), dtype=np.int32)
            ocp = np.empty(len(rows), dtype=np.int32)
            oml = np.empty(len(rows))
            oracle.lib().pco_sweep_batched(w.h, C.byref(c), C.c_uint64(seed), C.c_uint32(0), 2, C.c_int64(start),
                                           oracle._p(cur, C.c_int32), oracle._p(och, C.c_int32),
                                           oracle._p(ocp, C.c_int32), oracle._p(oml, C.c_double))
            assert np.array_equal(choice[:, rows], och), start
            assert np.array_equal(chosen[rows], ocp), start
            assert np.array_equal(logml[rows], oml), start
        # (c) conservation: every moved row takes one reference away and (unless NEW) adds one
        for bi, blk in enumerate(lw.blocks):
            t = tr.tables[blk["root_class"]]
            moved = choice[bi] != tr.cur[bi]
            want = -np.bincount(tr.cur[bi][moved], minlength=t.n)
            ex = moved & (choice[bi] >= 0)
            want = want + np.bincount(choice[bi][ex], minlength=t.n)
            assert np.array_equal(stats[bi], want)
            assert stats[bi].sum() == -int((choice[bi] < 0).sum())
            got_new = new_rows.get(bi, (np.zeros(0, np.int32), None))[0]
            assert np.array_equal(np.sort(got_new), np.nonzero(choice[bi] < 0)[0])
        # (d) started from the true entities: a sweep must leave almost every row where it is
        assert (choice[0] != tr.cur[0]).mean() < 0.01
        assert (choice[1] != tr.cur[1]).mean() < 0.05
        # chosen particle is uniform-ish over the 20 particles (weights equal within a context)
        hist = np.bincount(chosen, minlength=P) / n_rows
        assert hist.min() > 0.03 and hist.max() < 0.07
    finally:
        eng.close()


@pytest.mark.parametrize("dist", DIST_MODES)
def test_fast_root_kernel_equals_generic(oracle, dist):
    """The compact-table wave kernel (root_wave.hip) and the generic enumeration kernel give
    bit-identical sweeps; exclusions (incl. sole referrers), new rows and counts-only uploads covered."""
    sys.path.insert(0, ROOT)
    import bench
    from pclean_amd import _lib
    from pclean_amd.engine import Engine, InferenceConfig
    from pclean_amd.parallel import Comm, exchange_and_commit
    import helpers
    dirty, clean, lw, obs, tr = helpers.truth_workload(40_000, 1500, 7)
    eng = Engine(lw, obs, dist_mode=_dist(dist))
    comm = Comm()
    try:
        cfg = InferenceConfig(1, 6)
        for sweep in range(3):
            eng.upload_trace(tr)
            eng.hip.force_generic(False)
            a = eng.sweep(tr, cfg, 11, sweep)
            sa = eng.sweep_stats(tr)
            eng.hip.force_generic(True)
            b = eng.sweep(tr, cfg, 11, sweep)
            eng.hip.force_generic(False)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
            assert set(a[3]) == set(b[3])
            for k in a[3]:
                assert np.array_equal(a[3][k][0], b[3][k][0]) and np.array_equal(a[3][k][1], b[3][k][1])
            exchange_and_commit(tr, lw, comm, 0, a[0], sa, a[3])
            for bi, blk in enumerate(lw.blocks):
                t = tr.tables[blk["root_class"]]
                assert np.array_equal(np.bincount(tr.cur[bi], minlength=t.n), t.counts[:t.n])
    finally:
        eng.close()


@pytest.mark.parametrize("dist", DIST_MODES)
def test_million_row_own_init_state_parity(oracle, capsys, dist):
    """On the tables bench.py times (dist = dl: unrestricted Damerau-Levenshtein, the product's default) and on the restricted
    flavour of rounds 1-4.  The state bench.py actually times: the build's OWN initialize_trace from an empty trace + one full
    run_inference iteration at 1M rows (11-12k latent hospitals for 10k true ones, thousands of singleton measures,
    guess-and-refine groups, groups whose survivor list overflows, thousands of new rows per sweep) — then one
    observed-class sweep checked (a) bit for bit against the oracle on >= 8 windows that contain rows of every kind
    (pclean_debug_root_flags picks them), (b) compact-table wave kernels == generic enumeration kernels on all rows."""
    sys.path.insert(0, ROOT)
    import bench
    from pclean_amd import _lib
    from pclean_amd._lib import InferConfig
    from pclean_amd.engine import Engine, InferenceConfig
    from pclean_amd.inference import initialize_trace, run_inference
    from pclean_amd.trace import Trace
    n_rows, n_hosp, P, seed = 1_000_000, 10_000, 20, 20250926
    dirty, clean, lw, obs = bench.build_workload(n_rows, n_hosp, seed)
    eng = Engine(lw, obs, dist_mode=_dist(dist))
    try:
        cfg = InferenceConfig(1, P)
        tr = Trace(lw, n_rows, seed)
        initialize_trace(eng, tr, cfg, seed, max_batch=32768)
        run_inference(eng, tr, cfg, seed)
        tr.check_consistency()
        n_h, n_m = tr.tables["Hospital"].n_live, tr.tables["Measure"].n_live
        assert n_h > n_hosp and n_m > 28  # the messy state, not the ground truth
        eng.upload_trace(tr)
        choice, chosen, logml, new_rows = eng.sweep(tr, cfg, seed, 1)
        choice, chosen, logml = choice.copy(), chosen.copy(), logml.copy()
        new_rows = {k: (v[0].copy(), v[1].copy()) for k, v in new_rows.items()}
        flags = eng.hip.root_flags(n_rows)
        rs = eng.hip.get_root_stats()
        over, refine = np.flatnonzero(flags & 1), np.flatnonzero(flags & 2)
        moved = np.flatnonzero((choice != tr.cur).any(axis=0))
        fresh0, fresh1 = np.flatnonzero(choice[0] < 0), np.flatnonzero(choice[1] < 0)
        with capsys.disabled():
            print(f"\n[own-init 1M, {dist} tables] latent hospitals {n_h}, measures {n_m}; groups {rs.n_groups}, overflowed rows {len(over)}, "
                  f"guess-and-refine rows {len(refine)}, moved {len(moved)}, new hospital rows {len(fresh0)}, "
                  f"new measure rows {len(fresh1)}")
        assert rs.fast == 1 and len(refine) > 0 and len(moved) > 1000 and len(fresh0) + len(fresh1) > 100
        # (a) oracle parity on windows around rows of every kind
        W = 96
        starts = [0, n_rows // 2 - 37, n_rows - W]
        for special in (over[:1], over[-1:], refine[:1], refine[len(refine) // 2:len(refine) // 2 + 1], refine[-1:],
                        moved[len(moved) // 2:len(moved) // 2 + 1], fresh0[:1], fresh0[-1:], fresh1[:1], fresh1[-1:]):
            if len(special):
                starts.append(int(min(max(special[0] - W // 2, 0), n_rows - W)))
        assert len(starts) >= 8
        c = InferConfig(1, P, 1, 1, 0, 50, 100)
        kinds = np.zeros(4, dtype=np.int64)
        n_pairs_checked = 0
        for start in starts:
            rows = np.arange(start, start + W)
            w, _ = bench.oracle_world_for_rows(oracle, lw, obs, tr, eng, rows)
            if start == starts[1]:  # (one window: ~10^8 pairs, half a minute of the oracle's DP)
                # the rows of the FULL-SIZE pair tables this window is scored with (read back from the device for the
                # oracle's world) == the oracle's own distance computation on the same strings
                sym, off, _, _ = lw.pool.arrays()
                for key, (pid, odom, ldom) in lw.pair_id.items():
                    u = np.unique(obs[lw.obs_index[key[0]], rows])
                    if dist == "dl":  # (the oracle's full-matrix Lowrance-Wagner DP: a dozen observed values per column)
                        u = u[:12]
                    got = eng.hip.get_pair_rows(pid, u, len(ldom))
                    want = oracle.pair_table(sym, off, odom.id_array()[u], ldom.id_array(), eng.dist_mode)
                    assert np.array_equal(np.asarray(got, dtype=np.uint16), want), (start, key)
                    n_pairs_checked += got.size
            cur = np.ascontiguousarray(tr.cur[:, rows])
            och = np.empty((2, W), dtype=np.int32)
            ocp = np.empty(W, dtype=np.int32)
            oml = np.empty(W)
            oracle.lib().pco_sweep_batched(w.h, C.byref(c), C.c_uint64(seed), C.c_uint32(1), 2, C.c_int64(start),
                                           oracle._p(cur, C.c_int32), oracle._p(och, C.c_int32),
                                           oracle._p(ocp, C.c_int32), oracle._p(oml, C.c_double))
            assert np.array_equal(choice[:, rows], och), start
            assert np.array_equal(chosen[rows], ocp), start
            assert np.array_equal(logml[rows], oml), start
            # the new-row records of the window's rows (sub-choices of every node of the block)
            for b in range(2):
                k = oracle.lib().pco_new_rows_count(b)
                nn = len(lw.blocks[b]["nodes"])
                orows, ovals = np.empty(k, dtype=np.int32), np.empty((k, nn), dtype=np.int32)
                if k:
                    oracle.lib().pco_new_rows_get(b, nn, oracle._p(orows, C.c_int32), oracle._p(ovals, C.c_int32))
                g_rows, g_vals = new_rows.get(b, (np.zeros(0, np.int32), np.zeros((0, nn), np.int32)))
                sel = (g_rows >= start) & (g_rows < start + W)
                assert np.array_equal(g_rows[sel] - start, orows), (start, b)
                assert np.array_equal(g_vals[sel], ovals), (start, b)
            kinds += [int((flags[rows] & 1).sum()), int(((flags[rows] & 2) != 0).sum()),
                      int((choice[:, rows] != tr.cur[:, rows]).any(axis=0).sum()), int((choice[:, rows] < 0).any(axis=0).sum())]
        with capsys.disabled():
            print(f"[own-init 1M] {len(starts)} oracle windows of {W} rows: {kinds[0]} overflowed, {kinds[1]} guess-and-refine, "
                  f"{kinds[2]} moved, {kinds[3]} new-referent rows — all bit-identical; {n_pairs_checked} pairs of the full-size "
                  f"tables recomputed by the oracle")
        assert kinds[1] > 0 and kinds[2] > 0 and kinds[3] > 0 and (kinds[0] > 0 or len(over) == 0)
        # (b) wave kernels == generic kernels on this state, every row
        eng.hip.force_generic(True)
        b_choice, b_chosen, b_logml, b_new = eng.sweep(tr, cfg, seed, 1)
        eng.hip.force_generic(False)
        assert np.array_equal(choice, b_choice) and np.array_equal(chosen, b_chosen) and np.array_equal(logml, b_logml)
        assert set(new_rows) == set(b_new)
        for k in new_rows:
            assert np.array_equal(new_rows[k][0], b_new[k][0]) and np.array_equal(new_rows[k][1], b_new[k][1])
        # (c) the path bench.py TIMES: sweep + device-resident commit (Engine.sweep_commit_device), three sweeps in a row on
        # the device-resident state — after each, the pulled state == the host commit (inference._sweep_window with the
        # device commit off, a second engine that uploads the host trace afresh) of the same sweep on the same state:
        # tables with row ids, free lists, counts, live flags, columns, the 10^6 rows' referents, row origins
        import copy
        from pclean_amd import inference as inf
        from pclean_amd.parallel import Comm
        from test_gpu_commit import _same_state
        ref = Engine(lw, obs, dist_mode=_dist(dist))
        try:
            assert eng.enable_device_commit(tr), getattr(eng, "_dc_why", "")
            host = copy.deepcopy(tr)
            for t in host.tables.values():
                t.cols_dirty = True
            refused = []
            for sweep in range(2, 5):
                changed = inf._sweep_window(eng, tr, cfg, seed, sweep, 0, n_rows, Comm())
                if eng._dc["fallbacks"] > len(refused):
                    refused.append((sweep, eng._dc.get("last_fallback")))
                was = inf.DEVICE_COMMIT
                inf.DEVICE_COMMIT = False
                try:
                    hchanged = inf._sweep_window(ref, host, cfg, seed, sweep, 0, n_rows, Comm())
                finally:
                    inf.DEVICE_COMMIT = was
                assert changed == hchanged, (sweep, changed, hchanged)
                _same_state(tr, host, f"1M rows, sweep {sweep}: device-resident commit vs host commit")
            with capsys.disabled():
                print(f"[own-init 1M] 3 sweeps committed on the device == host commits ({changed} referents changed in the last one); "
                      f"device commits {eng._dc['commits']}, refused {refused} (bit 4 = PCC_FB_CAPACITY: the first commit after a "
                      f"table was uploaded may find its spare rows short; the host commits that sweep and the next upload gives room)")
            # a refusal is only ever the capacity of a freshly uploaded table (never a record / dummy refusal on this workload)
            assert len(refused) <= 1 and all(r[1] == 4 for r in refused), refused
            assert eng._dc["commits"] - eng._dc["fallbacks"] >= 2, eng._dc
        finally:
            ref.close()
    finally:
        eng.close()

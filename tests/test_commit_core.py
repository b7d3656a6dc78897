"""The algorithm of the device-resident commit (pclean_amd/csrc/commit_core.h, host build: tests/commit_host) against the
product's host commit (parallel.exchange_and_commit -> trace.materialise_bulk / delete_rows_bulk, the batched restatement
of row_inference.jl:169-185 + dependency_tracking.jl:6-258) on real sweeps of the three programs (CPU oracle engine):
same row ids, free lists, reference counts, live flags, columns, current referents and row origins after every commit
of a sequence of sweeps.  The HIP build of the same header runs in tests/test_gpu_commit.py."""
import copy

import numpy as np
import pytest

import commit_emul
import helpers
from oracle_engine import OracleEngine
from pclean_amd.engine import InferenceConfig
from pclean_amd.inference import initialize_trace
from pclean_amd.parallel import Comm, exchange_and_commit
from pclean_amd.trace import Trace


def _one_sweep(eng, tr, cfg, seed, sweep):
    choice, chosen, logml, new_rows = eng.sweep(tr, cfg, seed, sweep)
    stats = eng.sweep_stats(tr)
    moved = eng.sweep_moved()
    return choice, chosen, new_rows, stats, moved


def _drive(oracle, lw, obs, tr, cfg, n_sweeps, seed, slack=4096, kcap=4096, expect_supported=True):
    """n_sweeps sweeps; after each, commit on the host and with the emulated device commit; compare the states."""
    eng = OracleEngine(oracle, lw, obs)
    dev = commit_emul.EmulatedDevice(lw, tr, slack=slack)
    assert dev.supported == expect_supported, dev.why
    if not dev.supported:
        return None
    totals = dict(records=0, distinct=0, created=0, deleted=0, changed=0, nested=0)
    try:
        for sweep in range(n_sweeps):
            choice, chosen, new_rows, stats, moved = _one_sweep(eng, tr, cfg, seed, sweep)
            fb, n_changed, nrec, ndist = dev.commit(choice, chosen, new_rows, stats, sweep, kcap=kcap)
            assert fb == 0, fb
            n_before = {c: (t.n, int(t.live[:t.n].sum())) for c, t in tr.tables.items()}
            eng_locals = dict(tr.pending_locals)
            tr.pending_locals = {}
            changed = exchange_and_commit(tr, lw, Comm(), 0, choice, stats, new_rows, global_cur=True, moved_local=moved,
                                          n_local=choice.shape[1], sweep_idx=sweep)
            tr.pending_locals = eng_locals
            tr.commit_locals()
            assert changed == n_changed, (sweep, changed, n_changed)
            dev.assert_equals_trace(tr, f"sweep {sweep}")
            tr.check_consistency()
            totals["records"] += int(nrec.sum())
            totals["distinct"] += int(ndist.sum())
            totals["changed"] += n_changed
            for c, tb in dev.tab.items():
                totals["created"] += int(tb["state"][3])
                totals["deleted"] += int(tb["state"][4])
            for bi, (rows, vals) in new_rows.items():
                fk = [i for i, info in enumerate(lw.blocks[bi]["node_info"]) if info["kind"] == "fk" and i > 0]
                if fk:
                    totals["nested"] += int((np.asarray(vals)[:, fk] == -1).any(axis=1).sum())
    finally:
        dev.close()
    return totals


def test_commit_core_hospital(oracle):
    """hospital after its own batched initialisation (duplicate entities everywhere: rows merge, new rows appear, emptied
    rows are collected and their ids reused): 5 PG sweeps"""
    S = helpers.hospital_setup(n_rows=400)
    lw, obs = S["lw"], S["obs"]
    tr = Trace(lw, obs.shape[1], 1)
    eng = OracleEngine(oracle, lw, obs)
    cfg = InferenceConfig(1, 6)
    initialize_trace(eng, tr, cfg, 11, max_batch=64)
    tot = _drive(oracle, lw, obs, tr, cfg, 5, 123)
    print("hospital:", tot)
    assert tot["created"] > 0 and tot["deleted"] > 0 and tot["nested"] > 0 and tot["records"] > tot["created"]  # (reuse happened)


def _stats_and_moved(lw, tr, choice):
    """delta reference counts / moved rows of a sweep from its choices (what finalize_block_kernel produces)"""
    stats, moved = {}, {}
    for b, blk in enumerate(lw.blocks):
        if blk.get("score"):
            continue
        t = tr.tables[blk["root_class"]]
        ch, cur = choice[b], tr.cur[b]
        mv = ch != cur
        d = -np.bincount(cur[mv & (cur >= 0)], minlength=t.n).astype(np.int64)
        stats[b] = d + np.bincount(ch[mv & (ch >= 0)], minlength=t.n)
        rows = np.flatnonzero(mv).astype(np.int32)
        moved[b] = (rows, ch[rows].astype(np.int32))
    return stats, moved


def test_commit_core_duplicate_proposals(oracle):
    """identical new-row proposals of one sweep become ONE latent row (the first occurrence's), every proposer refers to
    it: the records of a real sweep are copied onto other rows (simple proposals AND proposals with a nested NEW
    referent), several sweeps in a row so that the free lists matter"""
    S = helpers.hospital_setup(n_rows=400)
    lw, obs = S["lw"], S["obs"]
    tr = Trace(lw, obs.shape[1], 1)
    eng = OracleEngine(oracle, lw, obs)
    cfg = InferenceConfig(1, 6)
    initialize_trace(eng, tr, cfg, 11, max_batch=64)
    dev = commit_emul.EmulatedDevice(lw, tr, slack=4096)
    rng = np.random.default_rng(5)
    dup_total = nested_dup = 0
    try:
        for sweep in range(5):
            choice, chosen, logml, new_rows = eng.sweep(tr, cfg, 77, sweep)
            choice = choice.copy()
            for bi in list(new_rows):
                rows, vals = (np.asarray(x) for x in new_rows[bi])
                fk = [i for i, info in enumerate(lw.blocks[bi]["node_info"]) if info["kind"] == "fk" and i > 0]
                others = np.setdiff1d(np.arange(choice.shape[1]), rows)
                extra_rows, extra_vals = [], []
                for j in range(len(rows)):
                    for r in rng.choice(others, size=int(rng.integers(0, 4)), replace=False):
                        if r in extra_rows:
                            continue
                        v = vals[j].copy()
                        v[0] = -1 - int(chosen[r])
                        extra_rows.append(int(r))
                        extra_vals.append(v)
                        choice[bi, r] = -1
                        dup_total += 1
                        nested_dup += int(fk != [] and (vals[j][fk] == -1).any())
                if extra_rows:
                    allr = np.concatenate([rows, np.array(extra_rows, dtype=rows.dtype)])
                    allv = np.concatenate([vals, np.array(extra_vals, dtype=vals.dtype)])
                    order = np.argsort(allr, kind="stable")
                    new_rows[bi] = (allr[order].astype(np.int32), allv[order])
            stats, moved = _stats_and_moved(lw, tr, choice)
            fb, n_changed, nrec, ndist = dev.commit(choice, chosen, new_rows, stats, sweep)
            assert fb == 0
            changed = exchange_and_commit(tr, lw, Comm(), 0, choice, stats, new_rows, global_cur=True, moved_local=moved,
                                          n_local=choice.shape[1], sweep_idx=sweep)
            assert changed == n_changed
            assert int(nrec.sum()) > int(ndist.sum()) or not dup_total
            dev.assert_equals_trace(tr, f"sweep {sweep}")
            tr.check_consistency()
    finally:
        dev.close()
    print(f"duplicates injected: {dup_total} ({nested_dup} of proposals with a nested NEW referent)")
    assert dup_total > 20 and nested_dup > 0


def test_commit_core_hospital_clean_state_mh(oracle):
    S = helpers.hospital_setup(n_rows=300)
    tot = _drive(oracle, S["lw"], S["obs"], S["trace"], InferenceConfig(1, 2, use_mh_instead_of_pg=True), 4, 5)
    print("hospital (clean state, MH):", tot)
    assert tot["changed"] > 0


def test_commit_core_flights(oracle):
    """flights: two reference-slot blocks + a scoring block, keyed TimePrior choices.  Its option lists CAN choose the
    ProposalDummyValue: commits whose created rows hold one are refused (fallback bit), the others must match."""
    S = helpers.flights_setup()
    lw, obs, tr = S["lw"], S["obs"], S["trace"]
    eng = OracleEngine(oracle, lw, obs)
    cfg = InferenceConfig(1, 4)
    dev = commit_emul.EmulatedDevice(lw, tr, slack=1024)
    assert dev.supported, dev.why
    done = refused = 0
    try:
        for sweep in range(4):
            choice, chosen, new_rows, stats, moved = _one_sweep(eng, tr, cfg, 3, sweep)
            before = copy.deepcopy(dev.tab), dev.cur.copy()
            fb, n_changed, nrec, ndist = dev.commit(choice, chosen, new_rows, stats, sweep)
            changed = exchange_and_commit(tr, lw, Comm(), 0, choice, stats, new_rows, global_cur=True, moved_local=moved,
                                          n_local=choice.shape[1], sweep_idx=sweep)
            if fb:
                assert fb == 2, fb  # PCC_FB_DUMMY
                refused += 1
                for c, tb in dev.tab.items():  # nothing was modified
                    for k in ("cols", "counts", "live", "free", "origin"):
                        assert np.array_equal(tb[k], before[0][c][k]), (c, k)
                    assert np.array_equal(tb["state"][:2], before[0][c]["state"][:2])
                assert np.array_equal(dev.cur, before[1])
                # the host went on (it drew the dummies' values): re-seed the emulated device from the host state
                from pclean_amd.inference import _after_commit
                _after_commit(eng, tr, 3)
                dev.close()
                dev = commit_emul.EmulatedDevice(lw, tr, slack=1024)
            else:
                assert changed == n_changed
                dev.assert_equals_trace(tr, f"sweep {sweep}")
                done += 1
    finally:
        dev.close()
    print(f"flights: {done} commits on the device, {refused} refused")
    assert done + refused == 4


def test_commit_core_rents(oracle):
    R = helpers.rents_setup(n_rows=600)
    tot = _drive(oracle, R["lw"], R["obs"], R["trace"], InferenceConfig(1, 4), 3, 9)
    print("rents:", tot)


def test_commit_core_clinic_and_capacity(oracle):
    """the clinic program (two reference slots in one model block, nested classes); then the capacity and record-count
    refusals: nothing is modified and the bits say why"""
    import clinic_program as cp
    P = cp.clinic_program()
    lw, obs = P["lw"], P["obs"]
    tr = Trace(lw, obs.shape[1], 2)
    eng = OracleEngine(oracle, lw, obs)
    cfg = InferenceConfig(1, 5)
    initialize_trace(eng, tr, cfg, 4, max_batch=16)
    tot = _drive(oracle, lw, obs, copy.deepcopy(tr), cfg, 4, 77)
    print("clinic:", tot)
    # refusals
    S = helpers.hospital_setup(n_rows=400)
    lw, obs = S["lw"], S["obs"]
    tr = Trace(lw, obs.shape[1], 1)
    eng = OracleEngine(oracle, lw, obs)
    cfg = InferenceConfig(1, 6)
    initialize_trace(eng, tr, cfg, 11, max_batch=64)
    for t in tr.tables.values():
        t.free = []  # (every created row needs the high-water mark)
    choice, chosen, new_rows, stats, moved = _one_sweep(eng, tr, cfg, 123, 0)
    assert sum(len(r) for r, _ in new_rows.values()) > 2
    for slack, kcap, want in ((0, 4096, 4), (4096, 1, 1)):
        dev = commit_emul.EmulatedDevice(lw, tr, slack=slack)
        try:
            before = copy.deepcopy(dev.tab), dev.cur.copy()
            fb, _, _, _ = dev.commit(choice, chosen, new_rows, stats, 0, kcap=kcap)
            assert fb & want, (slack, kcap, fb)
            for c, tb in dev.tab.items():
                for k in ("cols", "counts", "live", "free"):
                    assert np.array_equal(tb[k], before[0][c][k]), (c, k)
            assert np.array_equal(dev.cur, before[1])
        finally:
            dev.close()


def test_commit_core_refuses_chosen_dummies(oracle):
    """a created row that would hold a ProposalDummyValue needs the host (random(dist) draws its value,
    block_proposal.jl:58-60): the `people` program's short observed strings make the dummy win -> refused, nothing modified"""
    import dummy_program as dp
    m, q, dirty, lw, obs = dp.people_program()
    eng = OracleEngine(oracle, lw, obs)
    tr = Trace(lw, obs.shape[1], 0)
    cfg = InferenceConfig(1, 1)
    choice, chosen, logml, new_rows = eng.sweep(tr, cfg, 5, 0)
    stats, moved = _stats_and_moved(lw, tr, choice)
    dev = commit_emul.EmulatedDevice(lw, tr, slack=256)
    try:
        assert dev.supported, dev.why
        before = copy.deepcopy(dev.tab), dev.cur.copy()
        fb, _, nrec, _ = dev.commit(choice, chosen, new_rows, stats, 0)
        assert fb == 2 and int(nrec.sum()) == obs.shape[1], (fb, nrec)
        for c, tb in dev.tab.items():
            for k in ("cols", "counts", "live", "free", "state"):
                assert np.array_equal(tb[k], before[0][c][k]), (c, k)
        assert np.array_equal(dev.cur, before[1])
    finally:
        dev.close()


def _sharded_sweep(eng, tr, cfg, seed, sweep, bounds):
    """the same sweep, its rows block-partitioned over `bounds` = [(lo, hi)] (frozen tables: a shard's outputs do not depend
    on who else sweeps): per shard (lo, hi, choice, chosen, new_rows); plus the whole window's outputs for the host commit"""
    shards, nb = [], tr.cur.shape[0]
    for lo, hi in bounds:
        if hi <= lo:
            shards.append((lo, hi, np.zeros((nb, 0), np.int32), np.zeros(0, np.int32), {}))
            continue
        choice, chosen, logml, new_rows = eng.sweep(tr, cfg, seed, sweep, lo, hi)
        shards.append((lo, hi, choice.copy(), chosen.copy(), {b: (np.asarray(r).copy(), np.asarray(v).copy()) for b, (r, v) in new_rows.items()}))
    choice = np.concatenate([s[2] for s in shards], axis=1)
    chosen = np.concatenate([s[3] for s in shards])
    new_rows = {}
    for lo, hi, _, _, nr in shards:
        for b, (r, v) in nr.items():
            pr, pv = new_rows.get(b, (np.zeros(0, np.int32), np.zeros((0, v.shape[1]), np.int32)))
            new_rows[b] = (np.concatenate([pr, (r + lo).astype(np.int32)]), np.concatenate([pv, v]))
    return shards, choice, chosen, new_rows


@pytest.mark.parametrize("program", ["hospital", "rents"])
def test_commit_core_gathered_form_equals_single_rank(oracle, program):
    """Several ranks (pclean_commit_device_dist): every rank's moved rows and new-row records packed into its segment
    (pcc_pack), the segments side by side as the all-gather leaves them, concatenated (pcc_merge), then the SAME commit over
    the concatenation — equal to the host commit of the whole window (= the one-rank commit) after every sweep, with shards
    of unequal size and one rank that owns no row; a segment too small for a rank's lists refuses the commit on every rank."""
    if program == "hospital":
        S = helpers.hospital_setup(n_rows=400)
        lw, obs = S["lw"], S["obs"]
        tr = Trace(lw, obs.shape[1], 1)
        cfg = InferenceConfig(1, 6)
        initialize_trace(OracleEngine(oracle, lw, obs), tr, cfg, 11, max_batch=64)
    else:
        R = helpers.rents_setup(n_rows=600)
        lw, obs, tr = R["lw"], R["obs"], R["trace"]
        cfg = InferenceConfig(1, 4)
    eng = OracleEngine(oracle, lw, obs)
    n = obs.shape[1]
    dev = commit_emul.EmulatedDevice(lw, tr, slack=4096)
    assert dev.supported, dev.why
    created = 0
    try:
        for sweep, cuts in enumerate(([0, n // 3, n // 3, n], [0, n // 2, n], [0, 1, n - 1, n], [0, n])):
            bounds = list(zip(cuts[:-1], cuts[1:]))
            shards, choice, chosen, new_rows = _sharded_sweep(eng, tr, cfg, 31, sweep, bounds)
            stats, moved = _stats_and_moved(lw, tr, choice)
            if sweep == 1 and sum(len(r) for r, _ in new_rows.values()) + sum(len(m[0]) for m in moved.values()) > 8:
                before = copy.deepcopy(dev.tab), dev.cur.copy()
                fb, _, _, _ = dev.commit_gathered(shards, stats, sweep, cap_m=1, cap_k=1)  # segments that hold one entry
                assert fb & 1, fb  # PCC_FB_RECORDS
                for c, tb in dev.tab.items():
                    for k in ("cols", "counts", "live", "free"):
                        assert np.array_equal(tb[k], before[0][c][k]), (c, k)
                assert np.array_equal(dev.cur, before[1])
            fb, n_changed, nrec, ndist = dev.commit_gathered(shards, stats, sweep)
            assert fb == 0, fb
            eng_locals = dict(tr.pending_locals)
            tr.pending_locals = {}
            changed = exchange_and_commit(tr, lw, Comm(), 0, choice, stats, new_rows, global_cur=True, moved_local=moved,
                                          n_local=n, sweep_idx=sweep)
            tr.pending_locals = eng_locals
            if program != "rents":  # (the shards' own choices are not reassembled here: the trace keeps its old ones)
                tr.commit_locals()
            else:
                tr.pending_locals = {}
            assert changed == n_changed, (sweep, changed, n_changed)
            dev.assert_equals_trace(tr, f"sweep {sweep} over {len(bounds)} shards")
            created += sum(int(tb["state"][3]) for tb in dev.tab.values())
    finally:
        dev.close()
    assert program != "hospital" or created > 0

"""Device-resident commit (pclean_amd/csrc/commit.hip: pclean_commit_device) against the product's host commit
(parallel.exchange_and_commit) on the three programs: after every sweep of a sequence the pulled device state — latent
tables with row ids, free lists, counts, live flags, the observed rows' referents, row origins, Dirichlet counts, own
enumerated choices — equals the host trace that committed the SAME sweep outputs on the host; and the next sweep on the
device-resident state (tables uploaded once with spare capacity, CRP pieces refreshed on the device) equals the sweep
of a second engine that uploads the host trace afresh (no capacity padding), output for output."""
import copy

import numpy as np
import pytest

import helpers
from pclean_amd import _lib
from pclean_amd import inference as inf
from pclean_amd.engine import Engine, InferenceConfig
from pclean_amd.parallel import Comm, exchange_and_commit
from pclean_amd.trace import Trace

pytestmark = pytest.mark.gpu


def _same_state(a, b, what):
    for cname, ta in a.tables.items():
        tb = b.tables[cname]
        assert ta.n == tb.n, (what, cname, "high-water mark", ta.n, tb.n)
        assert list(ta.free) == list(tb.free), (what, cname, "free list")
        n = ta.n
        assert np.array_equal(ta.live[:n], tb.live[:n]), (what, cname, "live flags")
        assert np.array_equal(ta.counts[:n], tb.counts[:n]), (what, cname, "reference counts")
        assert np.array_equal(ta.cols[:, :n], tb.cols[:, :n]), (what, cname, "columns")
    assert np.array_equal(a.cur, b.cur), (what, "current referents")
    assert a.row_origin == b.row_origin, (what, "row origins")
    for k, pa in a.params.items():
        assert np.array_equal(pa.counts, b.params[k].counts), (what, k, "Dirichlet counts")
    for bi in a.locals:
        assert np.array_equal(a.locals[bi], b.locals[bi]), (what, "own enumerated choices", bi)


def _synthetic(n_rows, n_hosp):
    from pclean_amd import experiments as ex
    from pclean_amd.model import LoweredModel
    from pclean_amd.synth import synth_hospital
    dirty, clean, latent = synth_hospital(n_rows, n_hosp, 7)
    (dirty, clean), _ = ex.shuffle_rows([dirty, clean], 7)
    m = ex.hospital_model(ex.possibilities_of(dirty))
    lw = LoweredModel(m, ex.hospital_query(m), dirty)
    return lw, lw.encode_observations(dirty)


def _programs():
    S = helpers.hospital_setup(n_rows=500)
    yield "hospital", S["lw"], S["obs"], None, InferenceConfig(1, 8), 5
    # a table large enough for the compact-table root scan (>= 1024 candidates): the device commit refreshes the byte
    # rows of the candidates it wrote instead of rebuilding the tables (root_wave.hip: compact_update_kernel)
    lw, obs = _synthetic(40000, 2500)
    yield "synthetic", lw, obs, None, InferenceConfig(1, 6), 4
    S = helpers.flights_setup()
    yield "flights", S["lw"], S["obs"], S["trace"], InferenceConfig(1, 4), 4
    R = helpers.rents_setup(n_rows=2000)
    yield "rents", R["lw"], R["obs"], R["trace"], InferenceConfig(1, 4), 4


@pytest.mark.parametrize("program", ["hospital", "synthetic", "flights", "rents"])
def test_device_commit_equals_host_commit(program):
    name, lw, obs, tr, cfg, n_sweeps = next(p for p in _programs() if p[0] == program)
    eng = Engine(lw, obs, dist_mode=_lib.DIST_DL)
    ref_eng = Engine(lw, obs, dist_mode=_lib.DIST_DL)
    try:
        if tr is None:  # the build's own batched initialisation: duplicate entities everywhere, lots to merge and collect
            tr = Trace(lw, obs.shape[1], 1)
            inf.initialize_trace(eng, tr, cfg, 11, max_batch=64 if name == "hospital" else 4096)
        assert eng.enable_device_commit(tr), getattr(eng, "_dc_why", "")
        n = obs.shape[1]
        done = refused = 0
        for sweep in range(n_sweeps):
            host = copy.deepcopy(tr)  # (synchronises tr with the device first)
            # reference sweep: a plain engine, the host trace uploaded afresh without capacity padding (cols_dirty: the
            # flag is the trace's note to ITS engine; the copy goes to another one)
            for t in host.tables.values():
                t.cols_dirty = True
            ref_eng.upload_trace(host)
            _, _, _, ref_new = ref_eng.sweep(host, cfg, 42, sweep, light=True)
            ref_moved = ref_eng.sweep_moved()
            ref_stats = ref_eng.sweep_stats(host)
            ref_locals = dict(host.pending_locals)
            host.pending_locals = {}
            changed = eng.sweep_commit_device(tr, cfg, 42, sweep)
            if changed is None:
                refused += 1
            else:
                done += 1
                eng.hip.sweep_fetch()  # the committed sweep's outputs are still on the device: bring the lists over
            moved = eng.sweep_moved()
            new_rows = eng.fetched_new_rows(tr if changed is None else host, 0, n)
            if changed is not None:
                dev_locals, host.pending_locals = dict(host.pending_locals), {}
            else:
                dev_locals, tr.pending_locals = dict(tr.pending_locals), {}
            stats = {bi: eng.hip.get_stats(lw.table_id[blk["root_class"]], host.tables[blk["root_class"]].n)
                     for bi, blk in enumerate(lw.blocks) if not blk.get("score")}
            # the sweep on the device-resident state == the sweep on a fresh upload of the same state
            for bi in ref_moved:
                assert np.array_equal(moved[bi][0], ref_moved[bi][0]) and np.array_equal(moved[bi][1], ref_moved[bi][1]), (name, sweep, bi)
                assert np.array_equal(stats[bi], ref_stats[bi]), (name, sweep, bi, "delta counts")
                r0, v0 = new_rows.get(bi, (np.zeros(0, np.int32), None))
                r1, v1 = ref_new.get(bi, (np.zeros(0, np.int32), None))
                assert np.array_equal(r0, r1) and (v0 is None or np.array_equal(v0, v1)), (name, sweep, bi, "new-row records")
            for bi in ref_locals:
                assert np.array_equal(dev_locals[bi], ref_locals[bi]), (name, sweep, "locals")
            if changed is None:
                # refused (a created row would hold a ProposalDummyValue ...): the product path commits on the host and draws
                # the dummies' values; the next iteration starts from that state
                tr.pending_locals = dev_locals
                tr.commit_locals()
                exchange_and_commit(tr, lw, Comm(), 0, None, stats, new_rows, global_cur=True, moved_local=moved, n_local=n,
                                    sweep_idx=sweep)
                if inf._after_commit(eng, tr, 42):
                    ref_eng.reload()  # (the lowered model grew in place)
                tr.check_consistency()
                continue
            # host commit of the same outputs
            host.pending_locals = ref_locals
            host.commit_locals()
            hchanged = exchange_and_commit(host, lw, Comm(), 0, None, stats, new_rows, global_cur=True, moved_local=moved,
                                           n_local=n, sweep_idx=sweep)
            assert changed == hchanged, (name, sweep, changed, hchanged)
            _same_state(tr, host, f"{name} sweep {sweep}")
            tr.check_consistency()
        print(f"[device commit] {name}: {done} commits on the device, {refused} refused; engine stats {eng._dc['commits']} / "
              f"{eng._dc['fallbacks']}")
        assert done > 0
    finally:
        eng.close()
        ref_eng.close()


def test_run_inference_device_commit_equals_host_commit(monkeypatch):
    """run_inference (every class, two iterations) with the observed-class sweeps committed on the device == the same run
    committed on the host, bit for bit"""
    out = []
    for dev in (True, False):
        monkeypatch.setattr(inf, "DEVICE_COMMIT", dev)
        S = helpers.hospital_setup(n_rows=600)
        lw, obs = S["lw"], S["obs"]
        eng = Engine(lw, obs, dist_mode=_lib.DIST_DL)
        try:
            tr = Trace(lw, obs.shape[1], 3)
            cfg = InferenceConfig(2, 6)
            inf.initialize_trace(eng, tr, cfg, 5, max_batch=64)
            inf.run_inference(eng, tr, cfg, 5)
            if dev:
                assert eng._dc is not None and eng._dc["commits"] - eng._dc["fallbacks"] >= 1, eng._dc
            tr.check_consistency()
            out.append(copy.deepcopy(tr))
        finally:
            eng.close()
    _same_state(out[0], out[1], "run_inference device vs host commit")


def test_device_commit_capacity_refusal_and_regrowth():
    """a table about to outgrow its device capacity: the commit is refused (nothing modified), the host commits, the next
    upload gives the table more room and the device commit resumes — states equal to the all-host run throughout"""
    S = helpers.hospital_setup(n_rows=500)
    lw, obs = S["lw"], S["obs"]
    cfg = InferenceConfig(1, 8)
    eng = Engine(lw, obs, dist_mode=_lib.DIST_DL)
    ref = Engine(lw, obs, dist_mode=_lib.DIST_DL)
    try:
        tr = Trace(lw, obs.shape[1], 1)
        inf.initialize_trace(eng, tr, cfg, 11, max_batch=64)
        host = copy.deepcopy(tr)
        assert eng.enable_device_commit(tr)
        eng._capacity = lambda cname, t: t.n  # no spare rows at all
        eng._slack_min = lambda cname, t: 0
        for c in eng._dc["tables"]:
            eng._uploaded_shape.pop(c, None)
            eng._dc["cap"].pop(c, None)
            for t in tr.tables.values():
                t.free = []
        host = copy.deepcopy(tr)
        kinds = []
        for sweep in range(4):
            import os
            was = inf.DEVICE_COMMIT
            changed = inf._sweep_window(eng, tr, cfg, 9, sweep, 0, obs.shape[1], Comm())
            kinds.append(eng._dc.get("last_fallback", 0) if eng._dc["fallbacks"] else 0)
            inf.DEVICE_COMMIT = False
            try:
                hchanged = inf._sweep_window(ref, host, cfg, 9, sweep, 0, obs.shape[1], Comm())
            finally:
                inf.DEVICE_COMMIT = was
            assert changed == hchanged, (sweep, changed, hchanged)
            _same_state(tr, host, f"sweep {sweep}")
            if sweep == 0:  # from now on the default capacity policy
                del eng._capacity, eng._slack_min
        assert eng._dc["fallbacks"] >= 1 and eng._dc["commits"] > eng._dc["fallbacks"], eng._dc
    finally:
        eng.close()
        ref.close()


@pytest.mark.parametrize("program", ["hospital", "synthetic", "rents"])
def test_device_commit_through_the_rank_exchange_equals_plain_device_commit(program, monkeypatch):
    """pclean_commit_device_dist — the commit of several ranks: delta counts all-reduced in place, every rank's moved rows
    and new-row records packed, all-gathered (RCCL) and concatenated on the device, the commit kernel over the gathered
    lists — on a ONE-rank communicator (all a 1-GPU box can run; the gathered form itself is held against the host commit
    with several shards by tests/test_commit_core.py) == the plain one-rank device commit, state for state, sweep after
    sweep; including the capacities of the exchange segments adapting between sweeps."""
    name, lw, obs, tr0, cfg, n_sweeps = next(p for p in _programs() if p[0] == program)
    plain = Engine(lw, obs, dist_mode=_lib.DIST_DL)
    dist = Engine(lw, obs, dist_mode=_lib.DIST_DL)
    try:
        if tr0 is None:
            tr0 = Trace(lw, obs.shape[1], 1)
            inf.initialize_trace(plain, tr0, cfg, 11, max_batch=64 if name == "hospital" else 4096)
        a, b = copy.deepcopy(tr0), copy.deepcopy(tr0)
        for t in list(a.tables.values()) + list(b.tables.values()):
            t.cols_dirty = True
        dist.hip.comm_init(1, 0, dist.hip.comm_unique_id())
        dist._dev_comm = True
        monkeypatch.setenv("PCLEAN_FORCE_DIST", "1")
        n = obs.shape[1]
        for sweep in range(n_sweeps + 2):
            ca = inf._sweep_window(plain, a, cfg, 42, sweep, 0, n, Comm())
            cb = inf._sweep_window(dist, b, cfg, 42, sweep, 0, n, Comm())
            assert ca == cb, (name, sweep, ca, cb)
            _same_state(a, b, f"{name} sweep {sweep}: exchange path vs plain device commit")
        assert getattr(dist, "_dc_dist", False) and dist._dc["commits"] == plain._dc["commits"], (dist._dc, plain._dc)
        assert dist._dc["fallbacks"] == plain._dc["fallbacks"], (dist._dc, plain._dc)
        print(f"[dist commit] {name}: {dist._dc['commits']} commits through the exchange, {dist._dc['fallbacks']} refused")
    finally:
        plain.close()
        dist.close()

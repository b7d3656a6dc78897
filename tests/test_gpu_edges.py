"""Edge cases of the HIP path against the oracle: very long strings (16-bit distance tables), rows with
every cell missing, a one-row table, an empty row window, empty sampler batches, and the error paths of
the C ABI (bad ids, calls out of order) — status codes and messages, never a crash."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers
from pclean_amd import experiments as ex
from pclean_amd._lib import HipContext, InferConfig, PCleanHipError
from pclean_amd.encode import StringPool
from pclean_amd.engine import Engine, InferenceConfig
from pclean_amd.model import LoweredModel
from pclean_amd.trace import Trace

pytestmark = pytest.mark.gpu


def test_pair_table_strings_longer_than_255(oracle):
    """Distances above 255 need the uint16 table layout (hospital MeasureName reaches 184 symbols; this
    goes to 300)."""
    rnd = np.random.default_rng(1)
    alphabet = list("abcdefgh ")
    words = ["".join(rnd.choice(alphabet, size=n)) for n in (300, 290, 257, 256, 255, 3, 1)] + ["", "x" * 300]
    pool = StringPool()
    ids = pool.add_all(words)
    sym, off, _, _ = pool.arrays()
    hip = HipContext(0)
    try:
        hip.load_strings(sym, off)
        for mode in (0, 1):
            hip.build_pair_table(3 + mode, ids, ids, mode)
            got = hip.get_pair_table(3 + mode, len(ids), len(ids))
            want = oracle.pair_table(sym, off, ids, ids, mode)
            assert np.array_equal(got, want), mode
            assert got.max() > 255 and got[words.index(""), words.index("x" * 300)] == 300
    finally:
        hip.close()


def _hospital_with_holes(n_rows, holes):
    dirty, clean = ex.hospital_data()
    dirty = {c: list(v[:n_rows]) for c, v in dirty.items()}
    clean = {c: v[:n_rows] for c, v in clean.items()}
    m = ex.hospital_model(ex.possibilities_of(dirty))
    q = ex.hospital_query(m)
    for i, cols in holes.items():
        for c in (q.cleanmap if cols is None else cols):
            dirty[c][i] = None
    lw = LoweredModel(m, q, dirty)
    return dirty, clean, lw, lw.encode_observations(dirty)


def _oracle_sweep(oracle, world, c, seed, sweep, cur):
    nb, n = cur.shape
    choice = np.empty((nb, n), dtype=np.int32)
    chosen = np.empty(n, dtype=np.int32)
    logml = np.empty(n)
    oracle.lib().pco_sweep_batched(world.h, C.byref(c), C.c_uint64(seed), C.c_uint32(sweep), nb, C.c_int64(0),
                                   oracle._p(np.ascontiguousarray(cur), C.c_int32), oracle._p(choice, C.c_int32),
                                   oracle._p(chosen, C.c_int32), oracle._p(logml, C.c_double))
    return choice, chosen, logml


@pytest.mark.parametrize("n_rows", [1, 80])
def test_rows_with_missing_cells_and_tiny_tables(oracle, n_rows):
    """Row 0: every queried cell missing (nothing but the priors speaks); row 5: one cell left.  From an
    empty trace (every proposal is a new referent) and again after the commit."""
    holes = {0: None} if n_rows == 1 else {0: None, 5: ["ProviderNumber", "HospitalName", "City", "State", "ZipCode",
                                                         "CountyName", "PhoneNumber", "HospitalType", "HospitalOwner",
                                                         "EmergencyService", "Condition", "MeasureCode", "MeasureName"]}
    dirty, clean, lw, obs = _hospital_with_holes(n_rows, holes)
    assert (obs[:, 0] < 0).all()
    eng = Engine(lw, obs, dist_mode=1)
    try:
        tr = Trace(lw, n_rows, 3)
        cfg = InferenceConfig(1, 6)
        c = InferConfig(1, 6, 1, 1, 0, 50, 100)
        for sweep in range(2):
            eng.upload_trace(tr)
            world = helpers.mirror_world(oracle, lw, obs, tr, eng)
            choice, chosen, logml, new_rows = eng.sweep(tr, cfg, 9, sweep)
            och, ocp, oml = _oracle_sweep(oracle, world, c, 9, sweep, tr.cur)
            assert np.array_equal(choice, och) and np.array_equal(chosen, ocp) and np.array_equal(logml, oml)
            assert np.isfinite(logml).all()
            if sweep == 0:
                assert (choice < 0).all()  # empty tables: only the new-row candidate exists
                tr.commit_batch(0, n_rows, choice, new_rows, dedup=True)
            else:
                tr.commit(choice, new_rows)
            tr.check_consistency()
    finally:
        eng.close()


def test_empty_window_and_empty_batches(oracle):
    S = helpers.hospital_setup(n_rows=40)
    eng = Engine(S["lw"], S["obs"], dist_mode=1)
    try:
        eng.upload_trace(S["trace"])
        choice, chosen, logml, new_rows = eng.sweep(S["trace"], InferenceConfig(1, 4), 1, 0, lo=7, hi=7)
        assert choice.shape == (2, 0) and len(chosen) == 0 and len(logml) == 0 and new_rows == {}
        assert all(len(v[0]) == 0 for v in eng.sweep_moved().values())
        assert all(v.sum() == 0 for v in eng.sweep_stats(S["trace"]).values())
        hip = eng.hip
        from pclean_amd import sampling
        assert sampling.random_add_typos(hip, [], None) == [] and sampling.random_time_prior(hip, 0) == []
        assert len(hip.random_categorical(0, np.zeros(3), 1, 0)) == 0
        assert len(hip.random_normal(np.zeros(0), 1.0, 1.0, 1, 0)) == 0
    finally:
        eng.close()


def test_abi_error_paths():
    hip = HipContext(0)
    try:
        cfg = InferConfig(1, 2, 1, 1, 0, 50, 100)
        with pytest.raises(PCleanHipError):  # nothing loaded yet
            hip.sweep(cfg, 1, 0, np.zeros((1, 4), np.int32))
        with pytest.raises(PCleanHipError):
            hip.build_pair_table(0, np.zeros(1, np.int32), np.zeros(1, np.int32), 0)  # no strings loaded
        with pytest.raises(PCleanHipError):
            hip.get_moved(3)  # block never loaded
        with pytest.raises(PCleanHipError):
            hip.random_categorical(4, np.zeros(0), 1, 0)  # no options
        with pytest.raises(PCleanHipError):
            hip.load_score_block(99, [0], [0], [0, 0], [0, 0], [0], [0], 0, [0, 0], [0, 0])  # block id out of range
        msg = hip.lib.pclean_last_error(hip.h).decode()
        assert msg  # the message of the last failure is retrievable
        # the context is still usable after errors
        pool = StringPool()
        ids = pool.add_all(["abc", "abd"])
        sym, off, _, _ = pool.arrays()
        hip.load_strings(sym, off)
        hip.build_pair_table(0, ids, ids, 0)
        assert hip.get_pair_table(0, 2, 2)[0, 1] == 1
    finally:
        hip.close()



def test_use_dd_proposals_false_prior_proposals(oracle):
    """use_dd_proposals = false (block_proposal.jl:168): reference slots from the CRP prior, a new row's choices from
    their prior proposals, weight = likelihood of the sampled values (propose_non_enumerable!, 24-157).  HIP == oracle
    bit for bit on hospital (two blocks, the second with a JuliaNode context; PG and MH) and on the `people` program
    (prior draws of a StringPrior choice are the ProposalDummyValue: random strings weighed against the observation);
    equality constraints, MaybeSwap and scoring blocks: test_gpu_flights.py; a Gaussian term: the next test."""
    import dummy_program as dp
    S = helpers.hospital_setup(n_rows=300)
    cases = [(S["lw"], S["obs"], S["trace"], 2)]
    m, q, dirty, lw2, obs2 = dp.people_program()
    from pclean_amd.trace import Trace
    cases.append((lw2, obs2, Trace(lw2, obs2.shape[1], 0), 1))
    for lw, obs, tr, nb in cases:
        eng = Engine(lw, obs, dist_mode=1)
        try:
            eng.upload_trace(tr)
            world = helpers.mirror_world(oracle, lw, obs, tr, eng)
            for P, mh in ((6, 0), (2, 1)):
                cfg = InferenceConfig(1, P, use_dd_proposals=False, use_mh_instead_of_pg=bool(mh))
                choice, chosen, logml, new_rows = eng.sweep(tr, cfg, 77, 3)
                c = InferConfig(1, P, 0, 1, mh, 50, 100)
                och, ocp, oml = _oracle_sweep(oracle, world, c, 77, 3, tr.cur)
                assert np.array_equal(choice, och) and np.array_equal(chosen, ocp), (nb, P, mh)
                assert np.array_equal(logml, oml), (nb, P, mh, np.abs(logml - oml).max())
                for b in range(nb):
                    k = oracle.lib().pco_new_rows_count(b)
                    nn = len(lw.blocks[b]["nodes"])
                    orows, ovals = np.empty(k, dtype=np.int32), np.empty((k, nn), dtype=np.int32)
                    if k:
                        oracle.lib().pco_new_rows_get(b, nn, oracle._p(orows, C.c_int32), oracle._p(ovals, C.c_int32))
                    g = new_rows.get(b, (np.zeros(0, np.int32), np.zeros((0, nn), np.int32)))
                    assert np.array_equal(g[0], orows) and np.array_equal(g[1], ovals), (nb, P, mh, b)
            # the data-driven sweep gives different draws (same seed): the flag really switches the proposal — and it
            # equals the oracle's data-driven sweep: nothing the prior-proposal sweeps built (option-list caches keyed
            # by table versions) leaks into it (advisor r3)
            dd = eng.sweep(tr, InferenceConfig(1, 6), 77, 3)
            assert not np.array_equal(dd[2], logml)
            och, ocp, oml = _oracle_sweep(oracle, world, InferConfig(1, 6, 1, 1, 0, 50, 100), 77, 3, tr.cur)
            assert np.array_equal(dd[0], och) and np.array_equal(dd[1], ocp) and np.array_equal(dd[2], oml), nb
        finally:
            eng.close()


def test_use_dd_proposals_false_with_a_gaussian_term(oracle):
    """use_dd_proposals = false on rents: the own choices the data-driven proposal enumerates inside the candidate branch
    (room type, unit) are sampled from their priors by every particle, an observed room type is scored, the retained
    particle keeps the row's current ones (pclean_set_cur_locals) and the rent is scored at the particle's referent and
    own choices (gauss_prior_kernel).  HIP == oracle bit for bit — choices, chosen particles, log marginal likelihoods,
    new-row records and the chosen particles' own choices — for PG and MH, from a state with current own choices and from
    one without."""
    R = helpers.rents_setup(n_rows=400)
    lw, obs, tr = R["lw"], R["obs"], R["trace"]
    n = obs.shape[1]
    eng = Engine(lw, obs, dist_mode=1)
    try:
        eng.upload_trace(tr)
        for state in ("no current own choices", "current own choices"):
            if state == "current own choices":
                rng = np.random.default_rng(4)
                tr.locals[0][:, 0] = np.where(obs[3] >= 0, obs[3], rng.integers(0, 5, n))  # (column 3: the observed room type)
                tr.locals[0][:, 1] = rng.integers(0, 2, n)
            world = helpers.mirror_world(oracle, lw, obs, tr, eng)
            world.set_cur_locals(0, tr.locals[0])
            for P, mh in ((6, 0), (2, 1), (1, 0)):
                cfg = InferenceConfig(1, P, use_dd_proposals=False, use_mh_instead_of_pg=bool(mh))
                choice, chosen, logml, new_rows = eng.sweep(tr, cfg, 31, 2)
                got_loc = tr.pending_locals[0].copy()
                c = InferConfig(1, P, 0, 1, mh, 50, 100)
                och, ocp, oml = _oracle_sweep(oracle, world, c, 31, 2, tr.cur)
                assert np.array_equal(choice, och) and np.array_equal(chosen, ocp), (state, P, mh)
                assert np.array_equal(logml, oml), (state, P, mh, np.abs(logml - oml).max())
                assert np.array_equal(got_loc, world.get_locals(0, n)), (state, P, mh)
                k = oracle.lib().pco_new_rows_count(0)
                nn = len(lw.blocks[0]["nodes"])
                orows, ovals = np.empty(k, dtype=np.int32), np.empty((k, nn), dtype=np.int32)
                if k:
                    oracle.lib().pco_new_rows_get(0, nn, oracle._p(orows, C.c_int32), oracle._p(ovals, C.c_int32))
                g = new_rows.get(0, (np.zeros(0, np.int32), np.zeros((0, nn), np.int32)))
                assert np.array_equal(g[0], orows) and np.array_equal(g[1], ovals), (state, P, mh)
                if P == 1 and state == "current own choices":  # the retained particle kept everything
                    assert np.array_equal(choice, tr.cur) and np.array_equal(got_loc, tr.locals[0])
        # the data-driven sweep afterwards equals the oracle's: nothing of the prior-proposal sweeps leaks into it
        dd = eng.sweep(tr, InferenceConfig(1, 6), 31, 2)
        och, ocp, oml = _oracle_sweep(oracle, world, InferConfig(1, 6, 1, 1, 0, 50, 100), 31, 2, tr.cur)
        assert np.array_equal(dd[0], och) and np.array_equal(dd[1], ocp) and np.array_equal(dd[2], oml)
    finally:
        eng.close()


def test_use_dd_proposals_false_gaussian_block_followed_by_another_slot(oracle):
    """Prior proposals under particle Gibbs when ANOTHER reference-slot block follows the one with the Gaussian term
    (tests/rents_two_slots.py): the resampling step between the two blocks (row_inference.jl:139-151) permutes the
    particles, and the own choices each particle sampled for the Gaussian term follow it (apply_ancestors_kernel) — the
    chosen particle's own choices, the choices of both blocks and the log marginal likelihoods equal the oracle's bit for
    bit (P = 6 and 20: most rows are resampled; MH for the two-particle path)."""
    import rents_two_slots as r2
    R = r2.setup(n_rows=400)
    lw, obs, tr = R["lw"], R["obs"], R["trace"]
    n = obs.shape[1]
    eng = Engine(lw, obs, dist_mode=1)
    try:
        eng.upload_trace(tr)
        rng = np.random.default_rng(4)
        tr.locals[0][:, 0] = np.where(obs[3] >= 0, obs[3], rng.integers(0, 5, n))
        tr.locals[0][:, 1] = rng.integers(0, 2, n)
        world = helpers.mirror_world(oracle, lw, obs, tr, eng)
        world.set_cur_locals(0, tr.locals[0])
        spread = 0
        for P, mh in ((6, 0), (20, 0), (2, 1)):
            cfg = InferenceConfig(1, P, use_dd_proposals=False, use_mh_instead_of_pg=bool(mh))
            choice, chosen, logml, new_rows = eng.sweep(tr, cfg, 31, 2)
            got_loc = tr.pending_locals[0].copy()
            c = InferConfig(1, P, 0, 1, mh, 50, 100)
            och, ocp, oml = _oracle_sweep(oracle, world, c, 31, 2, tr.cur)
            assert np.array_equal(choice, och) and np.array_equal(chosen, ocp), (P, mh)
            assert np.array_equal(logml, oml), (P, mh, np.abs(logml - oml).max())
            assert np.array_equal(got_loc, world.get_locals(0, n)), (P, mh)
            spread += int((chosen > 0).sum())
        assert spread > n // 4  # (the chosen particles are not the retained ones: the permutation was exercised)
    finally:
        eng.close()


def test_new_branch_gate_light_outputs_and_profile(oracle):
    """The gate of the new-row branch (gate_new_kernel) only skips work whose fixed-point weight is exactly 0:
    a sweep with the gate forced on for every list equals the sweep without it and the oracle, bit for bit.
    A `light` sweep (no per-row outputs) reports the same moved rows / new-row records, and the per-phase
    profile is populated."""
    S = helpers.hospital_setup(n_rows=600)
    lw, tr, obs = S["lw"], S["trace"], S["obs"]
    cfg = InferenceConfig(1, 20)
    c = InferConfig(1, 20, 1, 1, 0, 50, 100)
    res = {}
    for mode in ("gate", "nogate"):
        os.environ.pop("PCLEAN_NO_GATE", None)
        if mode == "nogate":
            os.environ["PCLEAN_NO_GATE"] = "1"
        os.environ["PCLEAN_GATE_MIN"] = "1"  # the gate applies to lists of any length
        eng = Engine(lw, obs, dist_mode=1)
        try:
            eng.upload_trace(tr)
            res[mode] = eng.sweep(tr, cfg, 77, 0)
            res[mode + "_moved"] = eng.sweep_moved()
            if mode == "gate":
                world = helpers.mirror_world(oracle, lw, obs, tr, eng)
                och, ocp, oml = _oracle_sweep(oracle, world, c, 77, 0, tr.cur)
                eng.hip.set_profiling(True)
                choice_l, chosen_l, logml_l, new_l = eng.sweep(tr, cfg, 77, 0, reuse_buffers=True, light=True)
                prof = eng.hip.get_profile()
                eng.hip.set_profiling(False)
                assert choice_l is None and chosen_l is None and logml_l is None
                moved_l = eng.sweep_moved()
        finally:
            eng.close()
    os.environ.pop("PCLEAN_NO_GATE", None)
    os.environ.pop("PCLEAN_GATE_MIN", None)
    a, b = res["gate"], res["nogate"]
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.array_equal(a[0], och) and np.array_equal(a[1], ocp) and np.array_equal(a[2], oml)
    for bi in a[3]:
        assert np.array_equal(a[3][bi][0], b[3][bi][0]) and np.array_equal(a[3][bi][1], b[3][bi][1])
        assert np.array_equal(a[3][bi][0], new_l[bi][0]) and np.array_equal(a[3][bi][1], new_l[bi][1])
    for bi, (rows, ch) in res["gate_moved"].items():
        assert np.array_equal(rows, np.flatnonzero(a[0][bi] != tr.cur[bi]))
        assert np.array_equal(ch, a[0][bi][rows])
        assert np.array_equal(rows, moved_l[bi][0]) and np.array_equal(ch, moved_l[bi][1])
    assert prof and sum(v[0] for v in prof.values()) > 0 and "final_choice_and_outputs" in prof


def test_evidence_aggregation_paths_at_the_lds_capacity(oracle):
    """Latent sweeps of a 1 800-row synthetic table: HospitalType's single row is referred to by every observed row
    (1 800 evidence rows: the per-row LDS aggregation pads them to its 2 048-key capacity), Condition's rows by
    hundreds.  The LDS aggregation and the global radix sort + run-length encoding must give the same sweep, and
    both the oracle's."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from pclean_amd.inference import build_evidence, initialize_trace, latent_current_choices
    dirty, clean, lw, obs = bench.build_workload(1800, 20, 5)
    eng = Engine(lw, obs, dist_mode=0)
    try:
        cfg = InferenceConfig(1, 4)
        tr = Trace(lw, obs.shape[1], 5)
        initialize_trace(eng, tr, cfg, 5)
        seen_big = False
        for cname in lw.model.class_order:
            if cname not in lw.latent_plans:
                continue
            pl = lw.latent_plans[cname]
            live, ev_off, ev_rows, ev_ctx = build_evidence(lw, tr, cname)
            longest = int(np.max(np.diff(ev_off)))
            excl = latent_current_choices(lw, tr, cname, live, cfg)
            eng.upload_trace(tr)
            eng.hip.set_active_rows(0, -1)
            args = (cfg.as_c(), 11, 0, pl["block_id"], pl["roots"], live, ev_off, ev_rows, ev_ctx, excl, len(pl["nodes"]))
            a = eng.hip.sweep_latent(*args)
            eng.hip.global_evidence_sort(True)
            b = eng.hip.sweep_latent(*args)
            eng.hip.global_evidence_sort(False)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), cname
            if 1024 < longest <= 2048:
                seen_big = True
                world = helpers.mirror_world(oracle, lw, obs, tr, eng)
                want = world.sweep_latent(InferConfig(1, 4, 1, 1, 0, 50, 100), 11, 0, pl["block_id"], pl["roots"], live, ev_off,
                                          ev_rows, ev_ctx, excl, len(pl["nodes"]))
                assert np.array_equal(a[0], want[0]) and np.array_equal(a[1], want[1]), cname
        assert seen_big  # some class had an evidence set between 1 025 and 2 048 rows
    finally:
        eng.close()


def test_device_argsort_of_ids_is_numpys_stable_argsort():
    """pclean_argsort_ids (the evidence CSR's sort): same permutation as np.argsort(kind="stable") for every size, with ids of
    -1 (no referent), a single id, more than 2^16 ids"""
    hip = HipContext(0)
    try:
        rng = np.random.default_rng(5)
        for n, id_max in ((1, 0), (2, 5), (1000, 7), (70000, 70000), (300001, 10660), (1 << 20, 3)):
            ids = rng.integers(-1, id_max + 1, size=n).astype(np.int32)
            got = hip.argsort_ids(ids, id_max)
            assert np.array_equal(got, np.argsort(ids, kind="stable")), (n, id_max)
        assert len(hip.argsort_ids(np.zeros(0, np.int32), 0)) == 0
    finally:
        hip.close()


def test_device_evidence_equals_host():
    """pclean_build_evidence (the evidence sets of a latent class built on the device from the device-resident referents and
    tables) against inference.build_evidence (NumPy on the host): same live rows, offsets, ordered rows and per-row ctx values,
    element for element, for every latent class of the synthetic hospital program — after the initialisation, after a device
    commit of an observed sweep (the referents then exist on the device alone until the host pulls them) and after a latent
    class's own sweep moved referents; pclean_sweep_latent_resident gives pclean_sweep_latent's result; a full run_inference
    iteration ends in the same trace whichever path builds the sets."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from pclean_amd import inference as inf
    from pclean_amd._lib import _ctx_cols
    dirty, clean, lw, obs = bench.build_workload(6000, 60, 7)
    cfg = InferenceConfig(1, 4)

    def check_all(eng, tr, sweep_too):
        for cname in lw.model.class_order:
            if cname not in lw.latent_plans:
                continue
            pl = lw.latent_plans[cname]
            got = eng.build_evidence_device(tr, cname)  # (first: a trace the device is ahead of is pulled in here)
            assert got is not None, cname
            live, ev_off, ev_rows, ev_ctx = inf.build_evidence(lw, tr, cname)
            assert np.array_equal(got[0], live) and np.array_equal(got[1], ev_off) and got[2] is None, cname
            rows, cx = eng.hip.get_evidence(0, len(ev_rows))
            assert np.array_equal(rows, ev_rows), cname
            want_cx = _ctx_cols(ev_ctx)
            assert np.array_equal(cx, want_cx if want_cx is not None else np.zeros_like(cx)), cname
            if not sweep_too:
                continue
            excl = inf.latent_current_choices(lw, tr, cname, live, cfg)
            lo, hi = len(live) // 3, len(live)  # (a sub-range: ev_begin > 0)
            e0, e1 = int(ev_off[lo]), int(ev_off[hi])
            a = eng.sweep_latent(tr, cname, cfg, 11, 0, live[lo:hi], ev_off[lo:hi + 1] - e0, ev_rows[e0:e1],
                                 None if ev_ctx is None else ev_ctx[e0:e1], np.ascontiguousarray(excl[:, lo:hi]))
            b = eng.sweep_latent(tr, cname, cfg, 11, 0, live[lo:hi], ev_off[lo:hi + 1] - e0, None, None,
                                 np.ascontiguousarray(excl[:, lo:hi]), ev_begin=e0)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), cname

    eng = Engine(lw, obs, dist_mode=0)
    try:
        tr = Trace(lw, obs.shape[1], 7)
        inf.initialize_trace(eng, tr, cfg, 7)
        assert eng.enable_device_commit(tr), getattr(eng, "_dc_why", "")
        check_all(eng, tr, True)
        inf.observed_sweep(eng, tr, cfg, 7, 0)  # device commit: the trace is behind the device now
        check_all(eng, tr, False)
        inf.latent_sweep(eng, tr, "Hospital", cfg, 7, 0)
        inf.latent_sweep(eng, tr, "Place", cfg, 7, 0)
        check_all(eng, tr, True)
    finally:
        eng.close()
    # the same iteration with the sets built on the device and on the host: identical traces
    ends = []
    for min_rows in (0, 1 << 30):
        old = inf.DEVICE_EVIDENCE_MIN_ROWS
        inf.DEVICE_EVIDENCE_MIN_ROWS = min_rows
        eng = Engine(lw, obs, dist_mode=0)
        try:
            tr = Trace(lw, obs.shape[1], 7)
            inf.initialize_trace(eng, tr, cfg, 7)
            eng.prepare(tr)
            inf.run_inference(eng, tr, InferenceConfig(2, 4), 7)
            ends.append((tr.cur.copy(), {c: (t.cols[:, :t.n].copy(), t.counts[:t.n].copy(), t.live[:t.n].copy())
                                        for c, t in tr.tables.items()}))
        finally:
            inf.DEVICE_EVIDENCE_MIN_ROWS = old
            eng.close()
    assert np.array_equal(ends[0][0], ends[1][0])
    for c in ends[0][1]:
        for x, y in zip(ends[0][1][c], ends[1][1][c]):
            assert np.array_equal(x, y), c

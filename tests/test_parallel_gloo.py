"""CPU tests (gloo, world_size 2) of the row-sharded sweep: block partition, all-reduce of the
CRP sufficient statistics (delta reference counts), all-gather + ordered merge of new-row records.
The compute engine here is the CPU oracle (test infrastructure) standing in for the HIP engine;
the exchange/commit code under test is the product's pclean_amd/parallel.py.  Result must be
identical to the single-process sweep for any number of ranks (global-row-keyed RNG, integer stats)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_ROWS = 240


def _sweep_shard(orc, helpers, S, lo, hi, cfg, seed, sweep):
    """Oracle batched sweep of rows [lo, hi) against the current (replicated) trace."""
    from pclean_amd._lib import InferConfig
    lw, tr = S["lw"], S["trace"]
    obs = np.ascontiguousarray(S["obs"][:, lo:hi])
    logp = helpers.option_logp_cpu(orc, lw, tr)
    w = helpers.mirror_world(orc, lw, obs, tr, None, 1, logp)
    nb, n = tr.cur.shape
    choice = np.empty((nb, n), dtype=np.int32)
    c = InferConfig(1, cfg, 1, 1, 0, 50, 100)
    orc.lib().pco_sweep_batched(w.h, C.byref(c), C.c_uint64(seed), C.c_uint32(sweep), nb, C.c_int64(lo),
                                orc._p(np.ascontiguousarray(tr.cur), C.c_int32), orc._p(choice, C.c_int32), None, None)
    new_rows, stats = {}, {}
    for b, blk in enumerate(lw.blocks):
        k = orc.lib().pco_new_rows_count(b)
        if k:
            rows = np.empty(k, dtype=np.int32)
            vals = np.empty((k, len(blk["nodes"])), dtype=np.int32)
            orc.lib().pco_new_rows_get(b, len(blk["nodes"]), orc._p(rows, C.c_int32), orc._p(vals, C.c_int32))
            new_rows[b] = (rows, vals)
        t = tr.tables[blk["root_class"]]
        moved = choice[b] != tr.cur[b]
        d = -np.bincount(tr.cur[b][moved], minlength=t.n).astype(np.int64)
        ex = moved & (choice[b] >= 0)
        stats[b] = d + np.bincount(choice[b][ex], minlength=t.n)
    return choice, stats, new_rows


def _run(rank, world, port, out_path, use_moved=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import helpers
    import oracle as orc
    from pclean_amd.parallel import Comm, exchange_and_commit, shard_bounds
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    comm = Comm()
    S = helpers.hospital_setup(n_rows=N_ROWS, seed=3)
    lw, tr = S["lw"], S["trace"]
    lo, hi = shard_bounds(N_ROWS, rank, world)
    tr.cur = np.ascontiguousarray(tr.cur[:, lo:hi])
    changed = []
    for sweep in range(3):
        choice, stats, new_rows = _sweep_shard(orc, helpers, S, lo, hi, 6, 99, sweep)
        moved = None
        if use_moved:  # the pclean_get_moved lists instead of scanning the choice arrays (what bench.py does)
            moved = {}
            for b in range(choice.shape[0]):
                rows = np.flatnonzero(choice[b] != tr.cur[b]).astype(np.int32)
                moved[b] = (rows, choice[b][rows])
        changed.append(exchange_and_commit(tr, lw, comm, lo, choice, stats, new_rows, moved_local=moved))
    cur_all = comm.allgather_varlen_i32(tr.cur[0]), comm.allgather_varlen_i32(tr.cur[1])
    if rank == 0:
        np.savez(out_path, cur0=cur_all[0], cur1=cur_all[1], changed=np.array(changed),
                 **{f"cols_{c}": t.cols[:, :t.n] for c, t in tr.tables.items()},
                 **{f"counts_{c}": t.counts[:t.n] for c, t in tr.tables.items()})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def test_shard_bounds_cover_all_rows():
    from pclean_amd.parallel import shard_bounds
    for n in (0, 1, 7, 240, 1000003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


@pytest.mark.timeout(600)
@pytest.mark.parametrize("use_moved", [False, True])
def test_two_rank_sweep_equals_single_process(tmp_path, oracle, use_moved):
    import torch.multiprocessing as mp
    single = str(tmp_path / "single.npz")
    _run(0, 1, 0, single, False)
    multi = str(tmp_path / "multi.npz")
    port = 29500 + (os.getpid() % 2000) + (97 if use_moved else 0)
    mp.spawn(_run, args=(2, port, multi, use_moved), nprocs=2, join=True)
    a, b = np.load(single), np.load(multi)
    assert set(a.files) == set(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    assert a["changed"].sum() > 0
    # reference counts are consistent with the referents after the exchange
    assert np.array_equal(np.bincount(a["cur0"], minlength=len(a["counts_Hospital"])), a["counts_Hospital"])
    assert np.array_equal(np.bincount(a["cur1"], minlength=len(a["counts_Measure"])), a["counts_Measure"])


# ---------------------------------------------------------------------------
# whole inference (initialize_trace + run_inference over every class) on 2 ranks
def _program(name, cap=None):
    from pclean_amd import experiments as ex
    from pclean_amd.model import LoweredModel
    if name == "flights":
        dirty, clean = ex.flights_data()
        dirty = {c: v[:cap or 600] for c, v in dirty.items()}
        m = ex.flights_model(dirty)
        q = ex.flights_query(m)
    elif name == "rents":
        dirty, clean = ex.rents_data()
        dirty = {c: v[:400] for c, v in dirty.items()}
        m = ex.rents_model(dirty)
        q = ex.rents_query(m)
    else:
        dirty, clean = ex.hospital_data()
        dirty = {c: v[:cap or 200] for c, v in dirty.items()}
        m = ex.hospital_model(ex.possibilities_of(dirty))
        q = ex.hospital_query(m)
    lw = LoweredModel(m, q, dirty)
    return lw, lw.encode_observations(dirty)


def _run_inference(rank, world, port, out_path, name):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import oracle as orc
    from oracle_engine import OracleEngine
    from pclean_amd.engine import InferenceConfig
    from pclean_amd.inference import initialize_trace, run_inference
    from pclean_amd.parallel import Comm
    from pclean_amd.trace import Trace
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    comm = Comm()
    lw, obs = _program(name)
    eng = OracleEngine(orc, lw, obs)  # every rank holds all observations; the work is sharded
    tr = Trace(lw, obs.shape[1], 5)
    cfg = InferenceConfig(2, 3, rejuv_frequency=100)
    # hospital (file order): with the in-batch merge pass; flights: chosen dummy values get their prior draws
    initialize_trace(eng, tr, cfg, 17, max_batch=64, comm=comm, merge_rounds=2 if name == "hospital" else 0)
    tr.check_consistency()
    run_inference(eng, tr, cfg, 17, comm=comm)
    tr.check_consistency()
    if rank == world - 1:  # the LAST rank's replica: must equal the single-process result too
        extra = {}
        for bi, loc in tr.locals.items():
            extra[f"locals_{bi}"] = loc
        if tr.prob_param is not None:
            extra["prob"] = tr.prob_param.value
        if tr.mean_param is not None:
            extra["mean"] = tr.mean_param.value
        np.savez(out_path, cur=tr.cur,
                 **{f"cols_{c}": t.cols[:, :t.n] for c, t in tr.tables.items()},
                 **{f"counts_{c}": t.counts[:t.n] for c, t in tr.tables.items()},
                 **{f"param_{c}_{p}": v.value for (c, p), v in tr.params.items()}, **extra)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["flights", "hospital", "rents"])
def test_two_rank_inference_equals_single_process(tmp_path, oracle, name):
    """BASELINE.json configs[3] at plumbing scale: observed rows AND latent rows of every class sweep are
    block-partitioned over the ranks; the replicated traces must end bit-identical to one process."""
    import torch.multiprocessing as mp
    single = str(tmp_path / "single.npz")
    _run_inference(0, 1, 0, single, name)
    multi = str(tmp_path / "multi.npz")
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_run_inference, args=(2, port, multi, name), nprocs=2, join=True)
    a, b = np.load(single), np.load(multi)
    assert set(a.files) == set(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    assert (a["cur"] >= 0).all()


# ---------------------------------------------------------------------------
# the fused statistics path (exchange_and_commit(..., stats_reduced=True)): what Engine.sweep_stats_reduced feeds it on a
# multi-GPU run — the delta reference counts already summed over the ranks by ONE device-side all-reduce
# (pclean_allreduce_stats_fused) — driven here through a stand-in engine whose "device-side" all-reduce is a gloo one
def _run_fused(rank, world, port, out_path, name):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import oracle as orc
    from oracle_engine import OracleEngine
    from pclean_amd.engine import InferenceConfig
    from pclean_amd.inference import initialize_trace, observed_sweep, run_inference
    from pclean_amd.parallel import Comm
    from pclean_amd.trace import Trace
    if world > 1:
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    comm = Comm()

    class FusedStatsEngine(OracleEngine):
        """sweep_stats_reduced as the HIP engine provides it after init_device_comm: per block root table the delta counts
        summed over ALL ranks (a rank that swept nothing contributes zeros), in one collective for all tables; reload()
        re-binds the communicator (counted)."""
        reloads = 0
        fused_calls = 0

        def sweep_stats_reduced(self, trace):
            local = self.sweep_stats(trace)
            blocks = sorted(local)
            flat = np.concatenate([np.asarray(local[b], dtype=np.int64) for b in blocks]) if blocks else np.zeros(0, np.int64)
            total = comm.allreduce_sum_i64(flat)
            type(self).fused_calls += 1
            out, off = {}, 0
            for b in blocks:
                n = len(local[b])
                out[b] = total[off:off + n]
                off += n
            return out

        def reload(self):
            type(self).reloads += 1
            super().reload()

    lw, obs = _program(name, 60 if name == "hospital" else 120)
    eng = (FusedStatsEngine if world > 1 else OracleEngine)(orc, lw, obs)
    tr = Trace(lw, obs.shape[1], 5)
    cfg = InferenceConfig(1, 3, rejuv_frequency=100)
    initialize_trace(eng, tr, cfg, 17, max_batch=32, comm=comm)
    run_inference(eng, tr, cfg, 17, comm=comm, batch_rows=7)   # windows of 7 rows: shards of 4 + 3
    observed_sweep(eng, tr, cfg, 17, 5, comm, batch_rows=1)    # windows of ONE row: rank 1's shard is always empty
    tr.check_consistency()
    if world > 1:
        assert FusedStatsEngine.fused_calls > obs.shape[1], FusedStatsEngine.fused_calls
    if rank == world - 1:
        np.savez(out_path, cur=tr.cur, reloads=np.array([getattr(type(eng), "reloads", 0)]),
                 **{f"cols_{c}": t.cols[:, :t.n] for c, t in tr.tables.items()},
                 **{f"counts_{c}": t.counts[:t.n] for c, t in tr.tables.items()})
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("name", ["hospital", "flights"])
def test_fused_statistics_path_two_ranks(tmp_path, oracle, name):
    """exchange_and_commit with stats_reduced=True (the multi-GPU default once the device communicator is bound) on
    gloo world-size 2 == the single-process run, including windows where one rank owns no row and — flights: chosen
    dummy values — engine reloads in the middle of the run."""
    import torch.multiprocessing as mp
    single = str(tmp_path / "single.npz")
    _run_fused(0, 1, 0, single, name)
    multi = str(tmp_path / "multi.npz")
    port = 33500 + (os.getpid() % 2000) + (53 if name == "flights" else 0)
    mp.spawn(_run_fused, args=(2, port, multi, name), nprocs=2, join=True)
    a, b = np.load(single), np.load(multi)
    for k in a.files:
        if k != "reloads":
            assert np.array_equal(a[k], b[k]), k
    if name == "flights":
        assert b["reloads"][0] > 0, "flights chooses dummy values: the engine must have been reloaded mid-run"

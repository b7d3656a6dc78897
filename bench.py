#!/usr/bin/env python
"""bench.py — rows/sec per Gibbs sweep of the observed class on the synthetic
hospital-shaped table (BASELINE.json metric).

A "step" is one batched rejuvenation sweep of the Record class over all rows
(strong scaling: the 1M-row table is block-partitioned over the ranks), i.e.
upload of the replicated latent tables, the HIP sweep (proposal scoring, draws,
particle weights, final choice), the exchange of the CRP sufficient statistics
(all-reduce, RCCL for N>1) and of new-row records, and the host commit.

Prints ONE JSON line on rank 0 (contract in the task statement) with
`roofline` (dominant kernel = block-0 root enumeration, HIP-event timed on the
library's stream) and `cpu_baseline` (the CPU oracle's sequential-schedule sweep
on a bounded sample of the same workload, single thread).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def build_workload(n_rows, n_hosp, seed):
    from pclean_amd import experiments as ex
    from pclean_amd.model import LoweredModel
    from pclean_amd.synth import synth_hospital
    from pclean_amd.trace import Trace
    t0 = time.time()
    dirty, clean, latent = synth_hospital(n_rows, n_hosp, seed)
    poss = ex.possibilities_of(dirty)
    m = ex.hospital_model(poss)
    q = ex.hospital_query(m)
    lw = LoweredModel(m, q, dirty)
    obs = lw.encode_observations(dirty)
    # initial latent state = the generator's ground-truth entities (initialize_trace at this
    # scale is SURVEY §8f work); clean values that never occur undamaged fall back to the dirty cell
    by_path = [{}, {}]
    ocls = m.classes[q.cls]
    for col, ref in q.cleanmap.items():
        if "." not in ref:
            continue
        head, rest = ref.split(".", 1)
        bi = 0 if head == "hosp" else 1
        cname, attr = m.resolve(ocls.attr(head).target, rest)
        dom = lw.latent_dom[(cname, attr.name)]
        by_path[bi][rest] = [c if dom.get(c) >= 0 else d for c, d in zip(clean[col], dirty[col])]
    tr = Trace.from_clean_values(lw, by_path, n_rows, seed)
    log(f"[bench] workload built in {time.time() - t0:.1f}s: rows={n_rows} "
        + " ".join(f"{c}={t.n}" for c, t in tr.tables.items()))
    return dirty, clean, lw, obs, tr


def oracle_world_for_rows(orc, lw, obs_local, tr, eng, rows):
    """Oracle World (test infrastructure) holding the given observed rows and the full latent
    state currently uploaded to `eng`; pair-table rows and double tables are read back from the
    library so both sides score with bit-identical inputs.  Returns (world, py-params array)."""
    sub = obs_local[:, rows]
    w = orc.World()
    sub_local = np.empty_like(sub)
    remap = {}
    for j, dirty_attr in enumerate(lw.obs_cols):  # remap each column to the values present in the sample
        u, inv = np.unique(sub[j], return_inverse=True)
        assert u.size == 0 or u[0] >= 0, "missing observations are not expected in the synthetic table"
        remap[dirty_attr] = u
        sub_local[j] = inv
    w.set_obs(np.ascontiguousarray(sub_local))
    mr, md, ml, nb, logl = eng.hip.get_density_tables()
    w.set_density(mr, md, ml, nb, logl)
    for key, (pid, odom, ldom) in lw.pair_id.items():
        d = eng.hip.get_pair_rows(pid, remap[key[0]], len(ldom))
        w.set_pair(pid, d, lw.pool.lens[ldom.id_array()].astype(np.uint16))
    for fid, fn in lw.fn_tables.items():
        w.set_fn(fid, fn)
    py = np.zeros((64, 2))
    for cname, t in tr.tables.items():
        cols, counts = t.view()
        full, m1, scal = eng.hip.get_table_priors(lw.table_id[cname], len(counts))
        w.set_table(lw.table_id[cname], np.ascontiguousarray(cols), counts, full, m1, scal)
        py[lw.table_id[cname]] = (t.strength, t.discount)
    for (cname, aname), dom in lw.latent_dom.items():
        w.set_options(lw.option_id[(cname, aname)], lw.option_values[(cname, aname)], eng.option_logp[(cname, aname)])
    for bi in range(len(lw.blocks)):
        w.load_block(bi, *lw.block_arrays(bi))
    return w, py


def hbm_traffic(args, world):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes (profiles/collect.sh ->
    profiles/hbm_traffic.json); only valid for the configuration it was measured on."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        d = json.load(open(path))
        if d.get("rows") != args.rows or d.get("hospitals") != args.hospitals or d.get("particles") != args.particles \
                or world != 1:
            return None
        return float(d["bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(lw, obs_local, tr, eng, cfg, seed, target_seconds):
    """Oracle, sequential schedule, single thread, on a prefix sample of the rows."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as orc
    from pclean_amd._lib import InferConfig
    orc.build()
    eng.upload_trace(tr)  # the tables the oracle copies are the ones currently in the trace

    def run(n_sample):
        w, py = oracle_world_for_rows(orc, lw, obs_local, tr, eng, np.arange(n_sample))
        cur = np.ascontiguousarray(tr.cur[:, :n_sample].copy())
        c = InferConfig(1, cfg.num_particles, 1, 1, int(cfg.use_mh_instead_of_pg), 50, 100)
        moved, new = C.c_int64(), C.c_int64()
        t0 = time.perf_counter()
        orc.lib().pco_sweep_sequential(w.h, C.byref(c), C.c_uint64(seed), C.c_uint32(0), cur.shape[0], C.c_int64(0),
                                       orc._p(cur, C.c_int32), orc._p(py, C.c_double), C.byref(moved), C.byref(new))
        return time.perf_counter() - t0

    n_total = obs_local.shape[1]
    probe = min(64, n_total)
    t_probe = run(probe)
    n_sample = int(min(n_total, max(probe, target_seconds / max(t_probe / probe, 1e-9))))
    t = run(n_sample) if n_sample > probe else t_probe
    return dict(value=n_sample / t, unit="rows/s/sweep", cores=1, kind="port",
                sample=f"first {n_sample} rows of the same synthetic table, 1 sequential-schedule sweep of Record, "
                       f"{t:.1f}s, single thread of {os.cpu_count()} host cores; CPU restatement (oracle/), "
                       "not the Julia reference")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--hospitals", type=int, default=10_000)
    ap.add_argument("--particles", type=int, default=20)
    ap.add_argument("--seed", type=int, default=20250926)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--emulate-shard-of", type=int, default=0,
                    help="diagnostic, single process: sweep only the first 1/G of the rows (what one rank of a "
                         "G-GPU job does, without the collectives) and report the step time on stderr; no JSON")
    args = ap.parse_args()

    import torch
    from pclean_amd import _lib
    from pclean_amd.analysis import accuracy_counts, f1_from_counts
    from pclean_amd.engine import Engine, InferenceConfig
    from pclean_amd.parallel import Comm, exchange_and_commit, shard_bounds

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and pclean_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("PCLEAN_FORCE_DIST"):  # PCLEAN_FORCE_DIST: exercise RCCL with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    comm = Comm(device=f"cuda:{local_rank}")

    dirty, clean, lw, obs, tr = build_workload(args.rows, args.hospitals, args.seed)
    lo, hi = shard_bounds(args.rows, rank, world)
    if args.emulate_shard_of > 1 and world == 1:
        lo, hi = shard_bounds(args.rows, 0, args.emulate_shard_of)
    obs_local = np.ascontiguousarray(obs[:, lo:hi])
    tr.cur = np.ascontiguousarray(tr.cur[:, lo:hi])
    t0 = time.time()
    eng = Engine(lw, obs_local, device=local_rank, dist_mode=_lib.DIST_OSA, row_offset=lo)
    log(f"[bench] rank {rank}: pair tables + static upload in {time.time() - t0:.1f}s "
        f"({len(lw.pair_id)} tables, {sum(len(o) * len(l) for _, o, l in lw.pair_id.values()) / 1e9:.2f} G pairs)")
    cfg = InferenceConfig(args.warmup + args.steps, args.particles)

    def step(idx):
        t_a = time.perf_counter()
        eng.upload_trace(tr)
        t_b = time.perf_counter()
        choice, chosen, logml, new_rows = eng.sweep(tr, cfg, args.seed, idx, reuse_buffers=True)
        t_c = time.perf_counter()
        stats = eng.sweep_stats(tr)
        moved = eng.sweep_moved()
        tm = eng.hip.get_timing()
        t_d = time.perf_counter()
        if os.environ.get("PCLEAN_BENCH_DEBUG"):
            log("[bench] moved per block", (choice != tr.cur).sum(axis=1), "new per block", (choice < 0).sum(axis=1),
                "chosen particle hist", np.bincount(chosen, minlength=cfg.num_particles)[:6])
        if os.environ.get("PCLEAN_BENCH_PROFILE") and idx == args.warmup + args.steps - 1:
            import cProfile
            import pstats
            pr = cProfile.Profile()
            pr.enable()
            changed = exchange_and_commit(tr, lw, comm, lo, choice, stats, new_rows, moved_local=moved)
            pr.disable()
            pstats.Stats(pr, stream=sys.stderr).sort_stats("tottime").print_stats(12)
        else:
            changed = exchange_and_commit(tr, lw, comm, lo, choice, stats, new_rows, moved_local=moved)
        t_e = time.perf_counter()
        if os.environ.get("PCLEAN_BENCH_DEBUG"):
            log(f"[bench] host phases ms: upload {1e3 * (t_b - t_a):.2f} sweep call {1e3 * (t_c - t_b):.2f} "
                f"(device span {tm.total_ms:.2f}) stats {1e3 * (t_d - t_c):.2f} exchange+commit {1e3 * (t_e - t_d):.2f}")
        return tm, changed

    for i in range(args.warmup):
        tm, changed = step(i)
        log(f"[bench] warmup sweep {i}: device {tm.total_ms:.1f} ms, hot kernel {tm.hot_kernel_ms:.1f} ms, "
            f"{changed} referents changed, {tm.reserved} items re-run by the generic kernel")
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hot_ms, hot_launches, alg_bytes, dev_ms = 0.0, 0, 0.0, 0.0
    for i in range(args.steps):
        tm, changed = step(args.warmup + i)
        hot_ms += tm.hot_kernel_ms
        hot_launches += tm.hot_kernel_launches
        alg_bytes += tm.hot_kernel_alg_bytes
        dev_ms += tm.total_ms
    torch.cuda.synchronize()
    comm.barrier()
    elapsed = comm.max_float(time.perf_counter() - t0)

    if args.emulate_shard_of > 1 and world == 1:
        log(f"[bench] emulated rank 0 of {args.emulate_shard_of}: {hi - lo} rows, {1e3 * elapsed / args.steps:.2f} ms per step "
            f"(device span {dev_ms / args.steps:.2f} ms, root kernel {hot_ms / max(hot_launches, 1):.2f} ms), no collectives")
        eng.close()
        return
    cnt = accuracy_counts(lw, tr, {c: v[lo:hi] for c, v in dirty.items()}, {c: v[lo:hi] for c, v in clean.items()})
    cnt = comm.allreduce_sum_i64(cnt)
    acc = f1_from_counts(cnt)

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = args.rows * args.steps / elapsed
        K = tr.tables["Hospital"].n
        per_launch_bytes = alg_bytes / max(hot_launches, 1)
        per_launch_s = 1e-3 * hot_ms / max(hot_launches, 1)
        achieved = per_launch_bytes / per_launch_s / 1e9 if per_launch_s > 0 else 0.0
        out = {
            "metric": "rows/sec per Gibbs sweep on 1M-row synthetic hospital; F1 vs ground truth",
            "value": value, "unit": "rows/s/sweep", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic hospital x{args.rows // 1000}: {args.rows} dirty rows, "
                                   f"{args.hospitals} latent hospitals, Record class, PG n_particles={args.particles}, "
                                   "2 blocks, batched schedule", "rows": args.rows, "latent_hospitals": int(K),
                       "particles": args.particles, "parallelism": f"rows sharded over {world} GPU(s)",
                       "init": "ground-truth entities", "device_ms_per_step": dev_ms / args.steps},
            "f1": acc["f1"], "accuracy": acc,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": hbm_traffic(args, world),
                         "kernel": "fk_root_fast_kernel<12> (block 0 root: rows x candidate hospitals)",
                         "alg_bytes_per_launch": per_launch_bytes, "avg_launch_ms": 1e3 * per_launch_s,
                         "note": "achieved = SURVEY §8d algorithmic bytes (full enumeration, 920 296 B/row) / HIP-event "
                                 "kernel time of this rank. The kernel skips most of that work exactly (integer "
                                 "pre-filter, one workgroup per distinct row tuple), so frac > 1 is expected; "
                                 "traffic = measured HBM bytes per launch (rocprofv3 FETCH_SIZE x2 gfx950 correction "
                                 "+ WRITE_SIZE, profiles/hbm_traffic.json), DESIGN.md §5"},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lw, obs_local, tr, eng, cfg, args.seed, args.cpu_seconds)
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
        json_line = json.dumps(out)
    eng.close()
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio; flush that first so that the JSON line is the
        # LAST line on stdout
        try:
            C.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json_line, flush=True)


if __name__ == "__main__":
    main()

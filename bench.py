#!/usr/bin/env python
"""bench.py — rows/sec per Gibbs sweep of the observed class on the synthetic
hospital-shaped table (BASELINE.json metric).

Pipeline of one run (every rank; rows in random order, experiments.shuffle_rows):
  1. synthetic table -> model -> pair tables on the GPU (timed: `table_build`, DP cells/s);
  2. the build's OWN `initialize_trace` from an empty trace (timed: `config.init_s`, F1 after it);
  3. ONE full `run_inference` iteration over every class (latent classes + observed class; timed:
     `config.full_iteration_ms`);
  4. the timed region: W warmup + K "steps".  A step is one batched rejuvenation sweep of the Record class
     over all rows (strong scaling: the 1M-row table is block-partitioned over the ranks) through the
     product's own `inference.observed_sweep` with one sub-batch: upload of the replicated latent tables,
     the HIP sweep (proposal scoring, draws, particle weights, final choice), the exchange of the CRP
     sufficient statistics (all-reduce, RCCL for N>1) and of new-row records / moved rows, the host commit;
  5. one extra, untimed sweep with the library's per-phase HIP-event profile on (`phases_ms`).

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel = block 0's
root scan, HIP-event timed on the library's stream; algorithmic bytes = what THIS algorithm has to move, see
`roofline_model`) and `cpu_baseline` (the CPU oracle's sequential-schedule sweep on a bounded sample of the
same workload, single thread).

`python bench.py --gpus N` without a torchrun environment re-launches itself under
`python -m torch.distributed.run --nproc-per-node N` (one rank per GPU).
"""
import argparse
import ctypes as C
import json
import os
import subprocess
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
    t0 = time.time()
    dirty, clean, latent = synth_hospital(n_rows, n_hosp, seed)
    (dirty, clean), _ = ex.shuffle_rows([dirty, clean], seed)  # the generator emits a hospital's records consecutively
    poss = ex.possibilities_of(dirty)
    m = ex.hospital_model(poss)
    q = ex.hospital_query(m)
    lw = LoweredModel(m, q, dirty)
    obs = lw.encode_observations(dirty)
    log(f"[bench] workload built in {time.time() - t0:.1f}s: rows={n_rows}")
    return dirty, clean, lw, obs


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
    from pclean_amd.encode import load_lm_params
    sym, off, _, _ = lw.pool.arrays()
    w.set_strings(sym, off)
    w.set_lm(*load_lm_params(), lw.pool.letter_symbols())
    for key, (pid, odom, ldom) in lw.pair_id.items():
        d = eng.hip.get_pair_rows(pid, remap[key[0]], len(ldom))
        w.set_pair(pid, d, lw.pool.lens[ldom.id_array()].astype(np.uint16))
        w.set_pair_strings(pid, odom.id_array()[remap[key[0]]], eng.dist_mode)
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
    """HBM bytes per launch of the dominant kernel GROUP (group_desc_kernel + group_settle_kernel + fk_root_wave_kernel<12> + group_lse_kernel of
    block 0's root: the launches `alg_bytes_per_launch` models) from this round's committed PMC passes (profiles/collect_r04.sh
    -> profiles/hbm_traffic.json: rocprofv3 FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, separate --pmc runs of this same
    command); only valid for the configuration it was measured on, otherwise null.  Counters cannot be read from inside
    the process, so this figure is NOT measured in the run that prints it."""
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    try:
        d = json.load(open(path))
        if d.get("rows") != args.rows or d.get("hospitals") != args.hospitals or d.get("particles") != args.particles \
                or world != 1 or "pair_bytes_per_launch" not in d:
            return None, None, None
        hbm_traffic.measure_root = d.get("measure_root")  # (the longest single kernel of the step: counters of the same passes)
        return float(d["pair_bytes_per_launch"]), d.get("source"), d.get("pair_components")
    except Exception:
        return None, None, None


def step_traffic(args, world):
    """HBM bytes of one WHOLE sweep (every kernel of the step) from the committed all-kernel PMC passes
    (profiles/step_traffic.py -> profiles/hbm_traffic.json "step"); null off the measured configuration."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get("step")
        if not d or d.get("rows") != args.rows or d.get("hospitals") != args.hospitals or d.get("particles") != args.particles \
                or world != 1:
            return None
        return d
    except Exception:
        return None


LINE = 128  # bytes the memory system moves for one random gather: the L2 line.  The FETCH_SIZE calibration
# (profiles/calib/fetch_calib.hip, profiles/r03_fetch_calibration.txt) counts a single-byte gather at 64 B and the guide's
# gfx950 correction doubles FETCH_SIZE: 128 B per gather, the same rule the counter traffic is corrected with.


def roofline_model(rs, obs_local, particles):
    """Algorithmic bytes of ONE launch of the dominant kernel group — group_desc_kernel + group_settle_kernel +
    fk_root_wave_kernel<12> + group_lse_kernel of block 0's root (root_wave.hip) — as implemented, each byte counted once, every random gather at
    the 128-byte line the memory system moves for it (LINE):
      * coarse level of the scan: one block-minimum row (cstride bytes, one byte per 64 candidates) per DISTINCT
        observed value of the pre-filter columns among the swept rows;
      * fine level: the 64-candidate blocks the launch's scans actually read (rs.fine_blocks, counted by the kernel):
        three byte rows (one line each) + 8 B of the alive bitmap;
      * exact scores: group_desc_kernel gathers every term of the current referent once per group (one line each);
        the scan kernel gathers the (survivor, term) pairs it could not take from the descriptor (rs.scored_terms);
      * group descriptors: 128 B written and read per group; per group the representative's observed ids, grp_off /
        members (4 B per item + 4 B per group), the current referent (4 B per item);
      * outputs: 4 B per (row, particle) draw, 8 B log-marginal and 4 B overflow flag per row.
    Returns (bytes, bytes of round 2's model: every distinct pre-filter byte row streamed once — what the kernel read
    before the two-level scan).  The §8(d) figure of SURVEY.md (every candidate of every row gathered, 920 296 B/row)
    is reported beside both as `enumeration_equivalent`: work the kernel provably skips."""
    if not rs.fast:
        return None, None
    distinct = 0
    for p in range(rs.n_pre):
        col = rs.pre_obs_col[p]
        if col >= 0:
            distinct += int(np.unique(obs_local[col]).size)
    # (descriptor: written by group_desc_kernel, read by the scan kernel and — when it ran: rs.resolved_groups — by
    # group_settle_kernel, whose fine blocks are in rs.fine_blocks and whose draws are the rows' draws below)
    def model(line):
        # (rs.pre_scored: the current referent's exact score comes from group_gate_kernel — part of the timed launch group
        # since round 5 — instead of group_desc_kernel: the same gathers, + a flag and the score written and read per group)
        per_group = (3 if rs.resolved_groups > 0 else 2) * 128 + 4 * rs.n_terms + 4 + line * rs.n_terms + (20 if rs.pre_scored else 0)
        # (the last block's root leaves its survivor lists for ONE draw per row after the final choice instead of n_draws
        # draws per item: 12 bytes per (survivor, prefix) pair + the group's count and total)
        lazy = getattr(rs, "lazy_entries", 0)
        per_item = 4 + 4 + (0 if lazy else 4 * rs.n_draws) + 8 + 4
        common = rs.n_groups * (per_group + (12 if lazy else 0)) + rs.n_items * per_item + 12 * lazy
        two_level = distinct * rs.cstride + rs.fine_blocks * (3 * line + 8) + rs.scored_terms * line
        return common, two_level
    common, two_level = model(LINE)
    common64, two_level64 = model(64)  # (rounds 1-3 charged a gather the 64-byte sector; kept so that the rounds compare)
    rows_once = distinct * rs.kpad
    # Each byte counted ONCE: when the groups share their observed values (the Measure slot: 169 k groups over a few thousand
    # distinct values, ~60 survivors each) the per-scan / per-survivor charges above count the same short byte rows again and
    # again — every distinct row streamed once is then the smaller, and the honest, denominator (block 0: the two-level
    # figure is the smaller one by an order of magnitude, nothing changes there)
    roofline_model.sector64 = float(common64 + min(two_level64, rows_once))
    roofline_model.which = "two-level scan (blocks and gathers the kernel counted)" if two_level <= rows_once else \
        "every distinct pre-filter byte row streamed once (the groups share their rows)"
    return float(common + min(two_level, rows_once)), float(common + rows_once)


def step_byte_model(n, p, rs, n_groups_block1, n_obs_compact, kpad_compact, delta_rows):
    """Algorithmic bytes of the largest movers of one sweep besides the root scan of block 0 (profiles/r05_step_traffic.txt
    names them), per kernel group of the ROUND-5 step: what each has to read and write once, in bytes.  n rows, p particles.
    (Gone since round 4: the per-item gate of the new-row branch — the gate runs per group inside the root-scan launch
    group, roofline.alg_bytes_per_launch — and the resampling step after the first block, a no-op that is not launched.)"""
    np_ = n * p
    m = {}
    # block 0 (particle_update_kernel): draws [N][P] + lse + cur in, pchoice + weights out (the first block stores the
    # weights); block 1 (particle_update_final_kernel): the particle's item (4), its draw (4), its item's lse (8, mostly
    # shared), weights read, the final choice made in the same kernel: chosen particle + choice out per row
    m["particle_update_kernel + particle_update_final_kernel"] = np_ * (4 + 4 + 8) + n * 12 + np_ * (4 + 4 + 8 + 8) + n * 12
    # grouping through the hash table (eval.hip: make_item_groups_hash), three per sweep: the table (2n slots x 8 B) zeroed,
    # per item its key words (12 B) + a probe (line) + slot / position out (8 B), the scan over the slots (16 B each way),
    # the fill (slot, position, the slot's offsets: 8 + 16 B in, member / head / uid out: 12 B)
    m["hash grouping: table zero + insert + slot scan + fill (3 groupings)"] = 3 * (2 * n * 8 + n * (12 + LINE + 8) + 2 * n * 16 + n * (8 + 16 + 12))
    m["gather_ctx + ctx_count + ctx_fill"] = np_ * (4 + 4) + np_ * 4 + n * 8 + np_ * 4 + n * (4 + 16 + 4) + np_ * 4
    m["final choice of the flagged rows + finalize_block x2 + tail lists"] = n * 12 + 2 * n * (4 + 4 + 16) + 4 * n * 4
    m["compact table refresh (Measure)"] = sum(delta_rows * no * (LINE // 2) + no * kp * 1 for no, kp in zip(n_obs_compact, kpad_compact))
    m["root scans of the Measure slot + nested slots (group descriptors + draws)"] = n_groups_block1 * (2 * 128 + 4 * LINE) + n * (8 + 4 * p)
    return {k: float(v) for k, v in m.items()}


def cpu_baseline(lw, obs, tr, eng, cfg, seed, min_rows, target_seconds):
    """Oracle, sequential schedule, single thread, on a prefix sample of the (shuffled) rows."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as orc
    from pclean_amd._lib import InferConfig
    orc.build()
    eng.upload_trace(tr)  # the tables the oracle copies are the ones currently in the trace

    def run(n_sample):
        w, py = oracle_world_for_rows(orc, lw, obs, tr, eng, np.arange(n_sample))
        cur = np.ascontiguousarray(tr.cur[:, :n_sample].copy())
        c = InferConfig(1, cfg.num_particles, 1, 1, int(cfg.use_mh_instead_of_pg), 50, 100)
        moved, new = C.c_int64(), C.c_int64()
        t0 = time.perf_counter()
        orc.lib().pco_sweep_sequential(w.h, C.byref(c), C.c_uint64(seed), C.c_uint32(0), cur.shape[0], C.c_int64(0),
                                       orc._p(cur, C.c_int32), orc._p(py, C.c_double), C.byref(moved), C.byref(new))
        return time.perf_counter() - t0

    def run_pruned(n_sample):
        # the same sweep with the HIP path's two exact work savers on one thread (oracle/pruned.h): rows / particles that agree
        # on (observed values, contexts, current referent) share one enumeration, candidates and new-row branches whose
        # fixed-point weight is provably 0 are never scored; batched schedule (frozen tables: what makes the memo exact)
        w, _ = oracle_world_for_rows(orc, lw, obs, tr, eng, np.arange(n_sample))
        cur = np.ascontiguousarray(tr.cur[:, :n_sample].copy())
        c = InferConfig(1, cfg.num_particles, 1, 1, int(cfg.use_mh_instead_of_pg), 50, 100)
        t0 = time.perf_counter()
        st = w.sweep_batched(c, seed, 0, cur, pruned=True)[3]
        return time.perf_counter() - t0, st

    n_total = obs.shape[1]
    probe = min(64, n_total)
    t_probe = run(probe)
    n_sample = int(min(n_total, max(probe, min_rows, target_seconds / max(t_probe / probe, 1e-9))))
    t = run(n_sample) if n_sample > probe else t_probe
    out = dict(value=n_sample / t, unit="rows/s/sweep", cores=1, kind="port",
               sample=f"first {n_sample} rows of the same (shuffled) synthetic table against the full latent state, "
                      f"1 sequential-schedule sweep of Record, {t:.1f}s, single thread of {os.cpu_count()} host cores; "
                      "CPU restatement (oracle/), not the Julia reference")
    try:
        probe_p = min(2000, n_total)
        tp, _ = run_pruned(probe_p)
        n_p = int(min(n_total, 200_000, max(probe_p, target_seconds / max(tp / probe_p, 1e-9))))
        (tp, stp) = run_pruned(n_p) if n_p > probe_p else run_pruned(probe_p)
        out["pruned"] = dict(value=n_p / tp, unit="rows/s/sweep", cores=1, kind="port",
                             sample=f"first {n_p} rows, 1 batched-schedule sweep with grouping + exact pruning on one thread "
                                    f"(oracle/pruned.h: the HIP path's work savers restated; results == the plain oracle's, "
                                    f"tests/test_oracle_pruned.py), {tp:.1f}s", work=stp)
    except Exception as e:  # (the second leg must never cost the first)
        out["pruned"] = dict(error=repr(e))
    return out


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: one rank per GPU through torch.distributed.run."""
    port = int(os.environ.get("MASTER_PORT", "29533"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("[bench] spawning:", " ".join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--hospitals", type=int, default=10_000)
    ap.add_argument("--particles", type=int, default=20)
    ap.add_argument("--seed", type=int, default=20250926)
    ap.add_argument("--init-batch", type=int, default=32768, help="largest batch of the batched initialize_trace")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cpu-rows", type=int, default=10_000, help="minimum rows of the CPU baseline sample (SURVEY §8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-iteration", action="store_true")
    ap.add_argument("--no-steady-iterations", action="store_true",
                    help="skip the two steady-state full iterations after the timed steps (profiling runs: the tools read the last sweeps)")
    ap.add_argument("--distance", choices=("osa", "dl"), default="dl",
                    help="flavour of the AddTypos pair tables the workload runs on: unrestricted Damerau-Levenshtein (the default: "
                         "the Engine's default and what the three real programs use; 31 s of table build at 1M rows) or the "
                         "restricted one (OSA, bit-parallel, 0.45 s: the headline's tables in rounds 1-4; 0.48 %% of the pairs differ)")
    ap.add_argument("--no-dl-sample", action="store_true", help="skip timing the unrestricted-DL kernel on one table")
    ap.add_argument("--dl-sample-cells", type=float, default=6e10, help="largest table (in DP cells) the DL sample may pick")
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong",
                    help="strong (default, the BASELINE metric): --rows rows in total, sharded over the ranks; weak: --rows "
                         "rows PER RANK (the table grows with the ranks, the number of true hospitals stays: the pair "
                         "tables scale with unique values squared and would not fit otherwise)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    if args.gpus != world:
        log(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: running {world} rank(s), reported as n_gpus={world}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    from pclean_amd import _lib
    from pclean_amd.analysis import accuracy_counts, f1_from_counts
    from pclean_amd.engine import Engine, InferenceConfig
    from pclean_amd.inference import initialize_trace, observed_sweep, run_inference
    from pclean_amd.parallel import Comm, shard_bounds
    from pclean_amd.trace import Trace

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no GPU visible and pclean_amd has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1 or os.environ.get("PCLEAN_FORCE_DIST"):  # PCLEAN_FORCE_DIST: exercise RCCL with one rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    comm = Comm(device=f"cuda:{local_rank}")

    if args.scaling == "weak":
        args.rows *= world
    dirty, clean, lw, obs = build_workload(args.rows, args.hospitals, args.seed)
    t0 = time.time()
    # every rank holds all observation columns (60 MB) and the whole trace: latent-class sweeps and the
    # initialisation need every referring row; the observed-class sweep is sharded by rows
    eng = Engine(lw, obs, device=local_rank, dist_mode=_lib.DIST_OSA if args.distance == "osa" else _lib.DIST_DL)
    static_s = time.time() - t0
    if comm.dist is not None:  # the library's own RCCL communicator: fused device-side all-reduce of the CRP statistics
        eng.init_device_comm(comm)
    log(f"[bench] rank {rank}: pair tables {eng.pair_build_s:.1f}s ({len(lw.pair_id)} tables, {eng.pair_count / 1e9:.2f} G pairs, "
        f"{eng.pair_cells / 1e12:.2f} T DP cells), static upload total {static_s:.1f}s")

    # the unrestricted-DL kernel (what the three real programs build their tables with) on one table of the workload
    dl_sample = None
    if rank == 0 and not args.no_dl_sample:
        def cells_of(v):
            return int(lw.pool.lens[v[1].id_array()].astype(np.int64).sum()) * int(lw.pool.lens[v[2].id_array()].astype(np.int64).sum())

        # (bounded by DP cells: the sample must stay a few seconds whatever the kernel's speed on long strings)
        key, (pid, odom, ldom) = max(((k, v) for k, v in lw.pair_id.items() if cells_of(v) <= args.dl_sample_cells),
                                     key=lambda kv: cells_of(kv[1]))
        oi, li = odom.id_array(), ldom.id_array()
        spare = lw._next_pair  # (an unused pair-table id)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.hip.build_pair_table(spare, oi, li, _lib.DIST_DL)
        torch.cuda.synchronize()
        dl_s = time.perf_counter() - t0
        t0 = time.perf_counter()
        eng.hip.build_pair_table(spare, oi, li, _lib.DIST_OSA)
        torch.cuda.synchronize()
        osa_s = time.perf_counter() - t0
        cells = int(lw.pool.lens[oi].astype(np.int64).sum()) * int(lw.pool.lens[li].astype(np.int64).sum())
        dl_sample = {"table": f"{key[0]} x {key[1]}", "pairs": len(oi) * len(li), "dp_cells": cells,
                     "dl_seconds": dl_s, "dl_pairs_per_s": len(oi) * len(li) / dl_s, "dl_dp_cells_per_s": cells / dl_s,
                     "osa_seconds": osa_s, "osa_pairs_per_s": len(oi) * len(li) / osa_s}
        log(f"[bench] distance kernels on {dl_sample['table']} ({dl_sample['pairs'] / 1e9:.2f} G pairs): unrestricted DL "
            f"{dl_s:.2f}s = {dl_sample['dl_pairs_per_s'] / 1e9:.2f} G pairs/s, OSA {osa_s:.3f}s = {dl_sample['osa_pairs_per_s'] / 1e9:.1f} G pairs/s")

    def f1_now(tr):
        return f1_from_counts(accuracy_counts(lw, tr, dirty, clean))  # replicated trace: same counts on every rank

    # ---- the build's own initialisation + one full iteration ---------------------------------------------------
    cfg1 = InferenceConfig(1, args.particles)
    tr = Trace(lw, args.rows, args.seed)
    t0 = time.time()
    initialize_trace(eng, tr, cfg1, args.seed, max_batch=args.init_batch, comm=comm)
    torch.cuda.synchronize()
    init_s = time.time() - t0
    acc_init = f1_now(tr)
    log(f"[bench] initialize_trace {init_s:.1f}s: F1 {acc_init['f1']:.4f} " + " ".join(f"{c}={t.n_live}" for c, t in tr.tables.items()))
    full_ms = None
    from pclean_amd import inference as inf
    # one-time set-up of the device-resident commit (tables re-uploaded with spare capacity, scratch, log tables):
    # inference would do it at its first observed-class sweep
    t0 = time.time()
    dc_on = inf.DEVICE_COMMIT and eng.enable_device_commit(tr, comm)
    torch.cuda.synchronize()
    dc_enable_ms = 1e3 * (time.time() - t0)
    log(f"[bench] device-resident commit: {'on' if dc_on else 'off (' + getattr(eng, '_dc_why', 'several ranks') + ')'}, set-up {dc_enable_ms:.0f} ms")
    # one-time set-up of every class's compact tables and caches (Engine.prepare): what the first iteration would
    # otherwise allocate and build on its way (a few GB of byte tables)
    t0 = time.time()
    eng.prepare(tr, comm)
    torch.cuda.synchronize()
    prepare_ms = 1e3 * (time.time() - t0)
    log(f"[bench] prepare (compact tables and caches of every class): {prepare_ms:.0f} ms")
    if not args.no_full_iteration:
        inf.TIMERS.clear()
        eng.hip.set_profiling(True)
        t0 = time.time()
        run_inference(eng, tr, cfg1, args.seed, comm=comm)
        torch.cuda.synchronize()
        full_ms = 1e3 * (time.time() - t0)
        full_phases = eng.hip.get_profile()
        eng.hip.set_profiling(False)
        log("[bench] full iteration, host phases (ms): " + ", ".join(f"{k} {1e3 * v:.0f}" for k, v in sorted(inf.TIMERS.items(), key=lambda kv: -kv[1]) if v > 2e-3))
        log("[bench] full iteration, device phases (ms): " + ", ".join(f"{k} {v[0]:.1f}/{v[1]}" for k, v in sorted(full_phases.items(), key=lambda kv: -kv[1][0])))
        log(f"[bench] one full run_inference iteration (every class): {full_ms:.0f} ms "
            + " ".join(f"{c}={t.n_live}" for c, t in tr.tables.items()))

    # ---- timed region: observed-class sweeps through the product path ------------------------------------------
    cfg = InferenceConfig(args.warmup + args.steps, args.particles)

    def step(idx):
        # the product's DEFAULT schedule (inference.observed_sweep as run_inference calls it): Record declares no
        # learned parameter, so there is nothing to resample every rejuv_frequency rows and the class is swept in
        # one batch (inference.sub_batches)
        return observed_sweep(eng, tr, cfg, args.seed, 1 + idx, comm), eng.hip.get_timing()

    inf.TIMERS.clear()
    for i in range(args.warmup):
        changed, tm = step(i)
        log(f"[bench] warmup sweep {i}: device {tm.total_ms:.2f} ms, root scan {tm.hot_kernel_ms:.2f} ms, "
            f"{changed} referents changed, {tm.reserved} items re-run by the generic kernel")
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    hot_ms, hot_launches, enum_bytes, dev_ms = 0.0, 0, 0.0, 0.0
    for i in range(args.steps):
        changed, tm = step(args.warmup + i)
        hot_ms += tm.hot_kernel_ms
        hot_launches += tm.hot_kernel_launches
        enum_bytes += tm.hot_kernel_alg_bytes
        dev_ms += tm.total_ms
    torch.cuda.synchronize()
    comm.barrier()
    elapsed = comm.max_float(time.perf_counter() - t0)
    log(f"[bench] timed region, host phases (ms per step over {args.warmup + args.steps} sweeps): "
        + ", ".join(f"{k} {1e3 * v / (args.warmup + args.steps):.2f}" for k, v in sorted(inf.TIMERS.items(), key=lambda kv: -kv[1])))
    rs = eng.hip.get_root_stats()
    lo, hi = shard_bounds(args.rows, rank, world)
    alg_bytes, alg_bytes_rows_once = roofline_model(rs, obs[:, lo:hi], cfg.num_particles)

    # ---- the same sweep cut into 32 sub-batches (upload / sweep / exchange / commit per sub-batch): what a class WITH
    # learned parameters costs under the default max_sub_batches (reported beside the headline, not as it) --------------
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_sub = 2
    for i in range(n_sub):
        observed_sweep(eng, tr, cfg, args.seed, 1000 + i, comm, batch_rows=-(-args.rows // 32))
    torch.cuda.synchronize()
    comm.barrier()
    ms_32 = 1e3 * comm.max_float(time.perf_counter() - t0) / n_sub

    # ---- fixed / proportional split of the step: the same sweep as TWO half windows costs 2 x fixed + proportional ---
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n_half = 8  # (with three the estimate moved by a millisecond from run to run)
    for i in range(n_half):
        observed_sweep(eng, tr, cfg, args.seed, 2000 + i, comm, batch_rows=-(-args.rows // 2))
    torch.cuda.synchronize()
    comm.barrier()
    ms_2 = 1e3 * comm.max_float(time.perf_counter() - t0) / n_half

    # ... against the same number of whole-window sweeps timed the same way, right beside them
    comm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # (these sweeps also time the LAST reference slot's root launch group — the Measure slot, the longest kernel of the step —
    # with the same HIP events the timed region spent on block 0's: pclean_set_timed_block)
    last_slot = max(bi for bi, blk in enumerate(lw.blocks) if not blk.get("score"))
    eng.hip.set_timed_block(last_slot)
    m_ms, m_launches = 0.0, 0
    for i in range(n_half):
        observed_sweep(eng, tr, cfg, args.seed, 3000 + i, comm)
        tm = eng.hip.get_timing()
        m_ms += tm.hot_kernel_ms
        m_launches += tm.hot_kernel_launches
    torch.cuda.synchronize()
    comm.barrier()
    ms_1 = 1e3 * comm.max_float(time.perf_counter() - t0) / n_half
    rs_m = eng.hip.get_root_stats()
    eng.hip.set_timed_block(0)
    alg_bytes_m, _ = roofline_model(rs_m, obs[:, lo:hi], cfg.num_particles) if last_slot != 0 else (None, None)
    sector64_m, which_m = getattr(roofline_model, "sector64", None), getattr(roofline_model, "which", None)
    alg_bytes, alg_bytes_rows_once = roofline_model(rs, obs[:, lo:hi], cfg.num_particles)  # (restores roofline_model.sector64 for block 0)

    # ---- several ranks: what one small collective costs on this node (the per-sweep exchange is two of them) ----------
    coll_ms = None
    if getattr(eng, "_dev_comm", False):
        blocks_ = [bi for bi, blk in enumerate(lw.blocks) if not blk.get("score")]
        tids_ = [lw.table_id[lw.blocks[bi]["root_class"]] for bi in blocks_]
        ns_ = [tr.tables[lw.blocks[bi]["root_class"]].n for bi in blocks_]
        comm.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            eng.hip.allreduce_stats_fused(tids_, ns_, True)
        coll_ms = 1e3 * comm.max_float(time.perf_counter() - t0) / 20
        log(f"[bench] rank {rank}: one fused all-reduce of the delta counts ({sum(ns_)} int64, synchronised): {coll_ms:.3f} ms; "
            f"step as 1 window {ms_1:.2f} ms, as 2 half windows {ms_2:.2f} ms -> fixed {max(ms_2 - ms_1, 0):.2f} ms, "
            f"row-proportional {max(2 * ms_1 - ms_2, 0):.2f} ms per step and rank")

    # ---- several ranks: what every rank's exchange moved, how often its device commit was refused (rank 0 prints them) ------
    per_rank = None
    if comm.dist is not None:
        mine = {"rank": rank, "rows": int(shard_bounds(args.rows, rank, world)[1] - shard_bounds(args.rows, rank, world)[0]),
                "device_commits": (eng._dc or {}).get("commits"), "device_commit_refusals": (eng._dc or {}).get("fallbacks"),
                "step_fixed_ms": max(ms_2 - ms_1, 0.0), "step_proportional_ms": max(2 * ms_1 - ms_2, 0.0)}
        if getattr(eng, "_dev_comm", False):
            cs = eng.hip.comm_stats()
            mine.update({"allgather_calls": cs["allgather_calls"], "allgather_bytes_sent_last": cs["allgather_bytes_per_rank_last"],
                         "allgather_bytes_received_last": cs["allgather_bytes_per_rank_last"] * world,
                         "allgather_device_ms_mean": cs["allgather_device_ms"] / max(cs["allgather_calls"], 1),
                         "allreduce_calls": cs["allreduce_calls"], "allreduce_bytes_last": cs["allreduce_bytes_last"],
                         "allreduce_device_ms_mean": cs["allreduce_device_ms"] / max(cs["allreduce_calls"], 1)})
        per_rank = [None] * world
        comm.dist.all_gather_object(per_rank, mine)

    # ---- per-phase profile of one more (untimed) sweep ----------------------------------------------------------
    eng.hip.set_profiling(True)
    step(args.warmup + args.steps)
    phases = eng.hip.get_profile()
    eng.hip.set_profiling(False)
    acc = f1_now(tr)

    # ---- full iterations in the steady state (after everything the line reports): the first one after the initialisation
    # (full_iteration_ms) also pays the first build of every latent class's compact tables, caches and scratch buffers ---
    full_steady_ms = None
    if not args.no_full_iteration and not args.no_steady_iterations:
        comm.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n_full = 2
        for i in range(n_full):
            run_inference(eng, tr, cfg1, args.seed + 100 + i, comm=comm)
        torch.cuda.synchronize()
        comm.barrier()
        full_steady_ms = 1e3 * comm.max_float(time.perf_counter() - t0) / n_full
        log(f"[bench] full run_inference iteration, steady state (mean of {n_full}): {full_steady_ms:.0f} ms; F1 after them "
            f"{f1_now(tr)['f1']:.4f}")

    if rank == 0:
        ms_per_step = 1e3 * elapsed / args.steps
        value = args.rows * args.steps / elapsed
        per_launch_s = 1e-3 * hot_ms / max(hot_launches, 1)
        achieved = (alg_bytes / per_launch_s / 1e9) if (alg_bytes and per_launch_s > 0) else None
        traffic, traffic_src, traffic_parts = hbm_traffic(args, world)
        st = step_traffic(args, world)
        step_model = None
        # per-kernel byte model of the step's largest movers besides the root scans (an algorithmic denominator for the
        # whole step, beside the counters)
        blk1 = lw.blocks[1]
        n_obs_c, kpad_c = [], []
        for t_ in blk1["terms"][blk1["nodes"][0][2]:blk1["nodes"][0][2] + blk1["nodes"][0][3]]:
            if t_[5] < 0:  # plain (compact-table) terms of the Measure slot
                pid = t_[2]
                n_obs_c.append(next(len(v[1]) for v in lw.pair_id.values() if v[0] == pid))
                kpad_c.append(((eng._dc["cap"].get(blk1["root_class"], 0) if eng._dc else tr.tables[blk1["root_class"]].n) + 15) // 16 * 16)
        created = (eng._dc or {}).get("created", {}).get(blk1["root_class"], 0)
        per_kernel = step_byte_model(hi - lo, cfg.num_particles, rs, 0, n_obs_c, kpad_c, max(created, 1))
        per_kernel.pop("root scans of the Measure slot + nested slots (group descriptors + draws)", None)
        if alg_bytes:
            per_kernel["block-0 root scan (roofline.alg_bytes_per_launch)"] = alg_bytes
        if st:  # the whole step against the HBM roof: counter bytes of every kernel of a sweep / this run's device time
            dev_s = 1e-3 * dev_ms / args.steps
            step_model = {"hbm_bytes_per_step": st["hbm_bytes_per_step"], "device_ms_per_step": 1e3 * dev_s,
                          "GBps": st["hbm_bytes_per_step"] / dev_s / 1e9, "frac_of_hbm_peak": st["hbm_bytes_per_step"] / dev_s / 8e12,
                          "dispatches_per_step": st.get("dispatches"), "source": st.get("source"),
                          "by_kernel_counters": st.get("by_kernel")}
        step_model = dict(step_model or {}, alg_bytes_model=per_kernel, alg_bytes_modelled=sum(per_kernel.values()),
                          note="alg_bytes_model: what each kernel group of the round-5 step has to read and write once "
                               "(bench.step_byte_model); the Measure slot's root scan, the new-row sampling and the commit are "
                               "covered by counters only, so hbm_bytes_per_step / alg_bytes_modelled over-states the waste",
                          traffic_over_alg_modelled=(st["hbm_bytes_per_step"] / sum(per_kernel.values())) if st and per_kernel else None)
        out = {
            "metric": "rows/sec per Gibbs sweep on 1M-row synthetic hospital; F1 vs ground truth",
            "value": value, "unit": "rows/s/sweep", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"synthetic hospital x{args.rows // 1000}: {args.rows} dirty rows (random order), "
                                   f"{args.hospitals} true hospitals, Record class, PG n_particles={args.particles}, "
                                   "2 blocks, batched schedule, the product's default observed_sweep (Record has no "
                                   "learned parameter: one batch per sweep), AddTypos pair tables of the "
                                   + ("OSA" if args.distance == "osa" else "unrestricted-DL") + " flavour", "rows": args.rows,
                       "latent_hospitals": int(tr.tables["Hospital"].n_live), "particles": args.particles,
                       "parallelism": f"rows sharded over {world} GPU(s)",
                       "init": f"the build's own initialize_trace from an empty trace (batches <= {args.init_batch})"
                               + ("" if args.no_full_iteration else " + 1 full run_inference iteration"),
                       "init_s": init_s, "f1_after_init": acc_init["f1"],
                       "f1_reference": f"none at this setting: the +-0.5 pt band against the sequential references is tested with the "
                                       f"product's default initialisation batches (<= 256 rows) on <= 50 000 rows "
                                       f"(tests/test_gpu_f1_vs_sequential.py); this run initialises in batches <= {args.init_batch}, "
                                       "and a larger initialisation batch moves F1 upwards (DESIGN.md §9) — the f1 below "
                                       "is a sanity figure for the timed state, not a parity claim", "full_iteration_ms": full_ms,
                       "full_iteration_steady_ms": full_steady_ms, "prepare_ms": prepare_ms,
                       "device_ms_per_step": dev_ms / args.steps,
                       "commit": (("device-resident on every rank (pclean_commit_device_dist): delta counts all-reduced, moved rows and "
                                   "new-row records all-gathered in HBM (RCCL), the same commit kernel everywhere"
                                   if getattr(eng, "_dc_dist", False) else
                                   "device-resident (pclean_commit_device): tables, counts, free lists and referents stay in HBM")
                                  + ", one synchronisation per step" if dc_on else "host (parallel.exchange_and_commit)"),
                       "device_commit_setup_ms": dc_enable_ms,
                       "device_commits": (eng._dc or {}).get("commits"), "device_commit_refusals": (eng._dc or {}).get("fallbacks"),
                       "ms_per_step_32_sub_batches": ms_32,
                       # what does not shrink with the rows a rank sweeps (launches, count read-backs, the commit kernel, table
                       # refreshes) vs what does: from the same sweep run as two half windows (2 x fixed + proportional)
                       "step_fixed_ms": max(ms_2 - ms_1, 0.0), "step_proportional_ms": max(2 * ms_1 - ms_2, 0.0),
                       "ms_per_step_2_windows": ms_2, "ms_per_step_1_window_same_loop": ms_1, "fixed_split_sweeps": n_half,
                       "collective_ms_fused_allreduce": coll_ms,
                       # several ranks (also PCLEAN_FORCE_DIST=1 on one): per rank its shard, refused device commits, the fixed
                       # capacity all-gather of moved rows + new-row records (bytes of the last one, mean device time from HIP events
                       # on the library's stream) and the fused all-reduce of the delta counts
                       "per_rank": per_rank,
                       # the decomposition a strong-scaling number comes with: on N ranks a step costs fixed + collectives +
                       # proportional / N; `--scaling weak` in a separate invocation measures rows-per-rank fixed instead
                       "scaling_model": {"fixed_ms": max(ms_2 - ms_1, 0.0), "proportional_ms": max(2 * ms_1 - ms_2, 0.0),
                                         "collectives_device_ms": (sum((r or {}).get(k, 0.0) or 0.0 for r in per_rank[:1]
                                                                       for k in ("allgather_device_ms_mean", "allreduce_device_ms_mean"))
                                                                   if per_rank else None),
                                         "projected_speedup_at": {str(n): ms_1 / (max(ms_2 - ms_1, 0.0) + max(2 * ms_1 - ms_2, 0.0) / n)
                                                                  for n in (2, 4, 8)} if ms_1 > 0 and world == 1 else None,
                                         "note": "one-GPU split of the same sweep run as two half windows; the projection leaves the "
                                                 "collectives out (two small ones per step) and is printed for world == 1 only"}},
            "f1": acc["f1"], "accuracy": acc,
            "table_build": {"seconds": eng.pair_build_s, "pairs": eng.pair_count, "dp_cells": eng.pair_cells,
                            "dp_cells_per_s": eng.pair_cells / max(eng.pair_build_s, 1e-9),
                            "distance": ("OSA (restricted DL, --distance osa): the tables of the headline in rounds 1-4; the three real "
                                         "programs and bench.py's default use unrestricted DL, 0.48 % of the synthetic pairs differ "
                                         "(DESIGN.md §3)") if args.distance == "osa"
                            else "unrestricted Damerau-Levenshtein (dl_seg_kernel: the linear-space recurrence of csrc/dl_cell.h, 1-8 lanes "
                                 "per pair, latent strings sorted by length): the product's default distance; --distance osa builds "
                                 "the restricted (bit-parallel) tables of rounds 1-4 in 0.45 s instead",
                            "unrestricted_dl_sample": dl_sample},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": (achieved / 8000.0) if achieved else None, "traffic": traffic,
                         "traffic_source": traffic_src, "traffic_components": traffic_parts,
                         "traffic_over_alg": (traffic / alg_bytes) if (traffic and alg_bytes) else None,
                         "measure_root": getattr(hbm_traffic, "measure_root", None),
                         "kernel": "group_gate_kernel + group_desc_kernel + group_settle_kernel + worklist_pack_kernel + fk_root_wave_kernel<12> + "
                                   "group_lse_kernel (block 0 root: rows x candidate hospitals); alg bytes, launch time and counter "
                                   "traffic all cover these launches",
                         "alg_bytes_per_launch": alg_bytes, "avg_launch_ms": 1e3 * per_launch_s,
                         "gather_model": {"bytes_per_gather": LINE,
                                          "alg_bytes_per_launch_64B_sector": getattr(roofline_model, "sector64", None),
                                          "frac_64B_sector": (getattr(roofline_model, "sector64", 0.0) / per_launch_s / 8e12)
                                          if (per_launch_s > 0 and getattr(roofline_model, "sector64", None)) else None,
                                          "note": "rounds 1-3 charged a random gather the 64-byte sector, rounds 4-5 the 128-byte "
                                                  "line the counters see (FETCH_SIZE x 2 on gfx950): `frac` uses the line; the "
                                                  "sector figure is kept so that the rounds compare"},
                         "groups": rs.n_groups, "items": rs.n_items, "kpad": rs.kpad, "overflow_items": rs.overflow_items,
                         "full_scans": rs.full_scans, "fine_blocks": rs.fine_blocks, "scored_terms": rs.scored_terms,
                         "settled_groups": rs.resolved_groups,
                         "rows_streamed_once_model": {"bytes_per_launch": alg_bytes_rows_once,
                                                      "GBps": (alg_bytes_rows_once / per_launch_s / 1e9) if (alg_bytes_rows_once and per_launch_s > 0) else None,
                                                      "note": "round 2's byte model (every distinct pre-filter byte row "
                                                              "streamed once): what the scan read before its block-minimum "
                                                              "level; kept for comparison across rounds"},
                         "step": step_model,
                         "enumeration_equivalent": {"bytes_per_launch": enum_bytes / max(hot_launches, 1),
                                                    "GBps": enum_bytes / max(hot_launches, 1) / max(per_launch_s, 1e-12) / 1e9,
                                                    "note": "SURVEY §8d full-enumeration bytes (920 296 B/row) / kernel time: "
                                                            "work the kernel provably skips, not a bandwidth claim"},
                         "note": "achieved = bytes the implemented algorithm has to move once (bench.roofline_model: block-minimum "
                                 "rows + the fine blocks and gathers the kernel counted, every random gather at the 128-byte line "
                                 "+ descriptors + ids + outputs) / HIP-event time of the launch group on "
                                 "the library's stream; traffic = HBM bytes per launch from this round's rocprofv3 PMC passes "
                                 "(profiles/), not measured in this run"},
            "phases_ms": {k: {"ms": round(v[0], 4), "intervals": v[1]} for k, v in sorted(phases.items(), key=lambda kv: -kv[1][0])},
        }
        # ---- the Measure slot's root launch group, timed and modelled the same way; `roofline` describes whichever of the two
        # launch groups is LONGER per launch, the other one rides along
        if alg_bytes_m and m_launches:
            per_m = 1e-3 * m_ms / m_launches
            mr = getattr(hbm_traffic, "measure_root", None) or {}
            meas = {"bound": "hbm", "achieved": alg_bytes_m / per_m / 1e9, "peak": 8000.0, "unit": "GB/s",
                    "frac": alg_bytes_m / per_m / 8e12, "traffic": mr.get("group_bytes_per_launch"),
                    "traffic_source": mr.get("source"), "traffic_components": mr.get("group_components"),
                    "traffic_over_alg": (mr["group_bytes_per_launch"] / alg_bytes_m) if mr.get("group_bytes_per_launch") else None,
                    "kernel": "group_gate_kernel (when it runs) + group_desc_kernel + group_settle_kernel + fk_root_wave_kernel<4> + "
                              f"group_lse_kernel (root of block {last_slot}, the Measure slot: (row, context) items x candidate measures); "
                              "alg bytes, launch time and counter traffic all cover these launches",
                    "alg_bytes_per_launch": alg_bytes_m, "alg_bytes_model": which_m, "avg_launch_ms": 1e3 * per_m, "launches_timed": m_launches,
                    "timed_over": f"the {n_half} whole-window sweeps run right after the timed region (config.ms_per_step_1_window_same_loop), "
                                  "HIP events on the library's stream (pclean_set_timed_block)",
                    "gather_model": {"bytes_per_gather": LINE, "alg_bytes_per_launch_64B_sector": sector64_m,
                                     "frac_64B_sector": (sector64_m / per_m / 8e12) if sector64_m else None},
                    "groups": rs_m.n_groups, "items": rs_m.n_items, "kpad": rs_m.kpad, "overflow_items": rs_m.overflow_items,
                    "full_scans": rs_m.full_scans, "fine_blocks": rs_m.fine_blocks, "scored_terms": rs_m.scored_terms,
                    "settled_groups": rs_m.resolved_groups, "lazy_entries": rs_m.lazy_entries,
                    "counters": {k: v for k, v in mr.items() if k not in ("group_components", "group_avg_ms")} or None,
                    "note": "bench.roofline_model, each byte once: descriptors, ids, log marginals, the survivor lists left for the lazy "
                            "draw (12 B per entry) + the smaller of (block-minimum rows + the fine blocks and gathers the kernel "
                            "counted at the 128-byte line) and (every distinct pre-filter byte row once); latency- and "
                            "instruction-bound (DESIGN.md §5), not a bandwidth-bound kernel: frac says how far"}
            b0 = out["roofline"]
            if per_m > per_launch_s:  # the longer launch group is the line's `roofline`
                step_ = b0.pop("step", None)
                out["roofline"] = dict(meas, step=step_, block0_root_group=b0,
                                       chosen="the longest launch group of the step (this one: "
                                              f"{1e3 * per_m:.3f} ms against {1e3 * per_launch_s:.3f} ms of block 0's root group)")
            else:
                b0["measure_root_group"] = meas
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(lw, obs, tr, eng, cfg, args.seed, args.cpu_rows, args.cpu_seconds)
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            if out["cpu_baseline"].get("pruned", {}).get("value"):  # what the GPU buys over ONE core running the same savers
                out["cpu_baseline"]["pruned"]["gpu_over_cpu"] = value / out["cpu_baseline"]["pruned"]["value"]
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

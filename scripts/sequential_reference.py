"""Sequential-schedule REFERENCE runs on the CPU (test infrastructure; writes tests/golden/sequential_f1.json).

The reference updates one row at a time (row_inference.jl:169-185; inference.jl:20-58, 60-81): row i sees every commit
of the rows before it, new latent rows are created and unreferenced ones collected on the spot.  Here that schedule is
the product's own host code (initialize_trace with batches of ONE row, run_inference with batch_rows=1) driven by the
CPU oracle (tests/oracle_engine.OracleEngine): the oracle scores one row against the current tables, the row is committed
(creation / garbage collection in pclean_amd.trace), parameters move every rejuv_frequency rows.  The F1 of these runs is
what the batched GPU schedule is compared with (tests/test_gpu_inference.py, ±tolerance written there).

usage: python scripts/sequential_reference.py [hospital] [flights] [rents] [--seeds 0,1,2]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np

import oracle as orc
from oracle_engine import OracleEngine
from pclean_amd import experiments as ex
from pclean_amd.analysis import evaluate_accuracy
from pclean_amd.engine import InferenceConfig
from pclean_amd.inference import initialize_trace, run_inference
from pclean_amd.model import LoweredModel
from pclean_amd.trace import Trace

OUT = os.environ.get("PCLEAN_SEQ_OUT") or os.path.join(ROOT, "tests", "golden", "sequential_f1.json")  # (PCLEAN_SEQ_OUT: runs in
# parallel write their own file; merge the entries into the golden afterwards)


def program(name, n_rows=None):
    if name.startswith("synth"):  # hospital-shaped synthetic table (pclean_amd.synth), ~100 rows per true hospital
        from pclean_amd.synth import synth_hospital
        n_hosp = 3000 if "k3000" in name else max(n_rows // 100, 1)  # (k3000: many entities, ~10 rows each)
        dirty, clean, _ = synth_hospital(n_rows, n_hosp, 20250926)
        return dirty, clean, ex.hospital_model, ex.hospital_query
    name = name.split("_")[0]
    if name == "hospital":
        dirty, clean = ex.hospital_data()
        mk_model, mk_query = ex.hospital_model, ex.hospital_query
    elif name == "flights":
        dirty, clean = ex.flights_data()
        mk_model, mk_query = ex.flights_model, ex.flights_query
    else:
        dirty, clean = ex.rents_data()
        mk_model, mk_query = ex.rents_model, ex.rents_query
    if n_rows:
        dirty = {c: v[:n_rows] for c, v in dirty.items()}
        clean = {c: v[:n_rows] for c, v in clean.items()}
    return dirty, clean, mk_model, mk_query


def run(name, seed, iters, mh, particles, n_rows=None, shuffle=True, batch_rows=1):
    dirty, clean, mk_model, mk_query = program(name, n_rows)
    if shuffle:
        (dirty, clean), _ = ex.shuffle_rows([dirty, clean], seed)
    m = mk_model(ex.possibilities_of(dirty)) if mk_model is ex.hospital_model else mk_model(dirty)
    lw = LoweredModel(m, mk_query(m), dirty)
    obs = lw.encode_observations(dirty)
    eng = OracleEngine(orc, lw, obs, cached=True)
    tr = Trace(lw, obs.shape[1], seed)
    cfg = InferenceConfig(iters, particles, use_mh_instead_of_pg=mh, rejuv_frequency=rejuv_of(name))
    t0 = time.time()
    initialize_trace(eng, tr, cfg, seed, max_batch=batch_rows)
    t1 = time.time()
    f_init = evaluate_accuracy(lw, tr, dirty, clean)["f1"]
    run_inference(eng, tr, cfg, seed, batch_rows=batch_rows)
    tr.check_consistency()
    acc = evaluate_accuracy(lw, tr, dirty, clean)
    print(f"{name} seed {seed}: init {t1 - t0:.0f}s F1 {f_init:.4f}; + {iters} iterations {time.time() - t1:.0f}s F1 {acc['f1']:.4f} "
          f"(precision {acc['precision']:.4f} recall {acc['recall']:.4f})", {c: t.n_live for c, t in tr.tables.items()}, flush=True)
    return dict(f1=acc["f1"], precision=acc["precision"], recall=acc["recall"], f1_after_init=f_init,
                latent_rows={c: int(t.n_live) for c, t in tr.tables.items()})


def rejuv_of(name):
    """rejuv_frequency of the experiment scripts (hospital: the default 50; flights / rents: 500)"""
    return 50 if name.split("_")[0] in ("hospital", "synth") else 500


CONFIGS = {  # the experiment scripts' configurations (experiments/*/run.jl), rows as the GPU tests use them
    "hospital": dict(iters=3, mh=True, particles=2, n_rows=None),
    "flights": dict(iters=5, mh=True, particles=2, n_rows=None),
    "rents": dict(iters=1, mh=True, particles=2, n_rows=None),
    # BASELINE.json configs[1], [2]: particle Gibbs with 20 particles (hospital: 2 rejuvenation sweeps)
    "hospital_pg20": dict(iters=2, mh=False, particles=20, n_rows=None),
    "rents_pg20": dict(iters=1, mh=False, particles=20, n_rows=None),
    # the headline workload's shape at a size the sequential schedule can finish: 30 000 rows, 300 true hospitals
    "synth_pg20": dict(iters=1, mh=False, particles=20, n_rows=30000),
    # ... and with MANY entities (3 000 true hospitals, ~10 rows each): the table a batched sweep freezes is large
    "synth_k3000_pg20": dict(iters=1, mh=False, particles=20, n_rows=30000),
}

if __name__ == "__main__":
    orc.build()
    names = [a for a in sys.argv[1:] if a in CONFIGS] or list(CONFIGS)
    seeds = [0, 1, 2]
    for a in sys.argv[1:]:
        if a.startswith("--seeds"):
            seeds = [int(x) for x in sys.argv[sys.argv.index(a) + 1].split(",")]
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name in names:
        runs = dict(res.get(name, {}).get("runs", {})) if res.get(name, {}).get("config") == CONFIGS[name] else {}
        for sd in seeds:  # (seeds already recorded for this configuration are kept)
            if str(sd) not in runs:
                runs[str(sd)] = run(name, sd, **CONFIGS[name])
                res[name] = dict(config=CONFIGS[name], schedule="sequential (batch_rows=1), CPU oracle engine, rows shuffled with the seed",
                                 runs=runs)
                res[name]["f1_mean"] = float(np.mean([r["f1"] for r in runs.values()]))
                json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
        res[name] = dict(config=CONFIGS[name], schedule="sequential (batch_rows=1), CPU oracle engine, rows shuffled with the seed",
                         runs=runs)
        res[name]["f1_mean"] = float(np.mean([r["f1"] for r in res[name]["runs"].values()]))
        json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v["f1_mean"] for k, v in res.items()}))

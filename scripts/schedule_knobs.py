"""Which knob of the batched schedule moves F1 away from the sequential reference?  (test infrastructure: CPU oracle engine)

The product's host code (initialize_trace / run_inference) driven by the CPU oracle — bit for bit what the HIP path
computes — with the two batch sizes of the schedule set independently:
  init   max_batch of initialize_trace (1 = the reference's sequential SMC initialisation, inference.jl:20-37)
  batch  batch_rows of run_inference   (1 = the reference's sequential sweeps, inference.jl:60-81; 0 = the product's
         default cut: max(rejuv_frequency, n / 32) rows per sub-batch)
One process per (program, init, batch, seed); results are appended to a JSON-lines file and summarised per cell.

usage: python scripts/schedule_knobs.py rents_pg20 --cells 1:0,1024:1 --seeds 0,1,2,3,4,5,6,7 --procs 6 --out /tmp/knobs.jsonl
       python scripts/schedule_knobs.py --summary /tmp/knobs.jsonl"""
import json
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "scripts")]


def one(task):
    name, init, batch, seed, out = task
    import numpy as np  # noqa: F401
    import oracle as orc
    import sequential_reference as sr
    from oracle_engine import OracleEngine
    from pclean_amd import experiments as ex
    from pclean_amd.analysis import evaluate_accuracy
    from pclean_amd.engine import InferenceConfig
    from pclean_amd.inference import initialize_trace, run_inference
    from pclean_amd.model import LoweredModel
    from pclean_amd.trace import Trace
    c = sr.CONFIGS[name]
    dirty, clean, mk_model, mk_query = sr.program(name, c["n_rows"])
    (dirty, clean), _ = ex.shuffle_rows([dirty, clean], seed)
    m = mk_model(ex.possibilities_of(dirty)) if mk_model is ex.hospital_model else mk_model(dirty)
    lw = LoweredModel(m, mk_query(m), dirty)
    obs = lw.encode_observations(dirty)
    eng = OracleEngine(orc, lw, obs, cached=True)
    tr = Trace(lw, obs.shape[1], seed)
    cfg = InferenceConfig(c["iters"], c["particles"], use_mh_instead_of_pg=c["mh"], rejuv_frequency=sr.rejuv_of(name))
    t0 = time.time()
    initialize_trace(eng, tr, cfg, seed, max_batch=init)
    t1 = time.time()
    f_init = evaluate_accuracy(lw, tr, dirty, clean)["f1"]
    rows_init = {cn: int(t.n_live) for cn, t in tr.tables.items()}
    run_inference(eng, tr, cfg, seed, batch_rows=batch or None)
    tr.check_consistency()
    acc = evaluate_accuracy(lw, tr, dirty, clean)
    rec = dict(name=name, init=init, batch=batch, seed=seed, f1=acc["f1"], precision=acc["precision"], recall=acc["recall"],
               f1_after_init=f_init, latent_rows_after_init=rows_init,
               latent_rows={cn: int(t.n_live) for cn, t in tr.tables.items()}, init_s=t1 - t0, infer_s=time.time() - t1)
    with open(out, "a") as f:
        f.write(json.dumps(rec) + "\n")
    return rec


def summary(path, ref_path=os.path.join(ROOT, "tests", "golden", "sequential_f1.json")):
    import numpy as np
    recs = [json.loads(l) for l in open(path) if l.strip()]
    ref = json.load(open(ref_path))
    cells = {}
    for r in recs:
        cells.setdefault((r["name"], r["init"], r["batch"]), {})[r["seed"]] = r
    for (name, init, batch), by_seed in sorted(cells.items()):
        seeds = sorted(by_seed)
        got = np.array([by_seed[s]["f1"] for s in seeds])
        want = np.array([ref[name]["runs"][str(s)]["f1"] for s in seeds if str(s) in ref[name]["runs"]])
        line = f"{name} init={init} batch={batch or 'default'}: F1 mean {got.mean():.4f} over seeds {seeds}"
        if len(want) == len(got):
            d = got - want
            se = d.std(ddof=1) / np.sqrt(len(d)) if len(d) > 1 else float("nan")
            line += f"; vs sequential {want.mean():.4f}: {100 * d.mean():+.2f} pt (paired s.e. {100 * se:.2f})"
        line += f"; after init {np.mean([by_seed[s]['f1_after_init'] for s in seeds]):.4f}"
        print(line)


if __name__ == "__main__":
    if "--summary" in sys.argv:
        summary(sys.argv[sys.argv.index("--summary") + 1])
        sys.exit(0)
    arg = lambda k, d: sys.argv[sys.argv.index(k) + 1] if k in sys.argv else d
    name = sys.argv[1]
    cells = [tuple(int(x) for x in c.split(":")) for c in arg("--cells", "1:0,1024:1").split(",")]
    seeds = [int(s) for s in arg("--seeds", "0,1,2").split(",")]
    out = arg("--out", "/tmp/knobs.jsonl")
    done = set()
    if os.path.exists(out):
        for l in open(out):
            r = json.loads(l)
            done.add((r["name"], r["init"], r["batch"], r["seed"]))
    import oracle as orc
    orc.build()
    tasks = [(name, i, b, s, out) for s in seeds for (i, b) in cells if (name, i, b, s) not in done]
    with mp.Pool(int(arg("--procs", "4"))) as pool:
        for rec in pool.imap_unordered(one, tasks):
            print(f"{rec['name']} init={rec['init']} batch={rec['batch']} seed={rec['seed']}: F1 {rec['f1']:.4f} "
                  f"(init {rec['init_s']:.0f}s + {rec['infer_s']:.0f}s)", flush=True)
    summary(out)

"""experiments/flights/run.jl on the HIP path."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pclean_amd import experiments as ex
from pclean_amd.analysis import evaluate_accuracy
from pclean_amd.engine import Engine, InferenceConfig
from pclean_amd.inference import initialize_trace, run_inference
from pclean_amd.model import LoweredModel
from pclean_amd.trace import Trace


def main(n_rows=None, particles=2, mh=True, iters=1, seed=0, shuffle=True):
    dirty, clean = ex.flights_data()
    if shuffle:  # random row order for the batched initialisation (experiments.shuffle_rows)
        (dirty, clean), _ = ex.shuffle_rows([dirty, clean], seed)
    if n_rows:
        dirty = {c: v[:n_rows] for c, v in dirty.items()}
        clean = {c: v[:n_rows] for c, v in clean.items()}
    m = ex.flights_model(dirty)
    q = ex.flights_query(m)
    lw = LoweredModel(m, q, dirty)
    obs = lw.encode_observations(dirty)
    eng = Engine(lw, obs)
    tr = Trace(lw, obs.shape[1], seed)
    cfg = InferenceConfig(iters, particles, use_mh_instead_of_pg=mh, rejuv_frequency=500)
    t0 = time.time()
    initialize_trace(eng, tr, cfg, seed, max_batch=512)
    tr.check_consistency()
    t1 = time.time()
    acc0 = evaluate_accuracy(lw, tr, dirty, clean)
    print(f"init {t1 - t0:.2f}s F1 {acc0['f1']:.4f}", {c: (t.n, t.n_live) for c, t in tr.tables.items()}, flush=True)
    run_inference(eng, tr, cfg, seed, verbose=True)
    t2 = time.time()
    acc = evaluate_accuracy(lw, tr, dirty, clean)
    print(f"inference {t2 - t1:.2f}s", {c: (t.n, t.n_live) for c, t in tr.tables.items()})
    print(acc)
    eng.close()
    return acc


if __name__ == "__main__":
    a = sys.argv[1:]
    main(n_rows=int(a[0]) if a and int(a[0]) > 0 else None, particles=int(a[1]) if len(a) > 1 else 2,
         mh=(a[2] == "mh") if len(a) > 2 else True, iters=int(a[3]) if len(a) > 3 else 1,
         shuffle="sorted" not in a)

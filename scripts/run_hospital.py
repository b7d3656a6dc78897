"""experiments/hospital/run.jl on the HIP path: initialize_trace + run_inference! + evaluate_accuracy."""
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


def main(particles=2, mh=True, iters=1, seed=0, shuffle=True):
    dirty, clean = ex.hospital_data()
    if shuffle:  # random row order for the batched initialisation (experiments.shuffle_rows)
        (dirty, clean), _ = ex.shuffle_rows([dirty, clean], seed)
    m = ex.hospital_model(ex.possibilities_of(dirty))
    q = ex.hospital_query(m)
    lw = LoweredModel(m, q, dirty)
    obs = lw.encode_observations(dirty)
    eng = Engine(lw, obs)
    tr = Trace(lw, obs.shape[1], seed)
    cfg = InferenceConfig(iters, particles, use_mh_instead_of_pg=mh)
    t0 = time.time()
    # the table ships sorted by entity: without the shuffle the in-batch merge pass stands in for the reference's
    # row-by-row visibility of new rows (inference.initialize_trace)
    initialize_trace(eng, tr, cfg, seed, merge_rounds=0 if shuffle else 2)
    tr.check_consistency()
    print('after init:', {c: (t.n, t.n_live) for c, t in tr.tables.items()}, flush=True)
    t1 = time.time()
    acc0 = evaluate_accuracy(lw, tr, dirty, clean)
    run_inference(eng, tr, cfg, seed, verbose=True)
    t2 = time.time()
    acc = evaluate_accuracy(lw, tr, dirty, clean)
    print("tables:", {c: (t.n, t.n_live) for c, t in tr.tables.items()})
    print(f"init {t1 - t0:.2f}s  F1 after init {acc0['f1']:.4f}; inference {t2 - t1:.2f}s")
    print(acc)
    eng.close()
    return acc


if __name__ == "__main__":
    a = sys.argv[1:]
    main(particles=int(a[0]) if a else 2, mh=(a[1] == "mh") if len(a) > 1 else True, iters=int(a[2]) if len(a) > 2 else 1,
         shuffle="sorted" not in a)

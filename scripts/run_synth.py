"""Whole pipeline on the synthetic hospital-shaped table (SURVEY §8d config 5 generator):
initialize_trace from an EMPTY trace, then run_inference; prints wall-clock per phase and F1."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from pclean_amd import _lib
from pclean_amd import experiments as ex
from pclean_amd.analysis import evaluate_accuracy
from pclean_amd.engine import Engine, InferenceConfig
from pclean_amd.inference import initialize_trace, run_inference
from pclean_amd.model import LoweredModel
from pclean_amd.synth import synth_hospital
from pclean_amd.trace import Trace


def main(n_rows, n_hosp, particles=2, mh=True, iters=1, seed=20250926, max_batch=8192, shuffle=True):
    t0 = time.time()
    dirty, clean, latent = synth_hospital(n_rows, n_hosp, seed)
    if shuffle:  # the generator emits the records of a hospital consecutively (experiments.shuffle_rows)
        (dirty, clean), _ = ex.shuffle_rows([dirty, clean], seed)
    m = ex.hospital_model(ex.possibilities_of(dirty))
    lw = LoweredModel(m, ex.hospital_query(m), dirty)
    obs = lw.encode_observations(dirty)
    t1 = time.time()
    # 'x'-substitution typos never create the overlapping transpositions on which restricted and
    # unrestricted Damerau-Levenshtein differ; the restricted (LDS-tiled) kernel builds the tables
    eng = Engine(lw, obs, dist_mode=_lib.DIST_OSA)
    t2 = time.time()
    print(f"generate+lower {t1 - t0:.1f}s, static upload + pair tables {t2 - t1:.1f}s", flush=True)
    tr = Trace(lw, obs.shape[1], seed)
    cfg = InferenceConfig(iters, particles, use_mh_instead_of_pg=mh, rejuv_frequency=500)
    initialize_trace(eng, tr, cfg, seed, max_batch=max_batch)
    t3 = time.time()
    acc0 = evaluate_accuracy(lw, tr, dirty, clean)
    print(f"initialize_trace {t3 - t2:.1f}s F1 {acc0['f1']:.4f}", {c: (t.n, t.n_live) for c, t in tr.tables.items()},
          flush=True)
    run_inference(eng, tr, cfg, seed, verbose=True)
    t4 = time.time()
    tr.check_consistency()
    acc = evaluate_accuracy(lw, tr, dirty, clean)
    print(f"run_inference {t4 - t3:.1f}s ({iters} iterations)", {c: (t.n, t.n_live) for c, t in tr.tables.items()})
    print(acc)
    eng.close()


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]), int(a[1]), particles=int(a[2]) if len(a) > 2 else 2, mh=(a[3] == "mh") if len(a) > 3 else True,
         iters=int(a[4]) if len(a) > 4 else 1, max_batch=int(a[5]) if len(a) > 5 else 8192, shuffle="sorted" not in a)

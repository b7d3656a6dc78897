"""Independent END-TO-END reference runs (test infrastructure; writes tests/golden/literal_sequential.json): the literal,
sequential sampler of oracle/literal_inference.py — the reference's schedule on a dict-of-rows trace of strings, scoring
through oracle/literal.py; nothing of pclean_amd's lowering, trace, inference or analysis code — on the programs it
covers (hospital-shaped), with the experiment configurations, seeds and row shuffles of scripts/sequential_reference.py.

flights: oracle/literal_inference_flights.py (slots with noise-free observations only, keyed TimePrior proposals with dummy
values, the MaybeSwap block, learned error probabilities); rents / rents_pg20 (BASELINE.json configs[2]):
oracle/literal_inference_rents.py (own choices enumerated inside the candidate branch, TransformedGaussian observation,
learned means).

usage: python scripts/literal_sequential_reference.py [hospital] [hospital_pg20] [flights] [rents] [rents_pg20] [--seeds 0,1,2]
       [--rows N] [--out FILE]   (--out: runs in parallel write their own file; scripts/merge_goldens.py-style merge by hand)"""
import functools
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
import numpy as np

import literal as L
import literal_inference as LI
from pclean_amd import experiments as ex

OUT = os.path.join(ROOT, "tests", "golden", "literal_sequential.json")
L.add_typos_logpdf = functools.lru_cache(maxsize=None)(L.add_typos_logpdf)  # (add_typos.jl:47,55: the reference memoises too)
L.string_prior_logpdf = functools.lru_cache(maxsize=None)(L.string_prior_logpdf)


class Cfg:
    def __init__(self, iters, particles, mh, rejuv=50):
        self.num_iters, self.num_particles, self.use_mh_instead_of_pg, self.rejuv_frequency = iters, particles, mh, rejuv


CONFIGS = {"hospital": dict(iters=3, mh=True, particles=2), "hospital_pg20": dict(iters=2, mh=False, particles=20),
           "flights": dict(iters=5, mh=True, particles=2), "rents": dict(iters=1, mh=True, particles=2),
           "rents_pg20": dict(iters=1, mh=False, particles=20),
           # the headline workload's shape (scripts/sequential_reference.py: 30 000 rows of pclean_amd.synth, 300 / 3 000 true hospitals)
           "synth_pg20": dict(iters=1, mh=False, particles=20), "synth_k3000_pg20": dict(iters=1, mh=False, particles=20)}
SYNTH_ROWS = 30000


def run(name, seed, iters, mh, particles, n_rows=None, restricted=False):
    flights, rents = name.startswith("flights"), name.startswith("rents")
    if name.startswith("synth"):  # the table of scripts/sequential_reference.py's synth configurations, same generator seed
        from pclean_amd.synth import synth_hospital
        total = n_rows or SYNTH_ROWS
        dirty, clean, _ = synth_hospital(total, 3000 if "k3000" in name else max(total // 100, 1), 20250926)
        n_rows = None
    else:
        dirty, clean = ex.flights_data() if flights else ex.rents_data() if rents else ex.hospital_data()
    if n_rows:
        dirty = {c: v[:n_rows] for c, v in dirty.items()}
        clean = {c: v[:n_rows] for c, v in clean.items()}
    (dirty, clean), _ = ex.shuffle_rows([dirty, clean], seed)
    if flights:  # (rejuv_frequency 500: experiments/flights/run.jl)
        import literal_inference_flights as LF
        m = ex.flights_model(dirty)
        q = ex.flights_query(m)
        s = LF.FlightsLiteralSampler(m, q, dirty, Cfg(iters, particles, mh, rejuv=500), seed)
    elif rents:
        import literal_inference_rents as LR
        m = ex.rents_model(dirty)
        q = ex.rents_query(m)
        s = LR.RentsLiteralSampler(m, q, dirty, Cfg(iters, particles, mh, rejuv=500), seed)
    else:
        m = ex.hospital_model(ex.possibilities_of(dirty))
        q = ex.hospital_query(m)
        s = LI.LiteralSampler(m, q, dirty, Cfg(iters, particles, mh), seed, restricted=restricted)
    t0 = time.time()
    s.initialize()
    f_init = s.accuracy(dirty, clean)["f1"]
    t1 = time.time()
    for _ in range(iters):
        s.sweep()
    s.check()
    acc = s.accuracy(dirty, clean)
    print(f"{name} seed {seed}: init {t1 - t0:.0f}s F1 {f_init:.4f}; + {iters} iterations {time.time() - t1:.0f}s F1 {acc['f1']:.4f} "
          f"(precision {acc['precision']:.4f} recall {acc['recall']:.4f})", s.latent_rows(), flush=True)
    return dict(f1=acc["f1"], precision=acc["precision"], recall=acc["recall"], f1_after_init=f_init, latent_rows=s.latent_rows(),
                errors=acc["errors"], changed=acc["changed"], cleaned=acc["cleaned"])


if __name__ == "__main__":
    names = [a for a in sys.argv[1:] if a in CONFIGS] or list(CONFIGS)
    seeds, rows = [0, 1, 2], None
    for i, a in enumerate(sys.argv):
        if a == "--seeds":
            seeds = [int(x) for x in sys.argv[i + 1].split(",")]
        if a == "--rows":
            rows = int(sys.argv[i + 1])
        if a == "--out":
            OUT = sys.argv[i + 1]
    res = json.load(open(OUT)) if os.path.exists(OUT) and not rows else {}
    for name in names:
        res[name] = dict(config=CONFIGS[name], schedule="sequential, literal sampler on strings (oracle/literal_inference.py), "
                                                        "rows shuffled with the seed, unrestricted Damerau-Levenshtein",
                         runs={str(sd): run(name, sd, n_rows=rows, **CONFIGS[name]) for sd in seeds})
        res[name]["f1_mean"] = float(np.mean([r["f1"] for r in res[name]["runs"].values()]))
        if not rows:
            json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v["f1_mean"] for k, v in res.items()}))

"""Where one full run_inference iteration goes at the headline size (run on the GPU box): the host-phase timers of
inference.py per class (build_evidence / upload / gpu_sweep / commit), then a cProfile of the same iteration.
usage: python scripts/profile_iteration.py [--rows N] [--hospitals H] [--no-cprofile]"""
import argparse
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--rows", type=int, default=1_000_000)
ap.add_argument("--hospitals", type=int, default=10_000)
ap.add_argument("--no-cprofile", action="store_true")
args = ap.parse_args()

import bench
from pclean_amd import _lib
from pclean_amd import inference as inf
from pclean_amd.engine import Engine, InferenceConfig
from pclean_amd.trace import Trace

seed = 20250926
dirty, clean, lw, obs = bench.build_workload(args.rows, args.hospitals, seed)
eng = Engine(lw, obs, dist_mode=_lib.DIST_OSA)
cfg = InferenceConfig(1, 20)
tr = Trace(lw, args.rows, seed)
inf.initialize_trace(eng, tr, cfg, seed, max_batch=32768)
inf.run_inference(eng, tr, cfg, seed)
inf.TIMERS.clear()
t0 = time.perf_counter()
inf.run_inference(eng, tr, cfg, seed + 1)
tot = time.perf_counter() - t0
print(f"full iteration: {1e3 * tot:.1f} ms; timers (ms): "
      + ", ".join(f"{k} {1e3 * v:.1f}" for k, v in sorted(inf.TIMERS.items(), key=lambda kv: -kv[1])))
print("live rows:", {c: int(t.n_live) for c, t in tr.tables.items()})
# per class: host wall-clock of the sweep and the library's HIP-event phases
for cname in lw.model.class_order:
    if cname not in lw.latent_plans:
        continue
    eng.hip.set_profiling(True)
    t0 = time.perf_counter()
    inf.latent_sweep(eng, tr, cname, cfg, seed + 3, 0)
    dt = time.perf_counter() - t0
    ph = eng.hip.get_profile()
    eng.hip.set_profiling(False)
    print(f"{cname}: {1e3 * dt:.1f} ms host; device phases (ms/intervals): "
          + ", ".join(f"{k} {v[0]:.1f}/{v[1]}" for k, v in sorted(ph.items(), key=lambda kv: -kv[1][0]) if v[0] > 0.05))
# the observed class's sweep as an iteration runs it: right after the latent classes' sweeps moved the tables
eng.hip.set_profiling(True)
t0 = time.perf_counter()
inf.observed_sweep(eng, tr, cfg, seed + 3, 0)
dt = time.perf_counter() - t0
ph = eng.hip.get_profile()
eng.hip.set_profiling(False)
print(f"{lw.query.cls} (after the latent sweeps): {1e3 * dt:.1f} ms host; device phases (ms/intervals): "
      + ", ".join(f"{k} {v[0]:.2f}/{v[1]}" for k, v in sorted(ph.items(), key=lambda kv: -kv[1][0]) if v[0] > 0.02))
t0 = time.perf_counter()
inf.observed_sweep(eng, tr, cfg, seed + 4, 0)
print(f"{lw.query.cls} (again, nothing moved in between): {1e3 * (time.perf_counter() - t0):.1f} ms host")
if not args.no_cprofile:
    pr = cProfile.Profile()
    pr.enable()
    inf.run_inference(eng, tr, cfg, seed + 2)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(40)
eng.close()

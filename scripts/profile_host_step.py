"""Where the HOST side of one observed-class sweep goes at the headline size (run on the GPU box):
cProfile over a few steps of bench.py's timed region (inference.observed_sweep: upload, HIP sweep, exchange, commit),
printed by internal time.  usage: python scripts/profile_host_step.py [--rows N] [--steps K]"""
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
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()

import bench
from pclean_amd import _lib
from pclean_amd import inference as inf
from pclean_amd.engine import Engine, InferenceConfig
from pclean_amd.parallel import Comm
from pclean_amd.trace import Trace

seed = 20250926
dirty, clean, lw, obs = bench.build_workload(args.rows, args.hospitals, seed)
eng = Engine(lw, obs, dist_mode=_lib.DIST_OSA)
comm = Comm()
cfg = InferenceConfig(1, 20)
tr = Trace(lw, args.rows, seed)
inf.initialize_trace(eng, tr, cfg, seed, max_batch=32768)
inf.run_inference(eng, tr, cfg, seed)
for i in range(3):
    inf.observed_sweep(eng, tr, cfg, seed, 1 + i, comm)
inf.TIMERS.clear()
t0 = time.perf_counter()
for i in range(args.steps):
    inf.observed_sweep(eng, tr, cfg, seed, 10 + i, comm)
print(f"plain: {1e3 * (time.perf_counter() - t0) / args.steps:.2f} ms per step; phases (ms per step): "
      + ", ".join(f"{k} {1e3 * v / args.steps:.2f}" for k, v in sorted(inf.TIMERS.items(), key=lambda kv: -kv[1])))
pr = cProfile.Profile()
pr.enable()
for i in range(args.steps):
    inf.observed_sweep(eng, tr, cfg, seed, 100 + i, comm)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
eng.close()

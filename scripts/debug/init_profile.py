"""cProfile of initialize_trace at the headline size (run on the GPU box)."""
import cProfile
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from pclean_amd import _lib
from pclean_amd import inference as inf
from pclean_amd.engine import Engine, InferenceConfig
from pclean_amd.trace import Trace

seed = 20250926
dirty, clean, lw, obs = bench.build_workload(1_000_000, 10_000, seed)
eng = Engine(lw, obs, dist_mode=_lib.DIST_OSA)
cfg = InferenceConfig(1, 20)
for rep in range(2):
    tr = Trace(lw, 1_000_000, seed)
    inf.TIMERS.clear()
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    if rep:
        pr.enable()
    inf.initialize_trace(eng, tr, cfg, seed, max_batch=32768)
    if rep:
        pr.disable()
    print(f"initialize_trace: {time.perf_counter() - t0:.3f} s; timers (ms): "
          + ", ".join(f"{k} {1e3 * v:.0f}" for k, v in sorted(inf.TIMERS.items(), key=lambda kv: -kv[1])[:12]), flush=True)
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
eng.close()

"""Full run_inference iterations at the headline size, one line per iteration (total, the largest host timers): run
under different PCLEAN_* environment switches to compare builds / code paths on the same sequence of iterations.
usage: python scripts/debug/iter_ab.py [n_iterations=4]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from pclean_amd import _lib
from pclean_amd import inference as inf
from pclean_amd.engine import Engine, InferenceConfig
from pclean_amd.trace import Trace

n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 4
seed = 20250926
dirty, clean, lw, obs = bench.build_workload(1_000_000, 10_000, seed)
eng = Engine(lw, obs, dist_mode=_lib.DIST_OSA)
cfg = InferenceConfig(1, 20)
tr = Trace(lw, 1_000_000, seed)
inf.initialize_trace(eng, tr, cfg, seed, max_batch=32768)
tag = " ".join(f"{k}={v}" for k, v in sorted(os.environ.items()) if k.startswith("PCLEAN_")) or "default"
tot = []
for it in range(n_it):
    inf.TIMERS.clear()
    t0 = time.perf_counter()
    inf.run_inference(eng, tr, cfg, seed + it)
    dt = 1e3 * (time.perf_counter() - t0)
    tot.append(dt)
    top = sorted(inf.TIMERS.items(), key=lambda kv: -kv[1])[:7]
    print(f"[{tag}] iteration {it}: {dt:.0f} ms; " + ", ".join(f"{k} {1e3 * v:.0f}" for k, v in top), flush=True)
print(f"[{tag}] last iteration, every timer (ms): " + ", ".join(f"{k} {1e3 * v:.1f}" for k, v in sorted(inf.TIMERS.items(), key=lambda kv: -kv[1])))
print(f"[{tag}] mean of iterations 1..: {sum(tot[1:]) / max(len(tot) - 1, 1):.0f} ms; rows "
      + str({c: int(t.n_live) for c, t in tr.tables.items()}), flush=True)
eng.close()

"""Times the sweep kernels of the bench workload under different PCLEAN_FAST_T settings (one process,
tables built once).  Usage: python scripts/tune_root.py [rows] [hospitals] T1 T2 ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
from pclean_amd import _lib
from pclean_amd.engine import Engine, InferenceConfig

rows, hosp = int(sys.argv[1]), int(sys.argv[2])
dirty, clean, lw, obs, tr = bench.build_workload(rows, hosp, 20250926)
eng = Engine(lw, obs, dist_mode=_lib.DIST_OSA)
cfg = InferenceConfig(4, 20)
eng.upload_trace(tr)
eng.sweep(tr, cfg, 1, 0)
for T in sys.argv[3:]:
    if T != "default":
        os.environ["PCLEAN_FAST_T"] = T
    else:
        os.environ.pop("PCLEAN_FAST_T", None)
    best = None
    for rep in range(2):
        choice, chosen, logml, new_rows = eng.sweep(tr, cfg, 1, 1)
        tm = eng.hip.get_timing()
        best = (tm.total_ms, tm.hot_kernel_ms, tm.reserved) if best is None or tm.total_ms < best[0] else best
    print(f"T={T}: device {best[0]:.1f} ms, root kernel {best[1]:.1f} ms, fallbacks {best[2]}", flush=True)
eng.close()

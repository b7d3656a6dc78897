"""Per-table build times of the 1M-row workload's AddTypos pair tables (unrestricted DL): the linear-space kernel
(dl_seg_kernel) and, for tables of at most --wave-cells DP cells, the LDS-matrix kernel it replaces (PCLEAN_DL_KERNEL=wave)
with a bit-for-bit comparison of the two tables.  usage: python scripts/debug/dl_build_time.py [--rows N] [--wave-cells X]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from pclean_amd import _lib
from pclean_amd._lib import HipContext

rows = int(sys.argv[sys.argv.index("--rows") + 1]) if "--rows" in sys.argv else 1_000_000
wave_cells = float(sys.argv[sys.argv.index("--wave-cells") + 1]) if "--wave-cells" in sys.argv else 3e11
dirty, clean, lw, obs = bench.build_workload(rows, max(rows // 100, 1), 20250926)
hip = HipContext(0)
sym, off, lm, _ = lw.pool.arrays()
hip.load_strings(sym, off)
tot = {"seg": 0.0, "osa": 0.0}
tot_cells = 0
for key, (pid, odom, ldom) in lw.pair_id.items():
    oi, li = odom.id_array(), ldom.id_array()
    cells = int(lw.pool.lens[oi].astype(np.int64).sum()) * int(lw.pool.lens[li].astype(np.int64).sum())
    tot_cells += cells
    line = f"{key[0]:18s} {len(oi):6d} x {len(li):6d} pairs, lens obs<= {lw.pool.lens[oi].max():3d} lat<= {lw.pool.lens[li].max():3d} mean {lw.pool.lens[li].mean():5.1f}, {cells / 1e9:8.1f} G cells:"
    for kern in ("osa", "seg"):
        os.environ["PCLEAN_DL_KERNEL"] = kern if kern != "osa" else "seg"
        t0 = time.perf_counter()
        hip.build_pair_table(pid, oi, li, _lib.DIST_OSA if kern == "osa" else _lib.DIST_DL)
        dt = time.perf_counter() - t0
        tot[kern] += dt
        line += f"  {kern} {dt:6.3f}s" + (f" ({cells / dt / 1e9:6.0f} G cells/s)" if kern == "seg" else "")
    if cells <= wave_cells:
        n_o = min(len(oi), 4000)
        got = hip.get_pair_rows(pid, np.arange(n_o, dtype=np.int32), len(li))
        os.environ["PCLEAN_DL_KERNEL"] = "wave"
        t0 = time.perf_counter()
        hip.build_pair_table(pid, oi, li, _lib.DIST_DL)
        dt = time.perf_counter() - t0
        old = hip.get_pair_rows(pid, np.arange(n_o, dtype=np.int32), len(li))
        line += f"  wave {dt:6.3f}s, seg == wave on {n_o} rows: {bool(np.array_equal(got, old))}"
        os.environ["PCLEAN_DL_KERNEL"] = "seg"
        hip.build_pair_table(pid, oi, li, _lib.DIST_DL)
    print(line, flush=True)
print(f"total: seg {tot['seg']:.2f}s  osa {tot['osa']:.2f}s  {tot_cells / 1e12:.2f} T cells -> {tot_cells / tot['seg'] / 1e9:.0f} G cells/s")
hip.close()

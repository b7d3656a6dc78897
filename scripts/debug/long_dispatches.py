"""the longest single dispatches of a rocprofv3 kernel-trace database (rocpd): usage: python long_dispatches.py <results.db> [top]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = list(db.execute("select name, grid_x, workgroup_x, end - start from kernels order by end - start desc limit ?", (top,)))
for name, g, w, d in rows:
    print(f"{d / 1e3:10.1f} us  grid {g:>9} wg {w:>5}  {name[:110]}")

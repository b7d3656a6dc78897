"""Kernel time of everything BETWEEN the last two observed-class sweeps of a rocprofv3 rocpd kernel trace, i.e. the
latent-class sweeps (and parameter moves) of the last full run_inference iteration: span, busy time, dispatch count,
idle gaps and the kernels by total time.
usage: python profiles/iteration_window.py <results.db> [top]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    rows = list(db.execute("select name, grid_x, workgroup_x, start, end from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if "final_choice_kernel" in r[0]]
    a, b = idx[-2], idx[-1]
    win = rows[a + 1:b]

    def short(n):
        n = re.sub(r"void rocprim::ROCPRIM_\d+_NS::detail::", "rp::", n)
        n = re.sub(r"trampoline_kernel<rocprim::ROCPRIM_\d+_NS::detail::", "", n)
        return n[:72]

    span = (win[-1][4] - win[0][3]) / 1e3
    busy = sum(r[4] - r[3] for r in win) / 1e3
    gaps = [(win[i + 1][3] - win[i][4]) / 1e3 for i in range(len(win) - 1)]
    print(f"window: {len(win)} dispatches, span {span / 1e3:.1f} ms, busy {busy / 1e3:.1f} ms; gaps > 50 us: "
          f"{sum(1 for g in gaps if g > 50)} totalling {sum(g for g in gaps if g > 50) / 1e3:.1f} ms; "
          f"gaps 10-50 us: {sum(1 for g in gaps if 10 < g <= 50)} totalling {sum(g for g in gaps if 10 < g <= 50) / 1e3:.1f} ms")
    agg = {}
    for r in win:
        k = short(r[0])
        e = agg.setdefault(k, [0, 0.0, 0])
        e[0] += 1
        e[1] += (r[4] - r[3]) / 1e3
        e[2] = max(e[2], r[1])
    for k, (n, t, g) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"{t / 1e3:9.2f} ms  x{n:6d}  avg {t / n:8.1f} us  max grid {g:9d}  {k}")


main()

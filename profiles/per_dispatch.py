"""Per-dispatch counter values of the kernels matching a substring, from a rocprofv3 --pmc rocpd database:
    python profiles/per_dispatch.py <results.db> <kernel substring> [last_n=8]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2]
    last = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "info_kernel_symbol" in t][0]
    pe = [t for t in tabs if "pmc_event" in t][0]
    pi = [t for t in tabs if "info_pmc" in t][0]
    rows = cur.execute(f"select s.display_name, d.grid_size_x, d.start, d.end, d.event_id from {kd} d join {ks} s on d.kernel_id = s.id "
                       f"order by d.start").fetchall()
    val = {}
    for ev, name, v in cur.execute(f"select e.event_id, p.name, sum(e.value) from {pe} e join {pi} p on e.pmc_id = p.id group by e.event_id, p.name"):
        val.setdefault(ev, {})[name] = v
    sel = [r for r in rows if pat in r[0]]
    for name, grid, st, en, ev in sel[-last:]:
        print(f"{name[:50]:50s} grid {grid:9d} dur {(en - st) / 1e3:8.1f} us  " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(val.get(ev, {}).items())))


main()

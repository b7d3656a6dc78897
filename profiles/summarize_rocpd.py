"""Summarise a rocprofv3 rocpd sqlite database (kernel trace) as a --stats style table.
usage: python profiles/summarize_rocpd.py <results.db> [out.txt]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [d[1] for d in cur.execute("pragma table_info(kernels)")]
    grid = "grid_x" if "grid_x" in cols else None
    rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e6, min(end-start)/1e6, "
                            "max(end-start)/1e6 from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    lines = [f"{'kernel':58s} {'calls':>6s} {'total_ms':>10s} {'avg_ms':>10s} {'min_ms':>9s} {'max_ms':>9s} {'pct':>6s}"]
    for r in rows:
        lines.append(f"{r[0][:58]:58s} {r[1]:6d} {r[2]:10.2f} {r[3]:10.3f} {r[4]:9.3f} {r[5]:9.3f} {100 * r[2] / tot:6.1f}")
    lines.append(f"{'TOTAL':58s} {sum(r[1] for r in rows):6d} {tot:10.2f}")
    if grid:
        lines.append("")
        lines.append("largest dispatches:")
        for r in cur.execute(f"select name, {grid}, workgroup_x, lds_size, vgpr_count, (end-start)/1e6 from kernels "
                             "order by (end-start) desc limit 12"):
            lines.append(f"  {r[0][:50]:50s} grid={r[1]} wg={r[2]} lds={r[3]} vgpr={r[4]} {r[5]:.3f} ms")
    out = "\n".join(lines)
    print(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")


if __name__ == "__main__":
    main()

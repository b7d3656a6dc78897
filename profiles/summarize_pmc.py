"""Per-dispatch PMC values of one kernel from a rocprofv3 --pmc run (rocpd sqlite output).

    python profiles/summarize_pmc.py <results.db> <kernel substring> [--json out.json --counter FETCH_SIZE ...]

Prints, for every dispatch of the matching kernel, the summed counter value and the duration, and
optionally merges {counter: value of the LARGEST-grid dispatch} into a json file (bench.py reads
profiles/hbm_traffic.json for `roofline.traffic`).  FETCH_SIZE / WRITE_SIZE are reported in KiB by
rocprofv3; on gfx950 FETCH_SIZE counts wide coalesced reads at half their size
(/opt/skills/guides/MI355X_MICROARCH.md, "HBM"), so the corrected figure doubles it.
"""
import json
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2]
    out_json = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "info_kernel_symbol" in t][0]
    pe = [t for t in tabs if "pmc_event" in t][0]
    pi = [t for t in tabs if "info_pmc" in t][0]
    rows = cur.execute(
        f"select d.id, s.display_name, d.grid_size_x, d.group_segment_size, d.end - d.start, d.event_id "
        f"from {kd} d join {ks} s on d.kernel_id = s.id where s.display_name like ? order by d.start",
        (f"%{pat}%",)).fetchall()
    result = {}
    for did, name, grid, lds, dur, ev in rows:
        vals = cur.execute(f"select p.name, sum(e.value) from {pe} e join {pi} p on e.pmc_id = p.id "
                           f"where e.event_id = ? group by p.name", (ev,)).fetchall()
        for cname, v in vals:
            print(f"{cname} {name[:40]} grid={grid} lds={lds} value={v:.6g} dur={dur / 1e6:.3f} ms")
            key = (grid, cname)
            result[key] = (v, dur / 1e6, name)
    if out_json and result:
        gmax = max(g for g, _ in result)
        try:
            data = json.load(open(out_json))
        except Exception:
            data = {}
        for (g, cname), (v, dur, name) in result.items():
            if g == gmax:
                data[cname] = {"value": v, "kernel": name[:80], "grid": g, "duration_ms": dur}
        json.dump(data, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()

"""HBM traffic of ONE whole observed-class sweep (every kernel between the last two final_choice_kernel dispatches)
from two rocprofv3 --pmc passes over ALL kernels (FETCH_SIZE in one database, WRITE_SIZE in the other):

    python profiles/step_traffic.py <fetch.db> <write.db> [--json profiles/hbm_traffic.json]

HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 correction, /opt/skills/guides/MI355X_MICROARCH.md "HBM").
Prints the kernels by bytes and the step total; --json merges {"step": {...}} into the file bench.py echoes as
roofline.step (whole-step bytes / device time of the step)."""
import json
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"void rocprim::ROCPRIM_\d+_NS::detail::", "rp::", n)
    n = re.sub(r"trampoline_kernel<rocprim::ROCPRIM_\d+_NS::detail::", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:60]


def window(path, counter):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "info_kernel_symbol" in t][0]
    pe = [t for t in tabs if "pmc_event" in t][0]
    pi = [t for t in tabs if "info_pmc" in t][0]
    rows = cur.execute(f"select s.display_name, d.start, d.end, d.event_id from {kd} d join {ks} s on d.kernel_id = s.id "
                       f"order by d.start").fetchall()
    val = {}
    for ev, v in cur.execute(f"select e.event_id, sum(e.value) from {pe} e join {pi} p on e.pmc_id = p.id where p.name = ? "
                             f"group by e.event_id", (counter,)):
        val[ev] = v
    idx = [i for i, r in enumerate(rows) if "final_choice_kernel" in r[0]]
    a, b = idx[-2], idx[-1]
    per = {}
    busy = 0.0
    for name, st, en, ev in rows[a + 1:b + 1]:
        e = per.setdefault(short(name), [0, 0.0, 0.0])
        e[0] += 1
        e[1] += val.get(ev, 0.0)
        e[2] += (en - st) / 1e3
        busy += (en - st) / 1e3
    return per, busy, b - a


def main():
    args = sys.argv[1:]
    out_json = args[args.index("--json") + 1] if "--json" in args else None
    dbs = [a for a in args if a.endswith(".db")]
    f, busy_f, n_f = window(dbs[0], "FETCH_SIZE")
    w, busy_w, n_w = window(dbs[1], "WRITE_SIZE")
    names = sorted(set(f) | set(w), key=lambda k: -(2 * f.get(k, [0, 0, 0])[1] + w.get(k, [0, 0, 0])[1]))
    tot = 0.0
    print(f"one sweep: {n_f} dispatches (fetch pass), {n_w} (write pass); kernel time under the counters {busy_f / 1e3:.2f} ms")
    print(f"{'kernel':62s} {'disp':>5s} {'fetch MB':>10s} {'write MB':>10s} {'HBM MB':>10s}")
    for k in names:
        fe, wr = f.get(k, [0, 0.0, 0.0]), w.get(k, [0, 0.0, 0.0])
        b = (2 * fe[1] + wr[1]) * 1024
        tot += b
        if b > 2e6:
            print(f"{k:62s} {fe[0]:5d} {2 * fe[1] * 1024 / 1e6:10.1f} {wr[1] * 1024 / 1e6:10.1f} {b / 1e6:10.1f}")
    print(f"whole sweep: {tot / 1e9:.3f} GB of HBM traffic")
    if out_json:
        try:
            data = json.load(open(out_json))
        except Exception:
            data = {}
        by_kernel = {k: (2 * f.get(k, [0, 0.0, 0.0])[1] + w.get(k, [0, 0.0, 0.0])[1]) * 1024 for k in names[:12]}
        data["step"] = dict(hbm_bytes_per_step=tot, dispatches=n_f, rows=1000000, hospitals=10000, particles=20, by_kernel=by_kernel,
                            source="profiles/step_traffic.py over two rocprofv3 --pmc passes of bench.py over all kernels "
                                   "(FETCH_SIZE; WRITE_SIZE), one sweep = the dispatches between the last two final_choice_kernel")
        json.dump(data, open(out_json, "w"), indent=1)


main()

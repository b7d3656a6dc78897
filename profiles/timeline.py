"""Per-dispatch timeline of ONE observed-class sweep from a rocprofv3 rocpd database (kernel trace).
usage: python profiles/timeline.py <results.db> [min_us] [which]   (which: index of the sweep counted from the end, default 1)"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
    which = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rows = list(db.execute("select name, grid_x, workgroup_x, lds_size, vgpr_count, start, end from kernels order by start"))
    idx = [i for i, r in enumerate(rows) if "final_choice_kernel" in r[0]]
    TAIL = ("finalize_block", "select", "Select", "gather_moved", "gather_new_rows", "locals_tail", "partition", "pcc_commit_kernel",
            "pcc_refresh_kernel", "init_lookback_scan_state", "chosen_new_kernel", "fillBufferAligned", "copyBuffer")

    def tail_end(i):  # last dispatch of the sweep whose final choice is dispatch i (outputs, device commit)
        e = i
        while e + 1 < len(rows) and any(k in rows[e + 1][0] for k in TAIL) and rows[e + 1][5] - rows[e][6] < 400_000:
            e += 1
        return e

    b = idx[-which]
    a = idx[-which - 1] if len(idx) > which else 0
    s = tail_end(a) + 1 if len(idx) > which else 0
    e = tail_end(b)
    t0 = rows[s][5]

    def short(n):
        n = re.sub(r"void rocprim::ROCPRIM_\d+_NS::detail::", "rp::", n)
        n = re.sub(r"trampoline_kernel<rocprim::ROCPRIM_\d+_NS::detail::", "", n)
        return n[:64]

    prev_end = None
    tot = 0.0
    agg = {}
    for r in rows[s:e + 1]:
        d = (r[6] - r[5]) / 1e3
        gap = (r[5] - prev_end) / 1e3 if prev_end else 0.0
        prev_end = r[6]
        tot += d
        key = short(r[0])[:40]
        agg[key] = agg.get(key, [0, 0.0])
        agg[key][0] += 1
        agg[key][1] += d
        if d >= min_us or gap >= min_us:
            print(f"{(r[5] - t0) / 1e3:9.1f}us dur {d:8.1f} gap {gap:7.1f} grid={r[1]:>9} wg={r[2]:>4} lds={r[3]:>6} v={r[4]:>3} {short(r[0])}")
    print(f"span {(rows[e][6] - t0) / 1e3:.1f} us, kernels {e + 1 - s}, busy {tot:.1f} us")
    print("-- by kernel:")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"  {v[1]:9.1f} us  x{v[0]:4d}  {k}")


if __name__ == "__main__":
    main()

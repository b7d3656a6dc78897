"""One latent-class sub-batch of a rocprofv3 rocpd kernel trace: every dispatch between the n-th latent_choice_kernel
launch whose grid matches a sub-batch of about `rows` latent rows and the next latent_choice_kernel, with its queue —
shows what overlaps on the side streams of pclean_sweep_latent and where the device waits for the host.
usage: python profiles/latent_window.py <results.db> <rows_lo> <rows_hi> [which=3]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    lo, hi = int(sys.argv[2]), int(sys.argv[3])
    which = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = "name, grid_x, workgroup_x, start, end" + (", " + qcol if qcol else "")
    rows = list(db.execute(f"select {sel} from kernels order by start"))
    heads = [i for i, r in enumerate(rows) if "latent_choice_kernel" in r[0]]
    match = [i for i in heads if lo <= r_grid(rows[i]) <= hi]
    if not match:
        print("no latent_choice_kernel launch with a grid in range; grids seen:", sorted({rows[i][1] for i in heads})[:40])
        return
    a = match[min(which, len(match) - 1)]
    nxt = [i for i in heads if i > a]
    b = nxt[0] if nxt else len(rows)
    win = rows[a:b]
    t0 = win[0][3]
    last_end = t0
    busy_union = 0.0
    cur_end = t0
    for r in win:
        name = re.sub(r"void rocprim::ROCPRIM_\d+_NS::detail::", "rp::", r[0])[:70]
        gap = (r[3] - last_end) / 1e3
        print(f"{(r[3] - t0) / 1e3:9.1f}us dur {(r[4] - r[3]) / 1e3:8.1f} gap {gap:8.1f} q={r[5] if qcol else '-'} grid={r[1]:8d} wg={r[2]:4d} {name}")
        last_end = max(last_end, r[4])
        if r[3] > cur_end:
            cur_end = r[3]
        if r[4] > cur_end:
            busy_union += (r[4] - cur_end) / 1e3
            cur_end = r[4]
    span = (max(r[4] for r in win) - t0) / 1e3
    print(f"window: {len(win)} dispatches, span {span:.1f} us, device busy (union) {busy_union:.1f} us, sum of durations "
          f"{sum(r[4] - r[3] for r in win) / 1e3:.1f} us")
    if b < len(rows):
        print(f"next latent call starts {(rows[b][3] - t0) / 1e3:.1f} us after this one")


def r_grid(r):
    return r[1]


main()

"""Counter totals per KERNEL (top kernels by time) from rocprofv3 --pmc runs (rocpd sqlite output).

    python profiles/summarize_pmc_top.py <results.db> [<results.db> ...] [--top N] [--json profiles/hbm_traffic.json]

For every database (one --pmc pass each — FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950) prints, per kernel
name, the number of dispatches, the average duration and the average value of each counter per dispatch.  FETCH_SIZE /
WRITE_SIZE are KiB; on gfx950 FETCH_SIZE counts wide coalesced reads at half their size
(/opt/skills/guides/MI355X_MICROARCH.md "HBM"), so HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (x 1024).
--json merges {kernel, bytes_per_launch, ...} of the dominant sweep kernel (largest fk_root_wave_kernel) into the file
bench.py echoes as roofline.traffic."""
import json
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"void rocprim::ROCPRIM_\d+_NS::detail::", "rp::", n)
    n = re.sub(r"trampoline_kernel<rocprim::ROCPRIM_\d+_NS::detail::", "", n)
    n = re.sub(r"\(.*", "", n)
    return n[:60]


def main():
    args = [a for a in sys.argv[1:]]
    top = int(args[args.index("--top") + 1]) if "--top" in args else 14
    out_json = args[args.index("--json") + 1] if "--json" in args else None
    source = args[args.index("--source") + 1] if "--source" in args else "profiles/r02_pmc_top_kernels.txt"
    dbs = [a for a in args if a.endswith(".db")]
    merged = {}
    bygrid = {}  # (kernel, grid) -> the same totals: the group kernels run once per block, told apart by their grids
    for path in dbs:
        db = sqlite3.connect(path)
        cur = db.cursor()
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
        kd = [t for t in tabs if "kernel_dispatch" in t][0]
        ks = [t for t in tabs if "info_kernel_symbol" in t][0]
        pe = [t for t in tabs if "pmc_event" in t][0]
        pi = [t for t in tabs if "info_pmc" in t][0]
        rows = cur.execute(f"select s.display_name, d.grid_size_x, d.end - d.start, d.event_id from {kd} d join {ks} s "
                           f"on d.kernel_id = s.id").fetchall()
        pmc = {}
        for ev, cname, v in cur.execute(f"select e.event_id, p.name, sum(e.value) from {pe} e join {pi} p on e.pmc_id = p.id "
                                        f"group by e.event_id, p.name"):
            pmc.setdefault(ev, {})[cname] = v
        longest = {}
        for name, grid, dur, ev in rows:
            longest[short(name)] = max(longest.get(short(name), 0), dur)
        for name, grid, dur, ev in rows:
            k = short(name)
            # the sweep kernels run on lists of every size (initialisation batches, nested slots, ...): the launches
            # within a factor 2 of the kernel's longest launch are its full-size launches, the rest is "[small]"
            if any(t in k for t in ("fk_root_wave_kernel", "enum_node", "ev_leaf_wave", "group_desc", "group_settle", "group_lse", "group_gate", "gate_new", "particle_update", "hg_insert", "hg_fill", "overflow_lds")):
                k += " [full-size]" if dur * 2 >= longest[k] else " [small]"
            for m in (merged.setdefault(k, dict(n=0, dur=0.0, counters={})),
                      bygrid.setdefault((short(name), grid), dict(n=0, dur=0.0, counters={}))):
                m["n"] += 1
                m["dur"] += dur / 1e6
                for cname, v in pmc.get(ev, {}).items():
                    c = m["counters"].setdefault(cname, [0, 0.0])
                    c[0] += 1
                    c[1] += v
    names = sorted(merged, key=lambda k: -merged[k]["dur"])[:top]
    names += [k for k in merged if "fk_root_wave_kernel" in k and "full-size" in k and k not in names]  # the dominant sweep kernels, always
    print(f"{'kernel':72s} {'disp':>6s} {'avg_ms':>9s}  counters (average per dispatch)")
    for k in names:
        m = merged[k]
        cs = "  ".join(f"{c}={v[1] / max(v[0], 1):.4g}" for c, v in sorted(m["counters"].items()))
        print(f"{k:72s} {m['n']:6d} {m['dur'] / m['n']:9.3f}  {cs}")
        c = m["counters"]
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            b = (2 * c["FETCH_SIZE"][1] / c["FETCH_SIZE"][0] + c["WRITE_SIZE"][1] / c["WRITE_SIZE"][0]) * 1024
            dur = m["dur"] / m["n"]
            print(f"{'':72s}        -> HBM bytes per launch {b / 1e9:.3f} GB, {b / 1e9 / (dur / 1e3) / 1e3:.2f} TB/s of 8 TB/s")
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"][1] > 0:
            wc = c["SQ_WAVE_CYCLES"][1]
            parts = [f"{n2}/SQ_WAVE_CYCLES={c[n2][1] / wc:.2f}" for n2 in ("SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY") if n2 in c]
            print(f"{'':72s}        -> " + ", ".join(parts))
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"][1] + c["TCC_MISS_sum"][1] > 0:
            print(f"{'':72s}        -> L2 hit rate TCC_HIT/(TCC_HIT+TCC_MISS) = "
                  f"{c['TCC_HIT_sum'][1] / (c['TCC_HIT_sum'][1] + c['TCC_MISS_sum'][1]):.3f}")
    if out_json:
        # the dominant kernel GROUP of a sweep: the launches of block 0's root that bench.py's byte model and HIP-event time
        # cover — the full-size group_desc_kernel, fk_root_wave_kernel<12> and group_lse_kernel launches
        def hbm(k):
            c = merged[k]["counters"]
            if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                return None
            return (2 * c["FETCH_SIZE"][1] / c["FETCH_SIZE"][0] + c["WRITE_SIZE"][1] / c["WRITE_SIZE"][0]) * 1024

        keys = [next((k for k in merged if pat in k and "full-size" in k), None)
                for pat in ("group_desc_kernel", "fk_root_wave_kernel<12", "group_lse_kernel", "group_settle_kernel", "group_gate_kernel")]
        if keys[4] is None:  # (builds before round 5's gate on the groups)
            keys = keys[:4]
        if keys[3] is None:  # (builds before the settle kernel / PCLEAN_NO_SETTLE)
            keys = keys[:3]
        if all(keys) and all(hbm(k) is not None for k in keys):
            parts = {k: hbm(k) for k in keys}
            c = merged[keys[1]]["counters"]
            try:
                data = json.load(open(out_json))
            except Exception:
                data = {}
            data.update(kernel=" + ".join(k.split(" [")[0] for k in keys), pair_bytes_per_launch=sum(parts.values()),
                        pair_components=parts, bytes_per_launch=parts[keys[1]], rows=1000000, hospitals=10000, particles=20,
                        fetch_kib=c["FETCH_SIZE"][1] / c["FETCH_SIZE"][0], write_kib=c["WRITE_SIZE"][1] / c["WRITE_SIZE"][0],
                        avg_ms={k: merged[k]["dur"] / merged[k]["n"] for k in keys},
                        source=source + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py, "
                               "2 x FETCH_SIZE + WRITE_SIZE, KiB; full-size launches of the kernels of block 0's root)")
            # the Measure slot's root scan (the 4-term instance): the longest single kernel of the step gets its own entry
            km = next((k for k in merged if "fk_root_wave_kernel<4," in k and "full-size" in k), None)
            if km and hbm(km) is not None:
                ms = merged[km]["dur"] / merged[km]["n"]
                cm = merged[km]["counters"]
                ent = dict(kernel=km.split(" [")[0] + " (root scan of the Measure slot, block 1): the longest single kernel of the step",
                           avg_ms=ms, hbm_bytes_per_launch=hbm(km), GBps=hbm(km) / (ms * 1e-3) / 1e9,
                           frac_of_hbm_peak=hbm(km) / (ms * 1e-3) / 8e12, source=source)
                if "SQ_WAIT_ANY" in cm and "SQ_WAVE_CYCLES" in cm:
                    ent["SQ_WAIT_ANY_over_WAVE_CYCLES"] = cm["SQ_WAIT_ANY"][1] / cm["SQ_WAVE_CYCLES"][1]
                if "TCC_HIT_sum" in cm and "TCC_MISS_sum" in cm:
                    ent["L2_hit_rate"] = cm["TCC_HIT_sum"][1] / (cm["TCC_HIT_sum"][1] + cm["TCC_MISS_sum"][1])
                # ... and the launch GROUP bench.py times for it (pclean_set_timed_block): the group kernels of the Measure slot
                # are the instances whose grid is the second of the two large ones (block 0's root has more groups)
                def hbm_of(m):
                    c = m["counters"]
                    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
                        return None
                    return (2 * c["FETCH_SIZE"][1] / c["FETCH_SIZE"][0] + c["WRITE_SIZE"][1] / c["WRITE_SIZE"][0]) * 1024

                desc = sorted(((g, m) for (k, g), m in bygrid.items() if k.startswith("group_desc_kernel") and g > 50000),
                              key=lambda gm: -gm[1]["dur"])
                if desc:
                    g0 = max(g for g, _ in desc[:2])
                    gm_ = [g for g, _ in desc if g < 0.8 * g0]
                    if gm_:
                        gm = gm_[0]
                        comp = {km: hbm(km)}
                        ms_g = {km: ms}
                        for pat, scale in (("group_desc_kernel", 1), ("group_lse_kernel", 1), ("group_settle_kernel", 16)):
                            cand = [(g, m) for (k, g), m in bygrid.items() if k.startswith(pat) and abs(g / scale - gm) <= 0.05 * gm]
                            if cand:
                                g_, m_ = max(cand, key=lambda x: x[1]["n"])
                                if hbm_of(m_) is not None:
                                    comp[f"{pat} [grid {g_}]"] = hbm_of(m_)
                                    ms_g[f"{pat} [grid {g_}]"] = m_["dur"] / m_["n"]
                        ent["group_bytes_per_launch"] = sum(comp.values())
                        ent["group_components"] = comp
                        ent["group_avg_ms"] = ms_g
                prev = data.get("measure_root") or {}
                for stale in ("groups", "full_scans", "survivors_scored_exactly", "SQ_ACTIVE_INST_ANY_over_WAVE_CYCLES", "note"):
                    prev.pop(stale, None)  # (hand-added in round 5 from that round's phase clock: not this run's)
                data["measure_root"] = dict(prev, **ent)
            json.dump(data, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()

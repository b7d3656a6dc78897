"""Shared by tests/test_literal_fixtures.py (C++ oracle, CPU) and tests/test_gpu_literal.py (HIP path): compares
per-candidate scores of pclean_score_node-style scorers with the literal interpreter's fixtures
(tests/golden/literal_scores.json, generator tests/golden/make_literal_fixtures.py)."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _content_key(lw, trace, cname, k):
    """flattened values of latent row k as 'path=string|...' (path-sorted) — decoding only."""
    t = trace.tables[cname]
    flat = {}
    for j, col in enumerate(lw.layout[cname]):
        if col.kind == "val":
            cls, attr = lw.model.resolve(cname, col.name)
            flat[col.name] = lw.latent_dom[(cls, attr.name)].string(int(t.cols[j, k]))
    return "|".join(f"{p}={flat[p]}" for p in sorted(flat))


def check(S, score_node, rtol=1e-12):
    """score_node(block, rows, ctxv, excl) -> (lse [n], scores [n][K+1]) for node 0 of the block."""
    lw, tr = S["lw"], S["trace"]
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "literal_scores.json")))
    n_checked = 0
    for r in fx["rows"]:
        i = r["row"]
        for bi, fb in enumerate(r["blocks"]):
            blk = lw.blocks[bi]
            cname = blk["root_class"]
            t = tr.tables[cname]
            ctxv = np.zeros((1, 2), dtype=np.int32)
            for c, (sb, col) in enumerate(zip(blk.get("ctx_src_block", []), blk.get("ctx_src_col", []))):
                src = tr.tables[lw.blocks[sb]["root_class"]]
                ctxv[0, c] = src.cols[col, tr.cur[sb, i]]
            lse, scores = score_node(bi, np.array([i], np.int32), ctxv, np.array([tr.cur[bi, i]], np.int32), t.n)
            scores = np.asarray(scores).reshape(-1)
            assert len(scores) == t.n + 1
            seen = 0
            for k in range(t.n):
                key = _content_key(lw, tr, cname, k)
                if key in fb["cands"]:
                    want = fb["cands"][key]
                    assert abs(scores[k] - want) <= rtol * max(1.0, abs(want)), (i, bi, key, scores[k], want)
                    seen += 1
                else:  # the literal trace deleted it (it lost its last reference) or it is a free slot
                    assert scores[k] == -np.inf, (i, bi, key, scores[k])
            assert seen == len(fb["cands"]), (i, bi, seen, len(fb["cands"]))
            assert abs(scores[t.n] - fb["new"]) <= rtol * max(1.0, abs(fb["new"])), (i, bi, "new", scores[t.n], fb["new"])
            assert abs(lse[0] - fb["lse"]) <= 1e-9 * max(1.0, abs(fb["lse"])), (i, bi, "lse", lse[0], fb["lse"])
            n_checked += seen + 1
    return n_checked


def check_rents(S, score_node, rtol=1e-10):
    """rents fixtures (tests/golden/literal_scores_rents.json, generator make_literal_fixtures_rents.py): candidates
    the noise-free observations rule out must score -inf, the others and the new row within rtol (the fixtures use a
    plain log-sum-exp over the enumerated own choices, the product its fixed-point one: 2^-40 relative per term)."""
    lw, tr = S["lw"], S["trace"]
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "literal_scores_rents.json")))
    t = tr.tables["County"]
    n_checked = 0
    for r in fx["rows"]:
        i = r["row"]
        lse, scores = score_node(0, np.array([i], np.int32), np.zeros((1, 2), np.int32), np.array([tr.cur[0, i]], np.int32), t.n)
        scores = np.asarray(scores).reshape(-1)
        assert len(scores) == t.n + 1
        seen = impossible = 0
        for k in range(t.n):
            key = _content_key(lw, tr, "County", k)
            if key in r["cands"]:
                want = r["cands"][key]
                assert abs(scores[k] - want) <= rtol * max(1.0, abs(want)), (i, key, scores[k], want)
                seen += 1
            else:
                assert scores[k] == -np.inf, (i, key, scores[k])
                impossible += 1
        assert seen == len(r["cands"]), (i, seen, len(r["cands"]))
        assert abs(scores[t.n] - r["new"]) <= rtol * max(1.0, abs(r["new"])), (i, "new", scores[t.n], r["new"])
        assert abs(lse[0] - r["lse"]) <= 1e-9 * max(1.0, abs(r["lse"])), (i, "lse", lse[0], r["lse"])
        n_checked += seen + 1
    return n_checked


def check_flights(S, score_node, logml, rtol=1e-12):
    """flights fixtures (tests/golden/literal_scores_flights.json): per-candidate scores of the two reference-slot
    blocks through score_node, and logml [n_rows] of a ONE-particle conditional SMC sweep (nothing moves: the log
    marginal likelihood estimate is the sum of the two block marginals and the scoring block's value)."""
    lw, tr = S["lw"], S["trace"]
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "literal_scores_flights.json")))
    n_checked = 0
    for r in fx["rows"]:
        i = r["row"]
        for bi, fb in enumerate(r["blocks"]):
            cname = lw.blocks[bi]["root_class"]
            assert cname == fb["cls"]
            t = tr.tables[cname]
            lse, scores = score_node(bi, np.array([i], np.int32), np.zeros((1, 2), np.int32),
                                     np.array([tr.cur[bi, i]], np.int32), t.n)
            scores = np.asarray(scores).reshape(-1)
            seen = 0
            for k in range(t.n):
                key = _content_key(lw, tr, cname, k)
                if key in fb["cands"]:
                    want = fb["cands"][key]
                    assert abs(scores[k] - want) <= rtol * max(1.0, abs(want)), (i, bi, key, scores[k], want)
                    seen += 1
                else:
                    assert scores[k] == -np.inf, (i, bi, key, scores[k])
            assert seen == len(fb["cands"]), (i, bi, seen, len(fb["cands"]))
            assert abs(scores[t.n] - fb["new"]) <= 1e-10 * max(1.0, abs(fb["new"])), (i, bi, "new", scores[t.n], fb["new"])
            assert abs(lse[0] - fb["lse"]) <= 1e-9 * max(1.0, abs(fb["lse"])), (i, bi, "lse", lse[0], fb["lse"])
            n_checked += seen + 1
        assert abs(logml[i] - r["logml"]) <= 1e-9 * max(1.0, abs(r["logml"])), (i, "logml", logml[i], r["logml"])
        n_checked += 1
    return n_checked


def check_latent(S, eval_ev, rtol=1e-10):
    """latent-row fixtures (tests/golden/literal_scores_latent.json): eval_ev(block_id, node_id, ev_rows, ev_ctx, excl,
    n_scores) -> (lse, scores) for a node of a latent plan; evidence sets and ctx from the product's build_evidence."""
    from pclean_amd.inference import build_evidence
    lw, tr = S["lw"], S["trace"]
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "literal_scores_latent.json")))
    ev_cache = {}
    n_checked = 0
    for rec in fx["rows"]:
        cname = rec["cls"]
        if cname not in ev_cache:
            ev_cache[cname] = build_evidence(lw, tr, cname)
        live, ev_off, ev_rows, ev_ctx = ev_cache[cname]
        t = tr.tables[cname]
        (pos,) = [j for j, k in enumerate(live) if _content_key(lw, tr, cname, int(k)) == rec["content"]]
        k = int(live[pos])
        e0, e1 = int(ev_off[pos]), int(ev_off[pos + 1])
        assert e1 - e0 == rec["n_evidence"], (cname, rec["content"], e1 - e0, rec["n_evidence"])
        rows = ev_rows[e0:e1]
        ctx = None if ev_ctx is None else ev_ctx[e0:e1]
        pl = lw.latent_plans[cname]
        for root, attr in zip(pl["roots"], pl["root_attr"]):
            want = rec["roots"][attr]
            if want["kind"] == "leaf":
                opts = lw.option_values[(cname, attr)]
                dom = lw.latent_dom[(cname, attr)]
                lse, sc = eval_ev(pl["block_id"], root, rows, ctx, -1, len(opts))
                assert len(want["scores"]) == len(opts)
                for j, v in enumerate(opts):
                    w_ = want["scores"][dom.string(int(v))]
                    assert (sc[j] == w_) or abs(sc[j] - w_) <= rtol * max(1.0, abs(w_)), (cname, attr, dom.string(int(v)), sc[j], w_)
                n_checked += len(opts)
            else:
                tcls = pl["node_info"][root]["cls"]
                tt = tr.tables[tcls]
                excl = int(t.cols[lw.colidx[cname][attr], k])
                lse, sc = eval_ev(pl["block_id"], root, rows, ctx, excl, tt.n + 1)
                seen = 0
                for kk in range(tt.n):
                    key = _content_key(lw, tr, tcls, kk)
                    if key in want["cands"] and tt.live[kk] and not (kk == excl and tt.counts[kk] <= 1):
                        w_ = want["cands"][key]
                        assert abs(sc[kk] - w_) <= rtol * max(1.0, abs(w_)), (cname, attr, key, sc[kk], w_)
                        seen += 1
                    else:
                        assert sc[kk] == -np.inf, (cname, attr, key, sc[kk])
                assert seen == len(want["cands"]), (cname, attr, seen, len(want["cands"]))
                assert abs(sc[tt.n] - want["new"]) <= rtol * max(1.0, abs(want["new"])), (cname, attr, "new", sc[tt.n], want["new"])
                n_checked += seen + 1
            assert abs(lse - want["lse"]) <= 1e-9 * max(1.0, abs(want["lse"])), (cname, attr, "lse", lse, want["lse"])
    return n_checked


def check_latent_flights(S, eval_ev, rtol=1e-10):
    """flights latent fixtures (tests/golden/literal_scores_latent_flights.json): time attributes of Flight rows; the
    option table holds every key's atoms + dummies, the row's own key is selected by the equality term on the key
    column: options of other flights must score -inf."""
    from pclean_amd.inference import build_evidence
    lw, tr = S["lw"], S["trace"]
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "literal_scores_latent_flights.json")))
    live, ev_off, ev_rows, ev_ctx = build_evidence(lw, tr, "Flight")
    assert ev_ctx is not None  # the per-evidence-row error-probability index
    pl = lw.latent_plans["Flight"]
    n_checked = 0
    for rec in fx["rows"]:
        (pos,) = [j for j, k in enumerate(live) if _content_key(lw, tr, "Flight", int(k)) == rec["content"]]
        k = int(live[pos])
        e0, e1 = int(ev_off[pos]), int(ev_off[pos + 1])
        assert e1 - e0 == rec["n_evidence"]
        fid = int(tr.tables["Flight"].cols[lw.colidx["Flight"]["flight_id"], k])
        for root, attr in zip(pl["roots"], pl["root_attr"]):
            if attr not in rec["roots"]:
                continue
            want = rec["roots"][attr]
            vals, keys = lw.option_values[("Flight", attr)], lw.option_keycol[("Flight", attr)]
            dom = lw.latent_dom[("Flight", attr)]
            lse, sc = eval_ev(pl["block_id"], root, ev_rows[e0:e1], ev_ctx[e0:e1], -1, len(vals))
            seen = 0
            for j in range(len(vals)):
                if int(keys[j]) == fid:
                    w_ = want["scores"][dom.string(int(vals[j]))]
                    assert (sc[j] == w_) or abs(sc[j] - w_) <= rtol * max(1.0, abs(w_)), (rec["content"], attr, dom.string(int(vals[j])), sc[j], w_)
                    seen += 1
                else:
                    assert sc[j] == -np.inf
            assert seen == len(want["scores"])
            assert abs(lse - want["lse"]) <= 1e-9 * max(1.0, abs(want["lse"])), (rec["content"], attr, lse, want["lse"])
            n_checked += seen
    return n_checked


def check_latent_rents(S, eval_ev, rtol=1e-10):
    """rents latent fixtures (tests/golden/literal_scores_latent_rents.json): County.name (keyed options: other keys'
    options must score -inf) and County.state against the evidence set; S's trace must carry the fixture's own
    choices (br = row % 5, unit = row % 2)."""
    from pclean_amd.inference import build_evidence
    lw, tr = S["lw"], S["trace"]
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "literal_scores_latent_rents.json")))
    live, ev_off, ev_rows, ev_ctx = build_evidence(lw, tr, "County")
    assert ev_ctx is not None  # the referring rows' own choices
    pl = lw.latent_plans["County"]
    n_checked = 0
    for rec in fx["rows"]:
        (pos,) = [j for j, k in enumerate(live) if _content_key(lw, tr, "County", int(k)) == rec["content"]]
        k = int(live[pos])
        e0, e1 = int(ev_off[pos]), int(ev_off[pos + 1])
        assert e1 - e0 == rec["n_evidence"]
        ck = int(tr.tables["County"].cols[lw.colidx["County"]["countykey"], k])
        for root, attr in zip(pl["roots"], pl["root_attr"]):
            if attr not in rec["roots"]:
                continue
            want = rec["roots"][attr]
            vals = lw.option_values[("County", attr)]
            keys = lw.option_keycol.get(("County", attr))
            dom = lw.latent_dom[("County", attr)]
            lse, sc = eval_ev(pl["block_id"], root, ev_rows[e0:e1], ev_ctx[e0:e1], -1, len(vals))
            seen = 0
            for j in range(len(vals)):
                if keys is None or int(keys[j]) == ck:
                    w_ = want["scores"][dom.string(int(vals[j]))]
                    assert (sc[j] == w_) or abs(sc[j] - w_) <= rtol * max(1.0, abs(w_)), (rec["content"], attr, dom.string(int(vals[j])), sc[j], w_)
                    seen += 1
                else:
                    assert sc[j] == -np.inf
            assert seen == len(want["scores"]), (attr, seen, len(want["scores"]))
            if np.isfinite(want["lse"]):
                assert abs(lse - want["lse"]) <= 1e-9 * max(1.0, abs(want["lse"])), (rec["content"], attr, lse, want["lse"])
            n_checked += seen
    return n_checked

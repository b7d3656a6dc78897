"""evaluate_accuracy (src/analysis.jl:36-88) over the flat trace, vectorised.

Counts follow the reference exactly: `errors` over every column common to the
dirty and clean tables (queried or not), `changed` / `cleaned` over queried
columns, imputations for missing dirty cells; F1 = harmonic mean.
"""
import numpy as np


def reconstructed_pool_ids(lowered, trace, row_slice=None):
    """{query column: pool id of the value PClean currently believes, per local row}."""
    lw = lowered
    m, q = lw.model, lw.query
    ocls = m.classes[q.cls]
    fk_block = {blk["root_fk"]: bi for bi, blk in enumerate(lw.blocks) if not blk.get("score")}
    out = {}
    for col, ref in q.cleanmap.items():
        own = ocls.attr(ref) if "." not in ref else None
        if own is not None and own.kind == "choice":  # own discrete choice (e.g. br): option -> string
            bi = next(iter(lw.locals))
            li = lw.locals[bi].index(ref)
            dom = lw.latent_dom[(q.cls, ref)]
            out[col] = dom.id_array()[np.maximum(trace.locals[bi][:, li], 0)]
            continue
        if own is not None and own.kind == "julia" and getattr(lw, "gauss_spec", None) is not None \
                and lw.gauss_spec["gauss_attr"] == q.obsmap[col]:
            # corrected = round(unit.backward(x)) (experiments/rents/run.jl:25): numeric, compared as a number
            spec = lw.gauss_spec
            bi = next(iter(lw.locals))
            u = np.maximum(trace.locals[bi][:, spec["t_local"]], 0)
            out[col] = ("numeric", np.round(lw.gauss_backward(np.arange(trace.cur.shape[1]), u)))
            continue
        if "." in ref:
            head, rest = ref.split(".", 1)
            bi = fk_block[head]
            cname = lw.blocks[bi]["root_class"]
            t = trace.tables[cname]
            vals = t.cols[lw.colidx[cname][rest], trace.cur[bi]]
            c2, a2 = m.resolve(cname, rest)
            out[col] = lw.latent_dom[(c2, a2.name)].id_array()[vals]
        else:
            j = ocls.attr(ref)  # julia node: recompute from its arguments
            parts = []
            for arg in j.args:
                head, rest = arg.split(".", 1)
                bi = fk_block[head]
                cname = lw.blocks[bi]["root_class"]
                t = trace.tables[cname]
                c2, a2 = m.resolve(cname, rest)
                dom = lw.latent_dom[(c2, a2.name)]
                parts.append([dom.string(v) for v in t.cols[lw.colidx[cname][rest], trace.cur[bi]]])
            strs = [j.fn(*xs) for xs in zip(*parts)]
            out[col] = np.array([lw.pool.index.get(s, -2) for s in strs], dtype=np.int64)
    return out


def _pool_ids(index, values, unknown, missing):
    """Pool id of every value (object array of str / None): `unknown` for strings outside the pool,
    `missing` for None — looked up once per distinct value."""
    import pandas as pd
    codes, uniques = pd.factorize(values, use_na_sentinel=True)
    table = np.array([index.get(u, unknown) for u in uniques] + [missing], dtype=np.int64)
    return table[codes]  # code -1 (None) selects the last entry


def accuracy_counts(lowered, trace, dirty, clean):
    """The five counters of evaluate_accuracy for the rows held by `trace`."""
    lw = lowered
    ours = reconstructed_pool_ids(lw, trace)
    n = trace.cur.shape[1]
    errors = changed = cleaned = imputed = imputed_ok = 0
    for col in clean:
        if col not in dirty:
            continue
        d = np.asarray(dirty[col][:n], dtype=object)
        c = np.asarray(clean[col][:n], dtype=object)
        dmiss = np.equal(d, None)
        ne = d != c  # element-wise Python comparison (None == None, None != any string)
        errors += int(np.sum(ne & ~dmiss))
        if col not in ours:
            continue
        if isinstance(ours[col], tuple):  # numeric column: compare numbers (Float64 == Int in the reference)
            o = ours[col][1]
            dn = np.array([np.nan if v is None else float(v) for v in d])
            cn = np.array([np.nan if v is None else float(v) for v in c])
            errors -= int(np.sum(ne & ~dmiss))
            errors += int(np.sum((dn != cn) & ~dmiss))
            ch = (~dmiss) & (o != dn)
            changed += int(np.sum(ch))
            cleaned += int(np.sum(ch & (o == cn)))
            imputed += int(np.sum(dmiss & ~np.isnan(cn)))  # analysis.jl:52-60 counts numeric imputations too
            imputed_ok += int(np.sum(dmiss & (o == cn)))
            continue
        d_id = _pool_ids(lw.pool.index, d, -3, -4)
        c_id = _pool_ids(lw.pool.index, c, -5, -6)
        o = ours[col]
        cmiss = c_id == -6
        imputed += int(np.sum(dmiss & ~cmiss))
        imputed_ok += int(np.sum(dmiss & ~cmiss & (o == c_id)))
        ch = (~dmiss) & (o != d_id)
        changed += int(np.sum(ch))
        cleaned += int(np.sum(ch & (o == c_id)))
    return np.array([errors, changed, cleaned, imputed, imputed_ok], dtype=np.int64)


def f1_from_counts(cnt):
    errors, changed, cleaned, imputed, imputed_ok = [float(x) for x in cnt]
    num = cleaned + imputed_ok
    precision = num / (changed + imputed) if (changed + imputed) > 0 else float("nan")
    recall = num / (errors + imputed) if (errors + imputed) > 0 else float("nan")
    f1 = 2.0 / (1 / precision + 1 / recall) if num > 0 else 0.0
    return dict(f1=f1, errors=int(errors), changed=int(changed), cleaned=int(cleaned), precision=precision,
                recall=recall, imputed=int(imputed), correctly_imputed=int(imputed_ok))


def evaluate_accuracy(lowered, trace, dirty, clean):
    return f1_from_counts(accuracy_counts(lowered, trace, dirty, clean))


def reconstructed_table(lowered, trace, dirty):
    """The cleaned table: every queried column replaced by what PClean believes, the others copied
    (the DataFrame save_results writes, analysis.jl:21-30).  Values come back as strings / numbers."""
    lw = lowered
    ours = reconstructed_pool_ids(lw, trace)
    n = trace.cur.shape[1]
    out = {}
    for col, vals in dirty.items():
        if col not in ours:
            out[col] = list(vals[:n])
        elif isinstance(ours[col], tuple):
            out[col] = [None if np.isnan(v) else (int(v) if float(v).is_integer() else float(v)) for v in ours[col][1]]
        else:
            out[col] = [lw.pool.strings[i] if i >= 0 else None for i in ours[col]]
    return out


def latent_table(lowered, trace, cname):
    """One inferred latent table (save_tables, analysis.jl:8-13): row id + the value of every own attribute and
    reference slot of the class (flattened copies of referents' values are left out, as the reference does
    for its '#'-named inlined nodes)."""
    lw = lowered
    t = trace.tables[cname]
    live = np.nonzero(t.live[:t.n])[0]
    out = {"id": [int(k) for k in live]}
    for j, c in enumerate(lw.layout[cname]):
        if "." in c.name:
            continue
        if c.kind == "fk":
            out[c.name] = [int(v) for v in t.cols[j, live]]
        else:
            dom = lw.latent_dom[(cname, c.name)]
            out[c.name] = [dom.string(v) for v in t.cols[j, live]]
    return out


def save_results(directory, name, lowered, trace, dirty, timestamp=True):
    """save_results (analysis.jl:15-33): reconstructed_<class>.csv + inferred_<class>.csv per latent class."""
    import datetime
    import os

    import pandas as pd
    d = os.path.join(directory, f"{name}-{datetime.datetime.now().isoformat()}" if timestamp else name)
    os.makedirs(d, exist_ok=True)
    pd.DataFrame(reconstructed_table(lowered, trace, dirty)).to_csv(
        os.path.join(d, f"reconstructed_{lowered.query.cls}.csv"), index=False)
    for cname in trace.tables:
        pd.DataFrame(latent_table(lowered, trace, cname)).to_csv(os.path.join(d, f"inferred_{cname}.csv"), index=False)
    return d

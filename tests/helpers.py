"""Shared test scaffolding: build the hospital program, an initial trace, and
mirror the uploaded state into the CPU oracle's World."""
import numpy as np

from pclean_amd import experiments as ex
from pclean_amd.model import LoweredModel
from pclean_amd.trace import Trace


def hospital_setup(n_rows=None, seed=0):
    dirty, clean = ex.hospital_data()
    if n_rows is not None:
        dirty = {c: v[:n_rows] for c, v in dirty.items()}
        clean = {c: v[:n_rows] for c, v in clean.items()}
    poss = ex.possibilities_of(dirty)
    m = ex.hospital_model(poss)
    q = ex.hospital_query(m)
    lw = LoweredModel(m, q, dirty)
    obs = lw.encode_observations(dirty)
    n = obs.shape[1]
    # initial latent state: clean value where it is a possible latent value, else the dirty one
    by_path = [{}, {}]
    ocls = m.classes[q.cls]
    for col, ref in q.cleanmap.items():
        if "." not in ref:
            continue
        head, rest = ref.split(".", 1)
        bi = 0 if head == "hosp" else 1
        cname, attr = m.resolve(ocls.attr(head).target, rest)
        dom = lw.latent_dom[(cname, attr.name)]
        vals = []
        for i in range(n):
            v = clean[col][i]
            vals.append(v if (v is not None and dom.get(v) >= 0) else dirty[col][i])
        by_path[bi][rest] = vals
    tr = Trace.from_clean_values(lw, by_path, n, seed)
    return dict(dirty=dirty, clean=clean, model=m, query=q, lw=lw, obs=obs, trace=tr)


def rents_setup(n_rows=600, seed=3, units=None):
    """rents (experiments/rents/run.jl) on the first n_rows rows, latent state from the clean values (a clean county
    name that never occurs undamaged falls back to the dirty cell) — the deterministic state of the literal
    interpreter's rents fixtures (tests/golden/literal_scores_rents.json)."""
    dirty, clean = ex.rents_data()
    dirty = {c: v[:n_rows] for c, v in dirty.items()}
    clean = {c: v[:n_rows] for c, v in clean.items()}
    m = ex.rents_model(dirty, units)
    q = ex.rents_query(m)
    lw = LoweredModel(m, q, dirty)
    obs = lw.encode_observations(dirty)
    n = obs.shape[1]
    name_dom, state_dom = lw.latent_dom[("County", "name")], lw.latent_dom[("County", "state")]
    names = [c if (c is not None and name_dom.get(c) >= 0) else d for c, d in zip(clean["County"], dirty["County"])]
    states = []
    for i in range(n):
        v = clean["State"][i] if clean["State"][i] is not None and state_dom.get(clean["State"][i]) >= 0 else dirty["State"][i]
        states.append(v if v is not None else state_dom.string(0))
    tr = Trace.from_clean_values(lw, {0: {"countykey": list(dirty["CountyKey"]), "name": names, "state": states}}, n, seed)
    return dict(dirty=dirty, clean=clean, model=m, query=q, lw=lw, obs=obs, trace=tr)


def flights_setup(seed=5):
    """flights (experiments/flights/run.jl), latent state from the clean values: a Flight's time is the clean value
    when that is one of the flight's observed atoms, else the TimePrior dummy (the states inference can reach) — the
    deterministic state of the literal interpreter's flights fixtures (tests/golden/literal_scores_flights.json)."""
    dirty, clean = ex.flights_data()
    m = ex.flights_model(dirty)
    q = ex.flights_query(m)
    lw = LoweredModel(m, q, dirty)
    obs = lw.encode_observations(dirty)
    n = obs.shape[1]
    cols = {"sdt": "sched_dep_time", "sat": "sched_arr_time", "adt": "act_dep_time", "aat": "act_arr_time"}
    by0 = {"flight_id": list(dirty["flight"])}
    for f, c in cols.items():
        d = m.classes["Flight"].attr(f).dist
        by0[f] = [clean[c][i] if clean[c][i] in d.atoms[dirty["flight"][i]] else d.dummy_value() for i in range(n)]
    tr = Trace.from_clean_values(lw, {0: by0, 1: {"name": list(dirty["src"])}}, n, seed)
    return dict(dirty=dirty, clean=clean, model=m, query=q, lw=lw, obs=obs, trace=tr)


def density_tables_cpu(oracle, max_len):
    L = oracle.lib()
    ml = max(max_len, 64)
    mr, md = (ml + 4) // 5, ml
    nb = np.zeros((mr + 1, md + 1))
    nb[0, 1:] = -np.inf
    for r in range(1, mr + 1):
        for d in range(md + 1):
            nb[r, d] = L.pco_negbin_logpdf(float(r), 0.9, d)
    logl = np.zeros(ml + 1)
    logl[1:] = np.log(np.arange(1, ml + 1, dtype=np.float64))
    return mr, md, ml, nb, logl


def mirror_world(oracle, lw, obs, trace, engine=None, dist_mode=1, option_logp=None, row_lo=0):
    """Oracle World holding exactly what the product uploaded.  With an engine the
    double-valued tables are read back from the library (bit-identical inputs);
    without one (CPU tests) they are computed by the oracle itself."""
    w = oracle.World()
    w.set_obs(obs)
    sym, off, lm, _ = lw.pool.arrays()
    max_len = int(lw.pool.lens.max())
    if engine is not None:
        mr, md, ml, nb, logl = engine.hip.get_density_tables()
    else:
        mr, md, ml, nb, logl = density_tables_cpu(oracle, max_len)
    w.set_density(mr, md, ml, nb, logl)
    from pclean_amd.encode import load_lm_params
    w.set_strings(sym, off)
    w.set_lm(*load_lm_params(), lw.pool.letter_symbols())
    if engine is not None:
        dist_mode = engine.dist_mode
    for key, (pid, odom, ldom) in lw.pair_id.items():
        if engine is not None:
            d = engine.hip.get_pair_table(pid, len(odom), len(ldom))
        else:
            d = oracle.pair_table(sym, off, odom.id_array(), ldom.id_array(), dist_mode)
        w.set_pair(pid, d, lw.pool.lens[ldom.id_array()].astype(np.uint16))
        w.set_pair_strings(pid, odom.id_array(), dist_mode)
    for fid, fn in lw.fn_tables.items():
        w.set_fn(fid, fn)
    for key, (pid, n) in lw.eq_pairs.items():
        w.set_pair(pid, (1 - np.eye(n, dtype=np.uint16)), np.zeros(n, dtype=np.uint16))
    if getattr(lw, "xnum", None) is not None and lw.xnum.shape[0]:
        w.set_numeric(lw.xnum[:, row_lo:row_lo + obs.shape[1]])
    for cname, t in trace.tables.items():
        cols, counts = t.view()
        if engine is not None:
            full, m1, scal = engine.hip.get_table_priors(lw.table_id[cname], len(counts))
        else:
            full, m1, scal = oracle.table_priors(counts, t.strength, t.discount)
        w.set_table(lw.table_id[cname], np.ascontiguousarray(cols), counts, full, m1, scal)
    logps = engine.option_logp if engine is not None else option_logp
    for (cname, aname), dom in lw.latent_dom.items():
        if (cname, aname) not in logps:
            continue
        if (cname, aname) in lw.option_keycol:
            cols = [lw.option_values[(cname, aname)], lw.option_keycol[(cname, aname)]]
            if (cname, aname) in lw.option_ncol:
                cols.append(lw.option_ncol[(cname, aname)])
            w.set_options_cols(lw.option_id[(cname, aname)], np.stack(cols), logps[(cname, aname)])
        else:
            w.set_options(lw.option_id[(cname, aname)], lw.option_values[(cname, aname)], logps[(cname, aname)])
    for pid in lw.same_pairs:
        d = lw.same_pair_table(pid)
        w.set_pair(pid, d.astype(np.uint16), np.zeros(d.shape[1], dtype=np.uint16))
    if lw.prob_spec is not None:
        w.set_prob(trace.prob_table())
    lw.load_blocks_into(w)
    if getattr(lw, "gauss", None):
        from pclean_amd.engine import make_gauss
        w.set_mean(0, trace.mean_param.value)
        for (bid, nid), spec in lw.gauss.items():
            w.set_gauss(bid, nid, make_gauss(spec))
    return w


def option_logp_cpu(oracle, lw, trace):
    """discrete_proposal log-probabilities computed by the oracle (CPU tests)."""
    from pclean_amd.encode import load_lm_params
    from pclean_amd.model import ChooseProportionally, ChooseUniformly, StringPrior, TimePrior, Unmodeled
    init, trans = load_lm_params()
    _, off, lm, _ = lw.pool.arrays()
    out = {}
    for (cname, aname), dom in lw.latent_dom.items():
        d = lw.model.classes[cname].attr(aname).dist
        if isinstance(d, TimePrior):
            vals, keys = lw.option_values[(cname, aname)], lw.option_keycol[(cname, aname)]
            dummy = dom.get(d.dummy_value())
            sc = np.array([oracle.time_prior_atom(dom.string(v)) for v in vals])
            lp = sc.copy()
            for k in np.unique(keys):
                lp[(keys == k) & (vals == dummy)] = oracle.dummy_logmass(sc[(keys == k) & (vals != dummy)])
            out[(cname, aname)] = lp
        elif isinstance(d, StringPrior) and d.keyed_by:
            vals, keys = lw.option_values[(cname, aname)], lw.option_keycol[(cname, aname)]
            dummy = dom.get(d.dummy_value())
            ids = dom.id_array()[vals]
            sc = np.array([oracle.string_prior(lm[off[i]:off[i + 1]], d.min_len, d.max_len, init, trans) for i in ids])
            lp = sc.copy()
            for k in np.unique(keys):
                lp[(keys == k) & (vals == dummy)] = oracle.dummy_logmass(sc[(keys == k) & (vals != dummy)])
            out[(cname, aname)] = lp
        elif isinstance(d, StringPrior):
            ids = dom.id_array()[:dom.n_base() - 1]
            sc = np.array([oracle.string_prior(lm[off[i]:off[i + 1]], d.min_len, d.max_len, init, trans) for i in ids])
            out[(cname, aname)] = np.concatenate([sc, [oracle.dummy_logmass(sc)]])
        elif isinstance(d, Unmodeled) and cname != lw.query.cls:
            out[(cname, aname)] = np.zeros(len(dom))  # unmodeled.jl:7-10
        elif isinstance(d, ChooseUniformly):
            out[(cname, aname)] = np.full(len(dom), oracle.lib().pco_choose_uniformly(len(d.options)))
        elif isinstance(d, ChooseProportionally):
            with np.errstate(divide="ignore"):
                out[(cname, aname)] = np.log(trace.params[(cname, d.param)].value)
    return out


def truth_workload(n_rows, n_hosp, seed):
    """Synthetic hospital-shaped table (pclean_amd.synth) with the latent state set to the generator's
    ground-truth entities (clean values that never occur undamaged fall back to the dirty cell) — a known,
    cheap starting state for the full-size property tests.  Returns (dirty, clean, lowered, obs, trace)."""
    from pclean_amd import experiments as ex
    from pclean_amd.model import LoweredModel
    from pclean_amd.synth import synth_hospital
    from pclean_amd.trace import Trace
    dirty, clean, latent = synth_hospital(n_rows, n_hosp, seed)
    poss = ex.possibilities_of(dirty)
    m = ex.hospital_model(poss)
    q = ex.hospital_query(m)
    lw = LoweredModel(m, q, dirty)
    obs = lw.encode_observations(dirty)
    by_path = [{}, {}]
    ocls = m.classes[q.cls]
    for col, ref in q.cleanmap.items():
        if "." not in ref:
            continue
        head, rest = ref.split(".", 1)
        bi = 0 if head == "hosp" else 1
        cname, attr = m.resolve(ocls.attr(head).target, rest)
        dom = lw.latent_dom[(cname, attr.name)]
        by_path[bi][rest] = [c if dom.get(c) >= 0 else d for c, d in zip(clean[col], dirty[col])]
    tr = Trace.from_clean_values(lw, by_path, n_rows, seed)
    return dirty, clean, lw, obs, tr

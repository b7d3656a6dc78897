"""rents with a SECOND reference slot after the block that carries the Gaussian term (TEST INFRASTRUCTURE): under particle
Gibbs a resampling step sits between the two blocks, so with use_dd_proposals = false the own choices every particle
sampled for the Gaussian term (block_proposal.jl:42-56) have to follow their particles through it
(row_inference.jl:139-151).  None of the reference's three programs has this shape."""
import numpy as np

from pclean_amd import experiments as ex
from pclean_amd.model import AddTypos, LoweredModel, Query, StringPrior
from pclean_amd.trace import Trace

LANDLORDS = ["acme homes", "birch realty", "cedar llc", "dover estates", "elm street trust"]


def _damage(s, rng):
    if rng.random() < 0.25:
        i = int(rng.integers(0, len(s)))
        return s[:i] + "x" + s[i + 1:]
    return s


def setup(n_rows=400, seed=3):
    dirty, clean = ex.rents_data()
    dirty = {c: v[:n_rows] for c, v in dirty.items()}
    clean = {c: v[:n_rows] for c, v in clean.items()}
    rng = np.random.default_rng(11)
    who = [LANDLORDS[int(rng.integers(0, len(LANDLORDS)))] for _ in range(n_rows)]
    dirty["Landlord"] = [None if rng.random() < 0.05 else _damage(w, rng) for w in who]
    m = ex.rents_model(dirty)
    l = m.add_class("Landlord")
    l.choice("name", StringPrior(3, 25, LANDLORDS))
    o = m.cls("Obs") if hasattr(m, "cls") else m.classes["Obs"]
    with o.block():
        o.fk("landlord", "Landlord")
        o.choice("landlord_obs", AddTypos("landlord.name", 2))
    q = Query(m, "Obs", {
        "CountyKey": "county.countykey",
        "County": ("county.name", "county_name"),
        "State": "county.state",
        "Room Type": "br",
        "Monthly Rent": ("corrected", "rent"),
        "Landlord": ("landlord.name", "landlord_obs"),
    })
    lw = LoweredModel(m, q, dirty)
    obs = lw.encode_observations(dirty)
    n = obs.shape[1]
    name_dom, state_dom = lw.latent_dom[("County", "name")], lw.latent_dom[("County", "state")]
    names = [c if (c is not None and name_dom.get(c) >= 0) else d for c, d in zip(clean["County"], dirty["County"])]
    states = []
    for i in range(n):
        v = clean["State"][i] if clean["State"][i] is not None and state_dom.get(clean["State"][i]) >= 0 else dirty["State"][i]
        states.append(v if v is not None else state_dom.string(0))
    tr = Trace.from_clean_values(lw, {0: {"countykey": list(dirty["CountyKey"]), "name": names, "state": states},
                                      1: {"name": who}}, n, seed)
    return dict(dirty=dirty, model=m, query=q, lw=lw, obs=obs, trace=tr)

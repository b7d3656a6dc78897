"""The reference's experiment programs restated with the pclean_amd DSL mirror.

hospital: /root/reference/experiments/hospital/{load_data,run}.jl
"""
import os

import pandas as pd

from .model import (AddTypos, ChooseProportionally, ChooseUniformly, Model, ProportionsParameter, Query, StringPrior)

DATA_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "datasets")


def load_table(path):
    """CSV -> dict of column -> list of str | None (None = missing)."""
    df = pd.read_csv(path, dtype=str, keep_default_na=False)
    return {c: [None if v == "" else v for v in df[c].tolist()] for c in df.columns}


def shuffle_rows(tables, seed=0):
    """The same random permutation applied to every table of `tables` (dirty, clean, ...); returns
    (permuted tables, perm) with permuted[c][i] = original[c][perm[i]].

    Why: the batched initialize_trace lets the rows of one batch see only latent rows created by EARLIER
    batches.  The shipped tables are sorted by entity (all records of a hospital are consecutive), the worst
    case for that schedule — every batch meets only new entities and spawns duplicates (hospital: 349
    latent hospitals after the init instead of ~50).  In random order the small early batches create the
    entities and the large later ones join them.  The sequential reference is insensitive to the order."""
    import numpy as np
    n = len(next(iter(tables[0].values())))
    perm = np.random.default_rng(seed).permutation(n)
    return [{c: [v[i] for i in perm] for c, v in t.items()} for t in tables], perm


def possibilities_of(table):
    """load_data.jl:17-18: unique non-missing dirty values per column, first-seen order."""
    return {c: list(dict.fromkeys(v for v in vals if v is not None)) for c, vals in table.items()}


def hospital_data():
    dirty = load_table(os.path.join(DATA_DIR, "hospital_dirty.csv"))
    clean = load_table(os.path.join(DATA_DIR, "hospital_clean.csv"))
    return dirty, clean


def hospital_model(poss):
    """experiments/hospital/run.jl:5-56."""
    m = Model()
    c = m.add_class("County")
    c.param("state_proportions", ProportionsParameter())
    c.choice("state", ChooseProportionally(poss["State"], "state_proportions"))
    c.choice("county", StringPrior(3, 30, poss["CountyName"]))
    c = m.add_class("Place")
    c.fk("county", "County")
    c.choice("city", StringPrior(3, 30, poss["City"]))
    c = m.add_class("Condition")
    c.choice("desc", StringPrior(5, 35, poss["Condition"]))
    c = m.add_class("Measure")
    c.choice("code", ChooseUniformly(poss["MeasureCode"]))
    c.choice("name", ChooseUniformly(poss["MeasureName"]))
    c.fk("condition", "Condition")
    c = m.add_class("HospitalType")
    c.choice("desc", StringPrior(10, 30, poss["HospitalType"]))
    c = m.add_class("Hospital")
    c.param("owner_dist", ProportionsParameter())
    c.param("service_dist", ProportionsParameter())
    c.fk("loc", "Place")
    c.fk("type", "HospitalType")
    c.choice("provider", ChooseUniformly(poss["ProviderNumber"]))
    c.choice("name", StringPrior(3, 50, poss["HospitalName"]))
    c.choice("addr", StringPrior(10, 30, poss["Address1"]))
    c.choice("phone", StringPrior(10, 10, poss["PhoneNumber"]))
    c.choice("owner", ChooseProportionally(poss["HospitalOwner"], "owner_dist"))
    c.choice("zip", ChooseUniformly(poss["ZipCode"]))
    c.choice("service", ChooseProportionally(poss["EmergencyService"], "service_dist"))
    r = m.add_class("Record")
    with r.block():
        r.fk("hosp", "Hospital")
        r.choice("service", AddTypos("hosp.service"))
        r.choice("provider", AddTypos("hosp.provider"))
        r.choice("name", AddTypos("hosp.name"))
        r.choice("addr", AddTypos("hosp.addr"))
        r.choice("city", AddTypos("hosp.loc.city"))
        r.choice("state", AddTypos("hosp.loc.county.state"))
        r.choice("zip", AddTypos("hosp.zip"))
        r.choice("county", AddTypos("hosp.loc.county.county"))
        r.choice("phone", AddTypos("hosp.phone"))
        r.choice("type", AddTypos("hosp.type.desc"))
        r.choice("owner", AddTypos("hosp.owner"))
    with r.block():
        r.fk("metric", "Measure")
        r.choice("code", AddTypos("metric.code"))
        r.choice("mname", AddTypos("metric.name"))
        r.choice("condition", AddTypos("metric.condition.desc"))
        r.julia("stateavg", lambda state, code: f"{state}_{code}", ["hosp.loc.county.state", "metric.code"])
        r.choice("stateavg_obs", AddTypos("stateavg"))
    return m


def hospital_query(m):
    """experiments/hospital/run.jl:58-74."""
    return Query(m, "Record", {
        "ProviderNumber": ("hosp.provider", "provider"),
        "HospitalName": ("hosp.name", "name"),
        "HospitalType": ("hosp.type.desc", "type"),
        "HospitalOwner": ("hosp.owner", "owner"),
        "Address1": ("hosp.addr", "addr"),
        "PhoneNumber": ("hosp.phone", "phone"),
        "EmergencyService": ("hosp.service", "service"),
        "City": ("hosp.loc.city", "city"),
        "CountyName": ("hosp.loc.county.county", "county"),
        "State": ("hosp.loc.county.state", "state"),
        "ZipCode": ("hosp.zip", "zip"),
        "Condition": ("metric.condition.desc", "condition"),
        "MeasureCode": ("metric.code", "code"),
        "MeasureName": ("metric.name", "mname"),
        "Stateavg": ("stateavg", "stateavg_obs"),
    })


# ---------------------------------------------------------------------------
# rents: /root/reference/experiments/rents/{load_data,run}.jl
def rents_data():
    dirty = load_table(os.path.join(DATA_DIR, "rents_dirty.csv"))
    clean = load_table(os.path.join(DATA_DIR, "rents_clean.csv"))
    # load_data.jl:9 — CountyKey = first letter of the county + last letter of its first word
    dirty["CountyKey"] = [f"{x[0]}{x.split()[0][-1]}" for x in dirty["County"]]
    return dirty, clean


def rents_model(dirty, units=None):
    """experiments/rents/run.jl:5-27.  units: other Transformations for `unit` to choose from (tests of the lowering's
    non-linear path: the reference's type takes any forward / backward / deriv, transformed_gaussian.jl:5-9)."""
    from .model import IndexedLookup, IndexedMeanParameter, TransformedGaussian, Transformation, Unmodeled
    poss = {}
    for k, c in zip(dirty["CountyKey"], dirty["County"]):
        poss.setdefault(k, [])
        if c not in poss[k]:
            poss[k].append(c)
    states = list(dict.fromkeys(v for v in dirty["State"] if v is not None))  # load_data.jl:17
    room_types = ["studio", "1br", "2br", "3br", "4br"]
    if units is None:
        units = [Transformation(lambda x: x, lambda x: x, lambda x: 1.0),
                 Transformation(lambda x: x / 1000.0, lambda x: x * 1000.0, lambda x: 1 / 1000.0)]
    m = Model()
    c = m.add_class("County")
    c.param("state_pops", ProportionsParameter())
    c.choice("countykey", Unmodeled())
    c.choice("name", StringPrior(10, 35, poss, keyed_by="countykey"))
    c.choice("state", ChooseProportionally(states, "state_pops"))
    o = m.add_class("Obs")
    o.param("avg_rent", IndexedMeanParameter(1500, 1000))
    o.fk("county", "County")
    o.choice("county_name", AddTypos("county.name", 2))
    o.choice("br", ChooseUniformly(room_types))
    o.choice("unit", ChooseUniformly(units))
    o.julia("rent_base", IndexedLookup("avg_rent"), ["county.state", "county.countykey", "br"])
    o.choice("rent", TransformedGaussian("rent_base", 150.0, "unit"))
    o.julia("corrected", lambda unit, rent: round(unit.backward(rent)), ["unit", "rent"])
    return m


def rents_query(m):
    """experiments/rents/run.jl:29-35."""
    return Query(m, "Obs", {
        "CountyKey": "county.countykey",
        "County": ("county.name", "county_name"),
        "State": "county.state",
        "Room Type": "br",
        "Monthly Rent": ("corrected", "rent"),
    })


# ---------------------------------------------------------------------------
# flights: /root/reference/experiments/flights/{load_data,run}.jl
FLIGHT_FIELDS = ["sched_dep_time", "sched_arr_time", "act_dep_time", "act_arr_time"]


def flights_data():
    dirty = load_table(os.path.join(DATA_DIR, "flights_dirty.csv"))
    clean = load_table(os.path.join(DATA_DIR, "flights_clean.csv"))
    return dirty, clean


def flights_model(dirty):
    """experiments/flights/run.jl:5-36 (+ load_data.jl:9-17)."""
    from .model import IndexedProbParameter, MaybeSwap, ProbLookup, TimePrior
    flight_ids = list(dict.fromkeys(dirty["flight"]))
    websites = list(dict.fromkeys(dirty["src"]))
    times = {f: {fl: [] for fl in flight_ids} for f in FLIGHT_FIELDS}
    for i, fl in enumerate(dirty["flight"]):
        for f in FLIGHT_FIELDS:
            v = dirty[f][i]
            if v is not None and v not in times[f][fl]:
                times[f][fl].append(v)
    m = Model()
    w = m.add_class("TrackingWebsite")
    w.choice("name", StringPrior(2, 30, websites))
    fl = m.add_class("Flight")
    with fl.block():
        fl.choice("flight_id", StringPrior(10, 20, flight_ids))
    fl.choice("sdt", TimePrior(times["sched_dep_time"], "flight_id"))
    fl.choice("sat", TimePrior(times["sched_arr_time"], "flight_id"))
    fl.choice("adt", TimePrior(times["act_dep_time"], "flight_id"))
    fl.choice("aat", TimePrior(times["act_arr_time"], "flight_id"))
    o = m.add_class("Obs")
    o.param("error_probs", IndexedProbParameter(10.0, 50.0))
    with o.block():
        o.fk("flight", "Flight")
    o.fk("src", "TrackingWebsite")
    o.julia("error_prob", ProbLookup("error_probs", lambda src, fid: 1e-5 if src.lower() == fid[:2].lower() else src),
            ["src.name", "flight.flight_id"])
    with o.block():
        o.choice("sdt", MaybeSwap("flight.sdt", times["sched_dep_time"], "flight.flight_id", "error_prob"))
        o.choice("sat", MaybeSwap("flight.sat", times["sched_arr_time"], "flight.flight_id", "error_prob"))
        o.choice("adt", MaybeSwap("flight.adt", times["act_dep_time"], "flight.flight_id", "error_prob"))
        o.choice("aat", MaybeSwap("flight.aat", times["act_arr_time"], "flight.flight_id", "error_prob"))
    return m


def flights_query(m):
    """experiments/flights/run.jl:38-45."""
    return Query(m, "Obs", {
        "sched_dep_time": ("flight.sdt", "sdt"),
        "sched_arr_time": ("flight.sat", "sat"),
        "act_dep_time": ("flight.adt", "adt"),
        "act_arr_time": ("flight.aat", "aat"),
        "flight": "flight.flight_id",
        "src": "src.name",
    })

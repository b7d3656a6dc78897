"""The INDEPENDENT end-to-end reference (oracle/literal_inference.py: the reference's sequential schedule on a
dict-of-rows trace of strings, scored by oracle/literal.py — none of the product's lowering, trace, inference or analysis
code) against the product's own sequential-schedule runs: tests/golden/literal_sequential.json (generator:
scripts/literal_sequential_reference.py) vs tests/golden/sequential_f1.json (scripts/sequential_reference.py: the
product's host code with batch_rows=1 on the CPU oracle engine).  Same program, configurations, seeds and row shuffles;
the random numbers differ, so F1 is compared on the means (north_star: +-0.5 pt) and the number of latent rows per class
must agree to within a few rows.  A bug in the product's commit, garbage collection, parameter moves or evaluate_accuracy
would show here: it is no longer common to both sides."""
import functools
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "scripts"))


def test_literal_sampler_runs_and_keeps_its_database_consistent():
    import literal_sequential_reference as lsr
    import literal_inference as LI
    from pclean_amd import experiments as ex
    dirty, clean = ex.hospital_data()
    dirty = {c: v[:160] for c, v in dirty.items()}
    clean = {c: v[:160] for c, v in clean.items()}
    (dirty, clean), _ = ex.shuffle_rows([dirty, clean], 4)
    m = ex.hospital_model(ex.possibilities_of(dirty))
    q = ex.hospital_query(m)
    for mh, particles in ((True, 2), (False, 5)):
        s = LI.LiteralSampler(m, q, dirty, lsr.Cfg(1, particles, mh), 4)
        s.initialize()
        s.check()
        f0 = s.accuracy(dirty, clean)["f1"]
        s.sweep()
        s.check()
        acc = s.accuracy(dirty, clean)
        assert acc["f1"] > f0 and acc["f1"] > 0.8 and acc["precision"] > 0.95, (f0, acc)
        rows = s.latent_rows()
        assert rows["HospitalType"] == 1 and 4 <= rows["Hospital"] <= 12, rows


@pytest.mark.parametrize("name", ["hospital", "hospital_pg20"])
def test_product_sequential_runs_match_the_literal_reference(name):
    lit = json.load(open(os.path.join(ROOT, "tests", "golden", "literal_sequential.json")))[name]
    prod = json.load(open(os.path.join(ROOT, "tests", "golden", "sequential_f1.json")))[name]
    assert {k: lit["config"][k] for k in ("iters", "mh", "particles")} == {k: prod["config"][k] for k in ("iters", "mh", "particles")}
    assert sorted(lit["runs"]) == sorted(prod["runs"])
    assert abs(lit["f1_mean"] - prod["f1_mean"]) <= 0.005, (lit["f1_mean"], prod["f1_mean"])
    for sd in lit["runs"]:
        a, b = lit["runs"][sd]["latent_rows"], prod["runs"][sd]["latent_rows"]
        for cls in b:
            assert abs(a[cls] - b[cls]) <= max(3, b[cls] // 10), (sd, cls, a, b)
        assert abs(lit["runs"][sd]["f1"] - prod["runs"][sd]["f1"]) <= 0.01, sd


def test_flights_literal_sampler_runs_and_keeps_its_database_consistent():
    """oracle/literal_inference_flights.py on the first 400 rows: reference counts consistent after the initialisation and
    after every sweep, one latent Flight per flight id and one TrackingWebsite per source (noise-free observations), the
    Flights' times recovered from the MaybeSwap evidence (F1 from ~0 after the initialisation — new Flights draw their
    times from the prior proposal, i.e. the dummy — to > 0.5 on this small sample), and the learned error probabilities stay probabilities."""
    import literal_sequential_reference as lsr
    import literal_inference_flights as LF
    from pclean_amd import experiments as ex
    dirty, clean = ex.flights_data()
    dirty = {c: v[:400] for c, v in dirty.items()}
    clean = {c: v[:400] for c, v in clean.items()}
    (dirty, clean), _ = ex.shuffle_rows([dirty, clean], 3)
    m = ex.flights_model(dirty)
    q = ex.flights_query(m)
    for mh, particles in ((True, 2), (False, 4)):
        s = LF.FlightsLiteralSampler(m, q, dirty, lsr.Cfg(2, particles, mh, rejuv=100), 3)
        s.initialize()
        s.check()
        f0 = s.accuracy(dirty, clean)["f1"]
        for _ in range(3):
            s.sweep()
            s.check()
        acc = s.accuracy(dirty, clean)
        assert f0 < 0.05 and acc["f1"] > 0.5, (f0, acc)  # (400 rows: ~4 sources per flight; the full table reaches 0.89)
        rows = s.latent_rows()
        assert rows["Flight"] == len(set(dirty["flight"])) and rows["TrackingWebsite"] == len(set(dirty["src"])), rows
        probs = s.tr.params[("Obs", "error_probs")]
        assert probs and all(0.0 < p < 1.0 for p in probs.values())


def test_flights_product_sequential_runs_match_the_literal_reference():
    """eight seeds each: the literal flights sampler and the product's sequential-schedule runs agree within the north
    star's +-0.5 pt of F1 (their random numbers differ: means are compared) and on the latent tables' sizes."""
    lit = json.load(open(os.path.join(ROOT, "tests", "golden", "literal_sequential.json")))["flights"]
    prod = json.load(open(os.path.join(ROOT, "tests", "golden", "sequential_f1.json")))["flights"]
    assert {k: lit["config"][k] for k in ("iters", "mh", "particles")} == {k: prod["config"][k] for k in ("iters", "mh", "particles")}
    assert sorted(lit["runs"]) == sorted(prod["runs"]) and len(lit["runs"]) >= 8
    a = np.array([lit["runs"][s]["f1"] for s in sorted(lit["runs"])])
    b = np.array([prod["runs"][s]["f1"] for s in sorted(prod["runs"])])
    assert abs(a.mean() - b.mean()) <= 0.005, (a.mean(), b.mean())
    for sd in lit["runs"]:
        assert lit["runs"][sd]["latent_rows"] == prod["runs"][sd]["latent_rows"], sd


def test_rents_literal_sampler_runs_and_keeps_its_database_consistent():
    """oracle/literal_inference_rents.py on 1500 rows (MH and PG): reference counts and the referring-row index consistent
    after the initialisation and after a sweep; every row ends with a room type and a unit; the sweep improves F1; the
    learned means of the keys that are looked up moved from their prior towards the data."""
    import literal_sequential_reference as lsr
    import literal_inference_rents as LR
    from pclean_amd import experiments as ex
    dirty, clean = ex.rents_data()
    dirty = {c: v[:1500] for c, v in dirty.items()}
    clean = {c: v[:1500] for c, v in clean.items()}
    (dirty, clean), _ = ex.shuffle_rows([dirty, clean], 2)
    m = ex.rents_model(dirty)
    q = ex.rents_query(m)
    for mh, particles in ((True, 2), (False, 5)):
        s = LR.RentsLiteralSampler(m, q, dirty, lsr.Cfg(1, particles, mh, rejuv=500), 2)
        s.initialize()
        s.check()
        f0 = s.accuracy(dirty, clean)["f1"]
        s.sweep()
        s.check()
        acc = s.accuracy(dirty, clean)
        assert acc["f1"] > f0 > 0.3, (f0, acc)
        assert all(o is not None and o["br"] in ("studio", "1br", "2br", "3br", "4br") for o in s.own)
        seen = [i for i in range(s.n) if dirty["Room Type"][i] is not None]
        assert all(s.own[i]["br"] == dirty["Room Type"][i] for i in seen)  # an observed own choice is never changed
        vals = np.array(list(s.means.values()))
        assert len(vals) > 100 and 800 < np.median(vals) < 2200


@pytest.mark.parametrize("name", ["rents", "rents_pg20"])
def test_rents_product_sequential_runs_match_the_literal_reference(name):
    """BASELINE.json configs[2] (rents, PG-20) and the MH configuration of the experiment script: the literal rents sampler
    and the product's sequential-schedule runs, same seeds and shuffles, agree within +-0.5 pt of F1 on the means and
    within 2 % on the number of latent counties."""
    lit = json.load(open(os.path.join(ROOT, "tests", "golden", "literal_sequential.json")))[name]
    prod = json.load(open(os.path.join(ROOT, "tests", "golden", "sequential_f1.json")))[name]
    assert {k: lit["config"][k] for k in ("iters", "mh", "particles")} == {k: prod["config"][k] for k in ("iters", "mh", "particles")}
    assert sorted(lit["runs"]) == sorted(prod["runs"])
    a = np.array([lit["runs"][s]["f1"] for s in sorted(lit["runs"])])
    b = np.array([prod["runs"][s]["f1"] for s in sorted(prod["runs"])])
    assert abs(a.mean() - b.mean()) <= 0.005, (a.mean(), b.mean())
    for sd in lit["runs"]:
        ca, cb = lit["runs"][sd]["latent_rows"]["County"], prod["runs"][sd]["latent_rows"]["County"]
        assert abs(ca - cb) <= 0.02 * cb, (sd, ca, cb)


def test_synthetic_shape_stands_between_two_references():
    """The headline workload's shape (synthetic hospital program, PG-20).  F1 on this program is heavy-tailed — most seeds
    finish at 0.993, about a quarter lose several points to ONE wrongly cleaned shared value (a hospital's attribute is a
    column of ~100 rows) — so means over few seeds are uncertain to a point or more and the comparison is made where enough
    seeds can be run: 3 000 rows, 18 seeds, the independent literal sequential sampler against the product's sequential runs
    (tests/golden/synth3k_two_references.json).  The six 30 000-row literal runs (2.2 h each) are held to the same standard:
    two standard errors of the paired differences."""
    import numpy as np
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "synth3k_two_references.json")))
    seeds = sorted(d["literal"], key=int)
    assert len(seeds) >= 18 and seeds == sorted(d["product_sequential"], key=int)
    lit = np.array([d["literal"][s]["f1"] for s in seeds])
    seq = np.array([d["product_sequential"][s]["f1"] for s in seeds])
    diff = lit - seq
    se = diff.std(ddof=1) / np.sqrt(len(diff))
    assert abs(diff.mean()) <= 2 * se and abs(diff.mean()) <= 0.01, (diff.mean(), se)       # measured: +0.03 pt, s.e. 1.3 pt
    assert abs(np.median(lit) - np.median(seq)) <= 0.002, (np.median(lit), np.median(seq))  # measured: 0.9932 both
    bad_l, bad_s = int((lit < 0.97).sum()), int((seq < 0.97).sum())
    assert 1 <= bad_l <= len(seeds) // 2 and 1 <= bad_s <= len(seeds) // 2, (bad_l, bad_s)  # measured: 4 and 6 of 18
    for s in seeds:  # the same entities are found either way
        a, b = d["literal"][s]["latent_rows"], d["product_sequential"][s]["latent_rows"]
        assert a["Hospital"] == b["Hospital"] == 30 and abs(a["County"] - b["County"]) <= 2 and abs(a["Place"] - b["Place"]) <= 2, (s, a, b)
    # 30 000 rows, 6 seeds
    lit30 = json.load(open(os.path.join(ROOT, "tests", "golden", "literal_sequential.json")))["synth_pg20"]
    prod30 = json.load(open(os.path.join(ROOT, "tests", "golden", "sequential_f1.json")))["synth_pg20"]
    s30 = sorted(lit30["runs"], key=int)
    assert s30 == sorted(prod30["runs"], key=int) and len(s30) == 6
    d30 = np.array([lit30["runs"][s]["f1"] - prod30["runs"][s]["f1"] for s in s30])
    se30 = d30.std(ddof=1) / np.sqrt(len(d30))
    assert abs(d30.mean()) <= 2 * se30, (d30.mean(), se30)                                   # measured: -1.60 pt, s.e. 0.90 pt
    for s in s30:
        a, b = lit30["runs"][s]["latent_rows"], prod30["runs"][s]["latent_rows"]
        assert a["Hospital"] == b["Hospital"] == 300 and abs(a["County"] - b["County"]) <= 3 and abs(a["Place"] - b["Place"]) <= 3, (s, a, b)

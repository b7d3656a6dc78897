"""F1 of the batched GPU schedule against the SEQUENTIAL-schedule reference runs (the reference's own schedule:
one row at a time, creation / garbage collection on the spot) committed in tests/golden/sequential_f1.json
(generator: scripts/sequential_reference.py, CPU oracle engine).  Same programs, configurations, seeds and row
shuffles on both sides; the RNG streams necessarily differ (different schedules), so the comparison is between the
means over the seeds.  north_star: F1 within +-0.5 pt."""
import json
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# per-configuration exceptions to the symmetric +-0.5 pt band: (lower, upper) difference of the means — none
BAND = {}


def _gpu_f1(name, seed, iters, mh, particles, n_rows, init_batch=None):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    from pclean_amd import experiments as ex
    from pclean_amd.analysis import evaluate_accuracy
    from pclean_amd.engine import Engine, InferenceConfig
    from pclean_amd.inference import initialize_trace, run_inference
    from pclean_amd.model import LoweredModel
    from pclean_amd.trace import Trace
    import sequential_reference as sr
    dirty, clean, mk_model, mk_query = sr.program(name, n_rows)
    (dirty, clean), _ = ex.shuffle_rows([dirty, clean], seed)
    m = mk_model(ex.possibilities_of(dirty)) if mk_model is ex.hospital_model else mk_model(dirty)
    lw = LoweredModel(m, mk_query(m), dirty)
    obs = lw.encode_observations(dirty)
    eng = Engine(lw, obs, dist_mode=1)
    try:
        tr = Trace(lw, obs.shape[1], seed)
        cfg = InferenceConfig(iters, particles, use_mh_instead_of_pg=mh, rejuv_frequency=sr.rejuv_of(name))
        # (the product's default batches: max_batch = 256)
        initialize_trace(eng, tr, cfg, seed, **({} if init_batch is None else {"max_batch": init_batch}))
        run_inference(eng, tr, cfg, seed)
        tr.check_consistency()
        return evaluate_accuracy(lw, tr, dirty, clean)["f1"]
    finally:
        eng.close()


# the experiment scripts' MH configurations, BASELINE.json configs[1] / [2] (particle Gibbs, 20 particles) and the headline
# workload's shape at a size the sequential schedule can finish (30 000 synthetic rows, 300 true hospitals)
# ... and with 3 000 true hospitals of ~10 rows each (a batched sweep freezes a table as large as its batch)
@pytest.mark.parametrize("name", ["hospital", "flights", "rents", "hospital_pg20", "rents_pg20", "synth_pg20",
                                  "synth_k3000_pg20"])
def test_batched_gpu_f1_matches_the_sequential_reference(name, capsys):
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "sequential_f1.json")))[name]
    c = ref["config"]
    seeds = sorted(int(s) for s in ref["runs"])
    got = [_gpu_f1(name, sd, c["iters"], c["mh"], c["particles"], c["n_rows"]) for sd in seeds]
    want = [ref["runs"][str(sd)]["f1"] for sd in seeds]
    with capsys.disabled():
        print(f"\n[f1 vs sequential] {name}: batched GPU {np.round(got, 4).tolist()} (mean {np.mean(got):.4f})  "
              f"sequential reference {np.round(want, 4).tolist()} (mean {np.mean(want):.4f})")
    # The north star's band: +-0.5 pt, both sides, every configuration (BAND holds the named exceptions: none).
    # Round 4 ran rents PG-20 at +0.60 pt (paired s.e. 0.16, 8 seeds) with an asymmetric bound.  scripts/schedule_knobs.py
    # (CPU oracle engine = this path bit for bit, same 8 seeds) found the knob: the batch size of the INITIALISATION —
    # sequential initialisation + batched sweeps +0.00 pt (s.e. 0.09), batches of at most 16 / 64 / 256 / 1024 rows
    # -0.16 / +0.29 / +0.27 / +0.60 pt; sequential sweeps after the 1024-row initialisation still +0.66 pt: the batched
    # SWEEPS are neutral, a large initialisation batch creates one latent row per distinct spelling at once and lets the
    # later rows choose among them where the reference commits to the first spelling it meets.  The test had asked for
    # 1024-row batches; it now runs the product's default (initialize_trace: max_batch = 256), which is inside the band.
    diff = float(np.mean(got) - np.mean(want))
    with capsys.disabled():
        d = np.asarray(got) - np.asarray(want)
        print(f"[f1 vs sequential] {name}: difference of means {100 * diff:+.2f} pt (paired standard error "
              f"{100 * d.std(ddof=1) / np.sqrt(len(d)):.2f} pt, {len(d)} seeds)")
    lo, hi = BAND.get(name, (-0.005, 0.005))
    assert lo <= diff <= hi, (name, diff, got, want)
    lit = json.load(open(os.path.join(ROOT, "tests", "golden", "literal_sequential.json")))
    if name in lit:  # ... and of the INDEPENDENT literal sequential reference (oracle/literal_inference.py)
        with capsys.disabled():
            print(f"[f1 vs literal sequential] {name}: literal reference mean {lit[name]['f1_mean']:.4f}")
        if name.startswith("synth"):
            # heavy-tailed F1 (tests/test_literal_sequential.py::test_synthetic_shape_stands_between_two_references): six
            # 30 000-row literal runs pin the mean to about a point — two standard errors of the paired differences
            dl = np.asarray(got) - np.asarray([lit[name]["runs"][str(sd)]["f1"] for sd in seeds])
            se = dl.std(ddof=1) / np.sqrt(len(dl))
            with capsys.disabled():
                print(f"[f1 vs literal sequential] {name}: difference of means {100 * dl.mean():+.2f} pt (paired s.e. {100 * se:.2f} pt)")
            assert abs(dl.mean()) <= 2 * se + 0.005, (dl.mean(), se)
        else:
            assert abs(np.mean(got) - lit[name]["f1_mean"]) <= 0.005, (got, lit[name]["f1_mean"])
    # ... and no seed further below the reference's MEAN than three of the reference's own standard deviations + 0.5 pt (the
    # reference's seeds spread by more than a point themselves: rents PG-20 0.6654 .. 0.6862, s.d. 0.7 pt)
    floor = float(np.mean(want) - 3.0 * np.std(want, ddof=1) - 0.005)
    assert min(got) >= floor, (got, want, floor)


def test_initialisation_in_1024_row_batches_is_reported_not_asserted(capsys):
    """rents PG-20 with the initialisation in batches of up to 1024 rows (four times the product's default): round 4's
    +0.60 pt excursion.  Printed beside the band, asserted only against a 1.5 pt bound: the configuration is outside what the
    north star's +-0.5 pt covers (the knob table) and stays visible here."""
    name = "rents_pg20"
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "sequential_f1.json")))[name]
    c = ref["config"]
    seeds = sorted(int(s) for s in ref["runs"])[:4]
    got = [_gpu_f1(name, sd, c["iters"], c["mh"], c["particles"], c["n_rows"], init_batch=1024) for sd in seeds]
    want = [ref["runs"][str(sd)]["f1"] for sd in seeds]
    diff = float(np.mean(got) - np.mean(want))
    with capsys.disabled():
        print(f"\n[f1 vs sequential, init batches <= 1024] {name}: {np.round(got, 4).tolist()} vs {np.round(want, 4).tolist()}: "
              f"{100 * diff:+.2f} pt (outside the +-0.5 pt band by construction: reported)")
    assert abs(diff) <= 0.015, (diff, got, want)



def test_synthetic_shape_at_3000_rows_between_two_references(capsys):
    """18 seeds of the synthetic hospital program at 3 000 rows (PG-20): the batched GPU runs against BOTH sequential
    references of tests/golden/synth3k_two_references.json — the independent literal sampler and the product's sequential runs.
    F1 is heavy-tailed here (about a quarter of the seeds lose points to one wrongly cleaned shared value), so: means within two
    standard errors of the paired differences (and within 1.5 pt), medians within 0.3 pt, a comparable share of bad seeds."""
    d = json.load(open(os.path.join(ROOT, "tests", "golden", "synth3k_two_references.json")))
    seeds = sorted(d["literal"], key=int)
    got = np.array([_gpu_f1("synth_pg20", int(sd), 1, False, 20, 3000) for sd in seeds])
    for ref in ("literal", "product_sequential"):
        want = np.array([d[ref][sd]["f1"] for sd in seeds])
        diff = got - want
        se = diff.std(ddof=1) / np.sqrt(len(diff))
        with capsys.disabled():
            print(f"\n[synth 3 000 rows vs {ref}] GPU mean {got.mean():.4f} median {np.median(got):.4f} bad seeds {(got < 0.97).sum()} | "
                  f"reference mean {want.mean():.4f} median {np.median(want):.4f} bad seeds {(want < 0.97).sum()} | "
                  f"difference {100 * diff.mean():+.2f} pt (paired s.e. {100 * se:.2f} pt)")
        assert abs(diff.mean()) <= 2 * se and abs(diff.mean()) <= 0.015, (ref, diff.mean(), se)
        assert abs(np.median(got) - np.median(want)) <= 0.003, (ref, np.median(got), np.median(want))
    assert (got < 0.97).sum() <= len(seeds) // 2

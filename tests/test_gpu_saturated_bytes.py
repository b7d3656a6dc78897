"""The compact byte tables of the wave kernel saturate at 42 edits (root_wave.hip: PRE_CLAMP); a survivor whose
distance on a term is >= 42 has to fetch the true distance from the pair table.  In the three BASELINE programs the
long strings are exactly the three pre-filter terms, so that path never runs there; this model has FIVE attributes of
60-70 characters and rows whose fourth / fifth cell was replaced by an unrelated string (distance > 42) while the
others match: those candidates pass the pre-filter and need the fallback.  Fast path vs generic kernel, bit for bit,
and against the CPU oracle."""
import numpy as np
import pytest

import helpers
from pclean_amd import _lib
from pclean_amd.engine import Engine, InferenceConfig
from pclean_amd.model import AddTypos, LoweredModel, Model, Query, StringPrior
from pclean_amd.trace import Trace

pytestmark = pytest.mark.gpu
FIELDS = ["a", "b", "c", "d", "e"]


def long_string_setup(n_ent=1300, rows_per=3, seed=4):
    rng = np.random.default_rng(seed)
    letters = np.array(list("abcdefghijklmnopqrstuvwxyz "))

    def rand_str(lo, hi):
        return "".join(rng.choice(letters, int(rng.integers(lo, hi + 1)))).strip() or "x"

    ents = [{f: rand_str(60, 70) for f in FIELDS} for _ in range(n_ent)]
    dirty = {f.upper(): [] for f in FIELDS}
    owner = []
    for k, e in enumerate(ents):
        for r in range(rows_per):
            row = dict(e)
            kind = (k * rows_per + r) % 7
            if kind == 1:
                row["d"] = rand_str(60, 70)           # unrelated string in a non-pre-filter cell (distance > 42)
            elif kind == 2:
                row["e"] = rand_str(60, 70)
            elif kind == 3:
                row["d"], row["e"] = rand_str(60, 70), rand_str(60, 70)
            elif kind == 4:
                s_ = list(row["a"])
                s_[5], s_[17] = "q", "z"                # two typos in a pre-filter cell
                row["a"] = "".join(s_)
            elif kind == 5:
                row["b"] = None                        # missing cell
            for f in FIELDS:
                dirty[f.upper()].append(row[f])
            owner.append(k)
    m = Model()
    c = m.add_class("Ent")
    for f in FIELDS:
        c.choice(f, StringPrior(50, 80, list(dict.fromkeys(v for v in dirty[f.upper()] if v is not None))))
    o = m.add_class("Obs")
    with o.block():
        o.fk("ent", "Ent")
        for f in FIELDS:
            o.choice(f, AddTypos("ent." + f))
    q = Query(m, "Obs", {f.upper(): ("ent." + f, f) for f in FIELDS})
    lw = LoweredModel(m, q, dirty)
    obs = lw.encode_observations(dirty)
    n = obs.shape[1]
    by_path = {0: {f: [ents[k][f] for k in owner] for f in FIELDS}}
    for f in FIELDS:  # a clean value that never occurs undamaged is not a possible latent value: use the dirty cell
        dom = lw.latent_dom[("Ent", f)]
        by_path[0][f] = [v if dom.get(v) >= 0 else (d if d is not None else dom.string(0))
                         for v, d in zip(by_path[0][f], dirty[f.upper()])]
    tr = Trace.from_clean_values(lw, by_path, n, seed)
    return dirty, lw, obs, tr


def test_saturated_compact_bytes_fall_back_to_the_pair_table(oracle):
    dirty, lw, obs, tr = long_string_setup()
    assert tr.tables["Ent"].n >= 1024  # the wave kernel takes reference slots from 1024 candidates
    eng = Engine(lw, obs, dist_mode=_lib.DIST_OSA)
    try:
        cfg = InferenceConfig(1, 4)
        eng.upload_trace(tr)
        eng.hip.force_generic(False)
        a = eng.sweep(tr, cfg, 11, 0)
        rs = eng.hip.get_root_stats()
        assert rs.fast == 1 and rs.n_pre == 3
        eng.hip.force_generic(True)
        b = eng.sweep(tr, cfg, 11, 0)
        eng.hip.force_generic(False)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
        assert set(a[3]) == set(b[3])
        for k in a[3]:
            assert np.array_equal(a[3][k][0], b[3][k][0]) and np.array_equal(a[3][k][1], b[3][k][1])
        # the oracle agrees (scores through the pair tables only)
        world = helpers.mirror_world(oracle, lw, obs, tr, eng)
        import ctypes as C
        from pclean_amd._lib import InferConfig
        nb, n = tr.cur.shape
        choice, chosen, logml = np.empty((nb, n), np.int32), np.empty(n, np.int32), np.empty(n)
        c = InferConfig(1, 4, 1, 1, 0, 50, 100)
        oracle.lib().pco_sweep_batched(world.h, C.byref(c), C.c_uint64(11), C.c_uint32(0), nb, C.c_int64(0),
                                       oracle._p(np.ascontiguousarray(tr.cur), C.c_int32), oracle._p(choice, C.c_int32),
                                       oracle._p(chosen, C.c_int32), oracle._p(logml, C.c_double))
        assert np.array_equal(a[0], choice) and np.array_equal(a[1], chosen) and np.array_equal(a[2], logml)
        # the scenario is real: rows whose d / e cell is more than 42 edits from their referent's value
        ddom = lw.latent_dom[("Ent", "d")]
        far = 0
        for i in range(0, n - 1, 7):
            if dirty["D"][i + 1] is not None:
                ref = ddom.string(int(tr.tables["Ent"].cols[lw.colidx["Ent"]["d"], tr.cur[0, i + 1]]))
                far += sum(x != y for x, y in zip(ref, dirty["D"][i + 1])) > 42
        assert far > 50
    finally:
        eng.close()

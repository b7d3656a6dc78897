"""A fourth program (TEST INFRASTRUCTURE) that uses what the three experiment programs do not: a block of the observed
class with THREE reference slots, a single-argument JuliaNode, JuliaNodes across the slots of one block (two different
earlier values read by the third slot: two context slots), and latent classes that sit on the context side of several
cross-slot JuliaNodes (several per-evidence-row context sources).  Deterministic data with typos."""
import numpy as np

from pclean_amd.model import AddTypos, ChooseUniformly, LoweredModel, Model, Query, StringPrior

DOCTORS = [("alice marsh", "cardio"), ("bob lentil", "neuro"), ("carla voss", "cardio"), ("dmitri ozol", "ortho"),
           ("erin fay", "neuro"), ("farid khan", "derma")]
CLINICS = [("boston", "b12"), ("austin", "a07"), ("denver", "d33"), ("salem", "s05")]
ROOMS = ["exam", "lab", "scan"]
SPECS = ["cardio", "neuro", "ortho", "derma"]
CITIES = [c for c, _ in CLINICS]
CODES = [k for _, k in CLINICS]


def tag(code):
    return "C-" + code


def combo(spec, city):
    return spec[:3] + "@" + city


def rk(spec, kind):
    return kind + "/" + spec


def ck(city, kind):
    return city[:2] + ":" + kind


def _typo(rng, s):
    if rng.random() < 0.25 and len(s) > 2:
        i = int(rng.integers(len(s)))
        return s[:i] + "x" + s[i + 1:]
    return s


def clinic_program(n_rows=120, seed=5):
    rng = np.random.default_rng(seed)
    m = Model()
    c = m.add_class("Doctor")
    c.choice("name", StringPrior(3, 20, [d for d, _ in DOCTORS]))
    c.choice("spec", ChooseUniformly(SPECS))
    c = m.add_class("Clinic")
    c.choice("city", ChooseUniformly(CITIES))
    c.choice("code", ChooseUniformly(CODES))
    c = m.add_class("Room")
    c.choice("kind", ChooseUniformly(ROOMS))
    r = m.add_class("Visit")
    with r.block():
        r.fk("doctor", "Doctor")
        r.fk("clinic", "Clinic")
        r.fk("room", "Room")
        r.choice("dname", AddTypos("doctor.name"))
        r.choice("dspec", AddTypos("doctor.spec"))
        r.choice("ccity", AddTypos("clinic.city"))
        r.julia("tag", tag, ["clinic.code"])
        r.choice("tag_obs", AddTypos("tag"))
        r.julia("combo", combo, ["doctor.spec", "clinic.city"])
        r.choice("combo_obs", AddTypos("combo"))
        r.choice("rkind", AddTypos("room.kind"))
        r.julia("rk", rk, ["doctor.spec", "room.kind"])
        r.choice("rk_obs", AddTypos("rk"))
        r.julia("ck", ck, ["clinic.city", "room.kind"])
        r.choice("ck_obs", AddTypos("ck"))
    q = Query(m, "Visit", {"DName": ("doctor.name", "dname"), "DSpec": ("doctor.spec", "dspec"),
                           "City": ("clinic.city", "ccity"), "Tag": ("tag", "tag_obs"), "Combo": ("combo", "combo_obs"),
                           "Kind": ("room.kind", "rkind"), "RK": ("rk", "rk_obs"), "CK": ("ck", "ck_obs")})
    clean = {k: [] for k in q.obsmap}
    for i in range(n_rows):
        (dn, sp), (city, code), kind = DOCTORS[int(rng.integers(len(DOCTORS)))], CLINICS[int(rng.integers(len(CLINICS)))], \
            ROOMS[int(rng.integers(len(ROOMS)))]
        for k, v in (("DName", dn), ("DSpec", sp), ("City", city), ("Tag", tag(code)), ("Combo", combo(sp, city)), ("Kind", kind),
                     ("RK", rk(sp, kind)), ("CK", ck(city, kind))):
            clean[k].append(v)
    dirty = {k: [None if rng.random() < 0.04 else _typo(rng, v) for v in vals] for k, vals in clean.items()}
    lw = LoweredModel(m, q, dirty)
    obs = lw.encode_observations(dirty)
    return dict(model=m, query=q, dirty=dirty, clean=clean, lw=lw, obs=obs)

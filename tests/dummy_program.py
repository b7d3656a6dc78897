"""A small fourth program whose StringPrior atoms do NOT cover the observations (TEST INFRASTRUCTURE): the
ProposalDummyValue of `Person.name` competes with atoms that are several edits away from short observed strings,
so fresh particles regularly choose it WITH an observation below the node — the case block_proposal.jl:58-60 handles
by drawing random(StringPrior) and scoring the observation on the drawn string."""
import numpy as np

from pclean_amd.model import AddTypos, LoweredModel, Model, Query, StringPrior

ATOMS = ["ann", "bob", "cy", "dee"]
OBSERVED = ["zed", "al", "ann", "bo", "xx", None, "q", "dee", "bobby", "cyy", "zz", "ann", "k", None, "mo", "ed"]


def people_program(n_rows=64):
    m = Model()
    c = m.add_class("Person")
    c.choice("name", StringPrior(1, 6, ATOMS))
    r = m.add_class("Obs")
    with r.block():
        r.fk("person", "Person")
        r.choice("name_obs", AddTypos("person.name"))
    q = Query(m, "Obs", {"Name": ("person.name", "name_obs")})
    dirty = {"Name": [OBSERVED[i % len(OBSERVED)] for i in range(n_rows)]}
    lw = LoweredModel(m, q, dirty)
    obs = lw.encode_observations(dirty)
    return m, q, dirty, lw, obs

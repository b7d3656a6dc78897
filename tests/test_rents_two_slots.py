"""CPU checks of tests/rents_two_slots.py (the program whose Gaussian block is followed by another reference-slot block):
the lowering gives two engine blocks with the Gaussian term on the first, and the oracle's prior-proposal sweeps
(use_dd_proposals = false) under particle Gibbs resample between the blocks — the case in which a particle's own choices have
to follow it (oracle/sweep.h; the HIP side is held against it by tests/test_gpu_edges.py)."""
import ctypes as C

import numpy as np

import helpers
import rents_two_slots as r2
from pclean_amd._lib import InferConfig


def _sweep(oracle, world, cfg, seed, sweep, cur):
    nb, n = cur.shape
    choice = np.empty((nb, n), dtype=np.int32)
    chosen = np.empty(n, dtype=np.int32)
    logml = np.empty(n)
    oracle.lib().pco_sweep_batched(world.h, C.byref(cfg), C.c_uint64(seed), C.c_uint32(sweep), nb, C.c_int64(0),
                                   oracle._p(np.ascontiguousarray(cur), C.c_int32), oracle._p(choice, C.c_int32),
                                   oracle._p(chosen, C.c_int32), oracle._p(logml, C.c_double))
    return choice, chosen, logml


def test_two_slot_rents_lowering_and_oracle_prior_sweeps(oracle):
    S = r2.setup(n_rows=160)
    lw, obs, tr = S["lw"], S["obs"], S["trace"]
    assert len(lw.engine_blocks) == 2 and lw.engine_blocks[1] == ["landlord", "landlord_obs"]
    assert any(bid == 0 for (bid, _nid) in lw.gauss) and not any(bid == 1 for (bid, _nid) in lw.gauss)
    n = obs.shape[1]
    world = helpers.mirror_world(oracle, lw, obs, tr, option_logp=helpers.option_logp_cpu(oracle, lw, tr))
    world.set_cur_locals(0, tr.locals[0])
    cfg = InferConfig(1, 12, 0, 1, 0, 50, 100)  # PG, 12 particles, prior proposals
    a = _sweep(oracle, world, cfg, 31, 2, tr.cur)
    loc_a = world.get_locals(0, n).copy()
    b = _sweep(oracle, world, cfg, 31, 2, tr.cur)
    assert all(np.array_equal(x, y) for x, y in zip(a, b)) and np.array_equal(loc_a, world.get_locals(0, n))
    assert (a[1] > 0).sum() > n // 4          # chosen particles other than the retained one: the blocks were resampled between
    assert np.all(loc_a[:, 0] >= 0) and np.all(loc_a[:, 1] >= 0)  # the chosen particle's own choices came along
    room = obs[3]
    assert np.array_equal(loc_a[room >= 0, 0], room[room >= 0])   # an observed room type is kept, not sampled
    c = _sweep(oracle, world, InferConfig(1, 12, 0, 1, 0, 50, 100), 32, 2, tr.cur)
    assert not np.array_equal(a[1], c[1])     # another seed: other draws

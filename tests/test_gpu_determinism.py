"""Run-to-run and path-to-path determinism of the observed-class sweep (advisor r5): the hash grouping numbers its groups in
the order the workgroups arrive, the gate of the new-row branch pauses by its own history, the extra context items of a block
are numbered by an atomic counter — every consumer must be independent of all that.  The same sweeps (a 40 000-row synthetic
table from its own initialisation: several thousand groups, moved rows, new referents; sweep + host commit, three times) run
in fresh processes under
  * the product's defaults, twice,
  * PCLEAN_SORT_GROUPS=1 (radix-sort grouping), PCLEAN_GATE_ALWAYS=1 (no gate pauses),
  * the round-5 paths of this round's changes (PCLEAN_NO_LAZY_DRAWS, PCLEAN_NO_UNIFORM_W, PCLEAN_NO_SMALL_GENERIC,
    PCLEAN_NO_FUSED_CTX_ITEMS),
and every output (chosen referents, chosen particles, log marginal likelihood estimates, new-row records, the committed
state) must be bit-identical: one SHA-256 over all of them per process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import hashlib, sys
import numpy as np
sys.path[:0] = [ROOT, ROOT + "/tests"]
import helpers
from pclean_amd.engine import Engine, InferenceConfig
from pclean_amd.inference import initialize_trace
from pclean_amd.parallel import Comm, exchange_and_commit
from pclean_amd.trace import Trace
dirty, clean, lw, obs, _ = helpers.truth_workload(40000, 400, 11)
eng = Engine(lw, obs)
h = hashlib.sha256()
try:
    cfg = InferenceConfig(1, 20)
    tr = Trace(lw, obs.shape[1], 5)
    initialize_trace(eng, tr, cfg, 5, max_batch=4096)
    for sweep in range(3):
        eng.upload_trace(tr)
        choice, chosen, logml, new_rows = eng.sweep(tr, cfg, 77, sweep)
        stats = eng.sweep_stats(tr)
        for a in (choice, chosen, logml):
            h.update(np.ascontiguousarray(a).tobytes())
        for b in sorted(new_rows):
            h.update(np.ascontiguousarray(new_rows[b][0]).tobytes())
            h.update(np.ascontiguousarray(new_rows[b][1]).tobytes())
        exchange_and_commit(tr, lw, Comm(), 0, choice, stats, new_rows)
        h.update(np.ascontiguousarray(tr.cur).tobytes())
        for c in sorted(tr.tables):
            t = tr.tables[c]
            h.update(np.ascontiguousarray(t.counts[:t.n]).tobytes())
    print("DIGEST", h.hexdigest(), int((choice != tr.cur).sum()), {c: int(t.n_live) for c, t in tr.tables.items()})
finally:
    eng.close()
'''


def _run(extra_env):
    env = dict(os.environ)
    env.update(extra_env)
    out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + SCRIPT], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][-1]
    return line.split()[1], line


def test_sweeps_are_bit_identical_across_runs_and_paths(capsys):
    variants = {
        "default": {},
        "default again": {},
        "sorted groups": {"PCLEAN_SORT_GROUPS": "1"},
        "gate always": {"PCLEAN_GATE_ALWAYS": "1"},
        "round-5 paths": {"PCLEAN_NO_LAZY_DRAWS": "1", "PCLEAN_NO_UNIFORM_W": "1", "PCLEAN_NO_SMALL_GENERIC": "1",
                          "PCLEAN_NO_FUSED_CTX_ITEMS": "1"},
        "generic kernels": {"PCLEAN_NO_DEDUP": "1", "PCLEAN_NO_GATE": "1", "PCLEAN_NO_MEMO": "1"},
        # a first counter bank of 2 slots: every later counted launch takes its counter from a grown bank (fresh_counter)
        "counter bank exhausted": {"PCLEAN_CTR_BANK": "2"},
        "two-level scans only": {"PCLEAN_NO_DENSE_SCAN": "1"},
        # the sweep's delta reference counts by plain atomics for every table (finalize_block_kernel without its LDS histogram)
        "delta counts without the histogram": {"PCLEAN_NO_HIST": "1"},
        "read-backs by copies, separate alive kernel, hipMemsetAsync": {"PCLEAN_NO_PUBLISH_REGIONS": "1", "PCLEAN_NO_FUSED_PRIORS": "1",
                                                                        "PCLEAN_NO_ZERO_KERNEL": "1"},
    }
    digests = {}
    for name, env in variants.items():
        digests[name], line = _run(env)
        with capsys.disabled():
            print(f"\n[determinism] {name}: {line}")
    assert len(set(digests.values())) == 1, digests


# ---- a whole run_inference (observed class on the device-resident commit, every latent class with its sub-batches, host
# commits, parameter moves) under this round's latent-path switches ----------------------------------------------------------
SCRIPT_FULL = r'''
import hashlib, sys
import numpy as np
sys.path[:0] = [ROOT, ROOT + "/tests"]
import helpers
from pclean_amd.engine import Engine, InferenceConfig
from pclean_amd.inference import initialize_trace, run_inference
from pclean_amd.trace import Trace
dirty, clean, lw, obs, _ = helpers.truth_workload(20000, 200, 13)
eng = Engine(lw, obs)
h = hashlib.sha256()
try:
    cfg = InferenceConfig(2, 5)
    tr = Trace(lw, obs.shape[1], 3)
    initialize_trace(eng, tr, cfg, 3, max_batch=2048)
    run_inference(eng, tr, cfg, 3)
    tr.check_consistency()
    h.update(np.ascontiguousarray(tr.cur).tobytes())
    for c in sorted(tr.tables):
        t = tr.tables[c]
        h.update(np.ascontiguousarray(t.counts[:t.n]).tobytes())
        h.update(np.ascontiguousarray(t.cols[:, :t.n]).tobytes())
    print("DIGEST", h.hexdigest(), {c: int(t.n_live) for c, t in tr.tables.items()})
finally:
    eng.close()
'''


def test_full_inference_is_bit_identical_across_the_latent_path_switches(capsys):
    variants = {
        "default": {},
        "no changed-row deltas of re-uploaded tables": {"PCLEAN_NO_UPLOAD_DELTA": "1"},
        "reference slots of latent sweeps by the generic kernel": {"PCLEAN_NO_FAST_EV_SLOTS": "1"},
        "evidence scans for slots only from 1024 items on": {"PCLEAN_EV_SLOT_MIN_ITEMS": "1024"},
        "common prior in the evidence cut": {"PCLEAN_NO_EV_PRIOR_CUT": "1"},
        "no evidence scans at all": {"PCLEAN_NO_FAST_EV": "1"},
        "weighted sums of huge evidence sets by the row's own workgroup": {"PCLEAN_NO_EV_SPLIT": "1"},
        "read-backs by copies, hipMemsetAsync": {"PCLEAN_NO_PUBLISH_REGIONS": "1", "PCLEAN_NO_ZERO_KERNEL": "1", "PCLEAN_NO_FUSED_PRIORS": "1"},
        "evidence sets built on the device": {"PCLEAN_DEVICE_EVIDENCE_MIN_ROWS": "0"},
        "compact tables rebuilt, not refreshed along the chain of upload deltas": {"PCLEAN_NO_DELTA_CHAIN": "1"},
    }
    digests = {}
    for name, env in variants.items():
        e = dict(os.environ)
        e.update(env)
        out = subprocess.run([sys.executable, "-c", "ROOT = %r\n" % ROOT + SCRIPT_FULL], env=e, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][-1]
        digests[name] = line.split()[1]
        with capsys.disabled():
            print(f"\n[determinism, full inference] {name}: {line}")
    assert len(set(digests.values())) == 1, digests

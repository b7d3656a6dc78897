"""pclean_comm_* / pclean_allreduce_stats (the exchange a torch-less host would use), checked in a process
of its own: with a one-rank communicator the all-reduce is the identity on the delta reference counts of
the last sweep.  (Kept out of the pytest session on purpose: RCCL initialises process-wide state.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import helpers
from pclean_amd._lib import PCleanHipError
from pclean_amd.engine import Engine, InferenceConfig

S = helpers.hospital_setup(n_rows=200)
lw, tr = S["lw"], S["trace"]
eng = Engine(lw, S["obs"], dist_mode=1)
eng.upload_trace(tr)
try:
    eng.hip.allreduce_stats(lw.table_id["Hospital"], tr.tables["Hospital"].n)
    raise SystemExit("expected an error without a communicator")
except PCleanHipError:
    pass
uid = eng.hip.comm_unique_id()
assert len(uid) == 128
eng.hip.comm_init(1, 0, uid)
eng.sweep(tr, InferenceConfig(1, 4), 3, 0)
for cname in ("Hospital", "Measure"):
    tid, n = lw.table_id[cname], tr.tables[cname].n
    before = eng.hip.get_stats(tid, n)
    after = eng.hip.allreduce_stats(tid, n)
    assert np.array_equal(before, after) and np.array_equal(eng.hip.get_stats(tid, n), before)
tids = [lw.table_id["Hospital"], lw.table_id["Measure"]]
ns = [tr.tables["Hospital"].n, tr.tables["Measure"].n]
want = [eng.hip.get_stats(t, n) for t, n in zip(tids, ns)]
got = eng.hip.allreduce_stats_fused(tids, ns)
assert all(np.array_equal(a, b) for a, b in zip(got, want))
assert all(np.array_equal(eng.hip.get_stats(t, n), b) for t, n, b in zip(tids, ns, want))
zero = eng.hip.allreduce_stats_fused(tids, ns, local_is_zero=True)
assert all(not z.any() for z in zero)
eng.hip.comm_destroy()
eng.hip.comm_destroy()
eng.close()
print("rccl comm ok")

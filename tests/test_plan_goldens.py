"""The plan IR the lowering produces for the three experiment programs equals the committed JSON
(tests/golden/plans_*.json; generator scripts/make_plan_goldens.py) — the reference output for a lowering written in
another host language (julia/PCleanHIP.jl)."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("name", ["hospital", "flights", "rents"])
def test_lowering_matches_the_committed_plans(name):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import make_plan_goldens as mk
    lw, _ = mk.lowered(name)
    got = json.loads(json.dumps(mk.describe(lw), sort_keys=True))
    want = json.load(open(os.path.join(ROOT, "tests", "golden", f"plans_{name}.json")))
    for k, v in got.items():
        assert want[k] == v, (name, k)
    assert len(want["blocks"]) >= 1 and all(len(b.get("nodes", [1])) >= 1 for b in want["blocks"])

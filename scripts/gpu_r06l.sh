#!/bin/bash
# round 6, call L: timed block = the Measure slot (roofline of the longest launch group): quick bench + the sweep tests
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06l
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dl-sample --no-steady-iterations > "$OUT/bench.json" 2> "$OUT/bench.log"
echo "bench rc=$?"; tail -n 3 "$OUT/bench.log" | cut -c1-300
python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r06l/bench.json")).read().strip().split("\n")[-1])
r=d["roofline"]
print(d["ms_per_step"], r.get("chosen"), r["frac"], r["achieved"], r["alg_bytes_per_launch"], r["avg_launch_ms"], r.get("lazy_entries"), r.get("groups"), r.get("full_scans"), r.get("fine_blocks"), r.get("scored_terms"))
b=r.get("block0_root_group") or {}
print("block0:", b.get("frac"), b.get("avg_launch_ms"), b.get("alg_bytes_per_launch"))
PY
timeout 1200 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_determinism.py tests/test_gpu_edges.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 3 "$OUT/pytest.log"

#!/bin/bash
# round 6, call J: whole GPU suite + the one-rank RCCL line with the per-rank fields
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06j
mkdir -p "$OUT"
cd "$ROOT"
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x --durations=15 > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -n 30 "$OUT/pytest.log"
PCLEAN_FORCE_DIST=1 timeout 900 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-dl-sample --no-steady-iterations > "$OUT/bench_force_dist.json" 2> "$OUT/bench_force_dist.log"
echo "force_dist rc=$?"
tail -c 600 "$OUT/bench_force_dist.log"
python - <<'PY'
import json,os
d=json.loads(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out/r06j/bench_force_dist.json")).read().strip().split("\n")[-1])
print(d["ms_per_step"], d["config"]["per_rank"], d["config"]["scaling_model"])
PY

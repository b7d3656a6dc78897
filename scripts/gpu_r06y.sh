#!/bin/bash
# round 6, call Y: candidate_score with the terms' gathers in flight four at a time: whole GPU suite, bench, sweep timeline
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06y
mkdir -p "$OUT"
cd "$ROOT"
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 3 "$OUT/pytest.log"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample > "$OUT/bench.json" 2> "$OUT/bench.log"
echo "bench rc=$?"; python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]; c=d["config"]
print("ms/step", d["ms_per_step"], "value", d["value"], "roofline", r.get("achieved"), r.get("frac"), "ms", r.get("avg_launch_ms"), "dev", c.get("device_ms_per_step"), "iter", c.get("full_iteration_ms"), c.get("full_iteration_steady_ms"), "fixed", c.get("step_fixed_ms"), c.get("step_proportional_ms"))
PY
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-dl-sample --no-steady-iterations > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.log"
echo "trace rc=$?"
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/timeline.py "$T" 0 3 > "$OUT/sweep_timeline.txt" 2>&1
find "$OUT" -name "*.db" -delete
tail -32 "$OUT/sweep_timeline.txt"

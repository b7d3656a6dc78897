#!/bin/bash
# round 6, call AP: enum_node_kernel with candidate_score_batch — parity (whole GPU suite) + kernel trace of the bench + iteration
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06ap
mkdir -p "$OUT"
cd "$ROOT"
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 4 "$OUT/pytest.log"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-dl-sample --no-steady-iterations > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.log"
echo "trace rc=$?"
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/summarize_rocpd.py "$T" "$OUT/kernel_trace.txt" > /dev/null
python profiles/timeline.py "$T" 0 3 > "$OUT/sweep_timeline.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -n "overflow_lds\|enum_node\|ctx_items\|particle_update_final\|^span" "$OUT/sweep_timeline.txt" | head -20
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); c=d["config"]
print("ms/step %.3f f1 %.4f fixed %.2f prop %.2f iter %.1f/%.1f" % (d["ms_per_step"], d["f1"], c["step_fixed_ms"], c["step_proportional_ms"], c["full_iteration_ms"], c["full_iteration_steady_ms"]))
PY

#!/bin/bash
# round 6, call AQ: gate + descriptor scores with every term in flight, histogram below the high-water mark, lazy draws with hoisted loads and an 8-ary search — parity (determinism, full size, sweep) + kernel trace of the bench
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06aq
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_fullsize.py tests/test_gpu_sweep.py tests/test_gpu_edges.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 4 "$OUT/pytest.log"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-dl-sample --no-steady-iterations > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.log"
echo "trace rc=$?"
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/summarize_rocpd.py "$T" "$OUT/kernel_trace.txt" > /dev/null
python profiles/timeline.py "$T" 0 3 > "$OUT/sweep_timeline.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -n "group_gate\|group_desc\|finalize_block\|lazy_draw\|^span" "$OUT/sweep_timeline.txt" | head
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); c=d["config"]
print("ms/step %.3f f1 %.4f fixed %.2f prop %.2f iter %.1f/%.1f" % (d["ms_per_step"], d["f1"], c["step_fixed_ms"], c["step_proportional_ms"], c["full_iteration_ms"], c["full_iteration_steady_ms"]))
PY

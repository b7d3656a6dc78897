#!/bin/bash
# round 6, call N: read-backs through one kernel, priors + alive fused: tests + bench + timeline
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06n
mkdir -p "$OUT"
cd "$ROOT"
timeout 2400 python -m pytest tests/test_gpu_commit.py tests/test_gpu_determinism.py tests/test_gpu_sweep.py tests/test_gpu_inference.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_gpu_rents.py tests/test_gpu_flights.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 4 "$OUT/pytest.log"
for V in new old; do
  E=""; [ $V = old ] && E="PCLEAN_NO_PUBLISH_REGIONS=1 PCLEAN_NO_FUSED_PRIORS=1 PCLEAN_NO_ZERO_KERNEL=1"
  env $E timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample --distance osa > "$OUT/bench_$V.json" 2> "$OUT/bench_$V.log"
  python - "$OUT/bench_$V.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); c=d["config"]
print(sys.argv[1].split("/")[-1], "ms/step %.3f fixed %.2f prop %.2f iter %.0f/%.0f" % (d["ms_per_step"], c["step_fixed_ms"], c["step_proportional_ms"], c["full_iteration_ms"], c["full_iteration_steady_ms"]))
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-dl-sample --no-steady-iterations --distance osa > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.log"
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/timeline.py "$T" 0 3 > "$OUT/sweep_timeline.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -n "^span" "$OUT/sweep_timeline.txt"

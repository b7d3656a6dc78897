#!/bin/bash
# round 6, call Z: wide enumeration workgroups, evidence entries four at a time: tests, iteration profile, bench, timeline
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06z
mkdir -p "$OUT"
cd "$ROOT"
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x --deselect tests/test_gpu_fullsize.py > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 3 "$OUT/pytest.log"
for V in new; do
  timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter_$V.log" 2> "$OUT/iter_$V.err"
  echo "$V rc=$?"; grep -v "^\[pclean\]" "$OUT/iter_$V.log" | grep "full iteration\|^Hospital\|^County\|^Place\|^Measure" | cut -c1-420
done
for V in new nowide; do
  E="X=1"; [ $V = nowide ] && E="PCLEAN_NO_WIDE_ENUM=1"
  env $E timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample > "$OUT/bench_$V.json" 2> "$OUT/bench_$V.log"
  echo "bench $V rc=$?"; python - "$OUT/bench_$V.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]; c=d["config"]
print("ms/step", d["ms_per_step"], "value", d["value"], "roofline", r.get("achieved"), r.get("frac"), "ms", r.get("avg_launch_ms"), "dev", c.get("device_ms_per_step"), "iter", c.get("full_iteration_ms"), c.get("full_iteration_steady_ms"), "fixed", c.get("step_fixed_ms"), c.get("step_proportional_ms"))
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-dl-sample --no-steady-iterations > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.log"
echo "trace rc=$?"
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/timeline.py "$T" 0 3 > "$OUT/sweep_timeline.txt" 2>&1
find "$OUT" -name "*.db" -delete
tail -30 "$OUT/sweep_timeline.txt"

#!/bin/bash
# round 6, call X: unsplit lazy groups A/B on the headline state + sweep timeline
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06x
mkdir -p "$OUT"
cd "$ROOT"
for V in new lazysplit; do
  E="X=1"; [ $V = lazysplit ] && E="PCLEAN_LAZY_SPLIT=1"
  env $E timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample --no-steady-iterations > "$OUT/bench_$V.json" 2> "$OUT/bench_$V.log"
  echo "bench $V rc=$?"; python - "$OUT/bench_$V.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("ms/step", d["ms_per_step"], "value", d["value"], "roofline", r.get("achieved"), r.get("frac"), "ms", r.get("avg_launch_ms"), "groups", r.get("groups"), "dev", d["config"].get("device_ms_per_step"), "iter", d["config"].get("full_iteration_ms"))
PY
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-dl-sample --no-steady-iterations > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.log"
echo "trace rc=$?"
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/timeline.py "$T" 0 3 > "$OUT/sweep_timeline.txt" 2>&1
find "$OUT" -name "*.db" -delete
tail -40 "$OUT/sweep_timeline.txt"

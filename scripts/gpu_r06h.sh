#!/bin/bash
# round 6, call H: kernel trace + timeline of the current step (3 steps), plain bench line
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06h
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 2 --no-cpu-baseline --no-dl-sample --no-steady-iterations"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python "$ROOT/bench.py" $ARGS > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.log"
echo "trace rc=$?"
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/summarize_rocpd.py "$T" "$OUT/kernel_trace.txt" > /dev/null
python profiles/timeline.py "$T" 0 1 > "$OUT/sweep_timeline.txt" 2>&1
python profiles/timeline.py "$T" 0 0 > "$OUT/sweep_timeline_all.txt" 2>&1
find "$OUT" -name "*.db" -delete
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample > "$OUT/bench.json" 2> "$OUT/bench.log"
echo "bench rc=$?"
tail -c 1500 "$OUT/bench.json"
tail -n 45 "$OUT/sweep_timeline.txt"

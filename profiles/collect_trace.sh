#!/bin/bash
# Kernel trace (+stats) of a short bench run on the GPU box, summarised on the box (the database stays there):
# gpurun_out/prof_<tag>/{kernel_trace.txt, sweep_timeline.txt, bench_trace.json}
set -u
TAG=${1:-trace}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 2 --no-cpu-baseline $*"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python "$ROOT/bench.py" $ARGS \
  > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.log"
echo "trace rc=$?"
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/summarize_rocpd.py "$T" "$OUT/kernel_trace.txt" > /dev/null
python profiles/timeline.py "$T" 12 1 > "$OUT/sweep_timeline.txt" 2>&1
python profiles/timeline.py "$T" 0 1 > "$OUT/sweep_timeline_all.txt" 2>&1
find "$OUT" -name "*.db" -delete
tail -n 3 "$OUT/bench_trace.log"

#!/bin/bash
# Round-2 profiles on the GPU box (run through gpurun from the repo root):
#   kernel trace (+stats) of the bench command, then separate --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ counters) —
#   never combined with tracing domains other than --kernel-trace.  Databases under gpurun_out/prof_<tag>/;
#   summaries are written to profiles/ by the caller (summarize_rocpd.py, timeline.py, summarize_pmc_top.py).
set -u
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 2 --no-cpu-baseline"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python "$ROOT/bench.py" $ARGS \
  > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.log"
echo "trace rc=$?"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 1200 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_$C" -- python "$ROOT/bench.py" $ARGS \
    > "$OUT/bench_$C.json" 2> "$OUT/bench_$C.log"
  echo "$C rc=$?"
done
timeout 1200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY \
  -d "$OUT/pmc_SQ" -- python "$ROOT/bench.py" $ARGS > "$OUT/bench_SQ.json" 2> "$OUT/bench_SQ.log"
echo "SQ rc=$?"
cd "$ROOT"
# summaries on the box (the databases are too large to travel back), then drop the databases
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/summarize_rocpd.py "$T" "$OUT/kernel_trace.txt" > /dev/null
python profiles/timeline.py "$T" 15 1 > "$OUT/sweep_timeline.txt" 2>&1
python profiles/summarize_pmc_top.py $(find "$OUT"/pmc_* -name "*.db") --top 16 --json "$OUT/hbm_traffic.json" > "$OUT/pmc_top_kernels.txt" 2>&1
find "$OUT" -name "*.db" -delete
tail -n 3 "$OUT/bench_trace.log"
tail -c 600 "$OUT/bench_trace.json"
ls -la "$OUT"

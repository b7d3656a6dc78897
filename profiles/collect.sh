#!/bin/bash
# Collects the round's profiles on the GPU box (run through gpurun from the repo root):
#   kernel trace of the default bench command, then separate --pmc passes (FETCH_SIZE, WRITE_SIZE,
#   SQ counters) — never combined with tracing domains other than --kernel-trace.
# Outputs under gpurun_out/prof_<tag>/; summaries are written to profiles/ by the caller.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 3 --warmup 2 --cpu-seconds 20 \
  > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.log"
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_$C" -- python "$ROOT/bench.py" --steps 1 --warmup 2 --cpu-seconds 1 \
    > "$OUT/bench_$C.json" 2> "$OUT/bench_$C.log"
done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS \
  -d "$OUT/pmc_SQ" -- python "$ROOT/bench.py" --steps 1 --warmup 2 --cpu-seconds 1 > "$OUT/bench_SQ.json" 2> "$OUT/bench_SQ.log"
cd "$ROOT"
ls -R "$OUT" | head -40

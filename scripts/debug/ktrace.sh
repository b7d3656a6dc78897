#!/bin/bash
# kernel-trace stats of a short bench run: top kernels by total time (tag = output dir suffix; env passes through)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/kt_${1:-a}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-dl-sample --no-full-iteration \
  > "$OUT/b.json" 2> "$OUT/b.log"
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/timeline.py "$T" 0 1 > "$OUT/sweep_timeline.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -A12 "by kernel" "$OUT/sweep_timeline.txt" | cut -c1-100

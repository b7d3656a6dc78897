#!/bin/bash
# rocprofv3 kernel trace of a short bench run; the rocpd database lands in gpurun_out/<tag>/trace.
# usage: scripts/gpu_trace.sh <tag> [bench-args...]
set -u
TAG=${1:-trace}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 1500 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python "$ROOT/bench.py" "$@" > "$OUT/bench.json" 2> "$OUT/bench.log"
echo "rc=$?"
tail -n 8 "$OUT/bench.log"
find "$OUT/trace" -name "*.db" | head -3
du -sh "$OUT/trace"

#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of every dispatch of one kernel (regex $1) in a short bench run
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pk_${2:-a}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --kernel-include-regex "$1" --pmc $C -d "$OUT/$C" -- python "$ROOT/bench.py" --steps 2 --warmup 1 \
    --no-cpu-baseline --no-dl-sample --no-steady-iterations > "$OUT/b_$C.json" 2> "$OUT/b_$C.log"
  python "$ROOT/profiles/per_dispatch.py" $(find "$OUT/$C" -name "*.db" | head -1) "$1" 6
done
find "$OUT" -name "*.db" -delete

#!/bin/bash
# round 6, call AE: host profile of a full iteration (cProfile) + which path the new-row sampling of the step takes
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06ae
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python scripts/profile_iteration.py > "$OUT/iter.log" 2> "$OUT/iter.err"
echo "iter rc=$?"; grep -v "^\[pclean\]" "$OUT/iter.log" | grep "full iteration" | cut -c1-300

#!/bin/bash
# round 6, call AH: where the observed sweep of a full iteration goes (15 ms against 3 ms in the steady loop)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06ah
mkdir -p "$OUT"
cd "$ROOT"
PCLEAN_DEBUG_UPLOAD=1 timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter.log" 2> "$OUT/iter.err"
echo "iter rc=$?"; grep -v "^\[pclean\]" "$OUT/iter.log" | grep "full iteration\|^Record" | cut -c1-900
grep "upload\|set_table" "$OUT/iter.log" "$OUT/iter.err" | tail -20

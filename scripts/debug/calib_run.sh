#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/calib2
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $C -d "$OUT/$C" -- "$ROOT/profiles/calib/fetch_calib" > "$OUT/calib_$C.log" 2>&1
  python "$ROOT/profiles/summarize_pmc_top.py" $(find "$OUT/$C" -name "*.db") --top 12 > "$OUT/calib_$C.txt" 2>&1
  cat "$OUT/calib_$C.txt" | cut -c1-160
done
find "$OUT" -name "*.db" -delete

#!/bin/bash
# round 6, call AV: kernel timeline of a Hospital sub-batch and of the iteration on the current build
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06av
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d "$OUT/trace" -- python "$ROOT/scripts/profile_iteration.py" --no-cprofile > "$OUT/iter_trace.log" 2> "$OUT/iter_trace.err"
echo "trace rc=$?"
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/iteration_window.py "$T" 30 > "$OUT/iteration_window.txt" 2>&1
python profiles/latent_window.py "$T" 500 520 20 > "$OUT/latent_window.txt" 2>&1
find "$OUT" -name "*.db" -delete
head -34 "$OUT/iteration_window.txt"

#!/bin/bash
# round 6, call U: kernel trace of a full iteration: the latent part by kernel, one Hospital sub-batch dispatch by dispatch
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06u
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d "$OUT/trace" -- python "$ROOT/scripts/profile_iteration.py" --no-cprofile > "$OUT/iter.log" 2> "$OUT/iter.err"
echo "trace rc=$?"
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/iteration_window.py "$T" 45 > "$OUT/iteration_window.txt" 2>&1
python profiles/latent_window.py "$T" 300 360 20 > "$OUT/latent_window.txt" 2>&1
find "$OUT" -name "*.db" -delete
head -60 "$OUT/iteration_window.txt"

#!/bin/bash
# kernel trace of bench.py's full iteration; prints one Hospital sub-batch (about 340 latent rows) of the latent sweeps
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/lat_${1:-a}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 1 --warmup 0 --no-cpu-baseline --no-dl-sample \
  > "$OUT/b.json" 2> "$OUT/b.log"
echo "rc=$?"
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/latent_window.py "$T" 512 512 5 > "$OUT/hospital_subbatch.txt" 2>&1
python profiles/latent_window.py "$T" 256 256 5 > "$OUT/county_subbatch.txt" 2>&1
python profiles/iteration_window.py "$T" 30 > "$OUT/iteration_window.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -n "iteration" "$OUT/b.log" | cut -c1-300
tail -3 "$OUT/hospital_subbatch.txt"

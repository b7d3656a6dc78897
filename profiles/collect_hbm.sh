#!/bin/bash
# FETCH_SIZE and WRITE_SIZE passes only (separate rocprofv3 --pmc runs of a short bench), summarised on the box.
set -u
TAG=${1:-hbm}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline"
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_$C" -- python "$ROOT/bench.py" $ARGS \
    > "$OUT/bench_$C.json" 2> "$OUT/bench_$C.log"
  echo "$C rc=$?"
done
cd "$ROOT"
python profiles/summarize_pmc_top.py $(find "$OUT"/pmc_* -name "*.db") --top 16 --json "$OUT/hbm_traffic.json" > "$OUT/pmc_hbm_kernels.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -A1 "full-size" "$OUT/pmc_hbm_kernels.txt" | head -20
cat "$OUT/hbm_traffic.json"

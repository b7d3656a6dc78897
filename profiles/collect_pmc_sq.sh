#!/bin/bash
# One rocprofv3 --pmc pass (SQ counters) of a short bench run, summarised on the box.
# usage: profiles/collect_pmc_sq.sh <tag> "<counters>"
set -u
TAG=${1:-sq}
CTRS=${2:-"SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 1200 rocprofv3 --kernel-trace --pmc $CTRS -d "$OUT/pmc_SQ" -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline \
  > "$OUT/bench_SQ.json" 2> "$OUT/bench_SQ.log"
echo "SQ rc=$?"
cd "$ROOT"
python profiles/summarize_pmc_top.py $(find "$OUT"/pmc_* -name "*.db") --top 12 > "$OUT/pmc_top_kernels.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -A3 "fk_root_wave" "$OUT/pmc_top_kernels.txt" | head -40

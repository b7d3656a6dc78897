#!/bin/bash
# Round-6 profiles on the GPU box (run through gpurun from the repo root):
#   1. kernel trace (+stats) of the bench command;
#   2. separate --pmc passes RESTRICTED to the root-scan kernels (--kernel-include-regex): FETCH_SIZE; WRITE_SIZE;
#      the SQ set (wave / busy cycles, waits, issued instructions); TCC hit / miss — never combined with tracing
#      domains other than --kernel-trace;
#   3. the FETCH_SIZE calibration micro-kernels (profiles/calib/fetch_calib.hip) under the same counter.
# Databases under gpurun_out/prof_<tag>/; summarised on the box, only the summaries travel back.
set -u
TAG=${1:-r06}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (bench.py's default tables: unrestricted Damerau-Levenshtein, 4.3 s of table build per invocation since round 6; DIST=osa for the bit-parallel ones)
ARGS="--steps 3 --warmup 2 --no-cpu-baseline --no-dl-sample --no-steady-iterations --distance ${DIST:-dl}"
KRE="fk_root_wave_kernel|group_desc_kernel|group_settle_kernel|group_lse_kernel|group_gate_kernel|hg_insert_kernel|hg_fill_kernel|particle_update|lazy_draw_kernel|ctx_items_kernel|overflow_lds_kernel|pcc_commit"
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python "$ROOT/bench.py" $ARGS \
  > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.log"
echo "trace rc=$?"
pass() {  # name, counters...
  local name=$1; shift
  timeout 900 rocprofv3 --kernel-trace --kernel-include-regex "$KRE" --pmc "$@" -d "$OUT/pmc_$name" -- \
    python "$ROOT/bench.py" $ARGS > "$OUT/bench_$name.json" 2> "$OUT/bench_$name.log"
  echo "$name rc=$?"
}
pass FETCH FETCH_SIZE
pass WRITE WRITE_SIZE
pass SQ SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_INST_ANY
pass SQ2 SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS
pass TCC TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum
# whole-step traffic: the same two counters over EVERY kernel (no include filter), fewer sweeps
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d "$OUT/pmc_all_$C" -- python "$ROOT/bench.py" --steps 2 --warmup 1 --no-dl-sample --distance ${DIST:-dl} \
    --no-steady-iterations --no-cpu-baseline > "$OUT/bench_all_$C.json" 2> "$OUT/bench_all_$C.log"
  echo "all $C rc=$?"
done
if [ -x "$ROOT/profiles/calib/fetch_calib" ]; then
  timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d "$OUT/pmc_calib" -- "$ROOT/profiles/calib/fetch_calib" \
    > "$OUT/calib.log" 2>&1
  echo "calib rc=$?"
fi
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/summarize_rocpd.py "$T" "$OUT/kernel_trace.txt" > /dev/null
python profiles/timeline.py "$T" 0 3 > "$OUT/sweep_timeline.txt" 2>&1  # (a sweep inside the loop of whole-window sweeps: the last ones straddle host-side bookkeeping of bench.py)
python profiles/summarize_pmc_top.py $(find "$OUT"/pmc_FETCH "$OUT"/pmc_WRITE "$OUT"/pmc_SQ "$OUT"/pmc_SQ2 "$OUT"/pmc_TCC -name "*.db") \
  --top 12 --source "profiles/${TAG}_pmc_root_kernels.txt" --json "$OUT/hbm_traffic.json" > "$OUT/pmc_root_kernels.txt" 2>&1
python profiles/step_traffic.py $(find "$OUT"/pmc_all_FETCH_SIZE -name "*.db" | head -1) $(find "$OUT"/pmc_all_WRITE_SIZE -name "*.db" | head -1) \
  --json "$OUT/hbm_traffic.json" > "$OUT/step_traffic.txt" 2>&1
[ -d "$OUT/pmc_calib" ] && python profiles/summarize_pmc_top.py $(find "$OUT"/pmc_calib -name "*.db") --top 8 > "$OUT/fetch_calibration.txt" 2>&1
find "$OUT" -name "*.db" -delete
# the files as they are committed: profiles/<tag>_*.txt|json (the names hbm_traffic.json's `source` strings cite)
P=$ROOT/gpurun_out/profiles_$TAG
mkdir -p "$P"
cp "$OUT/kernel_trace.txt" "$P/${TAG}_kernel_trace.txt"
cp "$OUT/sweep_timeline.txt" "$P/${TAG}_sweep_timeline.txt"
cp "$OUT/pmc_root_kernels.txt" "$P/${TAG}_pmc_root_kernels.txt"
cp "$OUT/step_traffic.txt" "$P/${TAG}_step_traffic.txt"
[ -f "$OUT/fetch_calibration.txt" ] && cp "$OUT/fetch_calibration.txt" "$P/${TAG}_fetch_calibration.txt"
cp "$OUT/hbm_traffic.json" "$P/hbm_traffic.json"
tail -n 1 "$OUT/bench_trace.json" > "$P/${TAG}_bench_line_under_rocprof.json"
tail -n 3 "$OUT/bench_trace.log"
tail -c 400 "$OUT/bench_trace.json"
ls -la "$OUT"

#!/bin/bash
# round 6, call AR: hg_insert_kernel with the item's own key in registers — whole GPU suite, kernel trace, bench with and
# without the LDS histogram of finalize_block_kernel
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06ar
mkdir -p "$OUT"
cd "$ROOT"
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 3 "$OUT/pytest.log"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 10 --warmup 3 --no-cpu-baseline --no-dl-sample --no-steady-iterations > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.log"
echo "trace rc=$?"
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/summarize_rocpd.py "$T" "$OUT/kernel_trace.txt" > /dev/null
python profiles/timeline.py "$T" 0 3 > "$OUT/sweep_timeline.txt" 2>&1
find "$OUT" -name "*.db" -delete
grep -n "hg_insert\|finalize_block\|group_gate\|^span" "$OUT/sweep_timeline.txt" | head -12
for v in default nohist; do
  if [ $v = nohist ]; then export PCLEAN_NO_HIST=1; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample --no-steady-iterations > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.log"; echo "bench $v rc=$?"
  python - "$OUT/bench_$v.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); c=d["config"]
print("ms/step %.3f f1 %.4f fixed %.2f prop %.2f" % (d["ms_per_step"], d["f1"], c["step_fixed_ms"], c["step_proportional_ms"]))
PY
done

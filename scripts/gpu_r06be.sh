#!/bin/bash
# round 6, call BE: the round's final records — whole GPU suite, smoke, the bench lines (driver's flags; OSA tables; one-rank RCCL;
# no flags), kernel trace + per-dispatch timeline of one sweep
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06be
P=$ROOT/gpurun_out/profiles_r06be
mkdir -p "$OUT" "$P"
cd "$ROOT"
timeout 3000 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider --durations=8 > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"; tail -n 14 "$OUT/pytest.log"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -n 2 "$OUT/smoke.log"
timeout 1200 python bench.py --steps 20 --warmup 5 > "$P/r06_bench_line.json" 2> "$OUT/bench.log"; echo "bench rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 --distance osa > "$P/r06_bench_line_osa_tables.json" 2> "$OUT/bench_osa.log"; echo "osa rc=$?"
PCLEAN_FORCE_DIST=1 timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample > "$P/r06_bench_line_force_dist.json" 2> "$OUT/bench_fd.log"; echo "fd rc=$?"
timeout 900 python bench.py --no-cpu-baseline > "$P/r06_bench_line_default_3_steps.json" 2> "$OUT/bench_default.log"; echo "default rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-dl-sample --no-steady-iterations > "$P/r06_bench_line_under_rocprof.json" 2> "$OUT/bench_trace.log"
echo "trace rc=$?"
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/summarize_rocpd.py "$T" "$P/r06_kernel_trace.txt" > /dev/null
python profiles/timeline.py "$T" 0 3 > "$P/r06_sweep_timeline.txt" 2>&1
find "$OUT" -name "*.db" -delete
tail -n 4 "$P/r06_sweep_timeline.txt" | head -2; grep -n "^span" "$P/r06_sweep_timeline.txt"
for f in r06_bench_line r06_bench_line_osa_tables r06_bench_line_force_dist r06_bench_line_default_3_steps; do
python - "$P/$f.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); r=d["roofline"]; c=d["config"]
print(sys.argv[1].split("/")[-1], "ms/step %.3f value %.1fM f1 %.4f frac %.3f (%s) fixed %.2f prop %.2f iter %s/%s tables %.1fs cpu %s" % (d["ms_per_step"], d["value"]/1e6, d["f1"], r["frac"], "measure" if "chosen" in r else "block0", c["step_fixed_ms"], c["step_proportional_ms"], c["full_iteration_ms"], c["full_iteration_steady_ms"], d["table_build"]["seconds"], (d.get("cpu_baseline") or {}).get("value")))
PY
done

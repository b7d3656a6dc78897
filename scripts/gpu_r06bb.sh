#!/bin/bash
# round 6, call BB: the committed tree once more — whole GPU suite, smoke, the driver's bench command
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06bb
mkdir -p "$OUT"
cd "$ROOT"
timeout 2400 python -m pytest tests -x -q -m gpu -p no:cacheprovider > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 2 "$OUT/pytest.log"
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "smoke rc=$?"; tail -n 1 "$OUT/smoke.log"
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); c=d["config"]
print("ms/step %.3f value %.1fM f1 %.4f frac %.3f fixed %.2f prop %.2f iter %.1f/%.1f cpu %.0f" % (d["ms_per_step"], d["value"]/1e6, d["f1"], d["roofline"]["frac"], c["step_fixed_ms"], c["step_proportional_ms"], c["full_iteration_ms"], c["full_iteration_steady_ms"], d["cpu_baseline"]["value"]))
PY

#!/bin/bash
# round 6, call AS: ev_leaf_block_kernel with two evidence entries' loads in flight — whole GPU suite, the iteration's host
# timers + device phases, bench line with the steady iterations
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06as
mkdir -p "$OUT"
cd "$ROOT"
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 3 "$OUT/pytest.log"
timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iteration.txt" 2> "$OUT/iteration.err"; echo "iter rc=$?"
head -12 "$OUT/iteration.txt" | cut -c1-400
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample > "$OUT/bench.json" 2> "$OUT/bench.log"; echo "bench rc=$?"
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); c=d["config"]
print("ms/step %.3f f1 %.4f fixed %.2f prop %.2f iter %.1f/%.1f" % (d["ms_per_step"], d["f1"], c["step_fixed_ms"], c["step_proportional_ms"], c["full_iteration_ms"], c["full_iteration_steady_ms"]))
PY

#!/bin/bash
# round 6, call BA: A/B on one box — A = the build of call AY; B2 (in tree) = A + the lower bound of ev_leaf_block_kernel from a
# wavefront that spreads the best option's evidence entries over its lanes.  Bench line of each, then the whole GPU suite.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06ba
mkdir -p "$OUT"
cd "$ROOT"
for v in A B2 A B2; do
  if [ $v = B2 ]; then unset PCLEAN_HIP_LIB; else export PCLEAN_HIP_LIB=$ROOT/ab/lib_$v.so; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.log"; echo "bench $v rc=$?"
  python - "$OUT/bench_$v.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); c=d["config"]; r=d["roofline"]
print("ms/step %.3f f1 %.4f fixed %.2f prop %.2f iter %.1f/%.1f measure-group %.3f ms block0-group %.3f ms" % (d["ms_per_step"], d["f1"], c["step_fixed_ms"], c["step_proportional_ms"], c["full_iteration_ms"], c["full_iteration_steady_ms"], r.get("avg_launch_ms", 0), (r.get("block0_root_group") or {}).get("avg_launch_ms", 0)))
PY
done
unset PCLEAN_HIP_LIB
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 3 "$OUT/pytest.log"

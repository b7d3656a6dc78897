#!/bin/bash
# round 6, call AT: A/B on one box — A = the build of call AS; B1 = A + the root scan's exact scores with the plain terms'
# loads ahead of the context chains; B2 (in tree) = B1 + candidate_score_ev with every load of a step unconditional.
# Bench line of each (step + full iterations), the commit kernel's phase clock, then the whole GPU suite on B2.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06at
mkdir -p "$OUT"
cd "$ROOT"
for v in A B1 B2 A B2; do
  if [ $v = B2 ]; then unset PCLEAN_HIP_LIB; else export PCLEAN_HIP_LIB=$ROOT/ab/lib_$v.so; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.log"; echo "bench $v rc=$?"
  python - "$OUT/bench_$v.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); c=d["config"]; r=d["roofline"]
print("ms/step %.3f f1 %.4f fixed %.2f prop %.2f iter %.1f/%.1f measure-group %.3f ms block0-group %.3f ms" % (d["ms_per_step"], d["f1"], c["step_fixed_ms"], c["step_proportional_ms"], c["full_iteration_ms"], c["full_iteration_steady_ms"], r.get("avg_launch_ms", 0), (r.get("block0_root_group") or {}).get("avg_launch_ms", 0)))
PY
done
unset PCLEAN_HIP_LIB
PCLEAN_COMMIT_PROF=1 PCLEAN_COMMIT_ONE_WG=1 timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-dl-sample --no-steady-iterations > "$OUT/bench_prof.json" 2> "$OUT/bench_prof.log"
grep "commit kernel phases" "$OUT/bench_prof.log" | tail -3
timeout 2400 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 3 "$OUT/pytest.log"

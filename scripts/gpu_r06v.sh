#!/bin/bash
# round 6, call V: evidence scan with the entries in LDS and several quads in flight, split re-run of lists beyond LDS,
# unsplit groups for lazy launches (A/B each)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06v
mkdir -p "$OUT"
cd "$ROOT"
timeout 1500 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_inference.py tests/test_gpu_rents.py tests/test_gpu_flights.py tests/test_gpu_edges.py tests/test_gpu_sweep.py tests/test_gpu_literal.py tests/test_gpu_commit.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 3 "$OUT/pytest.log"
for V in new qb1 qb4 nosplit; do
  E="X=1"; [ $V = qb1 ] && E="PCLEAN_EV_QB=1"; [ $V = qb4 ] && E="PCLEAN_EV_QB=4"; [ $V = nosplit ] && E="PCLEAN_NO_ENUM_SPLIT=1"
  env $E timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter_$V.log" 2> "$OUT/iter_$V.err"
  echo "$V rc=$?"; grep -v "^\[pclean\]" "$OUT/iter_$V.log" | grep "full iteration\|^Hospital\|^County" | cut -c1-420
done
for V in new lazysplit; do
  E="X=1"; [ $V = lazysplit ] && E="PCLEAN_LAZY_SPLIT=1"
  env $E timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample --no-steady-iterations > "$OUT/bench_$V.json" 2> "$OUT/bench_$V.log"
  echo "bench $V rc=$?"; python - "$OUT/bench_$V.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("ms/step", d["ms_per_step"], "value", d["value"], "f1", d["config"].get("f1"), "roofline", r.get("achieved"), r.get("frac"), "ms", r.get("kernel_ms"), "iter", d["config"].get("full_iteration_ms"))
PY
done

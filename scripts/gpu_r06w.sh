#!/bin/bash
# round 6, call W: unsplit groups of lazy launches with per-member log-marginals / child scatters (A/B), full-size parity
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06w
mkdir -p "$OUT"
cd "$ROOT"
for V in new lazysplit; do
  E="X=1"; [ $V = lazysplit ] && E="PCLEAN_LAZY_SPLIT=1"
  env $E timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample --no-steady-iterations --no-full-iteration > "$OUT/bench_$V.json" 2> "$OUT/bench_$V.log"
  echo "bench $V rc=$?"; python - "$OUT/bench_$V.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d["roofline"]
print("ms/step", d["ms_per_step"], "value", d["value"], "roofline", r.get("achieved"), r.get("frac"), "ms", r.get("avg_launch_ms"), "groups", r.get("groups"), "dev", d["config"].get("device_ms_per_step"))
PY
done
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_determinism.py tests/test_gpu_sweep.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 3 "$OUT/pytest.log"

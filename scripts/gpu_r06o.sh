#!/bin/bash
# round 6, call O: dense tables skip the coarse level of the scan: parity + A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06o
mkdir -p "$OUT"
cd "$ROOT"
timeout 2400 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_sweep.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 3 "$OUT/pytest.log"
for V in new old new old; do
  E=""; [ $V = old ] && E="PCLEAN_NO_DENSE_SCAN=1"
  env $E timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample --no-steady-iterations --no-full-iteration --distance osa > "$OUT/bench_$V.json" 2> "$OUT/bench_$V.log"
  python - "$OUT/bench_$V.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); c=d["config"]; r=d["roofline"]
print(sys.argv[1].split("/")[-1], "ms/step %.3f measure group %.3f ms" % (d["ms_per_step"], r["avg_launch_ms"] if "chosen" in r else r["measure_root_group"]["avg_launch_ms"]))
PY
done

#!/bin/bash
# round 6, call BC: consecutive groups a wave of the root scans takes at a time (PCLEAN_WAVE_CHUNK; built-in: 8)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06bc
mkdir -p "$OUT"
cd "$ROOT"
for v in default 4 16 32 default 12; do
  if [ $v = default ]; then unset PCLEAN_WAVE_CHUNK; else export PCLEAN_WAVE_CHUNK=$v; fi
  timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-dl-sample --no-steady-iterations > "$OUT/bench_$v.json" 2> "$OUT/bench_$v.log"; echo "bench chunk=$v rc=$?"
  python - "$OUT/bench_$v.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); c=d["config"]; r=d["roofline"]
print("ms/step %.3f f1 %.4f measure-group %.3f ms (full scans %s) block0-group %.3f ms (full scans %s)" % (d["ms_per_step"], d["f1"], r.get("avg_launch_ms", 0), r.get("full_scans"), (r.get("block0_root_group") or {}).get("avg_launch_ms", 0), (r.get("block0_root_group") or {}).get("full_scans")))
PY
done

#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/tune
for W in 256 512 1024 1536 2048; do
  PCLEAN_WAVE_WGS=$W timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-iteration > gpurun_out/tune/w_$W.json 2> gpurun_out/tune/w_$W.log
  python - <<PY
import json
d=json.loads(open("gpurun_out/tune/w_$W.json").read().strip().splitlines()[-1])
ph=d["phases_ms"]
print("WGS=$W: root %.3f ms, slot_scan %.3f, step %.2f ms, device %.2f" % (d["roofline"]["avg_launch_ms"], ph["slot_scan"]["ms"], d["ms_per_step"], d["config"]["device_ms_per_step"]))
PY
done

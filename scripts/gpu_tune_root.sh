#!/bin/bash
# Tuning sweep of the root scan's compile-time knobs on the GPU box: rebuilds root_wave.o per variant and runs a short bench.
# usage: scripts/gpu_tune_root.sh "RB TC MINW CHUNK SLACK" ...
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/tune
for V in "$@"; do
  set -- $V
  T=$(echo $V | tr ' ' '_')
  touch pclean_amd/csrc/root_wave.hip
  PCLEAN_EXTRA_HIPCC_FLAGS="-DWAVE_RB=$1 -DWAVE_TC=$2 -DWAVE_MIN_WAVES=$3 -DWAVE_CHUNK=$4 -DWAVE_SLACK=${5}u" python -c "from pclean_amd import build as b; b.build(verbose=False)" 2> gpurun_out/tune/build_$T.log
  timeout 600 python bench.py --steps 6 --warmup 3 --no-cpu-baseline > gpurun_out/tune/b_$T.json 2> gpurun_out/tune/b_$T.log
  python - <<PY
import json
d=json.loads(open("gpurun_out/tune/b_$T.json").read().strip().splitlines()[-1])
ph=d["phases_ms"]
print("RB=$1 TC=$2 MINW=$3 CHUNK=$4 SLACK=$5: root %.3f ms, slot_scan %.3f, step %.2f ms, device %.2f" % (d["roofline"]["avg_launch_ms"], ph["slot_scan"]["ms"], d["ms_per_step"], d["config"]["device_ms_per_step"]))
PY
done

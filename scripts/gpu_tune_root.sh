#!/bin/bash
# Tuning sweep of the root scan's compile-time knobs on the GPU box: rebuilds root_wave.o per variant and runs a short bench.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/tune
for V in "4 6 4" "4 6 5" "4 6 6" "2 4 6" "2 4 8"; do
  set -- $V
  touch pclean_amd/csrc/root_wave.hip
  PCLEAN_EXTRA_HIPCC_FLAGS="-DWAVE_RB=$1 -DWAVE_TC=$2 -DWAVE_MIN_WAVES=$3" python -c "from pclean_amd import build as b; b.build(verbose=False)" 2> gpurun_out/tune/build_$1_$2_$3.log
  timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-full-iteration > gpurun_out/tune/b_$1_$2_$3.json 2> gpurun_out/tune/b_$1_$2_$3.log
  python - <<PY
import json
d=json.loads(open("gpurun_out/tune/b_$1_$2_$3.json").read().strip().splitlines()[-1])
ph=d["phases_ms"]
print("RB=$1 TC=$2 MINW=$3: root %.3f ms, slot_scan %.3f, step %.2f ms, device %.2f" % (d["roofline"]["avg_launch_ms"], ph["slot_scan"]["ms"], d["ms_per_step"], d["config"]["device_ms_per_step"]))
PY
done

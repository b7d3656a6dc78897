#!/bin/bash
# Timing experiments on the root scan kernel: phases switched off one at a time (results are wrong on purpose).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/tune
for V in "$@"; do
  touch pclean_amd/csrc/root_wave.hip
  PCLEAN_EXTRA_HIPCC_FLAGS="-DWAVE_DBG_SKIP=$V" python -c "from pclean_amd import build as b; b.build(verbose=False)" 2> gpurun_out/tune/build_dbg$V.log
  timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/tune/b_dbg$V.json 2> gpurun_out/tune/b_dbg$V.log
  python - <<PY
import json
d=json.loads(open("gpurun_out/tune/b_dbg$V.json").read().strip().splitlines()[-1])
ph=d["phases_ms"]
print("SKIP=$V: root %.3f ms, slot_scan %.3f, step %.2f ms, device %.2f" % (d["roofline"]["avg_launch_ms"], ph["slot_scan"]["ms"], d["ms_per_step"], d["config"]["device_ms_per_step"]))
PY
done

#!/bin/bash
# Measurement build of the root scan kernel (-DWAVE_PHASE_CLOCK): per-phase cycle shares of the big launches.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$ROOT"
mkdir -p gpurun_out/tune
touch pclean_amd/csrc/root_wave.hip
PCLEAN_EXTRA_HIPCC_FLAGS="-DWAVE_PHASE_CLOCK $*" python -c "from pclean_amd import build as b; b.build(verbose=False)" 2> gpurun_out/tune/build_clk.log
timeout 150 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/tune/b_clk.json 2> gpurun_out/tune/b_clk.log
echo "rc=$?"
grep "wave clk" gpurun_out/tune/b_clk.log | tail -6

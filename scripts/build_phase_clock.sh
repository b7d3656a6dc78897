#!/bin/bash
# Measurement build of the library: the root scan kernel compiled with -DWAVE_PHASE_CLOCK (per-phase cycle totals and
# scan counters printed to stderr for launches of > 100 000 groups), linked with the product objects into
# pclean_amd/libpclean_hip_clk.so.  Use:  PCLEAN_HIP_LIB=$PWD/pclean_amd/libpclean_hip_clk.so python bench.py ...
set -e
cd "$(dirname "$0")/.."
python -c "from pclean_amd import build; build.build()"
C=pclean_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DWAVE_PHASE_CLOCK -c $C/root_wave.hip -o /tmp/root_wave_clk.o
OBJS=$(ls $C/*.o | grep -v root_wave.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o pclean_amd/libpclean_hip_clk.so $OBJS /tmp/root_wave_clk.o -ldl
echo built pclean_amd/libpclean_hip_clk.so

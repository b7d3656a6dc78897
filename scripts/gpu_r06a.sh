#!/bin/bash
# round 6, call A: the new DL table kernel — parity tests + per-table build times at 1M rows
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06a
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python -m pytest tests/test_gpu_tables.py tests/test_gpu_sweep.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -n 15 "$OUT/pytest.log"
timeout 900 python scripts/debug/dl_build_time.py > "$OUT/dl_build.log" 2>&1
echo "dl_build rc=$?"
tail -n 25 "$OUT/dl_build.log"

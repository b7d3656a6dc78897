#!/bin/bash
# round 6, call AZ: the determinism test with the new variant (PCLEAN_NO_HIST)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06az
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_determinism.py -m gpu -q --tb=short -p no:cacheprovider -x -s > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; grep "determinism\]" "$OUT/pytest.log" | cut -c1-120; tail -n 2 "$OUT/pytest.log"

#!/bin/bash
# round 6, call AC: non-linear Transformations on the HIP path (parity with the oracle), rents / literal suites
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06ac
mkdir -p "$OUT"
cd "$ROOT"
timeout 1800 python -m pytest tests/test_gpu_rents.py tests/test_gpu_literal.py tests/test_gpu_edges.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 5 "$OUT/pytest.log"

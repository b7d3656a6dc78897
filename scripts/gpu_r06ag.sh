#!/bin/bash
# round 6, call AG: table re-uploads that do not wait for their copies (bump-allocated staging) — suites + iteration A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06ag
mkdir -p "$OUT"
cd "$ROOT"
timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter_new.log" 2> "$OUT/iter_new.err"
echo "iter rc=$?"; grep -v "^\[pclean\]" "$OUT/iter_new.log" | grep "full iteration" | cut -c1-600
timeout 2400 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_inference.py tests/test_gpu_commit.py tests/test_gpu_edges.py tests/test_gpu_rents.py tests/test_gpu_flights.py tests/test_gpu_literal.py tests/test_gpu_sweep.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 5 "$OUT/pytest.log"

#!/bin/bash
# round 6, call AJ: chain of upload deltas with per-row column masks
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06aj
mkdir -p "$OUT"
cd "$ROOT"
PCLEAN_DEBUG_CHAIN=1 timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter.log" 2> "$OUT/iter.err"
echo "rc=$?"; grep "\[chain\]" "$OUT/iter.err" | grep -v "set_table" | tail -8
grep -v "^\[pclean\]" "$OUT/iter.log" | grep "full iteration\|^Record\|^Hospital\|^Place" | cut -c1-520
timeout 2400 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_commit.py tests/test_gpu_inference.py tests/test_gpu_edges.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 6 "$OUT/pytest.log"

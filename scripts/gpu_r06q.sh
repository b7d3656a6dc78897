#!/bin/bash
# round 6, call Q: upload_trace compares before it pads: commit / inference tests + iteration profile
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06q
mkdir -p "$OUT"
cd "$ROOT"
timeout 1800 python -m pytest tests/test_gpu_commit.py tests/test_gpu_inference.py tests/test_gpu_determinism.py tests/test_gpu_rents.py tests/test_gpu_flights.py tests/test_gpu_f1_vs_sequential.py tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 3 "$OUT/pytest.log"
for V in a b; do
  timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter_$V.log" 2> "$OUT/iter_$V.err"
  echo "$V rc=$?"; grep -v "^\[pclean\]" "$OUT/iter_$V.log" | grep "full iteration" | cut -c1-700
done

#!/bin/bash
# round 6, call K: changed-row deltas of re-uploaded tables -> evidence scans for reference slots in sub-batches
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06k
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_inference.py tests/test_gpu_literal.py tests/test_gpu_flights.py tests/test_gpu_rents.py tests/test_gpu_commit.py tests/test_gpu_f1_vs_sequential.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -n 5 "$OUT/pytest.log"
for V in default min1024 nodelta; do
  case $V in
    default) E="" ;;
    min1024) E="PCLEAN_EV_SLOT_MIN_ITEMS=1024" ;;
    nodelta) E="PCLEAN_NO_UPLOAD_DELTA=1" ;;
  esac
  env $E timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter_$V.log" 2> "$OUT/iter_$V.err"
  echo "$V rc=$?"; grep -v "^\[pclean\]" "$OUT/iter_$V.log" | grep "full iteration\|^Hospital\|^Place" | cut -c1-700
done

#!/bin/bash
# round 6, call S: pass A's weighted sums kept in LDS for pass B of the evidence scan (A/B), and which items the evidence
# scan hands to the generic re-run (per launch: node, items, overflowed)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06s
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_inference.py tests/test_gpu_rents.py tests/test_gpu_flights.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 3 "$OUT/pytest.log"
for V in new old; do
  E=""; [ $V = old ] && E="PCLEAN_NO_EV_SUM_CACHE=1"
  env $E timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter_$V.log" 2> "$OUT/iter_$V.err"
  echo "$V rc=$?"; grep -v "^\[pclean\]" "$OUT/iter_$V.log" | grep "full iteration\|^Hospital\|^County" | cut -c1-420
done
env PCLEAN_NO_EV_LIST=1 PCLEAN_DEBUG_OVERFLOW=1 timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter_dbg.log" 2> "$OUT/iter_dbg.err"
echo "dbg rc=$?"
grep "re-run by the generic" "$OUT/iter_dbg.err" | sed 's/: [0-9]* of/: N of/' | sort | uniq -c | sort -rn | head -40
grep "re-run by the generic" "$OUT/iter_dbg.err" | tail -n 400 | head -n 120

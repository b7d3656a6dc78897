#!/bin/bash
# round 6, call I: per-option prior cut of the evidence scan: latent parity tests + iteration profile with / without it
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06i
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_inference.py tests/test_gpu_literal.py tests/test_gpu_flights.py tests/test_gpu_rents.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -n 5 "$OUT/pytest.log"
timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter.log" 2> "$OUT/iter.err"
echo "rc=$?"; grep -v "^\[pclean\]" "$OUT/iter.log" | tail -9
PCLEAN_NO_EV_PRIOR_CUT=1 timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter_nocut.log" 2> "$OUT/iter_nocut.err"
echo "rc=$?"; grep -v "^\[pclean\]" "$OUT/iter_nocut.log" | tail -9

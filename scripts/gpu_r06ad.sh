#!/bin/bash
# round 6, call AD: every node's evidence aggregation in one launch per latent call (A/B)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06ad
mkdir -p "$OUT"
cd "$ROOT"
timeout 1800 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_inference.py tests/test_gpu_rents.py tests/test_gpu_flights.py tests/test_gpu_literal.py tests/test_gpu_edges.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 3 "$OUT/pytest.log"
for V in new old; do
  E="X=1"; [ $V = old ] && E="PCLEAN_NO_AGG_ALL=1"
  env $E timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter_$V.log" 2> "$OUT/iter_$V.err"
  echo "$V rc=$?"; grep -v "^\[pclean\]" "$OUT/iter_$V.log" | grep "full iteration\|^Hospital\|^County\|^Place" | cut -c1-420
done

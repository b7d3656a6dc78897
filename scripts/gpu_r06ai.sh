#!/bin/bash
# round 6, call AI: compact tables refreshed along the chain of upload deltas (observed sweep after the latent classes' sweeps)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06ai
mkdir -p "$OUT"
cd "$ROOT"
for V in new old; do
  E="X=1"; [ $V = old ] && E="PCLEAN_NO_DELTA_CHAIN=1"
  env $E timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter_$V.log" 2> "$OUT/iter_$V.err"
  echo "$V rc=$?"; grep -v "^\[pclean\]" "$OUT/iter_$V.log" | grep "full iteration\|^Record" | cut -c1-420
done
timeout 2400 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_commit.py tests/test_gpu_inference.py tests/test_gpu_edges.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 6 "$OUT/pytest.log"

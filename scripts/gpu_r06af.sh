#!/bin/bash
# round 6, call AF: evidence sets of the latent classes built on the device (pclean_build_evidence) — parity with the host path,
# determinism / inference suites, full iteration A/B
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06af
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_edges.py -m gpu -q --tb=short -p no:cacheprovider -x -k "evidence or argsort" > "$OUT/pytest_ev.log" 2>&1
echo "pytest ev rc=$?"; tail -n 15 "$OUT/pytest_ev.log"
for V in new old; do
  E="X=1"; [ $V = old ] && E="PCLEAN_HOST_EVIDENCE=1"
  env $E timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter_$V.log" 2> "$OUT/iter_$V.err"
  echo "$V rc=$?"; grep -v "^\[pclean\]" "$OUT/iter_$V.log" | grep "full iteration" | cut -c1-700
done
timeout 1800 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_inference.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 5 "$OUT/pytest.log"

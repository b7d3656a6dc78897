#!/bin/bash
# Staged GPU check for a new kernel: a short hang guard first (fast-root vs generic bit-identity), then the full
# suite, smoke and a bench run.  usage: scripts/gpu_stage.sh <tag> [bench-args...]
set -u
TAG=${1:-stage}
shift || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
timeout 420 python -m pytest tests/test_gpu_fullsize.py -m gpu -q --tb=short -p no:cacheprovider -k fast_root > "$OUT/guard.log" 2>&1
RC=$?
echo "guard rc=$RC"; tail -n 25 "$OUT/guard.log"
if [ $RC -ne 0 ]; then exit 1; fi
exec scripts/gpu_check.sh "$TAG" tests "$@"

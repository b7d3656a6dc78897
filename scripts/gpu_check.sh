#!/bin/bash
# One gpurun call: GPU parity tests, smoke, a bench run (and optionally rocprofv3 passes).  Outputs under gpurun_out/<tag>/.
# usage: scripts/gpu_check.sh <tag> [tests|notests] [bench-args...]
set -u
TAG=${1:-chk}
DO_TESTS=${2:-tests}
shift 2 || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
if [ "$DO_TESTS" = "tests" ]; then
  timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > "$OUT/pytest.log" 2>&1
  echo "pytest rc=$?" >> "$OUT/pytest.log"
  tail -n 40 "$OUT/pytest.log"
  timeout 300 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1
  echo "smoke rc=$?"; tail -n 3 "$OUT/smoke.log"
fi
timeout 1500 python bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.log"
echo "bench rc=$?"
tail -n 25 "$OUT/bench.log"
tail -c 3000 "$OUT/bench.json"

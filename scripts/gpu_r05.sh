#!/bin/bash
# One gpurun call of round 5: GPU parity tests + smoke, the bench line, a kernel trace with the per-dispatch timeline of one
# sweep.  Outputs under gpurun_out/<tag>/.   usage: scripts/gpu_r05.sh <tag> [tests|notests] [trace|notrace] [bench-args...]
set -u
TAG=${1:-r05}
DO_TESTS=${2:-tests}
DO_TRACE=${3:-trace}
shift 3 || true
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$ROOT"
if [ "$DO_TESTS" = "tests" ]; then
  timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > "$OUT/pytest.log" 2>&1
  echo "pytest rc=$?" >> "$OUT/pytest.log"
  tail -n 60 "$OUT/pytest.log"
  timeout 300 python __graft_entry__.py --smoke > "$OUT/smoke.log" 2>&1
  echo "smoke rc=$?"; tail -n 3 "$OUT/smoke.log"
fi
timeout 1500 python bench.py "$@" > "$OUT/bench.json" 2> "$OUT/bench.log"
echo "bench rc=$?"
tail -n 12 "$OUT/bench.log"
tail -c 2500 "$OUT/bench.json"
if [ "$DO_TRACE" = "trace" ]; then
  cd /tmp && export TMPDIR=/tmp
  timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -- python "$ROOT/bench.py" --steps 3 --warmup 2 --no-cpu-baseline \
    --no-dl-sample --no-steady-iterations "$@" > "$OUT/bench_trace.json" 2> "$OUT/bench_trace.log"
  echo "trace rc=$?"
  cd "$ROOT"
  T=$(find "$OUT/trace" -name "*.db" | head -1)
  python profiles/summarize_rocpd.py "$T" "$OUT/kernel_trace.txt" > /dev/null
  python profiles/timeline.py "$T" 0 1 > "$OUT/sweep_timeline.txt" 2>&1
  find "$OUT" -name "*.db" -delete
  tail -n 40 "$OUT/sweep_timeline.txt"
fi

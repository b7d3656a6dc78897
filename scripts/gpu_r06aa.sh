#!/bin/bash
# round 6, call AA: evidence entries four at a time without the term-chunked candidate_score (it copied NodeDev to scratch)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06aa
mkdir -p "$OUT"
cd "$ROOT"
timeout 1800 python -m pytest tests/test_gpu_determinism.py tests/test_gpu_inference.py tests/test_gpu_rents.py tests/test_gpu_flights.py tests/test_gpu_edges.py tests/test_gpu_sweep.py tests/test_gpu_literal.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?"; tail -n 3 "$OUT/pytest.log"
for V in new; do
  timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter_$V.log" 2> "$OUT/iter_$V.err"
  echo "$V rc=$?"; grep -v "^\[pclean\]" "$OUT/iter_$V.log" | grep "full iteration\|^Hospital\|^County\|^Place\|^Measure" | cut -c1-420
done
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d "$OUT/trace" -- python "$ROOT/scripts/profile_iteration.py" --no-cprofile > "$OUT/iter_trace.log" 2> "$OUT/iter_trace.err"
echo "trace rc=$?"
cd "$ROOT"
T=$(find "$OUT/trace" -name "*.db" | head -1)
python profiles/iteration_window.py "$T" 30 > "$OUT/iteration_window.txt" 2>&1
python profiles/latent_window.py "$T" 500 520 20 > "$OUT/latent_window.txt" 2>&1
find "$OUT" -name "*.db" -delete
head -34 "$OUT/iteration_window.txt"

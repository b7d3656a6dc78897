#!/bin/bash
# round 6, call G: evidence-set pre-filter for reference slots, closed-form Pitman-Yor score: tests + iteration profile
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06g
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -n 5 "$OUT/pytest.log"
timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter.log" 2> "$OUT/iter.err"
echo "rc=$?"; grep -v "^\[pclean\]" "$OUT/iter.log" | tail -12
PCLEAN_NO_FAST_EV_SLOTS=1 timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter_noslots.log" 2> "$OUT/iter_noslots.err"
echo "rc=$?"; grep -v "^\[pclean\]" "$OUT/iter_noslots.log" | tail -12

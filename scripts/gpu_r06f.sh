#!/bin/bash
# round 6, call F: where a full iteration goes (per class device phases; overflow re-runs per plan node)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06f
mkdir -p "$OUT"
cd "$ROOT"
PCLEAN_DEBUG_OVERFLOW=1 timeout 900 python scripts/profile_iteration.py --no-cprofile > "$OUT/iter.log" 2> "$OUT/iter.err"
echo "rc=$?"
grep -v "^\[pclean\]" "$OUT/iter.log" | tail -20
python - <<'PY'
import re,collections,os
c=collections.defaultdict(lambda:[0,0,0])
for l in open(os.environ.get('OUTF','gpurun_out/r06f/iter.err')):
    m=re.search(r'block (\d+) node (\d+): (\d+) of (\d+) items re-run',l)
    if m:
        k=(int(m.group(1)),int(m.group(2))); c[k][0]+=1; c[k][1]+=int(m.group(3)); c[k][2]+=int(m.group(4))
for k,v in sorted(c.items()): print('block %d node %d: %d launches, %d of %d items re-run'%(k+tuple(v)))
PY

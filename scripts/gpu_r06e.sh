#!/bin/bash
# round 6, call E: commit on a workgroup per plan, early compact-table refresh, generic kernel for nested lists up to 160 KB
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06e
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests/test_gpu_commit.py tests/test_gpu_fullsize.py tests/test_gpu_sweep.py tests/test_gpu_determinism.py tests/test_gpu_inference.py -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -n 4 "$OUT/pytest.log"
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-steady-iterations --no-dl-sample > "$OUT/b_$name.json" 2> "$OUT/b_$name.log"
  tail -1 "$OUT/b_$name.json" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms']; c=d['config']
g=lambda k: p.get(k,{}).get('ms',0)
print('$name', 'ms/step %.3f dev %.3f fixed %.3f | root0 %.3f slot_scan %.3f pu %.3f final %.3f enum_fk %.3f gate %.3f compact %.3f f1 %.4f refusals %s' % (d['ms_per_step'], c['device_ms_per_step'], c['step_fixed_ms'], g('root_scan_block0'), g('slot_scan'), g('particle_update'), g('final_choice_and_outputs'), g('enum_fk_generic'), g('gate_new_branch'), g('compact_table_update'), d['f1'], c['device_commit_refusals']))"
}
run all X=1
run one_wg PCLEAN_COMMIT_ONE_WG=1
run late_pre PCLEAN_LATE_COMPACT_PREFETCH=1
run all2 X=2
run prof PCLEAN_COMMIT_PROF=1
grep "commit kernel phases" "$OUT/b_prof.log" | tail -2

#!/bin/bash
# round 6, call D: tests + A/B of the step changes (uniform block-0 weights, generic kernel for small nested lists, fused
# context items, lazy draws) + the commit kernel's phase stamps
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06d
mkdir -p "$OUT"
cd "$ROOT"
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > "$OUT/pytest.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest.log"
tail -n 6 "$OUT/pytest.log"; grep -n "determinism\]" "$OUT/pytest.log"
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-steady-iterations --no-dl-sample > "$OUT/b_$name.json" 2> "$OUT/b_$name.log"
  tail -1 "$OUT/b_$name.json" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms']; c=d['config']
g=lambda k: p.get(k,{}).get('ms',0)
print('$name', 'ms/step %.3f dev %.3f fixed %.3f | root0 %.3f slot_scan %.3f pu %.3f final %.3f lazy %.3f ctx_items %.3f enum_fk %.3f gate %.3f f1 %.4f' % (d['ms_per_step'], c['device_ms_per_step'], c['step_fixed_ms'], g('root_scan_block0'), g('slot_scan'), g('particle_update'), g('final_choice_and_outputs'), g('lazy_draws'), g('ctx_items'), g('enum_fk_generic'), g('gate_new_branch'), d['f1']))"
}
run all X=1
run no_uniw PCLEAN_NO_UNIFORM_W=1
run no_small PCLEAN_NO_SMALL_GENERIC=1
run no_ctxfuse PCLEAN_NO_FUSED_CTX_ITEMS=1
run no_lazy PCLEAN_NO_LAZY_DRAWS=1
run r5 PCLEAN_NO_UNIFORM_W=1 PCLEAN_NO_SMALL_GENERIC=1 PCLEAN_NO_FUSED_CTX_ITEMS=1 PCLEAN_NO_LAZY_DRAWS=1
run prof PCLEAN_COMMIT_PROF=1
grep "commit kernel phases" "$OUT/b_prof.log" | tail -3

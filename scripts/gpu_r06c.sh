#!/bin/bash
# round 6, call C: A/B of the Measure root changes (family order, lazy draws) + a phase-clock build
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06c
mkdir -p "$OUT"
cd "$ROOT"
run() {  # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-steady-iterations --no-dl-sample > "$OUT/b_$name.json" 2> "$OUT/b_$name.log"
  tail -1 "$OUT/b_$name.json" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms']; c=d['config']
print('$name', 'ms/step %.3f dev %.3f fixed %.3f | root0 %.3f slot_scan %.3f pu %.3f final %.3f lazy %.3f fam %.3f f1 %.4f' % (d['ms_per_step'], c['device_ms_per_step'], c['step_fixed_ms'], p['root_scan_block0']['ms'], p['slot_scan']['ms'], p['particle_update']['ms'], p['final_choice_and_outputs']['ms'], p.get('lazy_draws',{}).get('ms',0), p.get('family_order',{}).get('ms',0), d['f1']))"
}
run both X=1
run nofam PCLEAN_NO_FAMILY_ORDER=1
run nolazy PCLEAN_NO_LAZY_DRAWS=1
run neither PCLEAN_NO_FAMILY_ORDER=1 PCLEAN_NO_LAZY_DRAWS=1
run chunk8 PCLEAN_WAVE_CHUNK=8
bash scripts/gpu_phase_clock.sh > "$OUT/clk.log" 2>&1
cat "$OUT/clk.log"

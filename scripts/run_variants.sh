# usage (on the GPU box, from the repo root): bash scripts/run_variants.sh <name> ...   — one bench run per
# pclean_amd/libv_<name>.so (measurement builds of the library with different kernel macros), key timings per line
for v in "$@"; do
  PCLEAN_HIP_LIB=$PWD/pclean_amd/libv_$v.so python bench.py --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d['phases_ms']
print('$v', 'ms/step %.3f dev %.3f root0 %.3f slot_scan %.3f launch %.3f f1 %.4f' % (d['ms_per_step'], d['config']['device_ms_per_step'], p['root_scan_block0']['ms'], p['slot_scan']['ms'], d['roofline']['avg_launch_ms'], d['f1']))"
done

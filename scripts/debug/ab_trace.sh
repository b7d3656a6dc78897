set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in pad nopad; do
  if [ $v = nopad ]; then export PCLEAN_DC_NOPAD=1; fi
  OUT=$ROOT/gpurun_out/ab_$v; mkdir -p $OUT
  timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -- python $ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-dl-sample > $OUT/b.json 2> $OUT/b.log
  T=$(find $OUT/trace -name "*.db" | head -1)
  python $ROOT/profiles/summarize_rocpd.py "$T" $OUT/kernel_trace.txt > /dev/null
  find $OUT -name "*.db" -delete
done
cd $ROOT
for v in pad nopad; do echo "== $v"; grep -i "enum_node\|ev_leaf\|overflow_lds\|agg_item" gpurun_out/ab_$v/kernel_trace.txt | head -12; done

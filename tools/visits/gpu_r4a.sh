#!/bin/bash
# round 4, visit a: fused inter residuals (k_residual -> compact tiles -> k_inter_jobs write-back) vs the legacy order, ONE box.
# usage: tools/gpu_r4a.sh <tag> [tests]
TAG=$1; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [[ "$2" == *tests* ]]; then
  timeout 1500 python -m pytest tests -m gpu -x -q --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
fi
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end"
for f in 1 0 1 0; do
  export M355_RES_FUSED=$f
  for w in ${WORKLOADS:-c5_8k10_8tiles c3_4k_inter}; do
    for d in 3 1; do
      timeout 300 python bench.py $B --workload $w --steps ${STEPS:-200} --warmup 10 --pipeline-depth $d 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('fused=%s %-16s depth %d: %.4f ms/pic  one-at-a-time %.4f  %s' % ('$f', '$w', $d, d['ms_per_step'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $OUT/variants.txt
    done
  done
done
for f in 1 0; do
  export M355_RES_FUSED=$f
  cd /tmp; rocprofv3 --kernel-trace --stats -d $OUT/prof_f$f -o trace -- python $OLDPWD/bench.py $B --steps 60 --warmup 5 --pipeline-depth 1 > $OUT/prof_f$f.log 2>&1; cd $OLDPWD
  python tools/rocprof_summary.py $OUT/prof_f$f $OUT/kernel_stats_f$f.txt > /dev/null 2>&1 || true
  head -30 $OUT/kernel_stats_f$f.txt
done

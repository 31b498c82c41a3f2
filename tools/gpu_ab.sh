#!/bin/bash
# A/B of library builds on one box: tools/gpu_ab.sh <tag> "<variants: base = the product build, else libde265_amd/variants/<name>.so>" [tests] [diag]
TAG=$1; VARS=$2; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
if [[ "$3" == *tests* ]]; then
  timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
fi
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end"
for v in $VARS; do
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$PWD/libde265_amd/variants/$v.so; fi
  for w in ${WORKLOADS:-c2_1080p_intra c3_4k_inter c5_8k10_8tiles}; do
    for d in ${DEPTHS:-1 3 8}; do
      [ $w != c2_1080p_intra ] && [ $d -gt 3 ] && continue
      timeout 300 python bench.py $B --workload $w --steps ${STEPS:-200} --warmup 10 --pipeline-depth $d 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-16s depth %d: %.4f ms/pic  one-at-a-time %.4f  %s' % ('$v', '$w', $d, d['ms_per_step'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $OUT/variants.txt
    done
  done
  if [[ "$3" == *diag* ]]; then timeout 300 python tools/diag_intra.py 1 4 16 2>>$OUT/bench.err | sed "s/^/$v /" | tee -a $OUT/diag_intra.txt; fi
done

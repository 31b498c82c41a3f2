#!/bin/bash
# round 4, visit l: does leaving part of every SIMD's register file to the other pictures' kernels help? (k_inter_jobs capped to 2 / 1 workgroups per CU by unused LDS)
TAG=$1; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end"
run() { local label=$1; shift
  for d in 3 4 1; do
    env "$@" timeout 300 python bench.py $B --workload c5_8k10_8tiles --steps 200 --warmup 10 --pipeline-depth $d 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-26s depth %d: %.4f ms/pic  one-at-a-time %.4f  %s' % ('$label', $d, d['ms_per_step'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $OUT/variants.txt
  done
}
for rep in 1 2; do
  run "3 workgroups / CU (product)" X=1
  run "2 workgroups / CU" M355_INTER_LDS_PAD=56000
  run "1 workgroup / CU" M355_INTER_LDS_PAD=90000
done

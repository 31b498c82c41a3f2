#!/bin/bash
# round 4, visit h: what can be taken OUT of k_inter_jobs (the stage the others queue behind): pb_of written by k_meta_pb, luma / chroma as two launches
TAG=$1; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$PWD
timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 > $OUT/pytest.log 2>&1; echo "pytest exit $?" >> $OUT/pytest.log; tail -3 $OUT/pytest.log
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end"
run() { # label env...
  local label=$1; shift
  for d in 3 1; do
    env "$@" timeout 300 python bench.py $B --workload c5_8k10_8tiles --steps 200 --warmup 10 --pipeline-depth $d 2>>$OUT/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-22s depth %d: %.4f ms/pic  one-at-a-time %.4f  %s' % ('$label', $d, d['ms_per_step'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $OUT/variants.txt
  done
}
for rep in 1 2; do
  run base X=1
  run pb_of_in_meta M355_PB_OF_IN_META=1
  run split M355_INTER_SPLIT=1
  run split+pb_of M355_INTER_SPLIT=1 M355_PB_OF_IN_META=1
  run fused_all_depths M355_RES_FUSED=1
done
( time timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | grep real
python -c "
import json; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['ms_per_step_spread'], d['stage_ms'], d['roofline']['frac'], d['with_upload']['submit_only'], d['with_upload']['ms_per_step'], d['end_to_end'].get('speedup'), d['end_to_end']['with_output'].get('speedup'))"

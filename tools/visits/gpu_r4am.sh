#!/bin/bash
# round 4, visit am: both residual launches on the lane's main stream (new) vs side by side on its two streams (M355_RES_SIDE=1), C5, 3 in flight
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r4am; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain"
for rep in 1 2 3; do for e in 1 0; do
  M355_RES_SIDE=$e timeout 200 python bench.py $B --workload c5_8k10_8tiles --steps 200 --warmup 10 --pipeline-depth 3 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('residual side by side=$e  %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f  %s' % (d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $O/res_side.txt
done; done

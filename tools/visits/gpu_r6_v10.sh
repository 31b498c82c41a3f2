#!/bin/bash
# round 6, visit 10: what do k_tu_plan + k_intra cost a C5 picture with three in flight?  (diagnostic stage masks: 31 = all, 27 = without the intra stage and its planner)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v10; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain --no-verify"
for st in 31 27 31 27 25 29; do
  timeout 200 python bench.py $B --workload c5_8k10_8tiles --steps 200 --warmup 10 --pipeline-depth 3 --stages $st 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stages $st: %.4f ms/pic  one-at-a-time %.4f  %s' % (d['ms_per_step'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f' % kv for kv in d['stage_ms'].items())))" | tee -a $O/stage_masks.txt
done

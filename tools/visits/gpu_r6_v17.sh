#!/bin/bash
# round 6, visit 17: what evicts k_inter's reference frames from the Infinity Cache — plain / non-temporal streaming reads and writes (ub_kinter), and k_inter's
# launch time by the stages that run between two of its launches (diagnostic stage masks, one picture at a time)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v17; mkdir -p $O
timeout 300 tools/ubench/_build/ub_kinter 50 2>&1 | tee -a $O/ub_kinter.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain --no-verify"
for st in 1 3 7 15 31 1 9 17 25; do
  timeout 200 python bench.py $B --workload c5_8k10_8tiles --steps 200 --warmup 10 --pipeline-depth 1 --stages $st 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('stages %2d: %.4f ms/pic  %s' % ($st, d['ms_per_step'], ' '.join('%s=%.4f' % kv for kv in d['stage_ms'].items())))" | tee -a $O/stage_masks_depth1.txt
done

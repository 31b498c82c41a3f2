#!/bin/bash
# round 6, visit 20: pictures started in phase-aligned groups (the device drained after every N steps): do the k_inter launches of pictures that share their references, running together, find them warm?
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v20; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain --no-verify"
for cfg in "3 0" "3 3" "3 6" "6 6" "6 0" "4 4" "2 2" "3 0"; do set -- $cfg
  timeout 200 python bench.py $B --workload c5_8k10_8tiles --steps 198 --warmup 12 --pipeline-depth $1 --group-sync $2 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('depth $1 group-sync $2: %.4f ms/pic  %s' % (d['ms_per_step'], ' '.join('%s=%.4f' % kv for kv in d['stage_ms'].items())))" | tee -a $O/group_sync.txt
done

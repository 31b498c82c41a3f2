#!/bin/bash
# round 6, visit 22: all lanes' k_inter launches on one shared stream (M355_X_INTER_TRAIN): do back-to-back prediction launches on shared references run warm?
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v22; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('train %s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) verified %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d.get('verified')))"; }
run() { M355_X_INTER_TRAIN=$1 timeout 300 python bench.py $B --workload $2 --steps 200 --warmup 10 --pipeline-depth $3 2>>$O/bench.err | line $1 $2 $3 | tee -a $O/train_ab.txt; }
for d in 3 4 6; do for m in 0 1 0 1; do run $m c5_8k10_8tiles $d; done; done
for m in 0 2 0 2; do run $m c3_4k_inter 3; done

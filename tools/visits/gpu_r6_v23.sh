#!/bin/bash
# round 6, visit 23: m355_decode_batch for pictures with prediction blocks (stage by stage on one batch stream): parity, then the bench's timed region in batches against the lanes
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v23; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp parity
timeout 900 python -m pytest tests/test_gpu_batch_inter.py tests/test_gpu_batch.py -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-16s depth %s batch %s streams %s: %.4f ms/pic (p10 %.4f p90 %.4f) verified %s' % ('$1', '$2', '$3', '$4', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d.get('verified')))"; }
run() { # workload depth batch streams
  if [ "$4" = - ]; then unset M355_BATCH_STREAMS; else export M355_BATCH_STREAMS=$4; fi
  timeout 300 python bench.py $B --workload $1 --steps 198 --warmup 12 --pipeline-depth $2 --inter-batch $3 2>>$O/bench.err | line $1 $2 $3 $4 | tee -a $O/batch_ab.txt
  unset M355_BATCH_STREAMS
}
stamp C5
run c5_8k10_8tiles 3 0 -
for cfg in "3 3 -" "6 3 -" "9 3 -" "6 2 -" "8 2 -" "6 6 -" "12 3 -" "6 3 1" "9 3 2"; do set -- $cfg; run c5_8k10_8tiles $1 $2 $3; done
run c5_8k10_8tiles 3 0 -
stamp "C3 / C4"
for w in c3_4k_inter c4_4k_4tiles; do run $w 3 0 -; for cfg in "6 3 -" "9 3 -" "12 3 -" "12 6 -"; do set -- $cfg; run $w $1 $2 $3; done; done
stamp done

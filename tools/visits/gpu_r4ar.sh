#!/bin/bash
# round 4, visit ar: one-stream pictures (up to 4K) on stream number `lane` of the context's streams in creation order (M355_SPREAD_SINGLE=1)
# vs on their lane's main stream; C3 / C4 at depth 3 and 4; hardware parity of the pipeline tests with the switch on
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r4ar; mkdir -p $O
M355_SPREAD_SINGLE=1 timeout 200 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_random.py -m gpu -x -q 2>&1 | tail -1 | tee $O/parity.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain"
for w in c3_4k_inter c4_4k_4tiles; do for d in 3 4; do for s in 0 1; do
  M355_SPREAD_SINGLE=$s timeout 200 python bench.py $B --workload $w --steps 200 --warmup 10 --pipeline-depth $d 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-16s depth $d spread=$s: %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f' % ('$w', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight']))" | tee -a $O/spread.txt
done; done; done

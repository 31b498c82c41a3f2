#!/bin/bash
# round 4, visit u: batches of up to 32 pictures, persistent grid of the shared launch, two batches side by side
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4u
O=$REPO/gpurun_out/r4u
run() {  # workload depth batch [env...]
  local w=$1 d=$2 b=$3; shift 3
  env "$@" timeout 300 python bench.py --workload $w --steps 192 --warmup 32 --repeats 9 --pipeline-depth $d --intra-batch $b --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end 2>$O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w depth $d batch $b $*: %.4f ms/pic = %.3f M CTB64/s (p10 %.4f p90 %.4f; enqueue %.4f)' % (d['ms_per_step'], d['value']/1e6, d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['host_enqueue_ms_per_step']))" | tee -a $O/summary.txt
}
C2=c2_1080p_intra
run $C2 16 16 M355_BATCH_STREAMS=1
run $C2 16 16 M355_BATCH_STREAMS=1 M355_INTRA_GRID=256
run $C2 32 32 M355_BATCH_STREAMS=1
run $C2 32 32 M355_BATCH_STREAMS=1 M355_INTRA_GRID=384
run $C2 32 32 M355_BATCH_STREAMS=1 M355_INTRA_GRID=768
run $C2 32 16 M355_BATCH_STREAMS=2
run $C2 32 16 M355_BATCH_STREAMS=2 M355_INTRA_GRID=256
run $C2 32 16 M355_BATCH_STREAMS=2 M355_INTRA_GRID=192
run $C2 16 8 M355_BATCH_STREAMS=2
run $C2 16 8 M355_BATCH_STREAMS=2 M355_INTRA_GRID=256
run $C2 16 8 M355_BATCH_STREAMS=2 M355_INTRA_GRID=128
run $C2 32 8 M355_BATCH_STREAMS=4
run $C2 32 8 M355_BATCH_STREAMS=4 M355_INTRA_GRID=128
run $C2 32 8 M355_BATCH_STREAMS=3 M355_INTRA_GRID=160

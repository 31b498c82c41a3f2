#!/bin/bash
# round 4, visit p: hardware queues (GPU_MAX_HW_QUEUES) for intra pictures in flight and for batches; effect on the C5 headline
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4p
O=gpurun_out/r4p
run() {  # workload depth batch [env...]
  local w=$1 d=$2 b=$3; shift 3
  env "$@" timeout 300 python bench.py --workload $w --steps 192 --warmup 16 --repeats 9 --pipeline-depth $d --intra-batch $b --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end 2>$O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w depth $d batch $b $*: %.4f ms/pic = %.3f M CTB64/s (p10 %.4f p90 %.4f; enqueue %.4f)' % (d['ms_per_step'], d['value']/1e6, d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['host_enqueue_ms_per_step']))" | tee -a $O/summary.txt
}
C2=c2_1080p_intra
run $C2 8 0
for q in 8 16 32; do run $C2 8 0 GPU_MAX_HW_QUEUES=$q M355_LANE_PRIORITIES=0; done
run $C2 16 0 GPU_MAX_HW_QUEUES=32 M355_LANE_PRIORITIES=0
run $C2 16 0 GPU_MAX_HW_QUEUES=16
run $C2 16 16
run $C2 16 16 GPU_MAX_HW_QUEUES=16
run $C2 16 16 GPU_MAX_HW_QUEUES=32
run $C2 16 8
run $C2 16 8 GPU_MAX_HW_QUEUES=16
run $C2 16 8 GPU_MAX_HW_QUEUES=32
run $C2 16 8 GPU_MAX_HW_QUEUES=32 M355_BATCH_STREAMS=2 M355_BATCH_STREAM_PRIO=0
run $C2 16 4 GPU_MAX_HW_QUEUES=32
run $C2 16 4 GPU_MAX_HW_QUEUES=32 M355_BATCH_STREAMS=2 M355_BATCH_STREAM_PRIO=0
run c5_8k10_8tiles 3 0
run c5_8k10_8tiles 3 0 GPU_MAX_HW_QUEUES=8
run c5_8k10_8tiles 3 0 GPU_MAX_HW_QUEUES=16
run c5_8k10_8tiles 4 0 GPU_MAX_HW_QUEUES=8

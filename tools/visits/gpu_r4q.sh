#!/bin/bash
# round 4, visit q: whole batches on streams of their own (M355_BATCH_STREAMS), C2 by lanes x batch x streams
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4q
O=gpurun_out/r4q
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log | tee -a $O/summary.txt
run() {  # workload depth batch [env...]
  local w=$1 d=$2 b=$3; shift 3
  env "$@" timeout 300 python bench.py --workload $w --steps 192 --warmup 16 --repeats 9 --pipeline-depth $d --intra-batch $b --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end 2>$O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w depth $d batch $b $*: %.4f ms/pic = %.3f M CTB64/s (p10 %.4f p90 %.4f; enqueue %.4f)' % (d['ms_per_step'], d['value']/1e6, d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['host_enqueue_ms_per_step']))" | tee -a $O/summary.txt
}
C2=c2_1080p_intra
run $C2 16 16 M355_BATCH_STREAMS=0
run $C2 16 16
run $C2 16 8
run $C2 16 8 M355_BATCH_STREAMS=2
run $C2 16 4
run $C2 16 4 M355_BATCH_STREAMS=3
run $C2 16 2
run $C2 16 1
run $C2 12 4 M355_BATCH_STREAMS=3
run $C2 12 3
run $C2 8 4
run $C2 8 2
run $C2 8 1
run $C2 4 4
run $C2 4 2
run $C2 4 1
run $C2 4 0
run $C2 3 0

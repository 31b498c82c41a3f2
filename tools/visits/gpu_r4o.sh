#!/bin/bash
# round 4, visit o: m355_decode_batch — dedicated batch streams, persistent grid size, batch vs lanes (C2), one call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4o
O=gpurun_out/r4o
timeout 900 python -m pytest tests/test_gpu_batch.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log | tee -a $O/summary.txt
run() {  # depth batch [env...]
  local d=$1 b=$2; shift 2
  env "$@" timeout 300 python bench.py --workload c2_1080p_intra --steps 192 --warmup 16 --repeats 9 --pipeline-depth $d --intra-batch $b --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end 2>$O/err_${d}_${b}.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 depth $d batch $b $*: %.4f ms/pic = %.3f M CTB64/s (p10 %.4f p90 %.4f; enqueue %.4f)' % (d['ms_per_step'], 510/d['ms_per_step']/1e3, d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['host_enqueue_ms_per_step']))" | tee -a $O/summary.txt
}
run 8 0
run 8 8 M355_BATCH_STREAMS=0
run 8 8
run 16 16 M355_BATCH_STREAMS=0
run 16 16
for g in 128 192 256 384; do run 16 16 M355_INTRA_GRID=$g; done
run 16 8 M355_BATCH_STREAMS=0
run 16 8
for g in 96 128 192 256; do run 16 8 M355_INTRA_GRID=$g; done
run 16 4
run 16 4 M355_INTRA_GRID=64
run 12 6
run 12 6 M355_INTRA_GRID=128

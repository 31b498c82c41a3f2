#!/bin/bash
# round 4, visit n: m355_decode_batch (several intra pictures, one k_intra launch) — parity, then C2 by lanes and by batch in ONE call
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4n
O=gpurun_out/r4n
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_pipeline.py tests/test_gpu_encintra.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log | tee -a $O/summary.txt
run() {  # depth batch [env...]
  local d=$1 b=$2; shift 2
  env "$@" timeout 300 python bench.py --workload c2_1080p_intra --steps 200 --warmup 16 --pipeline-depth $d --intra-batch $b --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end 2>$O/err_${d}_${b}.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 depth $d batch $b $*: %.4f ms/pic = %.3f M CTB64/s (p10 %.4f p90 %.4f; one at a time %.4f; enqueue %.4f)' % (d['ms_per_step'], 510/d['ms_per_step']/1e3, d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], d['host_enqueue_ms_per_step']))" | tee -a $O/summary.txt
}
run 3 0
run 8 0
run 2 2
run 3 3
run 4 4
run 6 6
run 8 8
run 8 4
run 12 12
run 16 16
run 16 8
run 8 8 M355_INTRA_GRID=256
run 8 8 M355_INTRA_GRID=1024
run 8 0 M355_LANE_PRIORITIES=0
run 8 8 M355_LANE_PRIORITIES=0

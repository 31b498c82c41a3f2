#!/bin/bash
# round 4, visit w: m355_decode_batch with its defaults (streams by lanes / batch, grid by wavefront width): parity + C2 table
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4w
O=$REPO/gpurun_out/r4w
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_pipeline.py tests/test_gpu_synth.py tests/test_gpu_random.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
tail -3 $O/tests.log | tee -a $O/summary.txt
run() {  # workload depth batch [env...]
  local w=$1 d=$2 b=$3; shift 3
  env "$@" timeout 300 python bench.py --workload $w --steps 192 --warmup 32 --repeats 9 --pipeline-depth $d --intra-batch $b --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end 2>$O/err.log | tee $O/bench_${w}_${d}_${b}.json | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w depth $d batch $b $*: %.4f ms/pic = %.3f M CTB64/s (p10 %.4f p90 %.4f; enqueue %.4f)' % (d['ms_per_step'], d['value']/1e6, d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['host_enqueue_ms_per_step']))" | tee -a $O/summary.txt
}
C2=c2_1080p_intra
run $C2 3 0
run $C2 2 2
run $C2 4 4
run $C2 4 2
run $C2 8 0
run $C2 8 8
run $C2 8 4
run $C2 16 16
run $C2 16 8
run $C2 16 4
run $C2 32 32
run $C2 32 16
run $C2 32 8
run $C2 24 8
run $C2 24 12
run c5_8k10_8tiles 3 0

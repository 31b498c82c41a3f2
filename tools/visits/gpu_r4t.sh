#!/bin/bash
# round 4, visit t: m355_decode_batch with every stage batched (one launch per stage for the whole batch)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4t
O=$REPO/gpurun_out/r4t
timeout 900 python -m pytest tests/test_gpu_batch.py tests/test_gpu_pipeline.py tests/test_gpu_synth.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" | tee -a $O/summary.txt
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
run $C2 16 16 M355_BATCH_STREAMS=1
run $C2 16 8
run $C2 16 8 M355_BATCH_STREAMS=2
run $C2 16 8 M355_BATCH_STREAMS=1
run $C2 16 4
run $C2 16 4 M355_BATCH_STREAMS=2
run $C2 16 2
run $C2 8 8
run $C2 8 4
run $C2 8 4 M355_BATCH_STREAMS=2
run $C2 8 2
run $C2 4 4
run $C2 4 2
run $C2 4 2 M355_BATCH_STREAMS=2
run $C2 3 0
run c5_8k10_8tiles 3 0
cd /tmp && export TMPDIR=/tmp
for cfg in "16 16 1" "16 8 2"; do set -- $cfg
  rm -rf /tmp/kt; M355_BATCH_STREAMS=$3 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $REPO/bench.py --workload $C2 --steps 64 --warmup 16 --repeats 3 --pipeline-depth $1 --intra-batch $2 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end > /dev/null 2>$O/prof_err.log
  echo "--- kernel stats depth $1 batch $2 streams $3" >> $O/summary.txt
  python $REPO/tools/rocprof_summary.py /tmp/kt $O/kernel_stats_$1_$2_$3.txt | head -24 | cut -c1-180 >> $O/summary.txt
done

#!/bin/bash
# round 4, visit r: batch ring of 16; kernel trace of a 16-picture batch
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4r
O=gpurun_out/r4r
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
run $C2 16 2
run $C2 8 4
run $C2 8 2
run $C2 4 2
run $C2 4 1
cd /tmp && export TMPDIR=/tmp
for cfg in "16 16 0" "16 4 4"; do set -- $cfg
  rm -rf /tmp/kt; M355_BATCH_STREAMS=$3 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --workload $C2 --steps 64 --warmup 16 --repeats 3 --pipeline-depth $1 --intra-batch $2 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end > /dev/null 2>$GRAFT_REPO_ROOT/$O/prof_err.log
  f=$(find /tmp/kt -name '*kernel_stats.csv' | head -1)
  echo "--- kernel stats depth $1 batch $2 streams $3" >> $GRAFT_REPO_ROOT/$O/summary.txt
  head -14 "$f" | cut -c1-200 >> $GRAFT_REPO_ROOT/$O/summary.txt
done

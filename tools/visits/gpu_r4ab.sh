#!/bin/bash
# round 4, visit ab: k_intra_plan with the next record requested ahead and a grid row count by the largest CTB — kernel trace C2 / C5 / C2 batch
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
O=$REPO/gpurun_out/r4ab; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_synth.py tests/test_gpu_batch.py tests/test_gpu_encintra.py tests/test_gpu_girlshy.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$? $(tail -1 $O/tests.log)" | tee -a $O/summary.txt
cd /tmp
for cfg in "c2_1080p_intra 1 0" "c5_8k10_8tiles 1 0" "c2_1080p_intra 16 16"; do set -- $cfg
  rm -rf /tmp/kt; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $REPO/bench.py --workload $1 --steps 64 --warmup 16 --repeats 3 --pipeline-depth $2 --intra-batch $3 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end > /dev/null 2>$O/prof_err.log
  echo "--- kernel stats $1 depth $2 batch $3" >> $O/summary.txt
  python $REPO/tools/rocprof_summary.py /tmp/kt $O/kernel_stats_$1_$2_$3.txt | grep -i "kernel \|plan" | cut -c1-170 >> $O/summary.txt
done
cd $REPO
for cfg in "16 16" "32 8" "3 0"; do set -- $cfg
  timeout 300 python bench.py --workload c2_1080p_intra --steps 192 --warmup 32 --repeats 9 --pipeline-depth $1 --intra-batch $2 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end 2>$O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 depth $1 batch $2: %.4f ms/pic = %.3f M CTB64/s' % (d['ms_per_step'], d['value']/1e6))" | tee -a $O/summary.txt
done
timeout 300 python bench.py --workload c5_8k10_8tiles --steps 200 --warmup 10 --pipeline-depth 3 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end 2>$O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5 depth 3: %.4f ms/pic = %.3f M CTB64/s, one at a time %.4f' % (d['ms_per_step'], d['value']/1e6, d['ms_per_step_one_in_flight']))" | tee -a $O/summary.txt
cat $O/summary.txt

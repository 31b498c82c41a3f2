#!/bin/bash
# round 4, visit s: kernel traces of batches (how long is the shared k_intra launch with 4 / 8 / 16 pictures?)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4s
O=$REPO/gpurun_out/r4s
C2=c2_1080p_intra
cd /tmp && export TMPDIR=/tmp
for cfg in "16 16 0" "8 8 0" "4 4 0" "16 4 4" "16 8 2"; do set -- $cfg
  rm -rf /tmp/kt; M355_BATCH_STREAMS=$3 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt -- python $REPO/bench.py --workload $C2 --steps 64 --warmup 16 --repeats 3 --pipeline-depth $1 --intra-batch $2 --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end > /dev/null 2>$O/prof_err.log
  echo "--- kernel stats depth $1 batch $2 streams $3" >> $O/summary.txt
  python $REPO/tools/rocprof_summary.py /tmp/kt $O/kernel_stats_$1_$2_$3.txt | head -14 | cut -c1-180 >> $O/summary.txt
done
cat $O/summary.txt

#!/bin/bash
# round 5, visit 28: kernel traces with ONE picture in flight (the averages bench.py's roofline.launch_ms has to agree with: stage events are taken one picture
# at a time), C2 batches twice
#   gpurun --timeout 600 -- 'bash tools/gpu_r5_v28.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v28; mkdir -p $O
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --no-cold-refs"
cd /tmp
for w in c5_8k10_8tiles c3_4k_inter c4_4k_4tiles c2_1080p_intra; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$w -o x --output-format csv -- python $REPO/bench.py $B --workload $w --steps 50 --warmup 5 --pipeline-depth 1 > $O/trace_$w.log 2>&1
  f=$(find $O/trace_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-220 "$f" | head -24 > $O/${w}_depth1_kernel_stats.txt
done
cd $REPO
for i in 1 2; do timeout 300 python bench.py --workload c2_1080p_intra --steps 96 --warmup 8 --pipeline-depth 32 --intra-batch 8 $B 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 batch 32x8', d['value'], d['ms_per_step'], d['ms_per_step_spread'])" | tee -a $O/c2_batch.txt; done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +5M -delete

#!/bin/bash
# round 5, visit 4: the lean EDGE path (all clamped rows in flight) — parity, then C3 / C4 / C5 on this box
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v4; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain"
timeout 600 python -m pytest tests/test_inter_extremes.py tests/test_gpu_random.py tests/test_gpu_synth.py tests/test_gpu_girlshy.py tests/test_gpu_pipeline.py tests/test_streams.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/parity.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
for rep in 1 2; do for wd in "c5_8k10_8tiles 3" "c5_8k10_8tiles 1" "c3_4k_inter 3" "c3_4k_inter 1" "c4_4k_4tiles 3"; do set -- $wd
  timeout 200 python bench.py $B --workload $1 --steps 200 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | line base $1 $2 | tee -a $O/edge_lean.txt
done; done
cd /tmp
for w in c3_4k_inter; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$w -o x --output-format csv -- python $REPO/bench.py $B --workload $w --steps 50 --warmup 5 --pipeline-depth 1 > $O/trace_$w.log 2>&1
  f=$(find $O/trace_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-220 "$f" | head -24 > $O/${w}_kernel_stats.txt
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +5M -delete

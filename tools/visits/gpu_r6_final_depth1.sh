#!/bin/bash
# round 6, final tree: kernel traces with ONE picture in flight (what roofline.launch_ms is measured on), C5 and C3
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$REPO/gpurun_out/r6zz1; mkdir -p $O
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --no-cold-refs"
cd /tmp
for w in c5_8k10_8tiles c3_4k_inter; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$w -o x --output-format csv -- python $REPO/bench.py $B --workload $w --steps 50 --warmup 5 --pipeline-depth 1 > $O/trace_$w.log 2>&1
  f=$(find $O/trace_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-220 "$f" | head -24 > $O/${w}_depth1_kernel_stats.txt
  tail -1 $O/trace_$w.log | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['ms_per_step'], d['stage_ms'], d['roofline']['launch_ms'], d['roofline']['frac'])" | tee -a $O/lines.txt
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +5M -delete

#!/bin/bash
# round 6, visit 1: the tree after the ADVICE fixes — whole GPU tier, the driver's bench command, one-picture kernel trace of C5 (this round's baseline on this pool)
#   gpurun --timeout 1200 -- 'bash tools/visits/gpu_r6_v1.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v1; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "GPU tier"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -5 | tee $O/pytest_all.txt
stamp "driver's command"
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench.err ) 2>&1 | grep real | tee -a $O/timeline.txt
python -c "
import json; d=json.loads(open('$O/bench_driver_line.json').read().strip().splitlines()[-1]); print('driver-like:', d['value'], d['ms_per_step'], d['stage_ms'], 'frac', d['roofline']['frac'], 'chain', json.dumps(d.get('dependent_chain'))[:200])" | tee -a $O/timeline.txt
stamp "kernel trace C5 depth 1"
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --no-cold-refs"
cd /tmp
for w in c5_8k10_8tiles c3_4k_inter; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$w -o x --output-format csv -- python $REPO/bench.py $B --workload $w --steps 50 --warmup 5 --pipeline-depth 1 > $O/trace_$w.log 2>&1
  f=$(find $O/trace_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-220 "$f" | head -24 > $O/${w}_depth1_kernel_stats.txt
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +5M -delete
stamp done

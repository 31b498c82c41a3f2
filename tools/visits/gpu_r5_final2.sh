#!/bin/bash
# round 5, final tree after the k_intra work (clock levels, pruned plans, halo keeper + in-kernel planning at depth 1, shared 32x32): whole GPU tier,
# the driver's bench command, one JSON per BASELINE config, C2 at depth 1 / 3 / batches, kernel traces of C2
#   gpurun --timeout 1800 -- 'bash tools/gpu_r5_final2.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5zz; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "GPU tier"
timeout 900 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_all.txt
stamp "driver's command"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench.err ) 2>&1 | grep real | tee -a $O/timeline.txt
python -c "
import json; d=json.loads(open('$O/bench_driver_line.json').read().strip().splitlines()[-1]); print('driver-like:', d['value'], d['ms_per_step'], d['stage_ms'], 'frac', d['roofline']['frac'])" | tee -a $O/timeline.txt
stamp "other configs"
for w in c2_1080p_intra c3_4k_inter c4_4k_4tiles; do timeout 400 python bench.py --workload $w --steps 100 --warmup 10 --no-end-to-end > $O/bench_$w.json 2>>$O/bench.err; python -c "
import json; d=json.loads(open('$O/bench_$w.json').read().strip().splitlines()[-1]); print('$w', d['value'], d['ms_per_step'], d['ms_per_step_one_in_flight'], d['stage_ms'], 'with_upload', json.dumps(d.get('with_upload'))[:300], 'cpu', d.get('cpu_baseline'))" | tee -a $O/timeline.txt; done
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --no-cold-refs"
timeout 300 python bench.py --workload c2_1080p_intra --steps 100 --warmup 10 --pipeline-depth 1 $B > $O/bench_c2_depth1.json 2>>$O/bench.err
timeout 300 python bench.py --workload c2_1080p_intra --steps 64 --warmup 8 --pipeline-depth 32 --intra-batch 8 $B > $O/bench_c2_batch.json 2>>$O/bench.err
for f in c2_depth1 c2_batch; do python -c "
import json; d=json.loads(open('$O/bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['ms_per_step_one_in_flight'], d['stage_ms'])" | tee -a $O/timeline.txt; done
stamp "kernel traces"
cd /tmp
for dd in 1 3; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_c2_d$dd -o x --output-format csv -- python $REPO/bench.py $B --workload c2_1080p_intra --steps 50 --warmup 5 --pipeline-depth $dd > $O/trace_c2_d$dd.log 2>&1
  f=$(find $O/trace_c2_d$dd -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-220 "$f" | head -16 > $O/c2_depth${dd}_kernel_stats.txt
done
timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_c5 -o x --output-format csv -- python $REPO/bench.py $B --workload c5_8k10_8tiles --steps 50 --warmup 5 > $O/trace_c5.log 2>&1
f=$(find $O/trace_c5 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-220 "$f" | head -24 > $O/c5_kernel_stats.txt
cd $REPO
find $O -name "*.db" -delete; find $O -name "*counter_collection.csv" -size +5M -delete; find $O -name "*kernel_trace.csv" -size +5M -delete
stamp done

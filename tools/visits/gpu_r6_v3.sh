#!/bin/bash
# round 6, visit 3: k_inter_jobs with the reference windows through LDS — parity (whole GPU tier), then A/B against the per-lane kernel of the commit before (variants/perlane.so)
#   gpurun --timeout 1500 -- 'bash tools/visits/gpu_r6_v3.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v3; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "inter parity first"
timeout 600 python -m pytest tests/test_gpu_synth.py tests/test_gpu_random.py tests/test_inter_extremes.py tests/test_inter_narrow.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee $O/pytest_inter.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain --no-verify"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f stages %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f' % kv for kv in d['stage_ms'].items())))"; }
run() { # variant workload depth
  if [ "$1" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$1.so; fi
  timeout 200 python bench.py $B --workload $2 --steps 200 --warmup 10 --pipeline-depth $3 2>>$O/bench.err | line $1 $2 $3 | tee -a $O/inter_ab.txt
  unset M355_LIB
}
stamp "A/B"
for wd in "c5_8k10_8tiles 3" "c3_4k_inter 3" "c4_4k_4tiles 3" "c5x_cu64 3" "c5x_cu16 3"; do set -- $wd; for v in base perlane base perlane; do run $v $1 $2; done; done
stamp "GPU tier (rest)"
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_all.txt
stamp "kernel trace C5 depth 1"
cd /tmp
for w in c5_8k10_8tiles c3_4k_inter; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/trace_$w -o x --output-format csv -- python $REPO/bench.py $B --workload $w --steps 50 --warmup 5 --pipeline-depth 1 > $O/trace_$w.log 2>&1
  f=$(find $O/trace_$w -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cut -c1-220 "$f" | head -24 > $O/${w}_depth1_kernel_stats.txt
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -size +5M -delete
stamp done

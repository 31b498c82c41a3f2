#!/bin/bash
# round 5, visit 22: where does the prologue of an inter picture's CTB go (stamp 1 behind the plan / the residuals / the need scan's barrier / the staging),
# and w6 = the sparse kernel at 80 registers (6 workgroups per CU instead of 5; 8 bytes of scratch)
#   gpurun --timeout 900 -- 'bash tools/gpu_r5_v22.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v22; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "parity"
timeout 300 python -m pytest tests/test_gpu_synth.py tests/test_gpu_random.py tests/test_gpu_girlshy.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/parity.txt
M355_LIB=$REPO/libde265_amd/variants/w6.so timeout 300 python -m pytest tests/test_gpu_synth.py tests/test_gpu_random.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee -a $O/parity.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-6s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
run() { # variant workload depth
  if [ "$1" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$1.so; fi
  timeout 200 python bench.py $B --workload $2 --steps 200 --warmup 10 --pipeline-depth $3 2>>$O/bench.err | line $1 $2 $3 | tee -a $O/intra_w6_ab.txt
  unset M355_LIB
}
stamp "A/B"
for v in base w6 w6 base; do run $v c5_8k10_8tiles 3; done
for wd in "c5_8k10_8tiles 1" "c3_4k_inter 3" "c4_4k_4tiles 3"; do set -- $wd; for v in base w6; do run $v $1 $2; done; done
stamp "timelines"
for w in c3_4k_inter c5_8k10_8tiles; do for v in prof1 prof2 prof3 prof4 prof; do echo "== $v" | tee -a $O/intra_prologue_stamps.txt; M355_LIB=$REPO/libde265_amd/variants/$v.so timeout 120 python tools/prof_timeline_sparse.py $w 2>&1 | grep -E "CTBs with|prologue|block loop|whole CTB" | tee -a $O/intra_prologue_stamps.txt; done; done
stamp done

#!/bin/bash
# round 5, visit 31: a chain picture transforms its residuals in its front part (tiles) and adds them behind k_inter (v30 = k_residual behind k_inter as for every picture); glue at three lanes
#   gpurun --timeout 900 -- 'bash tools/gpu_r5_v31.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v31; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "parity"
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_all.txt
M355_TEST_CHAIN_RESIDUALS=1 timeout 300 python -m pytest tests/test_gpu_synth.py tests/test_gpu_random.py tests/test_gpu_girlshy.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -1 | sed "s/^/every picture in the chain order: /" | tee -a $O/pytest_all.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
dc=d.get('dependent_chain') or {}
print('%-6s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f chain %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], json.dumps(dc)[:160]))"; }
run() { # variant workload depth
  if [ "$1" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$1.so; fi
  timeout 200 python bench.py $B --workload $2 --steps 200 --warmup 10 --pipeline-depth $3 2>>$O/bench.err | line $1 $2 $3 | tee -a $O/chain_residuals_front_ab.txt
  unset M355_LIB
}
stamp "A/B"
for wd in "c3_4k_inter 3" "c4_4k_4tiles 3" "c5_8k10_8tiles 3"; do set -- $wd; for v in v30 base base v30; do run $v $1 $2; done; done

stamp done

#!/bin/bash
# round 5, visit 25: a picture whose reference is still being written follows it onto that lane (stream order instead of a cross-queue event wait) —
# dependent chains with three lanes, v24 = lanes strictly in rotation
#   gpurun --timeout 900 -- 'bash tools/gpu_r5_v25.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v25; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "parity"
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_all.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
dc=d.get('dependent_chain') or {}
print('%-6s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f chain %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], json.dumps(dc)[:160]))"; }
run() { # variant workload depth
  if [ "$1" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$1.so; fi
  timeout 200 python bench.py $B --workload $2 --steps 200 --warmup 10 --pipeline-depth $3 2>>$O/bench.err | line $1 $2 $3 | tee -a $O/chain_affinity_ab.txt
  unset M355_LIB
}
stamp "A/B"
for wd in "c3_4k_inter 3" "c5_8k10_8tiles 3" "c4_4k_4tiles 3"; do set -- $wd; for v in v24 base base v24; do run $v $1 $2; done; done
for wd in "c3_4k_inter 2" "c5_8k10_8tiles 2" "c3_4k_inter 1" "c5_8k10_8tiles 1"; do set -- $wd; for v in v24 base; do run $v $1 $2; done; done
stamp done

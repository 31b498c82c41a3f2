#!/bin/bash
# round 6, visit 13: (1) hipExtAnyOrderLaunch on this stack (tools/ubench/ub_anyorder.hip); (2) k_inter_jobs: what a list costs a workgroup (all PBs from one list / from two,
# no weights, no out-of-picture vectors) and the two main classes alone at 3 and at 4 waves per SIMD (118-120 registers without the weighted / EDGE paths in the kernel);
# (3) the row pipeline's depth re-measured on the round's kernel
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v13; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "any-order launch"
timeout 60 tools/ubench/_build/ub_anyorder 2>&1 | tee $O/ub_anyorder.txt
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
stamp "lists"
for w in c5x_uni c5x_bi c5x_plain; do for v in base main_w3 main_w4 base main_w3 main_w4; do run $v $w 1; done; done
stamp "three in flight"
for v in base main_w3 main_w4 base main_w3 main_w4; do run $v c5x_plain 3; done
stamp "pipeline depth"
for wd in "c5_8k10_8tiles 1" "c5_8k10_8tiles 3" "c3_4k_inter 3"; do set -- $wd; for v in base pipe1 pipe3 base pipe1 pipe3; do run $v $1 $2; done; done
stamp done

#!/bin/bash
# round 6, visit 48: which hardware queue each lane's main / side stream sits on (variant qmap, M355_QMAP), C5, three and four lanes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v48; mkdir -p $O
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --no-cold-refs"
run() { # name depth qmap
  L=$GRAFT_REPO_ROOT/libde265_amd/variants/qmap.so; [ "$1" = product ] && L=
  M355_QMAP=$3 M355_LIB=$L timeout 200 python bench.py $B --workload c5_8k10_8tiles --steps 200 --warmup 20 --pipeline-depth $2 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('depth $2 $1 [$3]: %.4f ms/picture (p10 %.4f p90 %.4f), one at a time %.4f' % (d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight']))" | tee -a $O/qmap_ab.txt
}
for rep in 1 2; do
run product 3 ""
run as_product 3 "0,1,2,3,0,1"
run sides_on_q3 3 "0,3,1,3,2,3"
run own_queue 3 "0,0,1,1,2,2"
run cross 3 "0,1,1,2,2,0"
run mains_0_1_0 3 "0,2,1,3,0,3"
run two_queues 3 "0,1,0,1,0,1"
run pair_then_single 3 "0,1,0,1,2,3"
done
run product 4 ""
run sides_on_q3 4 "0,3,1,3,2,3,0,3"
run d4_pairs 4 "0,1,2,3,0,1,2,3"
run d4_sides_q3 4 "0,3,1,3,2,3,1,3"
run product 5 ""
run d5_sides_q3 5 "0,3,1,3,2,3,0,3,1,3"

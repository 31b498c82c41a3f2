#!/bin/bash
# round 6, visit 47: the lanes' side streams created only when a picture needs them (variant lazy2: at 4K the three lanes' main streams are then the first three streams
# the runtime maps onto its hardware queues) against the product, C3 / C4 / C5, 3 .. 5 lanes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v47; mkdir -p $O
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --no-cold-refs"
for w in c3_4k_inter c4_4k_4tiles c5_8k10_8tiles; do for d in 3 4 5; do for v in product lazy2 product lazy2; do
  L=; [ $v = lazy2 ] && L=$GRAFT_REPO_ROOT/libde265_amd/variants/lazy2.so
  M355_LIB=$L timeout 200 python bench.py $B --workload $w --steps 200 --warmup 20 --pipeline-depth $d 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w depth $d $v: %.4f ms/picture (p10 %.4f p90 %.4f), one at a time %.4f' % (d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight']))" | tee -a $O/lazy2_ab.txt
done; done; done

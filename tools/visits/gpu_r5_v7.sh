#!/bin/bash
# round 5, visit 7: sparse k_intra per-CTB timeline (C3, C5); pictures in flight at 4K
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v7; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain"
for w in c3_4k_inter c5_8k10_8tiles; do M355_LIB=$REPO/libde265_amd/variants/prof.so timeout 120 python tools/prof_timeline_sparse.py $w 2>&1 | tail -12 | tee -a $O/intra_sparse_timeline.txt; done
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
for rep in 1 2; do for wd in "c3_4k_inter 2" "c3_4k_inter 3" "c3_4k_inter 4" "c3_4k_inter 6" "c4_4k_4tiles 4" "c5_8k10_8tiles 4"; do set -- $wd
  timeout 200 python bench.py $B --workload $1 --steps 200 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | line base $1 $2 | tee -a $O/depths.txt
done; done

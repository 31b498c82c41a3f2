#!/bin/bash
# round 5, visit 6: EDGE jobs through LDS row extension, tap tables from memory, planes + job list in one launch: timeline, parity, variants
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v6; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain"
for w in c3_4k_inter c5_8k10_8tiles; do M355_LIB=$REPO/libde265_amd/variants/prof.so timeout 120 python tools/prof_inter_timeline.py $w 2>&1 | tail -34 | tee -a $O/inter_timeline.txt; done
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_all.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-12s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
for rep in 1 2; do for v in base p0w3 p2w3 p3w3 p1w4 p3w4; do for wd in "c5_8k10_8tiles 3" "c3_4k_inter 3" "c3_4k_inter 1" "c4_4k_4tiles 3"; do set -- $wd
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  timeout 200 python bench.py $B --workload $1 --steps 200 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | line $v $1 $2 | tee -a $O/inter_variants.txt
done; done; done
unset M355_LIB

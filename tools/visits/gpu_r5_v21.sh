#!/bin/bash
# round 5, visit 21: inter pictures' CTBs on the intra picture's block code (LDS residuals inside the body, class loops, shared 32x32), chain-first
# work order, four-round-trip prologue — prev = the tree before (eb9be66), nw6 = six waves per sparse CTB (4 luma)
#   gpurun --timeout 1500 -- 'bash tools/gpu_r5_v21.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v21; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "GPU tier"
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_all.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-6s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
run() { # variant workload depth
  if [ "$1" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$1.so; fi
  timeout 200 python bench.py $B --workload $2 --steps 200 --warmup 10 --pipeline-depth $3 2>>$O/bench.err | line $1 $2 $3 | tee -a $O/intra_sparse_ab.txt
  unset M355_LIB
}
stamp "A/B"
for wd in "c3_4k_inter 3" "c5_8k10_8tiles 3"; do set -- $wd; for v in prev base nw6 base prev; do run $v $1 $2; done; done
for wd in "c3_4k_inter 1" "c5_8k10_8tiles 1" "c4_4k_4tiles 3" "c2_1080p_intra 1" "c2_1080p_intra 3"; do set -- $wd; for v in prev base; do run $v $1 $2; done; done
stamp "timelines"
for w in c3_4k_inter c5_8k10_8tiles; do M355_LIB=$REPO/libde265_amd/variants/prof.so timeout 120 python tools/prof_timeline_sparse.py $w 2>&1 | tail -12 | tee -a $O/intra_sparse_timeline.txt; done
stamp done

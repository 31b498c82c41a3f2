#!/bin/bash
# round 5, visit 8: the two prepared patches of round 4 (k_sao scalar-base addressing, k_meta_planes batched neighbour loads) against the tree,
# weighted write-back as dot2, rotating references, clean timelines
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v8; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain"
timeout 600 python -m pytest tests/test_inter_extremes.py tests/test_gpu_random.py tests/test_gpu_synth.py tests/test_gpu_girlshy.py tests/test_streams.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -2 | tee $O/parity.txt
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r=d.get('rotating_references')
print('%-12s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) %s %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items()), ('rotating_refs=%.4f' % r['ms_per_step']) if r else ''))"; }
for rep in 1 2 3; do for v in base sao_saddr meta_sao; do for wd in "c5_8k10_8tiles 3" "c3_4k_inter 3" "c4_4k_4tiles 3"; do set -- $wd
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  timeout 200 python bench.py $B --workload $1 --steps 200 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | line $v $1 $2 | tee -a $O/patches_ab.txt
done; done; done
unset M355_LIB
for w in c3_4k_inter c5_8k10_8tiles; do M355_LIB=$REPO/libde265_amd/variants/prof.so timeout 120 python tools/prof_inter_timeline.py $w 2>&1 | tail -34 | tee -a $O/inter_timeline.txt; done
for w in c3_4k_inter c5_8k10_8tiles; do M355_LIB=$REPO/libde265_amd/variants/prof.so timeout 120 python tools/prof_timeline_sparse.py $w 2>&1 | tail -12 | tee -a $O/intra_sparse_timeline.txt; done

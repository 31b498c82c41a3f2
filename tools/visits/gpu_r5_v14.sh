#!/bin/bash
# round 5, visit 14: k_intra — border entries a mode never reads pruned from the plan, the halo keeper wave (13th wave of an intra picture's
# workgroup), against the tree of visit r5z (variants: nw12 = pruned plan, no keeper; old = neither)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v14; mkdir -p $O
timeout 600 python -m pytest tests/test_intra_halo_late.py tests/test_intra_one_sided.py tests/test_gpu_random.py tests/test_gpu_synth.py tests/test_gpu_girlshy.py tests/test_encintra_streams.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/parity.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
for rep in 1 2; do for v in old nw12 base; do for wd in "c2_1080p_intra 1" "c2_1080p_intra 3"; do set -- $wd
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  timeout 200 python bench.py $B --workload $1 --steps 100 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | line $v $1 $2 | tee -a $O/intra_keeper_ab.txt
done; done; done
for v in old base; do for wd in "c3_4k_inter 3" "c5_8k10_8tiles 3"; do set -- $wd
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  timeout 200 python bench.py $B --workload $1 --steps 200 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | line $v $1 $2 | tee -a $O/intra_keeper_ab.txt
done; done
unset M355_LIB

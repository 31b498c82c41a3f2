#!/bin/bash
# round 5, visit 18: only 32x32 blocks shared among a component's waves (noshare = one wave per block, clock costs 2 / 5)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v18; mkdir -p $O
timeout 600 python -m pytest tests/test_intra_halo_late.py tests/test_intra_one_sided.py tests/test_gpu_synth.py tests/test_gpu_random.py tests/test_gpu_girlshy.py tests/test_encintra_streams.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/parity.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
for rep in 1 2; do for v in noshare base; do for wd in "c2_1080p_intra 1" "c2_1080p_intra 3"; do set -- $wd
  if [ "$v" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$v.so; fi
  timeout 200 python bench.py $B --workload $1 --steps 100 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | line $v $1 $2 | tee -a $O/intra_share32_ab.txt
done; done; done
unset M355_LIB

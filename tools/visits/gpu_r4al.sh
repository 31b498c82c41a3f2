#!/bin/bash
# round 4, visit al: launches per picture — zero fill inside k_job_count, CU + SAO planes in one launch, job scan inside k_meta_pb (on top of
# the one mark per decode) against variants/prev2.so (commit 69591ea, before both); M355_SINGLE_STREAM=1 on the new build; hardware parity
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r4al; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_random.py tests/test_gpu_synth.py tests/test_gpu_girlshy.py tests/test_gpu_pipeline.py tests/test_gpu_shard.py tests/test_gpu_batch.py -m gpu -x -q 2>&1 | tail -2 | tee $O/parity.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain"
run() { # tag env workload
  timeout 200 env $2 python bench.py $B --workload $3 --steps 200 --warmup 10 --pipeline-depth 3 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-14s %-16s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f host enqueue %.4f  %s' % ('$1', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], d['host_enqueue_ms_per_step'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $O/launches.txt
}
for rep in 1 2; do for w in c5_8k10_8tiles c3_4k_inter c4_4k_4tiles; do
  run prev2 M355_LIB=$REPO/libde265_amd/variants/prev2.so $w
  run new M355_X=0 $w
  run new-1stream M355_SINGLE_STREAM=1 $w
done; done

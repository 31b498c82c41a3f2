#!/bin/bash
# first hardware visit of M355_INTRA_ONE_SIDED (≈12 GPU-minutes; written in round 4's last session, emulator-verified only): gpurun --timeout 1500 -- 'bash tools/gpu_r5c.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5c; mkdir -p $O
# k_intra's dependency levels from what each intra mode can read (M355_INTRA_ONE_SIDED=1:
# 42 % fewer levels on the C2 picture): parity, then C2 one picture at a time / three in flight and C5, alternating
M355_TEST_INTRA_ONE_SIDED=1 timeout 600 python -m pytest tests/test_intra_one_sided.py -m gpu -x -q 2>&1 | tail -3 | sed "s/^/test_intra_one_sided: /" | tee -a $O/parity.txt
M355_INTRA_ONE_SIDED=1 timeout 900 python -m pytest tests/test_gpu_random.py tests/test_gpu_synth.py tests/test_gpu_girlshy.py tests/test_gpu_encintra.py tests/test_gpu_pipeline.py tests/test_gpu_batch.py tests/test_streams.py -m gpu -x -q 2>&1 | tail -1 | sed "s/^/M355_INTRA_ONE_SIDED=1: /" | tee -a $O/parity.txt
for rep in 1 2; do for m in 0 1; do for wd in "c2_1080p_intra 1" "c2_1080p_intra 3" "c5_8k10_8tiles 3" "c3_4k_inter 3"; do set -- $wd
  M355_INTRA_ONE_SIDED=$m timeout 200 python bench.py --no-cpu-baseline --no-end-to-end --no-with-upload --no-dependent-chain --workload $1 --steps 100 --warmup 10 --pipeline-depth $2 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('M355_INTRA_ONE_SIDED=$m %-16s depth $2 %.4f ms/pic (p10 %.4f p90 %.4f) %s' % ('$1', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))" | tee -a $O/one_sided.txt
done; done; done

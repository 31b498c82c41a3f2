#!/bin/bash
# round 6, visit 21: three cheap A/Bs — pictures in flight (2 / 3 / 4) on this round's tree; k_inter_jobs' + k_sao's stores plain instead of non-temporal; four instead of eight coefficient
# pairs per lane and batch for 4x4 blocks in k_residual
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v21; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-verify"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ch=d.get('dependent_chain') or {}
print('%-9s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f chain %.4f stages %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], ch.get('ms_per_step') or 0, ' '.join('%s=%.4f' % kv for kv in d['stage_ms'].items())))"; }
run() { # variant workload depth
  if [ "$1" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$1.so; fi
  timeout 300 python bench.py $B --workload $2 --steps 200 --warmup 10 --pipeline-depth $3 2>>$O/bench.err | line $1 $2 $3 | tee -a $O/ab.txt
  unset M355_LIB
}
for d in 2 3 4 2 3 4; do run base c5_8k10_8tiles $d; done
for v in base plainst resgb4 base plainst resgb4; do run $v c5_8k10_8tiles 3; done
for v in base plainst resgb4 base plainst resgb4; do run $v c3_4k_inter 3; done
M355_LIB=$REPO/libde265_amd/variants/resgb4.so timeout 600 python -m pytest tests/test_gpu_synth.py tests/test_gpu_random.py -m gpu -q -x 2>&1 | tail -2 | tee $O/pytest_resgb4.txt

#!/bin/bash
# round 5, visit 24: the 12-wave kernel of intra pictures in flight at 80 / 96 registers (d6 / d5: two workgroups per CU instead of one) — C2 with three
# in flight, nine in flight, batches of 8; the tree after visit 23's revert (base) on C3 / C4 / C5
#   gpurun --timeout 900 -- 'bash tools/gpu_r5_v24.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v24; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "parity"
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_all.txt
for v in d6 d5; do M355_LIB=$REPO/libde265_amd/variants/$v.so timeout 300 python -m pytest tests/test_gpu_synth.py tests/test_gpu_batch.py tests/test_gpu_encintra.py tests/test_gpu_girlshy.py -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -1 | sed "s/^/$v: /" | tee -a $O/pytest_all.txt; done
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-6s %-16s depth %-5s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f'%(k,v) for k,v in d['stage_ms'].items())))"; }
run() { # variant workload depth [extra]
  if [ "$1" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$1.so; fi
  timeout 300 python bench.py $B --workload $2 --steps ${5:-200} --warmup 10 --pipeline-depth $3 $4 2>>$O/bench.err | line $1 $2 "$3$4" | tee -a $O/intra_dense_regs_ab.txt
  unset M355_LIB
}
stamp "A/B"
for v in base d6 d5 d5 d6 base; do run $v c2_1080p_intra 3; done
for v in base d6 d5; do run $v c2_1080p_intra 9; done
for v in base d6 d5 d6 base; do run $v c2_1080p_intra 32 "--intra-batch 8" 96; done
for wd in "c3_4k_inter 3" "c4_4k_4tiles 3" "c5_8k10_8tiles 3" "c2_1080p_intra 1"; do set -- $wd; run base $1 $2; done
stamp done

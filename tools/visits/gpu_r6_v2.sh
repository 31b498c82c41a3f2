#!/bin/bash
# round 6, visit 2: the forced-schedule parity tests (tests/test_gpu_chain_forced.py), the bench line with its frame checks, and the first k_inter A/B of the round:
# VOP3P chain heads (product) against the v_mov + v_dot2c heads (nodot2z), window-row touches at three levels (touch1..3)
#   gpurun --timeout 1800 -- 'bash tools/visits/gpu_r6_v2.sh'
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v2; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "forced-schedule tests"
timeout 1500 python -m pytest tests/test_gpu_chain_forced.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -8 | tee $O/pytest_forced.txt
stamp "GPU tier (rest)"
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_chain_forced.py 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_all.txt
stamp "driver's command"
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_line.json 2> $O/bench.err ) 2>&1 | grep real | tee -a $O/timeline.txt
python -c "
import json; d=json.loads(open('$O/bench_driver_line.json').read().strip().splitlines()[-1]); print('driver-like:', d['value'], d['ms_per_step'], d['stage_ms'], 'frac', d['roofline']['frac'], 'verified', d.get('verified'), 'chain', json.dumps(d.get('dependent_chain'))[:300], 'rot', json.dumps(d.get('rotating_references'))[:100], 'up', d['with_upload'].get('verified'), d['with_upload']['with_transfers'].get('verified'))" | tee -a $O/timeline.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain --no-verify"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-8s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f stages %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], ' '.join('%s=%.4f' % kv for kv in d['stage_ms'].items())))"; }
run() { # variant workload depth
  if [ "$1" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$1.so; fi
  timeout 200 python bench.py $B --workload $2 --steps 200 --warmup 10 --pipeline-depth $3 2>>$O/bench.err | line $1 $2 $3 | tee -a $O/inter_ab.txt
  unset M355_LIB
}
stamp "A/B"
for wd in "c5_8k10_8tiles 3" "c3_4k_inter 3"; do set -- $wd; for v in base nodot2z touch1 touch2 touch3 base nodot2z touch2; do run $v $1 $2; done; done
stamp done

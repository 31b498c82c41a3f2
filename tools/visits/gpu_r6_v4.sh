#!/bin/bash
# round 6, visit 4: window kernel, second cut (column-per-lane staging, one chroma phase for both lists, 8-byte LDS reads): where does it differ from the oracle (tools/diag_inter.py),
# parity suites, A/B against the per-lane kernel
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v4; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "diag"
timeout 300 python tools/diag_inter.py 0 1 2 5 > $O/diag_inter.txt 2>&1; grep -c DIFFERS $O/diag_inter.txt; head -40 $O/diag_inter.txt
stamp "inter parity"
timeout 600 python -m pytest tests/test_gpu_synth.py tests/test_gpu_random.py tests/test_inter_extremes.py tests/test_inter_narrow.py -m gpu -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -12 | tee $O/pytest_inter.txt
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
for wd in "c5_8k10_8tiles 3" "c3_4k_inter 3" "c5x_cu64 3" "c5x_cu16 3"; do set -- $wd; for v in base perlane $EXTRA_VARIANTS; do run $v $1 $2; done; done
stamp done

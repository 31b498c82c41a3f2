#!/bin/bash
# round 6, visit 11: k_inter_jobs' luma window rows fetched by the lanes of a run together (d_mc_luma_coop, 16-bit planes) against the per-lane fetch (variants/perlane.so): parity, then A/B
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v11; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "diag + inter parity"
timeout 300 python tools/diag_inter.py 2 4 7 > $O/diag_inter.txt 2>&1; grep -c DIFFERS $O/diag_inter.txt; grep -A4 DIFFERS $O/diag_inter.txt | head -20
timeout 600 python -m pytest tests/test_gpu_synth.py tests/test_gpu_random.py tests/test_inter_extremes.py tests/test_inter_narrow.py -m gpu -q 2>&1 | grep -E "passed|failed|error|Error|assert" | tail -6 | tee $O/pytest_inter.txt
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
for wd in "c5_8k10_8tiles 3" "c5x_cu64 3" "c5x_cu16 3" "c5_8k10_8tiles 1"; do set -- $wd; for v in base perlane base perlane; do run $v $1 $2; done; done
stamp done

#!/bin/bash
# round 6, visit 18: streaming accesses of the filter / residual kernels marked non-temporal (variants nt1 k_sao loads, nt2 k_deblock luma, nt4 k_residual rows + pairs, nt7 all three):
# do k_inter_jobs' reference frames stay in the Infinity Cache, and what do the filters lose?
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v18; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-verify"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
rot=d.get('rotating_references') or {}; ch=d.get('dependent_chain') or {}
print('%-5s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f rotating %.4f chain %.4f stages %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], rot.get('ms_per_step') or 0, ch.get('ms_per_step') or 0, ' '.join('%s=%.4f' % kv for kv in d['stage_ms'].items())))"; }
run() { # variant workload depth
  if [ "$1" = base ]; then unset M355_LIB; else export M355_LIB=$REPO/libde265_amd/variants/$1.so; fi
  timeout 300 python bench.py $B --workload $2 --steps 200 --warmup 10 --pipeline-depth $3 2>>$O/bench.err | line $1 $2 $3 | tee -a $O/nt_ab.txt
  unset M355_LIB
}
stamp "C5"
for d in 1 3; do for v in base nt1 nt2 nt4 nt7 base nt7; do run $v c5_8k10_8tiles $d; done; done
stamp "C3 / C4"
for w in c3_4k_inter c4_4k_4tiles; do for v in base nt7 base nt7; do run $v $w 3; done; done
stamp "parity nt7"
M355_LIB=$REPO/libde265_amd/variants/nt7.so timeout 600 python -m pytest tests/test_gpu_synth.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt
stamp done

#!/bin/bash
# round 6, visit 16: the reference frames streamed into the Infinity Cache in front of a picture's kernels (k_ref_prefetch; M355_X_REF_PREFETCH 0 / 1 side stream / 2 main stream / 3 any lane)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v16; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-verify"
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
rot=d.get('rotating_references') or {}; ch=d.get('dependent_chain') or {}
print('prefetch %s %-16s depth %s %.4f ms/pic (p10 %.4f p90 %.4f) one-at-a-time %.4f rotating %s chain %s stages %s' % ('$1', '$2', '$3', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['ms_per_step_one_in_flight'], rot.get('ms_per_step'), ch.get('ms_per_step'), ' '.join('%s=%.4f' % kv for kv in d['stage_ms'].items())))"; }
run() { # mode workload depth
  M355_X_REF_PREFETCH=$1 timeout 300 python bench.py $B --workload $2 --steps 200 --warmup 10 --pipeline-depth $3 2>>$O/bench.err | line $1 $2 $3 | tee -a $O/prefetch_ab.txt
}
stamp "C5"
for d in 1 3; do for m in 0 1 2 0 1 2; do run $m c5_8k10_8tiles $d; done; done
stamp "C3 / C4"
for w in c3_4k_inter c4_4k_4tiles; do for m in 0 3 0 3; do run $m $w 3; done; done
stamp "parity with the prefetch on"
M355_X_REF_PREFETCH=1 timeout 600 python -m pytest tests/test_gpu_synth.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | tail -3 | tee $O/pytest.txt
stamp done

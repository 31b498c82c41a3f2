#!/bin/bash
# round 4, visit aj: the seven stage-timing events per decode, recorded always (M355_ALWAYS_TIME=1, as before) vs only inside a
# m355_timing_reset .. m355_timing_collect window; C5 / C3, three in flight, alternating (one box)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r4aj; mkdir -p $O
B="--no-cpu-baseline --no-dependent-chain --no-end-to-end"
for rep in 1 2 3; do for a in 1 0; do for w in c5_8k10_8tiles c3_4k_inter; do
  M355_ALWAYS_TIME=$a timeout 200 python bench.py $B --workload $w --steps 200 --warmup 10 --pipeline-depth 3 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); u=d.get('with_upload') or {}
print('events always=$a %-16s %.4f ms/pic (p10 %.4f p90 %.4f) host enqueue %.4f  submit_only %.4f with_upload %.4f' % ('$w', d['ms_per_step'], d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90'], d['host_enqueue_ms_per_step'], u.get('submit_only',{}).get('ms_per_step',0), u.get('ms_per_step',0)))" | tee -a $O/events.txt
done; done; done

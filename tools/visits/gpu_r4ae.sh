#!/bin/bash
# round 4, visit ae: which hardware queue do the lanes' main streams get?  idle streams in front of every second lane (M355_X_STREAM_PAD)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r4ae; mkdir -p $O
for w in c5_8k10_8tiles; do for d in 3 4 5 6; do for s in 0 1 2 3; do
  M355_X_STREAM_PAD=$s timeout 300 python bench.py --workload $w --steps 200 --warmup 10 --pipeline-depth $d --no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end 2>$O/err.log | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w depth $d pad=$s: %.4f ms/pic = %.3f M CTB64/s (p10 %.4f p90 %.4f)' % (d['ms_per_step'], d['value']/1e6, d['ms_per_step_spread']['p10'], d['ms_per_step_spread']['p90']))" | tee -a $O/summary.txt
done; done; done

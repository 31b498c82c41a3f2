#!/bin/bash
# round 4, visit ai: staging-arena ring of the submit path, 3 vs 4 vs depth + 3 (= 6), alternating, three times each (one box)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r4ai; mkdir -p $O
for rep in 1 2 3; do for r in 3 4 6; do
  M355_TRANSIENT_RING=$r timeout 300 python bench.py --no-cpu-baseline --no-dependent-chain --no-end-to-end --steps 20 --warmup 5 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); u=d['with_upload']
print('ring $r: with_upload %.4f ms  submit_only %.4f ms  copying %.4f ms  (resident lists %.4f ms)' % (u['ms_per_step'], u['submit_only']['ms_per_step'], u['copying_submit']['ms_per_step'], d['ms_per_step']))" | tee -a $O/ring.txt
done; done

#!/bin/bash
# round 6, visit 46: pictures in flight at 4K (C3 / C4) and 1080p-sized inter pictures: 3 .. 7 lanes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v46; mkdir -p $O
B="--no-cpu-baseline --no-with-upload --no-dependent-chain --no-end-to-end --no-cold-refs"
for w in c3_4k_inter c4_4k_4tiles; do for d in 3 4 5 6 7 3 5; do
  timeout 200 python bench.py $B --workload $w --steps 200 --warmup 20 --pipeline-depth $d 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w depth $d: %.4f ms/picture (spread %s), one at a time %.4f' % (d['ms_per_step'], d.get('ms_per_step_spread'), d['ms_per_step_one_in_flight']))" | tee -a $O/depth_4k.txt
done; done

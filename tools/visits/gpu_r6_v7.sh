#!/bin/bash
# round 6, visit 7: why does bench.py's tile-sharded leg fail over the ipc transport when the test's rank processes do not?
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v7; mkdir -p $O
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain --no-verify"
M355_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 $B > $O/bench_share_2.json 2>$O/bench_share_2.err
grep -v "^\[W\|Gloo\|amdgpu.ids" $O/bench_share_2.err | tail -30
python -c "
import json; d=json.loads(open('$O/bench_share_2.json').read().strip().splitlines()[-1]); print('N=2 value', d['value'], 'tile_sharded', json.dumps(d.get('tile_sharded'))[:900])"

#!/bin/bash
# round 6, visit 54: the multi-rank bench path on the last tree — bench.py --gpus 2 / 4 as the driver launches it (torch.distributed.run) and self-launched, the ranks sharing the GPU
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=$REPO/gpurun_out/r6v54; mkdir -p $O
for n in 2 4; do
  M355_BENCH_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29600 + n)) bench.py --gpus $n --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > $O/bench_share_$n.json 2> $O/bench_share_$n.err
  python -c "
import json; d=json.loads(open('$O/bench_share_$n.json').read().strip().splitlines()[-1]); print('N=$n (driver launch) value', d['value'], 'ms', d['ms_per_step'], 'n_gpus', d['n_gpus'], 'scaling', d['scaling'], 'verified', d.get('verified'), 'tile_sharded', json.dumps(d.get('tile_sharded'))[:700])" 2>&1 | tee -a $O/lines.txt
  grep -E "Error|error" $O/bench_share_$n.err | grep -v "hostname" | tail -3
done

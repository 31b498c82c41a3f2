#!/bin/bash
# round 6, visit 8: the interprocess transport with its own flag words (signal / wait kernels instead of HIP's interprocess events): rank processes sharing the GPU, the in-process group
# (its repack wait now also covers non-reference pictures), bench.py --gpus 2 / 4 / 8 with the ranks sharing the GPU
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v8; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "ipc transport, rank processes"
timeout 1500 python -m pytest tests/test_gpu_shard_ipc.py tests/test_gpu_shard.py -m gpu -q 2>&1 | grep -E "passed|failed|error|Error|assert|ok.: False" | tail -8 | tee $O/pytest_ipc.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain --no-verify"
stamp "bench --gpus N, ranks share the GPU"
for n in 2 4 8; do
  M355_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus $n --steps 30 --warmup 5 $B > $O/bench_share_$n.json 2>$O/bench_share_$n.err
  python -c "
import json; d=json.loads(open('$O/bench_share_$n.json').read().strip().splitlines()[-1]); print('N=$n value', d['value'], 'ms', d['ms_per_step'], 'tile_sharded', json.dumps(d.get('tile_sharded'))[:900])" 2>&1 | tee -a $O/timeline.txt
  grep -E "Error|error" $O/bench_share_$n.err | grep -v "hostname" | tail -3
done
stamp done

#!/bin/bash
# round 6, visit 6: the interprocess tile-sharding transport on the hardware — rank PROCESSES sharing the one GPU (tests/test_gpu_shard_ipc.py), bench.py --gpus 2 / 4 / 8 with the ranks
# sharing the GPU (M355_BENCH_SHARE_GPU=1: plumbing of the driver's multi-GPU command, ipc transport in the tile-sharded leg), and whether more hardware queues cure the second-runtime slowdown
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v6; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "ipc transport, rank processes"
timeout 1500 python -m pytest tests/test_gpu_shard_ipc.py -m gpu -q -x 2>&1 | grep -E "passed|failed|error|Error|assert|ok.: False" | tail -8 | tee $O/pytest_ipc.txt
B="--no-cpu-baseline --no-end-to-end --no-with-upload --no-cold-refs --no-dependent-chain --no-verify"
stamp "bench --gpus N, ranks share the GPU"
for n in 2 4 8; do
  M355_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus $n --steps 30 --warmup 5 $B > $O/bench_share_$n.json 2>$O/bench_share_$n.err
  python -c "
import json; d=json.loads(open('$O/bench_share_$n.json').read().strip().splitlines()[-1]); print('N=$n value', d['value'], 'ms', d['ms_per_step'], 'tile_sharded', json.dumps(d.get('tile_sharded'))[:700])" 2>&1 | tee -a $O/timeline.txt
  tail -3 $O/bench_share_$n.err
done
stamp "hardware queues"
for q in 4 8 16; do echo "GPU_MAX_HW_QUEUES=$q"; M355_AB_ARMS=base,torch,streams,nccl GPU_MAX_HW_QUEUES=$q timeout 400 python tools/rccl_idle_ab.py c5_8k10_8tiles 60 2>&1 | grep -E "^(base|torch|streams|nccl) " ; done | tee $O/hw_queues_ab.txt
stamp done

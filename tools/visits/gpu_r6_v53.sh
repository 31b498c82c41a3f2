#!/bin/bash
# round 6, visit 53: the soaks once more on the last tree (k_sao / k_residual with the XCD-contiguous orders)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v53; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
timeout 600 python tools/soak_gpu.py 400000 8000 48 2>&1 | tail -8 | tee $O/soak_gpu.txt | cut -c1-300
SOAK_SCALE=8 timeout 600 python tools/soak_gpu.py 410000 1600 32 2>&1 | tail -8 | tee $O/soak_gpu_scale8.txt | cut -c1-300
SOAK_SCALE=12 timeout 600 python tools/soak_gpu.py 420000 500 16 2>&1 | tail -8 | tee $O/soak_gpu_scale12.txt | cut -c1-300
timeout 600 python tools/soak_shard.py 430000 2400 24 2>&1 | tail -8 | tee $O/soak_shard.txt | cut -c1-300
SOAK_SCALE=6 timeout 600 python tools/soak_shard.py 440000 600 16 2>&1 | tail -8 | tee $O/soak_shard_scale6.txt | cut -c1-300
timeout 600 python tools/soak_streams.py 450000 2400 24 2>&1 | tail -8 | tee $O/soak_streams.txt | cut -c1-300
SOAK_BIG=4 timeout 600 python tools/soak_streams.py 460000 300 24 2>&1 | tail -8 | tee $O/soak_streams_big4.txt | cut -c1-300
timeout 600 python tools/soak_chain.py 470000 3000 24 2>&1 | tail -8 | tee $O/soak_chain.txt | cut -c1-300
SOAK_SCALE=8 timeout 600 python tools/soak_chain.py 480000 400 16 2>&1 | tail -8 | tee $O/soak_chain_scale8.txt | cut -c1-300

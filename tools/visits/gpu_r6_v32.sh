#!/bin/bash
# round 6, visit 32: the soaks at SIZE — random pictures 8x and 12x as wide and high (up to 5120 x 2880 / 7680 x 4320: the launch orders the runtime picks by size), streams 4x
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v32; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
SOAK_SCALE=8 timeout 1200 python tools/soak_gpu.py 200000 4000 32 2>&1 | tail -25 | tee $O/soak_gpu_scale8.txt
SOAK_SCALE=12 timeout 1200 python tools/soak_gpu.py 210000 1000 16 2>&1 | tail -25 | tee $O/soak_gpu_scale12.txt
SOAK_BIG=4 timeout 1200 python tools/soak_streams.py 10000 400 24 2>&1 | tail -25 | tee $O/soak_streams_big4.txt
free -g | head -2 | tee $O/mem.txt

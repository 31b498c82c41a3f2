#!/bin/bash
# round 6, visit 27: the soak again, 100 000 further random pictures (seeds 5000 .. 104999) on 64 processes sharing the GPU
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v27; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
timeout 2400 python tools/soak_gpu.py 5000 100000 64 2>&1 | tee $O/soak_gpu.txt

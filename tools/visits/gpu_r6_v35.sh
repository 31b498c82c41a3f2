#!/bin/bash
# round 6, visit 35: the 8x soak's differences with a second look at the same device frame (tools/soak_recheck.py), 32 processes sharing the GPU
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v35; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
SOAK_SCALE=8 timeout 900 python tools/soak_recheck.py 200000 1600 32 2>&1 | tail -45 | tee $O/soak_recheck.txt | cut -c1-300

#!/bin/bash
# round 6, visit 38: soak of the interprocess transport — groups of 2..4 rank processes sharing the GPU on random tiled pictures (tools/soak_ipc.py), two groups at a time
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v38; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
timeout 1200 python tools/soak_ipc.py 300000 16 12 2 2>&1 | tail -14 | tee $O/soak_ipc.txt | cut -c1-500

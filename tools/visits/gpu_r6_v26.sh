#!/bin/bash
# round 6, visit 26: soak — 4000 random pictures beyond the suite's seeds on the GPU against the oracle (tools/soak_gpu.py), 32 processes sharing the GPU
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v26; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
timeout 1500 python tools/soak_gpu.py 1000 4000 32 2>&1 | tee $O/soak_gpu.txt

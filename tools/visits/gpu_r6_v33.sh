#!/bin/bash
# round 6, visit 33: the four pictures of visit 32's 12x soak that differed — by stage, three runs each; then the 8x soak again (an error of the library ended a worker there)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v33; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
SOAK_SCALE=12 timeout 900 python tools/diag_picture.py 3 210700 210716 210732 210780 2>&1 | tee $O/diag_picture.txt | cut -c1-250
SOAK_SCALE=8 timeout 600 python tools/soak_gpu.py 200000 1200 32 2>&1 | tail -25 | tee $O/soak_gpu_scale8.txt | cut -c1-600

#!/bin/bash
# round 6, visit 45: one context for a long time (tools/soak_long.py): ~1 M decodes of small pictures, then 240 k of 1664x960 pictures
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v45; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
timeout 900 python tools/soak_long.py 400 2>&1 | tail -50 | tee $O/soak_long_small.txt | tail -6
timeout 900 python tools/soak_long.py 100 4 2>&1 | tail -20 | tee $O/soak_long_x4.txt | tail -5

#!/bin/bash
# round 6, visit 34: two of visit 33's failing workers' sequences again, alone on the GPU (tools/diag_sequence.py)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v34; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
SOAK_SCALE=8 timeout 900 python tools/diag_sequence.py 200000 12 32 201200 > $O/seq12.txt 2>&1 &
SOAK_SCALE=8 timeout 900 python tools/diag_sequence.py 200000 16 32 201200 > $O/seq16.txt 2>&1 &
wait
grep -c identical $O/seq12.txt $O/seq16.txt

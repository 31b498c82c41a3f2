#!/bin/bash
# round 6, visit 55: soak of m355_decode_batch (tools/soak_batch.py): random all-intra pictures in random batches, also 4x the size
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v55; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
timeout 900 python tools/soak_batch.py 0 3600 24 2>&1 | tail -12 | tee $O/soak_batch.txt | cut -c1-500
SOAK_SCALE=4 timeout 900 python tools/soak_batch.py 10000 400 16 2>&1 | tail -12 | tee $O/soak_batch_scale4.txt | cut -c1-500

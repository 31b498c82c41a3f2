#!/bin/bash
# round 6, visit 43: the stream soak driven the ways applications drive a decoder (SOAK_APP=1: pieces / NAL by NAL / reset in mid-stream), 2400 streams
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v43; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
SOAK_APP=1 timeout 1200 python tools/soak_streams.py 20000 2400 24 2>&1 | tail -12 | tee $O/soak_app.txt | cut -c1-600

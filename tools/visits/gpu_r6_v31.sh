#!/bin/bash
# round 6, visit 31: the tile-sharded soak (tools/soak_shard.py, in-process groups of 2..8 virtual ranks on random tiled pictures) and the stream soak with one
# bitstream spread over 3 ranks (M355_GLUE_RANKS) and with one lane
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v31; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
timeout 1500 python tools/soak_shard.py 0 6000 24 2>&1 | tail -25 | tee $O/soak_shard.txt
M355_GLUE_RANKS=3 timeout 900 python tools/soak_streams.py 4000 1200 24 2>&1 | tail -25 | tee $O/soak_streams_ranks3.txt
M355_PIPELINE_DEPTH=1 timeout 900 python tools/soak_streams.py 6000 1200 24 2>&1 | tail -25 | tee $O/soak_streams_depth1.txt

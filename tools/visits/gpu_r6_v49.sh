#!/bin/bash
# round 6, visit 49: every way of dealing three lanes' six streams to the four hardware queues (tools/qmap_search.py), C5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v49; mkdir -p $O
timeout 2000 python tools/qmap_search.py c5_8k10_8tiles 3 60 2>&1 | tee $O/qmap_search_c5_depth3.txt | tail -20

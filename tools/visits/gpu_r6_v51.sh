#!/bin/bash
# round 6, visit 51: the deal of three lanes' six streams to the four hardware queues again, by what a DEPENDENT CHAIN gets (QMAP_CHAIN=1 tools/qmap_search.py), C5
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v51; mkdir -p $O
QMAP_CHAIN=1 timeout 2300 python tools/qmap_search.py c5_8k10_8tiles 3 40 2>&1 | tee $O/qmap_search_chain_c5_depth3.txt | tail -20

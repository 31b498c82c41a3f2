#!/bin/bash
# round 5, visit 19: where k_intra's time goes under the clock levels — per-CTB timeline and the per-level profile of a CTB in the middle of the picture
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r5v19; mkdir -p $O
M355_LIB=$REPO/libde265_amd/variants/prof.so timeout 120 python tools/prof_timeline.py 2>&1 | tee $O/timeline.txt
M355_LIB=$REPO/libde265_amd/variants/prof255.so timeout 120 python tools/prof_intra.py 2>&1 | tee $O/levels_item255_wave0.txt
M355_LIB=$REPO/libde265_amd/variants/prof255w3.so timeout 120 python tools/prof_intra.py 2>&1 | tee $O/levels_item255_wave3.txt

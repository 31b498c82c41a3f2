#!/bin/bash
# round 6, visit 12: tools/ubench/ub_tile.hip — k_inter_jobs' traffic model on linear planes against 8 x 8 line tiles, 2 / 3 / 4 / 8 workgroups per CU
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v12; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
for w in 3 2 4 8; do stamp "wg_per_cu $w"; timeout 300 tools/ubench/_build/ub_tile $w 2>&1 | tee -a $O/ub_tile.txt; done
stamp done

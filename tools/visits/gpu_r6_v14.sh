#!/bin/bash
# round 6, visit 14: k_inter_jobs<u16> outside the library (tools/ubench/ub_kinter.hip: the product kernel and cut-down outer kernels on a synthetic picture)
# and its traffic model with the product's structure as knobs (tools/ubench/ub_inter_model.hip)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v14; mkdir -p $O
T0=$(date +%s); stamp() { echo "[$(( $(date +%s) - T0 )) s] $*" | tee -a $O/timeline.txt; }
stamp "product kernel in the harness"
for b in 100 50 0; do timeout 300 tools/ubench/_build/ub_kinter $b 2>&1 | tee -a $O/ub_kinter.txt; done
stamp "model"
timeout 300 tools/ubench/_build/ub_inter_model 2>&1 | tee $O/ub_inter_model.txt
stamp done

#!/bin/bash
# round 6, visit 15: ub_kinter again, the product kernel by the state of the caches (warm / behind a 1 GiB streaming write / that + the reference frames streamed in)
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v15; mkdir -p $O
for b in 50 100; do timeout 300 tools/ubench/_build/ub_kinter $b 2>&1 | tee -a $O/ub_kinter.txt; done

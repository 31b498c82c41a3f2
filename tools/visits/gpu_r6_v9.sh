#!/bin/bash
# round 6, visit 9: the interprocess transport's GPU tests, one by one with their times
cd "$GRAFT_REPO_ROOT" || exit 1
REPO=$GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=$REPO/gpurun_out/r6v9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_shard_ipc.py -m gpu -v --durations=0 2>&1 | grep -E "PASSED|FAILED|passed|failed|s call|error.:|Error" | cut -c1-600 | tee $O/pytest_ipc.txt

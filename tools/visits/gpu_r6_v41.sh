#!/bin/bash
# round 6, visit 41: the interprocess soak inside the transport's envelope on ONE GPU (device-side flag words: up to four rank processes polling beside each other) — one group at a time
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v41; mkdir -p $O
make -s -C oracle >/dev/null 2>&1
M355_IPC_TIMEOUT=30 SOAK_IPC_RANK_TIMEOUT=200 timeout 1500 python tools/soak_ipc.py 300000 60 12 1 2>&1 | tail -10 | tee $O/soak_ipc_one_group_at_a_time.txt | cut -c1-1200

#!/bin/bash
# round 6, visit 39: the interprocess soak's failures again with a log per rank (open file descriptors per picture, the Python stack of a rank that hangs)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp; O=$GRAFT_REPO_ROOT/gpurun_out/r6v39; mkdir -p $O/diag
make -s -C oracle >/dev/null 2>&1
ulimit -n | tee $O/ulimit.txt
M355_IPC_DIAG_DIR=$O/diag timeout 1000 python tools/soak_ipc.py 300000 16 12 2 2>&1 | tail -14 | tee $O/soak_ipc.txt | cut -c1-300
for f in $O/diag/*.log; do echo "$f: $(head -1 $f | cut -c1-40) ... $(tail -1 $f | cut -c1-60)"; done | tee $O/fd_summary.txt | tail -50
ls -la $O/diag/*.stack | awk '$5 > 0' | tee $O/stacks.txt
